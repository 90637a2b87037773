#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call12; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
LAB=1 SAVE_NPZ=$OUT/dev.npz SGA_DEBUG_DUMP=/tmp/x3d SGA_DEBUG_DUMP_BUFS=1 timeout 900 python scripts/x3_race4.py 400 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/summary.log
