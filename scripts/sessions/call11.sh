#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call11; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
LAB=1 SGA_DEBUG_DUMP=/tmp/x3d SGA_DEBUG_DUMP_BUFS=1 timeout 900 python scripts/x3_race4.py 200 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
echo "=== without the final evaluation between runs" | tee -a $OUT/summary.log
LAB=1 METRICS=0 SGA_DEBUG_DUMP=/tmp/x3d SGA_DEBUG_DUMP_BUFS=1 timeout 900 python scripts/x3_race4.py 200 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
