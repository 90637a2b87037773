#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call10; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "ordering probe: does the branch's first kernel see the main chain's marker?"
LAB=1 SGA_DEBUG_PROBE=1 SGA_DEBUG_DUMP=/tmp/x3dump timeout 900 python scripts/x3_race3.py 200 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
step "the same in f32"
LAB=1 PREC=f32 SGA_DEBUG_PROBE=1 SGA_DEBUG_DUMP=/tmp/x3dump timeout 900 python scripts/x3_race3.py 200 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
step "f32 without the probe (checksums only)"
LAB=1 PREC=f32 SGA_DEBUG_DUMP=/tmp/x3dump timeout 900 python scripts/x3_race3.py 200 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
