#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call9; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "which buffer first (x3, two streams, graph, checksums)"
LAB=1 SGA_DEBUG_DUMP=/tmp/x3dump timeout 900 python scripts/x3_race3.py 200 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
LAB=1 SGA_DEBUG_DUMP=/tmp/x3dump timeout 900 python scripts/x3_race3.py 200 40 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
