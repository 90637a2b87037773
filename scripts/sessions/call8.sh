#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call8; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
run() { echo "--- $*" | tee -a $OUT/summary.log; env "$@" 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/summary.log; }
step "x3 race at 1x512x768, 40-iteration runs"
run LAB=1 python scripts/x3_race2.py 120 40
run LAB=1 SGA_X3_FORK=0 python scripts/x3_race2.py 120 40
run LAB=1 SGA_X3_ONLY=1 python scripts/x3_race2.py 120 40
run LAB=1 SGA_X3_ONLY=2 python scripts/x3_race2.py 120 40
run LAB=1 SGA_NO_GRAPH=1 python scripts/x3_race2.py 120 40
run LAB=1 SGA_X3_VARIANTS=0 python scripts/x3_race2.py 120 40
run LAB=1 PREC=f32 python scripts/x3_race2.py 120 40
run LAB=1 python scripts/x3_race2.py 60 40 256 256 8
step "rANS throughput at Tecnick size"
python scripts/rans_throughput.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
