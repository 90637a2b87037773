#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call14; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
run() { echo "--- $*" | tee -a $OUT/summary.log; env "$@" 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $OUT/summary.log; }
step "library built without packed-f32 instructions: the bf16x3 two-stream races"
run LAB=1 SGA_DEBUG_DUMP=/tmp/x3dump python scripts/x3_race3.py 300 40
run LAB=1 python scripts/x3_race2.py 200 40
run python scripts/x3_race2.py 200 40
run python scripts/x3_fork_race.py 20 300
step "perf"
python scripts/perf_modes.py f32 2>&1 | tail -1 | tee -a $OUT/summary.log
python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
step "FULL GPU suite"
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -12 $OUT/tests.log | tee -a $OUT/summary.log
