#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call3; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "perf f32 / bf16x3 (variants + two-stream) / bf16x3 single-stream / bf16x3 128-row only"
python scripts/perf_modes.py f32 2>&1 | tail -1 | tee -a $OUT/summary.log
python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
SGA_X3_FORK=0 python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
SGA_X3_VARIANTS=0 python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
step "per-layer bf16x3"
PREC=bf16x3 python scripts/profile_layers.py > $OUT/layers_x3.txt 2>&1; cat $OUT/layers_x3.txt | tee -a $OUT/summary.log
step "bf16x3 two-stream determinism (timed fork point, never at the root)"
timeout 600 python scripts/x3_fork_race.py 20 300 > $OUT/x3_race.log 2>&1; tail -4 $OUT/x3_race.log | tee -a $OUT/summary.log
step "tests: fused / step / ops / fullsize / host"
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_step.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -15 $OUT/tests.log | tee -a $OUT/summary.log
