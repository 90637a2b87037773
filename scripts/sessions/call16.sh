#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call16; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "bf16x3 post-phase contraction on the bf16 pipe: tests"
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -6 $OUT/tests.log | tee -a $OUT/summary.log
step "perf"
python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
python scripts/perf_modes.py f32 2>&1 | tail -1 | tee -a $OUT/summary.log
PREC=bf16x3 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -12 | tee -a $OUT/summary.log
step "acceptance bf16x3"
timeout 900 python -m pytest tests/test_gpu_acceptance.py -m gpu -q -k "bf16x3" > $OUT/acc.log 2>&1; tail -4 $OUT/acc.log | tee -a $OUT/summary.log
