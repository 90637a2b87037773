#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call5; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "x3 pipelined loop == single-stage loop (bitwise)"
timeout 600 python scripts/x3_loop_equal.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.log
step "perf bf16x3"
python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
SGA_LIB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_x3ref.so python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
step "per-layer bf16x3"
PREC=bf16x3 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | tee $OUT/layers_x3.txt | head -24 | tee -a $OUT/summary.log
step "FULL GPU suite (no -x)"
timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -12 $OUT/tests.log | tee -a $OUT/summary.log
