#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call4; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "bf16 / f32 MFMA sustained rate"
scripts/mfma_clock_bf16.bin 2>&1 | tee $OUT/mfma_clock_bf16.txt | tee -a $OUT/summary.log
step "FULL GPU suite"
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -15 $OUT/tests.log | tee -a $OUT/summary.log
step "bench"
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "rc $?" | tee -a $OUT/summary.log; cat $OUT/bench.json | tee -a $OUT/summary.log
