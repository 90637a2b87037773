#!/bin/bash
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call13; mkdir -p $OUT
echo "=== default build (packed f32 ops allowed)" | tee -a $OUT/summary.log
timeout 600 scripts/trans_mfma_repro.bin 200 13 2>&1 | tee -a $OUT/summary.log
echo "=== built with -target-feature -packed-fp32-ops" | tee -a $OUT/summary.log
timeout 600 scripts/trans_mfma_repro_nopk.bin 200 13 2>&1 | tee -a $OUT/summary.log
