#!/bin/bash
O=gpurun_out/call35; mkdir -p $O
{
echo "=== scalar tap-table read in the bf16 K loop"
for p in bf16x3 bf16x2; do
echo "=== $p"; PREC=$p python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -7 | tail -6
done
echo "=== perf (product)"; for p in bf16x2 bf16x3; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
} > $O/summary.log 2>&1
