#!/bin/bash
O=gpurun_out/call31; mkdir -p $O
{
for p in bf16x3 bf16x2; do
echo "=== $p, K loop without its global loads (wrong results; timing only)"; LAB=1 SGA_X3_NOLOAD=1 PREC=$p python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -8
echo "=== $p, as shipped (lab build)"; LAB=1 PREC=$p python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -8
done
} > $O/summary.log 2>&1
