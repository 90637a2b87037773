#!/bin/bash
O=gpurun_out/call40; mkdir -p $O
{
echo "=== channel-major K walk: correctness"; SGA_KORDER=1 python tests/tools/korder_check.py 2>&1 | grep -v amdgpu.ids
for p in f32 bf16x3 bf16x2; do for ko in 1 0 1 0; do
echo "=== lab SGA_KORDER=$ko $p"; LAB=1 SGA_KORDER=$ko PREC=$p python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -7 | tail -6
done; done
echo "=== iteration (lab, graph replay)"; python scripts/ab_iter.py --rounds 2 "LAB=1" "LAB=1 SGA_KORDER=1"
} > $O/summary.log 2>&1
