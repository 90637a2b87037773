#!/bin/bash
O=gpurun_out/call22; mkdir -p $O
{
echo "=== bf16x2 perf"; for p in bf16x2 bf16x3; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
echo "=== bf16x2 layers"; PREC=bf16x2 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -14
echo "=== gz probe"; python tests/tools/gz_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/summary.log 2>&1
tail -40 $O/summary.log
