#!/bin/bash
O=gpurun_out/call28; mkdir -p $O
{
echo "=== bf16x2 with the two-plane post-phase: tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fused.py -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_acceptance.py -q -k "bf16x2 or bf16x3" 2>&1 | tail -4
echo "=== perf"; for p in bf16x2 bf16x3 f32; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
echo "=== layers bf16x2"; PREC=bf16x2 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -24
echo "=== races"; PREC=bf16x2 python scripts/x3_race2.py 100 40 2>&1 | tail -1; PREC=bf16x2 python scripts/x3_race2.py 40 60 256 256 8 2>&1 | tail -1
} > $O/summary.log 2>&1
