#!/bin/bash
O=gpurun_out/call29; mkdir -p $O
{
echo "=== swizzled bf16 LDS stages: tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -q 2>&1 | tail -4
echo "=== perf"; for p in bf16x2 bf16x3 f32; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
echo "=== layers bf16x3"; PREC=bf16x3 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -8
echo "=== layers bf16x2"; PREC=bf16x2 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -8
} > $O/summary.log 2>&1
