#!/bin/bash
O=gpurun_out/call32; mkdir -p $O
{
echo "=== two staging register sets in the bf16 K loop: tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fused.py tests/test_gpu_fullsize.py -q 2>&1 | tail -4
echo "=== perf"; for p in bf16x2 bf16x3 f32; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
echo "=== layers bf16x3"; PREC=bf16x3 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -12
echo "=== layers bf16x2"; PREC=bf16x2 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -12
echo "=== configs"; python scripts/perf_configs.py bf16x3 2>&1 | grep -v amdgpu.ids; python scripts/perf_configs.py bf16x2 2>&1 | grep -v amdgpu.ids
} > $O/summary.log 2>&1
