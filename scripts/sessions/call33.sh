#!/bin/bash
O=gpurun_out/call33; mkdir -p $O
{
echo "=== ping-pong: tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py tests/test_gpu_fused.py -q 2>&1 | tail -3
for pp in 1 0 2; do for p in bf16x3 bf16x2; do
echo "=== SGA_X3_PINGPONG=$pp $p"; LAB=1 SGA_X3_PINGPONG=$pp PREC=$p python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -7 | tail -6
done; done
echo "=== perf (product)"; for p in bf16x2 bf16x3; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
} > $O/summary.log 2>&1
