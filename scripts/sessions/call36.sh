#!/bin/bash
O=gpurun_out/call36; mkdir -p $O
{
echo "=== ping-pong K loop (two copies of the loop, one per wave role)"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -q 2>&1 | tail -3
for p in bf16x3 bf16x2; do for pp in 1 0 1 0; do
echo "=== lab SGA_X3_PINGPONG=$pp $p"; LAB=1 SGA_X3_PINGPONG=$pp PREC=$p python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -4 | tail -3
done; done
echo "=== perf (product)"; for p in bf16x2 bf16x3; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
} > $O/summary.log 2>&1
