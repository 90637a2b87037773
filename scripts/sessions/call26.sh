#!/bin/bash
O=gpurun_out/call26; mkdir -p $O
{
echo "=== plan sweep"; timeout 1400 python -m pytest tests/test_gpu_configs.py -q -k across_launch_plans 2>&1 | tail -15
echo "=== bf16x2 ops + step"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -q -k bf16x2 2>&1 | tail -15
echo "=== bf16x2 acceptance"; timeout 900 python -m pytest tests/test_gpu_acceptance.py -q -k bf16x2 2>&1 | tail -15
echo "=== bench (short its)"; python bench.py --no-cpu-baseline --no-other-configs --no-other-input --steps 1 --warmup 1 2>/dev/null | tail -c 2500
} > $O/summary.log 2>&1
