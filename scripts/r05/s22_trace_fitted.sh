#!/bin/bash
# GPU box, round 5: the 2000-iteration trace tests on both goldens (synthetic cfg2 and the fitted C=192 set) in the three modes.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_acceptance.py -q -k trace_2000 -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/s22_trace.log
cat gpurun_out/s22_trace.log
