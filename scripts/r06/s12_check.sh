#!/bin/bash
# GPU box, round 6: the tests whose bounds were tightened after measurement; a longer A/B of the hyper branch's split-K target
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_s12
timeout 900 python -m pytest tests/test_gpu_acceptance.py -q -k "real_size or (trace_2000 and cfg2_fitted)" 2>&1 | tail -3
python scripts/ab_iter.py --rounds 5 "LAB=1" "LAB=1 SGA_SIDE_TARGET=256" "LAB=1 SGA_SIDE_TARGET=320" 2>&1 | tee gpurun_out/r06_s12/side_target.txt
