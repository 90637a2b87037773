#!/bin/bash
# GPU box, round 5: the tests touched by this round's host changes (graph cache, base_compress scratch, driver outputs,
# fork-name rule, persistent igdn2.bwd in the lab build, acceptance per-image bounds on two small sets)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s10; mkdir -p $OUT
timeout 2400 python -X faulthandler -m pytest tests/test_gpu_configs.py tests/test_gpu_step.py tests/test_gpu_bb.py tests/test_gpu_c_abi.py -x -q -m gpu > $OUT/tests_a.log 2>&1; echo "rc $?" >> $OUT/tests_a.log; tail -8 $OUT/tests_a.log
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "persistent or lab_build" > $OUT/tests_b.log 2>&1; echo "rc $?" >> $OUT/tests_b.log; tail -5 $OUT/tests_b.log
SGA_FORK_NAME=gs0.fwd SGA_FORK_VERBOSE=1 timeout 300 python scripts/ab_iter.py --rounds 1 "SGA_FORK_NAME=gs0.fwd" "SGA_FORK_NAME=start" "SGA_NO_OVERLAP=1" > $OUT/fork_first.txt 2>&1; cat $OUT/fork_first.txt
