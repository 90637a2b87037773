#!/bin/bash
# GPU box, round 5: the shortest reproducer so far of "defect (a)" (host SIGSEGV in the NEXT run on a handle's process after mid-life
# hipGraphExecDestroy calls; DESIGN_EXPERIMENTS.md A.8a).  Under SGA_GRAPH_DROP=destroy the file tests/test_gpu_configs.py ALONE
# crashed 2 of 2 times at its 36th test (test_full_run_cfg4_tecnick, second complete run at 1200 x 1200, C = 256: the round-3 / round-4
# spot), ~100 s in; subsets of up to three of its tests (cfg 4 alone twice; cfg 3 + cfg 4; the Tecnick-size step tests + cfg 4; cfg 1 +
# cfg 3 + cfg 4) passed, and so did two eviction soaks with 1 872 immediate destroys on ONE handle (soak_evictions.py).  So what it
# takes is the HISTORY of the process (35 tests = ~60 handles created and destroyed, ~100 graphs destroyed), not the cfg-4 run.
# Under the default policy (retire; bounded at 256) the same file and the full suite are green (11 full runs over three rounds).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
K=${1:-}
SGA_GRAPH_DROP=destroy timeout 600 python -X faulthandler -m pytest tests/test_gpu_configs.py -q -p no:cacheprovider ${K:+-k "$K"} > gpurun_out/destroy_repro.log 2>&1
echo "rc $?  (139 = SIGSEGV)"; head -12 gpurun_out/destroy_repro.log | cut -c1-200
