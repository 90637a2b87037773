#!/bin/bash
# GPU box, round 6: the GPU suite once more, under freed-memory poisoning (MALLOC_PERTURB_: a use-after-free or an out-of-bounds read of
# freed / uninitialised host memory faults instead of "depending on the heap layout") -- second of the three green runs asked for
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s10; mkdir -p $OUT
MALLOC_PERTURB_=165 timeout 3000 python -X faulthandler -m pytest tests/ -q -m gpu > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -12 $OUT/tests.log
