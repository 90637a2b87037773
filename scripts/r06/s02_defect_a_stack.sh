#!/bin/bash
# GPU box, round 6: defect (a) -- is the host SIGSEGV a STACK OVERFLOW inside the HIP runtime?  (Session 1: the crash reproduces with the
# lab build, 3 of 3, but its SIGSEGV handler -- on the faulting stack -- printed nothing; under rocgdb and with ROCm 7.2's runtime
# preloaded the same file ran through.)
#  A  destroy policy, unlimited main-thread stack            -> expected to PASS if the fault is the stack
#  B  destroy policy, the lab build's handler on its own stack: fault address vs rsp, depth, innermost / outermost frames
#  C  RETIRE (default) policy with a 1 MiB stack              -> does the default only survive by margin?
#  D  destroy policy, 2 MiB stack                             -> expected to crash earlier
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06_s02; mkdir -p $OUT
T=tests/test_gpu_configs.py
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
run() { # name, then the command
  local n=$1; shift
  ( "$@" ) > $OUT/$n.log 2>&1; local rc=$?
  echo "=== $n rc $rc"; grep -c PASSED $OUT/$n.log > /dev/null; tail -3 $OUT/$n.log | cut -c1-240
}
run A bash -c "ulimit -s unlimited; SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider --deselect $T::test_graph_cache_selects_instead_of_recapturing --deselect $T::test_graph_cache_eviction_keeps_results"
run B bash -c "SGA_LIB=$LAB SGA_DEBUG_SEGV=1 SGA_GRAPH_DROP=destroy timeout 420 python -m pytest $T -q -x -p no:cacheprovider"
grep -n "sga: fatal" -A100 $OUT/B.log | cut -c1-200 | head -120
run C bash -c "ulimit -s 1024; timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider"
run D bash -c "ulimit -s 2048; SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -v -m pytest $T -x -p no:cacheprovider"
grep -n "Fatal Python\|Segmentation" -B3 -A12 $OUT/D.log | cut -c1-200 | head -60
