#!/bin/bash
# GPU box, round 6: defect (a), session 3.  Session 2 ruled out a stack overflow (the crash survives `ulimit -s unlimited`; the default
# policy survives a 1 MiB stack) and showed why the lab handler printed nothing: pytest captures fd 2.  The report goes to a file now.
#  B  destroy policy, lab build, SGA_DEBUG_SEGV=<file>: registers, code at rip, native frames, /proc/self/maps; SGA_CALL_LOG=<file>:
#     every C-ABI call of the process up to the crash (input of a Python-free replay)
#  E  destroy policy with 4 hardware queues (the package's default is 2)
#  F  destroy policy under MALLOC_CHECK_=3 HSA_ENABLE_INTERRUPT=0
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s03; mkdir -p $OUT
T=tests/repro/defect_a_configs_r05_frozen.py      # tests/test_gpu_configs.py as of round 5: the 100-second reproducer
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
run() { local n=$1; shift; ( "$@" ) > $OUT/$n.log 2>&1; echo "=== $n rc $?"; tail -2 $OUT/$n.log | cut -c1-200; }
run B env SGA_LIB=$LAB SGA_DEBUG_SEGV=$OUT/B_segv.txt SGA_CALL_LOG=$OUT/B_calls.txt SGA_GRAPH_DROP=destroy timeout 420 python -m pytest $T -q -x -p no:cacheprovider
head -80 $OUT/B_segv.txt | cut -c1-220
grep -n "amdhip64\|hsa-runtime\|libsga\|\[heap\]\|\[stack\]" $OUT/B_segv.txt | head -40
wc -l $OUT/B_calls.txt; tail -5 $OUT/B_calls.txt
run E env GPU_MAX_HW_QUEUES=4 SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider
run F env MALLOC_CHECK_=3 HSA_ENABLE_INTERRUPT=0 SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider
