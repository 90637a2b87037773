#!/bin/bash
# GPU box, round 6: defect (a) -- native backtraces of the 100-second reproducer (tests/test_gpu_configs.py under SGA_GRAPH_DROP=destroy).
#  1. the lab build's own SIGSEGV handler (backtrace_symbols_fd)        2. the same run inside rocgdb (all threads)
#  3. the same run with ROCm 7.2's libamdhip64 preloaded in place of the 7.0.2 runtime that PyTorch bundles
set -u
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/r06_s01; mkdir -p $OUT
T=tests/test_gpu_configs.py
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
echo "=== 1 lab handler"
SGA_LIB=$LAB SGA_DEBUG_SEGV=1 SGA_GRAPH_DROP=destroy AMD_LOG_LEVEL=1 timeout 420 python -m pytest $T -q -x -p no:cacheprovider > $OUT/run1.log 2>&1; echo "rc $?" | tee -a $OUT/run1.log
grep -n "native backtrace" -A40 $OUT/run1.log | head -60
echo "=== 2 rocgdb"
SGA_GRAPH_DROP=destroy timeout 600 rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop print" -ex run -ex "bt 40" -ex "info registers rip rsp rax rdi rsi" -ex "x/8i \$pc" \
  -ex "info sharedlibrary" -ex "thread apply all bt 14" --args python -m pytest $T -q -x -p no:cacheprovider > $OUT/run2.log 2>&1; echo "rc $?" | tee -a $OUT/run2.log
grep -n "SIGSEGV" -A45 $OUT/run2.log | head -80
echo "=== 3 ROCm 7.2 runtime preloaded"
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7 SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider > $OUT/run3.log 2>&1; echo "rc $?" | tee -a $OUT/run3.log
tail -5 $OUT/run3.log | cut -c1-300
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so.7 python -c "import torch,sga_amd; from sga_amd import _lib; _lib.load_library(); print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l][::6])" 2>&1 | tail -2
