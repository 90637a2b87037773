#!/bin/bash
# crash hunt 2 (A.8a): FULL collection (every test module imported), the tests up to the crashing one, destroy policy, the lab
# build with its SIGSEGV handler (native frames through backtrace_symbols_fd)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call15; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
for i in 1 2 3 4; do
  SGA_LIB=$LAB SGA_DEBUG_SEGV=1 SGA_GRAPH_DROP=destroy timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x \
    -k "acceptance or test_gpu_bb or c_abi or checkpoint or test_gpu_configs" > $OUT/hunt$i.log 2>&1
  rc=$?
  echo "attempt $i rc $rc: $(tail -1 $OUT/hunt$i.log | cut -c1-200)" | tee -a $OUT/summary.log
  if grep -q "native backtrace\|Segmentation fault" $OUT/hunt$i.log; then
    grep -n "native backtrace" -A 40 $OUT/hunt$i.log | head -60 | tee -a $OUT/summary.log
    grep -n "Current thread" -A 5 $OUT/hunt$i.log | head -8 | tee -a $OUT/summary.log
    cat /proc/self/maps > /dev/null
    break
  fi
done
