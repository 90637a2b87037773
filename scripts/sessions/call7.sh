#!/bin/bash
# crash hunt (DESIGN_EXPERIMENTS.md A.8): the test files up to test_gpu_configs in one process, destroy policy, core dumps on,
# repeated; a core is opened with rocgdb for the native backtrace of every thread.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/call7; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
ulimit -c unlimited
echo "core_pattern: $(cat /proc/sys/kernel/core_pattern)" | tee -a $OUT/summary.log
for i in 1 2 3 4; do
  rm -f $ROOT/core* /tmp/core*
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_acceptance.py tests/test_gpu_bb.py tests/test_gpu_c_abi.py tests/test_gpu_checkpoint.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/hunt$i.log 2>&1
  rc=$?
  echo "attempt $i rc $rc: $(tail -1 $OUT/hunt$i.log)" | tee -a $OUT/summary.log
  if [ $rc -ge 128 ] || grep -q "Segmentation fault" $OUT/hunt$i.log; then
    grep -n "Current thread" -A 6 $OUT/hunt$i.log | head -12 | tee -a $OUT/summary.log
    CORE=$(ls -t $ROOT/core* /tmp/core* 2>/dev/null | head -1)
    echo "core file: ${CORE:-none}" | tee -a $OUT/summary.log
    if [ -n "${CORE:-}" ]; then
      timeout 900 rocgdb -batch -ex "thread apply all bt 30" $(readlink -f $(which python3)) "$CORE" > $OUT/core_bt$i.txt 2>&1
      grep -E "^Thread|^#" $OUT/core_bt$i.txt | head -150 | tee -a $OUT/summary.log
      rm -f "$CORE"
    fi
    break
  fi
done
