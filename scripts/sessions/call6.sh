#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call6; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "bf16x3 with the 4-wave 256-row instance (one wave per SIMD) for gs2.bwd"
SGA_LIB=$LAB python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
SGA_LIB=$LAB SGA_X3_W4=1 python scripts/perf_modes.py bf16x3 2>&1 | tail -1 | tee -a $OUT/summary.log
SGA_LIB=$LAB SGA_X3_W4=1 PREC=bf16x3 python scripts/profile_layers.py 2>&1 | grep -E "gs2|gs1" | tee -a $OUT/summary.log
step "crash hunt: the files up to test_gpu_configs, core dump on"
ulimit -c unlimited
cd /tmp && rm -f core core.*
( cd $OLDPWD && timeout 1500 python -X faulthandler -m pytest tests/test_gpu_acceptance.py tests/test_gpu_bb.py tests/test_gpu_c_abi.py tests/test_gpu_checkpoint.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/hunt.log 2>&1; echo "rc $?" >> $OUT/hunt.log )
cd $OLDPWD
tail -5 $OUT/hunt.log | tee -a $OUT/summary.log
CORE=$(ls -t core* /tmp/core* 2>/dev/null | head -1)
if [ -n "$CORE" ]; then
  step "core: $CORE"
  timeout 600 rocgdb -batch -ex "thread apply all bt 25" $(which python3) "$CORE" > $OUT/core_bt.txt 2>&1
  grep -E "^Thread|^#" $OUT/core_bt.txt | head -120 | tee -a $OUT/summary.log
  rm -f "$CORE"
fi
step "FULL suite under xdist (-n 1: a crashed worker is reported and replaced)"
timeout 2400 python -m pytest tests -m gpu -q -n 1 > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -25 $OUT/tests.log | tee -a $OUT/summary.log
