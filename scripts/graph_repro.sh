#!/bin/bash
# GPU box: the two orchestration defects of DESIGN_EXPERIMENTS.md A.7 / A.3 under tools that make a use-after-free or a
# missing dependency show (see scripts/graph_repro.hip).  Writes gpurun_out/graph_repro/*.log
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/graph_repro; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
ASAN_RT=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so)
step() { echo "=== $*" | tee -a $OUT/summary.log; }

step "1 stand-alone repro, plain"
timeout 300 scripts/graph_repro.bin all > $OUT/repro_plain.log 2>&1; echo "rc $?" >> $OUT/repro_plain.log; tail -8 $OUT/repro_plain.log | tee -a $OUT/summary.log
step "2 stand-alone repro, MALLOC_PERTURB_"
MALLOC_PERTURB_=165 timeout 300 scripts/graph_repro.bin lifetime 96 > $OUT/repro_perturb.log 2>&1; echo "rc $?" >> $OUT/repro_perturb.log; tail -4 $OUT/repro_perturb.log | tee -a $OUT/summary.log
step "3 stand-alone repro, 4 hardware queues"
GPU_MAX_HW_QUEUES=4 MALLOC_PERTURB_=165 timeout 300 scripts/graph_repro.bin all > $OUT/repro_q4.log 2>&1; echo "rc $?" >> $OUT/repro_q4.log; tail -6 $OUT/repro_q4.log | tee -a $OUT/summary.log

step "4 GPU suite (orchestration-heavy files), destroy policy + MALLOC_PERTURB_ + faulthandler"
SGA_GRAPH_DROP=destroy MALLOC_PERTURB_=165 SGA_DEBUG_SEGV=1 timeout 900 python -X faulthandler -m pytest tests/test_gpu_step.py tests/test_gpu_configs.py tests/test_gpu_multi.py tests/test_gpu_bb.py -m gpu -x -q > $OUT/suite_destroy_perturb.log 2>&1
echo "rc $?" >> $OUT/suite_destroy_perturb.log; tail -30 $OUT/suite_destroy_perturb.log | tee -a $OUT/summary.log

step "5 the same files, destroy policy, ASan host build"
SGA_LIB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_asan.so LD_PRELOAD=$ASAN_RT \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:log_path=$PWD/$OUT/asan SGA_GRAPH_DROP=destroy \
  timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_configs.py tests/test_gpu_multi.py -m gpu -x -q > $OUT/suite_asan.log 2>&1
echo "rc $?" >> $OUT/suite_asan.log; tail -15 $OUT/suite_asan.log | tee -a $OUT/summary.log; ls $OUT | tee -a $OUT/summary.log

step "6 bf16x3 with the hyper branch on the second stream: identical runs?"
SGA_X3_FORK=1 timeout 600 python scripts/x3_fork_race.py 20 300 > $OUT/x3_fork_graph.log 2>&1; tail -8 $OUT/x3_fork_graph.log | tee -a $OUT/summary.log
SGA_X3_FORK=1 timeout 600 python scripts/x3_fork_race.py 12 200 eager > $OUT/x3_fork_eager.log 2>&1; tail -8 $OUT/x3_fork_eager.log | tee -a $OUT/summary.log
