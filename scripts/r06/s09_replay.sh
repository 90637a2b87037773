#!/bin/bash
# GPU box, round 6: the Python-free replay, done properly -- (1) record the C-ABI call log of the frozen reproducer with the fixed logger
# (h<n> = n-th sga_create) under the OLD behaviour (lab build, graphs on the caller's stream): the process dies at its 36th test, the log
# ends in the fatal call; (2) replay that log with tests/c_client/sga_replay.cpp -- no Python, no PyTorch in the process -- on the 7.0.2
# runtime PyTorch bundles and on ROCm 7.2's.
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s09; mkdir -p $OUT
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
T=tests/repro/defect_a_configs_r05_frozen.py
rm -f $OUT/calls.txt
SGA_LIB=$LAB SGA_LAUNCH_STREAM=caller SGA_CALL_LOG=$OUT/calls.txt timeout 420 python -m pytest $T -q -x -p no:cacheprovider > $OUT/record.log 2>&1; echo "record rc $? (139 expected)"
wc -l $OUT/calls.txt; tail -2 $OUT/calls.txt
for k in 1 2; do
  SGA_LAUNCH_STREAM=caller GPU_MAX_HW_QUEUES=2 MALLOC_PERTURB_=165 LD_PRELOAD=$TL/libamdhip64.so timeout 500 scripts/r06/sga_replay.bin $OUT/calls.txt $LAB > $OUT/replay_torch702_$k.log 2>&1
  echo "replay $k (7.0.2 runtime, no Python) rc $?"; tail -1 $OUT/replay_torch702_$k.log | cut -c1-200
done
SGA_LAUNCH_STREAM=caller GPU_MAX_HW_QUEUES=2 MALLOC_PERTURB_=165 timeout 500 scripts/r06/sga_replay.bin $OUT/calls.txt $LAB > $OUT/replay_rocm72.log 2>&1
echo "replay (7.2 runtime) rc $?"; tail -1 $OUT/replay_rocm72.log | cut -c1-200
GPU_MAX_HW_QUEUES=2 MALLOC_PERTURB_=165 LD_PRELOAD=$TL/libamdhip64.so timeout 500 scripts/r06/sga_replay.bin $OUT/calls.txt > $OUT/replay_shipped.log 2>&1
echo "replay, shipped library (7.0.2 runtime) rc $?"; tail -1 $OUT/replay_shipped.log | cut -c1-200
