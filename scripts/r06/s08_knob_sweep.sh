#!/bin/bash
# GPU box, round 6: does the new launch-stream class move any of the round-3..5 placement optima?  cfg 2, us per iteration
# (scripts/ab_iter.py: separate processes, alternating, best of 3 x 400 replays), laboratory build.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_s08
python scripts/ab_iter.py --rounds 2 "" "LAB=1" "LAB=1 GPU_MAX_HW_QUEUES=3" "LAB=1 GPU_MAX_HW_QUEUES=4" "LAB=1 SGA_SIDE_TARGET=256" "LAB=1 SGA_SIDE_TARGET=512" \
   "LAB=1 SGA_REDUCE_BATCH=1" "LAB=1 SGA_REDUCE_BATCH=0" "LAB=1 SGA_FORK_NAME=gs2.fwd" "LAB=1 SGA_FORK_NAME=start" "LAB=1 SGA_SIDE_LAST=0" 2>&1 | tee gpurun_out/r06_s08/knob_sweep.txt
