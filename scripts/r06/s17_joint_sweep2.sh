#!/bin/bash
# GPU box, round 6: second joint sweep (the launch planner's thresholds and a finer side target) under the new launch-stream class
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_s17
export SGA_LIB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so SGA_FORK_NAME=gs3.fwd
python scripts/joint_sweep.py '{"SGA_SIDE_TARGET": ["224", "256", "288"], "SGA_SMALL_TILES": ["0", "64", "128"], "SGA_BM64_MAX": ["128", "256", "384"], "SGA_SPLIT256": ["0", "1"]}' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_s17/joint_sweep2.txt | tail -14
