#!/bin/bash
# GPU box, round 6: joint sweep of the placement / variant switches at cfg 2 under the new launch-stream class (laboratory build; the fork
# point pinned where the timed choice puts it at cfg 2, because the sweep's 60-iteration warm-up is too short to time it)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_s15
export SGA_LIB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so SGA_FORK_NAME=gs3.fwd
python scripts/joint_sweep.py '{"SGA_SIDE_TARGET": ["192", "256", "320"], "SGA_MAIN_TARGET": ["384", "512", "640"], "SGA_REDUCE_BATCH": ["1", "2"], "SGA_FUSED_POST64": ["0", "1"], "SGA_GS3_GEMM": ["0", "1"], "SGA_BN96_AS_192": ["0", "1"]}' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_s15/joint_sweep.txt | tail -20
