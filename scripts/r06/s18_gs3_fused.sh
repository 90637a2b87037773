#!/bin/bash
# GPU box, round 6: VERDICT r5 #4c -- the C -> 3 layer in ONE launch (laboratory, SGA_GS3_FUSED=1): bit-equality with the shipped pair,
# the layer alone (hipEvents per layer), the iteration (A/B)
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s18; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fused.py -q -k "gs3_in_one_launch" 2>&1 | tail -6
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
for f in 0 1; do echo "=== SGA_GS3_FUSED=$f, per layer"; SGA_LIB=$LAB SGA_GS3_FUSED=$f python scripts/profile_layers.py 2>&1 | grep -i "gs3\|total" ; done | tee $OUT/layers.txt
python scripts/ab_iter.py --rounds 4 "LAB=1" "LAB=1 SGA_GS3_FUSED=1" 2>&1 | tee $OUT/ab.txt
