#!/bin/bash
# GPU box, round 5: loader-wave form of the 64-row convolution instance (lab, SGA_DEEP64_KIND=2; SGA_DEEP64: 1 main chain, 2 hyper branch)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s15; mkdir -p $OUT
timeout 600 python scripts/r05/loader_check.py 2>&1 | grep -v amdgpu.ids > $OUT/check.txt; cat $OUT/check.txt
export SGA_DEEP64_KIND=2
timeout 1500 python scripts/ab_iter.py --rounds 2 "LAB=1 SGA_DEEP64=0" "LAB=1 SGA_DEEP64=1" "LAB=1 SGA_DEEP64=2" "LAB=1 SGA_DEEP64=3" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
B=1 timeout 900 python scripts/ab_iter.py --rounds 2 "LAB=1 SGA_DEEP64=0" "LAB=1 SGA_DEEP64=3" > $OUT/ab_b1.txt 2>&1; cat $OUT/ab_b1.txt
LAB=1 SGA_DEEP64=3 timeout 300 python scripts/profile_layers.py > $OUT/layers.txt 2>&1; grep "gs0\|gs1\|hs" $OUT/layers.txt
