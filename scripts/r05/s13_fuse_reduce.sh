#!/bin/bash
# GPU box, round 5: split-K slab sums inside the convolution launch: bit-equality with the reduce launches, then the iteration time
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s13; mkdir -p $OUT
timeout 900 python scripts/r05/fuse_reduce_equal.py > $OUT/equal.txt 2>&1; grep -v amdgpu.ids $OUT/equal.txt
timeout 1500 python scripts/ab_iter.py --rounds 3 "LAB=1 SGA_FUSE_REDUCE=0" "LAB=1 SGA_FUSE_REDUCE=1" "LAB=1 SGA_FUSE_REDUCE=2" "LAB=1 SGA_FUSE_REDUCE=3" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
B=1 timeout 900 python scripts/ab_iter.py --rounds 2 "LAB=1 SGA_FUSE_REDUCE=0" "LAB=1 SGA_FUSE_REDUCE=3" > $OUT/ab_b1.txt 2>&1; cat $OUT/ab_b1.txt
