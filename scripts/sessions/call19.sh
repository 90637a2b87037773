#!/bin/bash
# GPU box: cfg 2 at B = 1, per-layer times under different split-K targets (laboratory build)
O=gpurun_out/call19; mkdir -p $O
{
for t in 512 384 256 192; do echo "=== main target $t"; LAB=1 B=1 SGA_MAIN_TARGET=$t python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -22; done
for t in 128 256; do echo "=== side target $t"; LAB=1 B=1 SGA_SIDE_TARGET=$t python scripts/profile_layers.py 2>&1 | grep "hs[0-2]"; done
echo "=== combos"
B=1 python scripts/ab_iter.py --rounds 2 "LAB=1" "LAB=1 SGA_MAIN_TARGET=256 SGA_SIDE_TARGET=128" "LAB=1 SGA_MAIN_TARGET=256 SGA_SIDE_TARGET=256" "LAB=1 SGA_MAIN_TARGET=320" "LAB=1 SGA_MAIN_TARGET=288 SGA_SIDE_TARGET=128"
B=2 python scripts/ab_iter.py --rounds 1 "LAB=1" "LAB=1 SGA_MAIN_TARGET=256" "LAB=1 SGA_MAIN_TARGET=384" "LAB=1 SGA_SIDE_TARGET=128"
B=4 python scripts/ab_iter.py --rounds 1 "LAB=1" "LAB=1 SGA_MAIN_TARGET=256" "LAB=1 SGA_MAIN_TARGET=384" "LAB=1 SGA_SIDE_TARGET=128"
} > $O/summary.log 2>&1
tail -5 $O/summary.log
