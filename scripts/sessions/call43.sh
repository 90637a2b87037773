#!/bin/bash
O=gpurun_out/call43; mkdir -p $O
{
echo "=== small-grid split target (<= 64 tiles -> 256 workgroups): us per iteration, lab build, rule on (default) / off"
for b in 1 2 3 4 8; do echo "--- B=$b 256x256"; B=$b python scripts/ab_iter.py --rounds 2 "LAB=1" "LAB=1 SGA_SMALL_TILES=0"; done
echo "--- Kodak 1 x 512x768"; B=1 H=512 W=768 python scripts/ab_iter.py --rounds 2 "LAB=1" "LAB=1 SGA_SMALL_TILES=0"
echo "--- 2 x 128x128"; B=2 H=128 W=128 python scripts/ab_iter.py --rounds 2 "LAB=1" "LAB=1 SGA_SMALL_TILES=0"
echo "--- C=256 1 x 448x320"; C=256 B=1 H=448 W=320 python scripts/ab_iter.py --rounds 1 "LAB=1" "LAB=1 SGA_SMALL_TILES=0"
} > $O/summary.log 2>&1
