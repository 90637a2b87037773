#!/bin/bash
# GPU box, round 5: the persistent igdn2.bwd (8 memory waves, PF 4, AQ 2) in the real iteration: alone-time by layer, A/B of graph replays
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s06; mkdir -p $OUT
timeout 300 python scripts/profile_layers.py > $OUT/layers_ws.txt 2>&1; grep "igdn2\|total" $OUT/layers_ws.txt
SGA_IGDN_WS=0 timeout 300 python scripts/profile_layers.py > $OUT/layers_tile.txt 2>&1; grep "igdn2\|total" $OUT/layers_tile.txt
timeout 900 python scripts/ab_iter.py --rounds 2 "SGA_IGDN_WS=0" "SGA_IGDN_WS=1" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
