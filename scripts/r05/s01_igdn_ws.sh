#!/bin/bash
# GPU box, round 5 session 1: the persistent wave-specialised igdn2.bwd (csrc/igdn_bwd_ws.hip): bit-equality with the tile
# kernel, its time alone (per-layer hipEvents) and in the iteration (A/B of graph replays)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s01; mkdir -p $OUT
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_fused.py -x -q -k "persistent" > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -15 $OUT/tests.log
timeout 300 python scripts/profile_layers.py > $OUT/layers_ws.txt 2>&1; head -12 $OUT/layers_ws.txt
SGA_IGDN_WS=0 timeout 300 python scripts/profile_layers.py > $OUT/layers_tile.txt 2>&1; grep igdn2 $OUT/layers_tile.txt
timeout 900 python scripts/ab_iter.py --rounds 2 "SGA_IGDN_WS=0" "SGA_IGDN_WS=1" "SGA_IGDN_WS=1 SGA_IGDN_WS_SCHED=static" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
