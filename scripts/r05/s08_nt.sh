#!/bin/bash
# GPU box, round 5: non-temporal loads (bit 0) / stores (bit 1) in igdn*.bwd, compile-time (libsga_hip_nt{1,2,3}.so), both
# kernels (SGA_IGDN_WS=0 tile kernel, =1 persistent kernel), in the real iteration
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s08; mkdir -p $OUT
L=$(pwd)/improving-inference-for-neural-image-compression_amd
timeout 2000 python scripts/ab_iter.py --rounds 2 "SGA_IGDN_WS=0" "SGA_IGDN_WS=0 SGA_LIB=$L/libsga_hip_nt1.so" "SGA_IGDN_WS=0 SGA_LIB=$L/libsga_hip_nt2.so" "SGA_IGDN_WS=0 SGA_LIB=$L/libsga_hip_nt3.so" "SGA_IGDN_WS=1" "SGA_IGDN_WS=1 SGA_LIB=$L/libsga_hip_nt1.so" "SGA_IGDN_WS=1 SGA_LIB=$L/libsga_hip_nt2.so" "SGA_IGDN_WS=1 SGA_LIB=$L/libsga_hip_nt3.so" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
SGA_LIB=$L/libsga_hip_nt3.so timeout 300 python scripts/profile_layers.py > $OUT/layers_ws_nt3.txt 2>&1; grep "igdn\|gs2.bwd\|gs1.bwd" $OUT/layers_ws_nt3.txt
SGA_IGDN_WS=0 SGA_LIB=$L/libsga_hip_nt3.so timeout 300 python scripts/profile_layers.py > $OUT/layers_tile_nt3.txt 2>&1; grep "igdn\|gs2.bwd\|gs1.bwd" $OUT/layers_tile_nt3.txt
