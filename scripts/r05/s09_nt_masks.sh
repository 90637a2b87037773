#!/bin/bash
# GPU box, round 5: SGA_NT masks (sga_common.h) in the real iteration, tile-kernel igdn2.bwd (SGA_IGDN_WS=0)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s09; mkdir -p $OUT
L=$(pwd)/improving-inference-for-neural-image-compression_amd
export SGA_IGDN_WS=0
timeout 2400 python scripts/ab_iter.py --rounds 3 "A=0" "SGA_LIB=$L/libsga_hip_nt15.so" "SGA_LIB=$L/libsga_hip_nt12.so" "SGA_LIB=$L/libsga_hip_nt8.so" "SGA_LIB=$L/libsga_hip_nt31.so" "SGA_LIB=$L/libsga_hip_nt47.so" "SGA_LIB=$L/libsga_hip_nt63.so" > $OUT/ab2.txt 2>&1; cat $OUT/ab2.txt
