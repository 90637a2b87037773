#!/bin/bash
# GPU box, round 5: LDS pitch of the col2im kernel (80 -> 76 / 84) in the iteration and alone
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s11; mkdir -p $OUT
L=$(pwd)/improving-inference-for-neural-image-compression_amd
timeout 1200 python scripts/ab_iter.py --rounds 3 "SGA_LIB=$L/libsga_hip_p80.so" "SGA_LIB=$L/libsga_hip_p76.so" "SGA_LIB=$L/libsga_hip_p84.so" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
for p in 80 76 84; do SGA_LIB=$L/libsga_hip_p$p.so timeout 300 python scripts/profile_layers.py 2>&1 | grep "gs3.fwd"; done
cd /tmp; export TMPDIR=/tmp
for p in 80 76; do SGA_LIB=$L/libsga_hip_p$p.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p$p -- python $OLDPWD/bench.py --roofline-only > /dev/null 2>&1; grep "col2im\|deconv3_gemm" /tmp/prof_p$p/*/*kernel_stats.csv | cut -c1-160; done
