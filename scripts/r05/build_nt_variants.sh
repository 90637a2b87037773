#!/bin/bash
# builds libsga_hip_ntNN.so for each SGA_NT mask given (objects of the four files that read the macro in /tmp/lib_ntNN, the rest from the product build)
cd "$(dirname "$0")/../../improving-inference-for-neural-image-compression_amd/csrc"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
for nt in "$@"; do
  mkdir -p /tmp/lib_nt$nt
  for f in gdn_fused igdn_bwd_ws conv_mfma deconv3_gemm; do hipcc $FL -DSGA_NT=$nt -c $f.hip -o /tmp/lib_nt$nt/$f.o 2>/tmp/lib_nt$nt/$f.err & done
done
wait
for nt in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/lib_nt$nt/conv_mfma.o /tmp/lib_nt$nt/gdn_fused.o /tmp/lib_nt$nt/igdn_bwd_ws.o deconv3.o /tmp/lib_nt$nt/deconv3_gemm.o elementwise.o msssim.o rans.o sga_api.o -o ../libsga_hip_nt$nt.so && echo built nt$nt
done
