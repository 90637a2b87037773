#!/bin/bash
# builds scripts/r05/igdn_ws_bench[suffix] from the harness + the two kernel files; usage: build_igdn_ws_bench.sh <suffix> [-D...]
cd "$(dirname "$0")/../.."
SUF=$1; shift
CS=improving-inference-for-neural-image-compression_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -I$CS -DSGA_EXPERIMENTS=1 $*"
mkdir -p /tmp/wsb$SUF
hipcc $FL -c $CS/gdn_fused.hip -o /tmp/wsb$SUF/gdn.o 2>/tmp/wsb$SUF/err_$RANDOM.log &
hipcc $FL -c $CS/igdn_bwd_ws.hip -o /tmp/wsb$SUF/ws.o 2>/tmp/wsb$SUF/err_$RANDOM.log &
hipcc $FL -c scripts/r05/igdn_ws_bench.hip -o /tmp/wsb$SUF/b.o 2>/tmp/wsb$SUF/err_$RANDOM.log &
wait
rm -f scripts/r05/igdn_ws_bench$SUF.bin; hipcc --offload-arch=gfx950 /tmp/wsb$SUF/gdn.o /tmp/wsb$SUF/ws.o /tmp/wsb$SUF/b.o -o scripts/r05/igdn_ws_bench$SUF.bin && echo built scripts/r05/igdn_ws_bench$SUF.bin
