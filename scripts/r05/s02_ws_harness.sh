#!/bin/bash
# GPU box, round 5 session 2: the stand-alone harness (scripts/r05/igdn_ws_bench.hip): equality at several shapes and both
# schedules, then the per-phase stamps at the bench shape
cd "$(dirname "$0")"
for a in "192 8 128 128 30 1" "192 8 128 128 30 0" "128 2 256 256 20 1" "128 2 256 256 20 0" "128 2 256 256 20 1" "64 2 256 256 20 1" "192 3 256 384 10 1"; do
  timeout 120 ./igdn_ws_bench.bin $a
done
timeout 120 ./igdn_ws_bench_probe.bin 192 8 128 128 10 1
