#!/bin/bash
# GPU box, round 5: variants of the persistent igdn2.bwd (harness, probe build): a..d = (memory waves, B prefetch depth, A prefetch depth)
cd "$(dirname "$0")"
for v in a b c d; do echo "=== variant $v"; timeout 120 ./igdn_ws_bench_$v.bin 192 8 128 128 20 1 | grep -v "^   ph\|^  wg"; done
