#!/bin/bash
# GPU box, second pass (see scripts/graph_repro.sh): in-flight destroy, side-stream root node, fork-point variants of the
# bf16x3 two-stream mode, the FULL suite under the destroy policy.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/graph_repro2; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=2
step() { echo "=== $*" | tee -a $OUT/summary.log; }
step "1 side-stream ROOT node reads the previous replay's tail (stand-alone)"
timeout 300 scripts/graph_repro.bin rootfork 0 > $OUT/rootfork.log 2>&1; echo "rc $?" >> $OUT/rootfork.log; tail -5 $OUT/rootfork.log | tee -a $OUT/summary.log
step "1b the same with a head node before the fork"
timeout 300 scripts/graph_repro.bin rootfork 1 > $OUT/rootfork_head.log 2>&1; echo "rc $?" >> $OUT/rootfork_head.log; tail -5 $OUT/rootfork_head.log | tee -a $OUT/summary.log
step "2 bf16x3 two-stream, fork AFTER the first main-chain launch (SGA_FORK_AT=1)"
SGA_X3_FORK=1 SGA_FORK_AT=1 timeout 600 python scripts/x3_fork_race.py 20 300 > $OUT/x3_fork_at1.log 2>&1; tail -6 $OUT/x3_fork_at1.log | tee -a $OUT/summary.log
step "3 f32 two-stream with the fork pinned at the start (root node), 20 x 300"
SGA_FORK_NAME=start timeout 600 python scripts/x3_fork_race.py 20 300 > $OUT/f32_fork_start.log 2>&1; tail -6 $OUT/f32_fork_start.log | tee -a $OUT/summary.log
step "4 hipGraphExecDestroy with replays in flight (stand-alone)"
MALLOC_PERTURB_=165 timeout 300 scripts/graph_repro.bin inflight 50 40 > $OUT/inflight.log 2>&1; echo "rc $?" >> $OUT/inflight.log; tail -5 $OUT/inflight.log | tee -a $OUT/summary.log
step "5 FULL GPU suite, destroy policy + MALLOC_PERTURB_"
SGA_GRAPH_DROP=destroy MALLOC_PERTURB_=165 SGA_DEBUG_SEGV=1 timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/suite_full_destroy.log 2>&1
echo "rc $?" >> $OUT/suite_full_destroy.log; tail -12 $OUT/suite_full_destroy.log | tee -a $OUT/summary.log
