#!/bin/bash
# GPU box: cfg 2 at B = 1 -- the launch-plan knobs of the laboratory build on the iteration time, and one iteration's timeline
export TMPDIR=/tmp
O=gpurun_out/call18; mkdir -p $O
{
echo "=== B=1 sweep (us per iteration, 2 rounds)"
B=1 python scripts/ab_iter.py --rounds 2 "LAB=1" "LAB=1 SGA_MAIN_TARGET=128" "LAB=1 SGA_MAIN_TARGET=256" "LAB=1 SGA_MAIN_TARGET=384" "LAB=1 SGA_MAIN_TARGET=768" "LAB=1 SGA_MAIN_TARGET=1024" \
  "LAB=1 SGA_SIDE_TARGET=128" "LAB=1 SGA_SIDE_TARGET=256" "LAB=1 SGA_NO_SPLITK=1" "LAB=1 SGA_NO_OVERLAP=1" "LAB=1 SGA_GS3_GEMM=0" "LAB=1 SGA_FUSED_POST64=0" "LAB=1 SGA_BM64_MAX=0" "LAB=1 SGA_REDUCE_BATCH=0" "LAB=1 SGA_FUSED_GDN=0"
echo "=== B=1 timeline"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_b1 -- python $GRAFT_REPO_ROOT/bench.py --batch 1 --its 200 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-other-input --no-alt-precision --no-other-configs > /dev/null 2>&1 )
python scripts/timeline_from_trace.py $(ls $O/prof_b1/*/*kernel_trace.csv) 100
rm -rf $O/prof_b1
} > $O/summary.log 2>&1
tail -60 $O/summary.log
