#!/bin/bash
# GPU box: rocprofv3 kernel trace of 120 graph-replayed iterations -> timeline of iteration 60 (gpurun_out/tl/$1.txt)
export TMPDIR=/tmp; R=$(pwd); mkdir -p gpurun_out/tl
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl/tr -- python $R/bench.py --its 120 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile > /dev/null 2>&1 )
python scripts/timeline_from_trace.py $(ls gpurun_out/tl/tr/*/*kernel_trace.csv) 60 > gpurun_out/tl/$1.txt 2>&1
rm -rf gpurun_out/tl/tr
