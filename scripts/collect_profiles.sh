#!/bin/bash
# GPU box: the measurement set committed under profiles/ (run from the repo root: bash scripts/collect_profiles.sh r02 <commit>)
R=${1:-r02}; COMMIT=${2:-unknown}
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/$R; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python scripts/profile_layers.py > $O/layers_hipevents.txt 2>&1
python scripts/perf_configs.py f32 > $O/configs.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -- python $ROOT/bench.py --roofline-only > $O/roofline_leg.json 2>/dev/null )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $ROOT/bench.py --its 200 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile > $O/graph_its200.json 2>/dev/null )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f --output-format csv -- python $ROOT/bench.py --roofline-only > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w --output-format csv -- python $ROOT/bench.py --roofline-only > /dev/null 2>&1 )
python scripts/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json $COMMIT > /dev/null
python scripts/pmc_kernels.py collect $O/pmc_kernels > $O/pmc_kernels.txt 2>&1
cp $O/prof_roofline/*/*kernel_stats.csv $O/roofline_leg_kernel_stats.csv
python scripts/timeline_from_trace.py $(ls $O/prof_graph/*/*kernel_trace.csv) 100 > $O/timeline_iteration.txt 2>&1
cp $O/prof_graph/*/*kernel_stats.csv $O/graph_its200_kernel_stats.csv
rm -rf $O/pmc_f $O/pmc_w $O/pmc_kernels/sq $O/pmc_kernels/lds $O/pmc_kernels/fetch $O/pmc_kernels/write
ls $O
