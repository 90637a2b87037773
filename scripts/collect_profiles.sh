#!/bin/bash
# GPU box: the measurement set committed under profiles/ (run from the repo root: bash scripts/collect_profiles.sh r03 <commit>)
# Order: the PMC passes first, then the benchmark line and the roofline leg, so that bench.py finds THIS run's
# rNN_pmc_traffic.json (copied into profiles/ right away) and one commit id appears in every file of the set.
R=${1:-r04}; COMMIT=${2:-unknown}
export TMPDIR=/tmp
ROOT=$(pwd); O=$ROOT/gpurun_out/$R; mkdir -p $O
# -- HBM traffic per kernel symbol: eager roofline leg with the hyper branch on its second stream (as in the run) and
#    single-stream (SGA_NO_OVERLAP=1: nothing else competes for L2 / MALL while the kernel runs)
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f --output-format csv -- python $ROOT/bench.py --roofline-only > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w --output-format csv -- python $ROOT/bench.py --roofline-only > /dev/null 2>&1 )
( cd /tmp && SGA_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f1 --output-format csv -- python $ROOT/bench.py --roofline-only > /dev/null 2>&1 )
( cd /tmp && SGA_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w1 --output-format csv -- python $ROOT/bench.py --roofline-only > /dev/null 2>&1 )
python scripts/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json $COMMIT $O/pmc_f1 $O/pmc_w1 > /dev/null
cp $O/pmc_traffic.json $ROOT/profiles/${R}_pmc_traffic.json
python scripts/pmc_kernels.py collect $O/pmc_kernels > $O/pmc_kernels.txt 2>&1
# -- the benchmark line, layers, configs
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python scripts/profile_layers.py > $O/layers_hipevents.txt 2>&1
python scripts/perf_configs.py f32 > $O/configs.txt 2>&1
# -- rocprofv3 kernel stats of the roofline leg (eager) and of the graph replay that the headline times
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roofline -- python $ROOT/bench.py --roofline-only > $O/roofline_leg.json 2>/dev/null )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python $ROOT/bench.py --its 200 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-other-input --no-alt-precision --no-other-configs > $O/graph_its200.json 2>/dev/null )
cp $O/prof_roofline/*/*kernel_stats.csv $O/roofline_leg_kernel_stats.csv
python scripts/timeline_from_trace.py $(ls $O/prof_graph/*/*kernel_trace.csv) 100 > $O/timeline_iteration.txt 2>&1
cp $O/prof_graph/*/*kernel_stats.csv $O/graph_its200_kernel_stats.csv
# -- the secondary precision mode (bf16x3, bench.py's `alt_precision`): layers, configs, kernel stats + timeline of the graph replay
PREC=bf16x3 python scripts/profile_layers.py > $O/layers_hipevents_bf16x3.txt 2>&1
python scripts/perf_configs.py bf16x3 > $O/configs_bf16x3.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph_x3 -- python $ROOT/bench.py --precision bf16x3 --its 200 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-other-input --no-alt-precision --no-other-configs > $O/graph_its200_bf16x3.json 2>/dev/null )
python scripts/timeline_from_trace.py $(ls $O/prof_graph_x3/*/*kernel_trace.csv) 100 > $O/timeline_iteration_bf16x3.txt 2>&1
cp $O/prof_graph_x3/*/*kernel_stats.csv $O/graph_its200_bf16x3_kernel_stats.csv
B=1 python scripts/profile_layers.py > $O/layers_hipevents_b1.txt 2>&1
# -- the fast precision mode (bf16x2, bench.py's `fast_precision`): layers, configs, kernel stats of the graph replay
PREC=bf16x2 python scripts/profile_layers.py > $O/layers_hipevents_bf16x2.txt 2>&1
python scripts/perf_configs.py bf16x2 > $O/configs_bf16x2.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph_x2 -- python $ROOT/bench.py --precision bf16x2 --its 200 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-other-input --no-alt-precision --no-other-configs > $O/graph_its200_bf16x2.json 2>/dev/null )
python scripts/timeline_from_trace.py $(ls $O/prof_graph_x2/*/*kernel_trace.csv) 100 > $O/timeline_iteration_bf16x2.txt 2>&1
cp $O/prof_graph_x2/*/*kernel_stats.csv $O/graph_its200_bf16x2_kernel_stats.csv
rm -rf $O/prof_graph_x2
rm -rf $O/prof_graph_x3
rm -rf $O/pmc_f $O/pmc_w $O/pmc_f1 $O/pmc_w1 $O/pmc_kernels/sq $O/pmc_kernels/lds $O/pmc_kernels/fetch $O/pmc_kernels/write
ls $O
