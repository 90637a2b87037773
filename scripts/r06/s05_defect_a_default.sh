#!/bin/bash
# GPU box, round 6: defect (a), session 5 -- the shipped form: LOW-priority own launch stream, mid-life destroy (retire list deleted).
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s05; mkdir -p $OUT
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
R=scripts/r06/graph_stream_collision_repro.bin
echo "=== 1 stand-alone program, 7.0.2 runtime, LOW-priority launch stream"
ok=0; died=""
for m in $(seq 0 63); do
  env MALLOC_PERTURB_=165 GPU_MAX_HW_QUEUES=2 LD_PRELOAD=$TL/libamdhip64.so timeout 30 $R $m 1 >> $OUT/standalone_torch702_low.log 2>&1 && ok=$((ok+1)) || died="$died $m"
done
echo "standalone torch702 prio +1 (low): $ok of 64 masks survived; died:${died:- none}"
echo "=== 2 in-process harness (PyTorch + libsga_hip): shipped library, then the control (graphs on the caller's stream)"
MALLOC_PERTURB_=165 timeout 600 python tests/tools/graph_stream_stress.py > $OUT/stress_fixed.log 2>&1; echo "fixed rc $?"; tail -1 $OUT/stress_fixed.log
MALLOC_PERTURB_=165 timeout 600 python tests/tools/graph_stream_stress.py --control > $OUT/stress_control.log 2>&1; echo "control rc $? (139 = SIGSEGV expected on this runtime)"; tail -2 $OUT/stress_control.log | cut -c1-160
echo "=== 3 the frozen 100-second reproducer with the shipped library"
T=tests/repro/defect_a_configs_r05_frozen.py
timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider --deselect $T::test_graph_cache_selects_instead_of_recapturing --deselect $T::test_graph_cache_eviction_keeps_results > $OUT/repro_shipped.log 2>&1; echo "rc $?"; tail -1 $OUT/repro_shipped.log
echo "=== 4 iteration time, cfg 2"
python scripts/ab_iter.py --rounds 3 "" "LAB=1 SGA_LAUNCH_STREAM=caller" "LAB=1 SGA_LAUNCH_STREAM=high" 2>&1 | tee $OUT/ab_launch_stream.txt
echo "=== 5 the GPU suite, smoke, bench"
timeout 2400 python -X faulthandler -m pytest tests/ -x -q -m gpu --durations=25 > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -40 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json
cp gpurun_out/acceptance_real_size_*.json $OUT/ 2>/dev/null
