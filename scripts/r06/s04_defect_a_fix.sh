#!/bin/bash
# GPU box, round 6: defect (a), session 4 -- the mechanism (scripts/r06/graph_stream_collision_repro.hip) and the fix (the library's own
# high-priority launch stream, sga_handle::sG).
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s04; mkdir -p $OUT
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
R=scripts/r06/graph_stream_collision_repro.bin
sweep() { # tag, priority, extra env...
  local tag=$1 prio=$2; shift 2
  local died=() ok=0
  for m in $(seq 0 63); do
    env MALLOC_PERTURB_=165 GPU_MAX_HW_QUEUES=2 "$@" timeout 30 $R $m $prio >> $OUT/standalone_$tag.log 2>&1; rc=$?
    if [ $rc -eq 0 ]; then ok=$((ok+1)); else died+=("$m:$rc"); fi
  done
  echo "standalone $tag prio $prio: $ok of 64 masks survived; died (mask:rc): ${died[*]:-none}"
}
echo "=== 1 stand-alone program, ROCm 7.0.2 runtime as bundled with PyTorch"
sweep torch702_normal 0 LD_PRELOAD=$TL/libamdhip64.so
sweep torch702_high -1 LD_PRELOAD=$TL/libamdhip64.so
echo "=== ... ROCm 7.2 runtime (/opt/rocm)"
sweep rocm72_normal 0
echo "=== 2 the 100-second reproducer, destroy policy, library with its own launch stream (2 runs)"
T=tests/repro/defect_a_configs_r05_frozen.py
for k in 1 2; do
  SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider --deselect $T::test_graph_cache_selects_instead_of_recapturing --deselect $T::test_graph_cache_eviction_keeps_results > $OUT/repro_fixed_$k.log 2>&1
  echo "fixed run $k rc $?"; tail -1 $OUT/repro_fixed_$k.log
done
echo "=== ... control: the same library build with graphs on the CALLER's stream (lab switch)"
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
SGA_LIB=$LAB SGA_LAUNCH_STREAM=caller SGA_GRAPH_DROP=destroy timeout 420 python -X faulthandler -m pytest $T -q -x -p no:cacheprovider > $OUT/repro_control.log 2>&1
echo "control rc $? (139 = SIGSEGV)"; tail -1 $OUT/repro_control.log | cut -c1-200
echo "=== 3 iteration time, cfg 2: own high-priority launch stream vs the caller's stream vs low priority"
python scripts/ab_iter.py --rounds 2 "" "LAB=1" "LAB=1 SGA_LAUNCH_STREAM=caller" "LAB=1 SGA_LAUNCH_STREAM=low" 2>&1 | tee $OUT/ab_launch_stream.txt
echo "=== 4 Python-free replay of the recorded call sequence (old behaviour: caller's stream, destroy policy)"
SGA_LAUNCH_STREAM=caller SGA_GRAPH_DROP=destroy GPU_MAX_HW_QUEUES=2 LD_PRELOAD=$TL/libamdhip64.so timeout 400 scripts/r06/sga_replay.bin tests/repro/defect_a_calls_r05.txt $LAB > $OUT/replay_torch702.log 2>&1
echo "replay (7.0.2 runtime) rc $?"; tail -2 $OUT/replay_torch702.log | cut -c1-200
SGA_LAUNCH_STREAM=caller SGA_GRAPH_DROP=destroy GPU_MAX_HW_QUEUES=2 timeout 400 scripts/r06/sga_replay.bin tests/repro/defect_a_calls_r05.txt $LAB > $OUT/replay_rocm72.log 2>&1
echo "replay (7.2 runtime) rc $?"; tail -2 $OUT/replay_rocm72.log | cut -c1-200
echo "=== 5 tests touched by the host changes"
timeout 600 python -m pytest tests/test_gpu_c_abi.py tests/test_gpu_bb.py tests/test_gpu_multi.py "tests/test_gpu_entropy.py::test_stream_names_the_precision_mode_of_its_encoder" "tests/test_gpu_configs.py::test_other_size_call_inside_an_open_run_keeps_the_zero_borders" tests/test_gpu_configs.py::test_base_compress_inside_an_open_run tests/test_gpu_configs.py::test_graph_cache_selects_instead_of_recapturing -q -x -p no:cacheprovider 2>&1 | tail -4
