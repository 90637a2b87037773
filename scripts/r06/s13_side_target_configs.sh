#!/bin/bash
# GPU box, round 6: the hyper branch's split-K target 384 (rounds 3-5) vs 256 at every bench configuration (laboratory build)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_s13
LAB=$PWD/improving-inference-for-neural-image-compression_amd/libsga_hip_lab.so
for k in 1 2; do
  for t in 384 256; do
    echo "=== side target $t (pass $k)"; SGA_LIB=$LAB SGA_SIDE_TARGET=$t python scripts/perf_configs.py f32 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r06_s13/side_target_configs.txt
for t in 384 256; do echo "=== bb, side target $t"; SGA_LIB=$LAB SGA_SIDE_TARGET=$t python scripts/perf_bb.py 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r06_s13/side_target_configs.txt
