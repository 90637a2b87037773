// Measurement only (scripts/coresident_probe.py): a register-only MFMA "filler" kernel with a small footprint (4 waves,
// 32 KB LDS, < 64 VGPRs) that can be co-resident with one 8-wave convolution workgroup (117 KB LDS, 2 x 176 VGPRs per SIMD).
// Question: how much matrix-pipe time does a co-resident low-footprint wave find in the bubbles of the big K loop?
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void filler_kernel(float* out, int n, int prio) {
  __shared__ float lds[8192];      // 32 KB: at most one filler workgroup fits beside a 117 KB convolution workgroup
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float a = 1.0f + 1e-3f * (threadIdx.x & 63), b = 0.5f + 1e-3f * (threadIdx.x >> 2);
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += acc[k][r];
  lds[threadIdx.x] = s;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[(threadIdx.x + 1) & 255];
}

extern "C" int filler_launch(float* out, int grid, int n, int prio, void* stream) {
  hipLaunchKernelGGL(filler_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n, prio);
  return (int)hipGetLastError();
}
