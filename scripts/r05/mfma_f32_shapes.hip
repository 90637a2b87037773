// Micro-benchmark (round 5): does the SHAPE of the f32-input MFMA change what the chip sustains on random operands?
// v_mfma_f32_32x32x2_f32 (64 cycles, 16 accumulator registers) against v_mfma_f32_16x16x4_f32 (32 cycles, 4 accumulator
// registers): same 64 FLOP / clk / SIMD nominal rate; the clock under load is set by the package's power limit.
//   hipcc --offload-arch=gfx950 -O3 scripts/r05/mfma_f32_shapes.hip -o scripts/r05/mfma_f32_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, const float* opnd, int steps, unsigned long long* clk) {
  const int tid = threadIdx.x;
  float a[4], b[4];
  for (int k = 0; k < 4; ++k) { a[k] = opnd[(size_t)(blockIdx.x * 256 + tid) * 8 + k]; b[k] = opnd[(size_t)(blockIdx.x * 256 + tid) * 8 + 4 + k]; }
  float v = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  if constexpr (SHAPE == 32) {
    f32x16 acc[6];
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + j) & 3], b[k], acc[j], 0, 0, 0);
      asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) v += acc[j][r];
  } else {
    f32x4 acc[12];
    for (int j = 0; j < 12; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + j) & 3], b[k], acc[j], 0, 0, 0);
      asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    for (int j = 0; j < 12; ++j) for (int r = 0; r < 4; ++r) v += acc[j][r];
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * 256 + tid] = v;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int blocks_max = 512, n = blocks_max * 256 * 8;
  float* out; float* opnd; unsigned long long* clk;
  hipMalloc(&out, blocks_max * 256 * 4); hipMalloc(&opnd, (size_t)n * 4); hipMalloc(&clk, blocks_max * 16);
  float* h = (float*)malloc((size_t)n * 4);
  unsigned long long* hc = (unsigned long long*)malloc(blocks_max * 16);
  const char* names[3] = {"zeros", "random |x| ~ 1", "random |x| ~ 1e-4 (gradient-like)"};
  for (int shape : {32, 16})
    for (int blocks : {256, 512})
      for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < n; ++i) h[i] = mode == 0 ? 0.f : (mode == 1 ? 1.f : 1e-4f) * (float)(rand() % 2001 - 1000) * 1e-3f;
        hipMemcpy(opnd, h, (size_t)n * 4, hipMemcpyHostToDevice);
        const int steps = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int st) {
          if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, out, opnd, st, clk);
          else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, out, opnd, st, clk);
        };
        launch(steps / 10); hipDeviceSynchronize();
        hipEventRecord(e0); launch(steps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
        double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += hc[2 * i]; w += hc[2 * i + 1]; }
        const double flops = (double)blocks * 4 * steps * (shape == 32 ? 24 * 4096.0 : 48 * 2048.0);
        printf("f32 %s  %d workgroups (%d wave(s)/SIMD) %-34s %8.1f TFLOP/s  %6.0f MHz in the loop\n", shape == 32 ? "32x32x2" : "16x16x4",
               blocks, blocks / 256, names[mode], flops / (ms * 1e-3) / 1e12, w > 0 ? 100.0 * c / w : 0.0);
      }
  return 0;
}
