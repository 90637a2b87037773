// Micro-benchmark: the rate v_mfma_f32_32x32x16_bf16 SUSTAINS on this chip (register-only loop, 6 independent accumulators per
// wave, 1 or 2 waves per SIMD), for zero / constant / random operands -- the ceiling of the bf16x3 precision mode (6 such
// MFMAs replace 8 v_mfma_f32_32x32x2_f32: 2.67x the f32 matrix rate at EQUAL clocks; the clock under dense bf16 MFMAs is
// what this measures).  Issue-bound: 32 cycles per MFMA per SIMD at the nominal rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_clock_bf16.hip -o scripts/mfma_clock_bf16.bin && scripts/mfma_clock_bf16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF16>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, const unsigned* opnd, int steps, unsigned long long* clk) {
  const int tid = threadIdx.x;
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  u32x4 a[4], b[4];
  for (int k = 0; k < 4; ++k) {
    a[k] = *reinterpret_cast<const u32x4*>(opnd + ((size_t)(blockIdx.x * 256 + tid) * 8 + k) * 4);
    b[k] = *reinterpret_cast<const u32x4*>(opnd + ((size_t)(blockIdx.x * 256 + tid) * 8 + 4 + k) * 4);
  }
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if constexpr (BF16)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + j) & 3]), __builtin_bit_cast(bf16x8, b[k]), acc[j], 0, 0, 0);
        else
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[(k + j) & 3].x), __builtin_bit_cast(float, b[k].x), acc[j], 0, 0, 0);
      }
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float v = 0.f;
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) v += acc[j][r];
  out[blockIdx.x * 256 + tid] = v;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int blocks_max = 512, n = blocks_max * 256 * 32;
  float* out; unsigned* opnd; unsigned long long* clk;
  hipMalloc(&out, blocks_max * 256 * 4); hipMalloc(&opnd, (size_t)n * 4); hipMalloc(&clk, blocks_max * 16);
  unsigned* h = (unsigned*)malloc((size_t)n * 4);
  unsigned long long* hc = (unsigned long long*)malloc(blocks_max * 16);
  const char* names[3] = {"zeros", "constant 1.5", "random (finite, |x| ~ 1)"};
  for (int bf = 1; bf >= 0; --bf)
    for (int blocks : {256, 512})
      for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < n; ++i) {
          if (mode == 0) h[i] = 0;
          else if (mode == 1) h[i] = bf ? 0x3fc03fc0u : 0x3fc00000u;
          else if (bf) { unsigned lo = (rand() & 0x807f) | ((120 + rand() % 8) << 7), hi = (rand() & 0x807f) | ((120 + rand() % 8) << 7); h[i] = lo | (hi << 16); }
          else { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); h[i] = (u & 0x807fffffu) | ((unsigned)(120 + rand() % 8) << 23); }
        }
        hipMemcpy(opnd, h, (size_t)n * 4, hipMemcpyHostToDevice);
        const int steps = bf ? 40000 : 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&](int st) {
          if (bf) hipLaunchKernelGGL(mfma_loop<true>, dim3(blocks), dim3(256), 0, 0, out, opnd, st, clk);
          else hipLaunchKernelGGL(mfma_loop<false>, dim3(blocks), dim3(256), 0, 0, out, opnd, st, clk);
        };
        launch(steps / 10); hipDeviceSynchronize();
        hipEventRecord(e0); launch(steps); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
        double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += hc[2 * i]; w += hc[2 * i + 1]; }
        const double flops = (double)blocks * 4 * steps * 24 * (bf ? 32768.0 : 4096.0);
        printf("%s %d workgroups (%d wave(s)/SIMD) %-26s %8.1f TFLOP/s  %6.0f MHz in the loop  (%.2f ms)\n", bf ? "bf16 32x32x16" : "f32  32x32x2 ",
               blocks, blocks / 256, names[mode], flops / (ms * 1e-3) / 1e12, w > 0 ? 100.0 * c / w : 0.0, ms);
      }
  return 0;
}
