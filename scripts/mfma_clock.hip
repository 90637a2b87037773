// Micro-benchmark: does the fp32 MFMA rate depend on the operand DATA (i.e. on power)?
// A register-only loop of v_mfma_f32_32x32x2_f32 (6 independent accumulators per wave, no LDS, no global
// memory inside the loop) is issue-bound at 64 cycles per MFMA per SIMD, so its TFLOP/s is a direct reading of
// the shader clock the chip sustains under that load.  Operand sets: zeros, one constant, random floats.
// Also read: s_memtime / s_memrealtime deltas of wave 0 of each workgroup (reported as a ratio).
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_clock.hip -o scripts/mfma_clock.bin && scripts/mfma_clock.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, const float* opnd, int steps, unsigned long long* clk) {
  const int tid = threadIdx.x;
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {
    a[k] = opnd[(blockIdx.x * 256 + tid) * 16 + k];
    b[k] = opnd[(blockIdx.x * 256 + tid) * 16 + 8 + k];
  }
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + j) & 7], b[k], acc[j], 0, 0, 0);
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float v = 0.f;
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) v += acc[j][r];
  out[blockIdx.x * 256 + tid] = v;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int blocks_max = 512, n = blocks_max * 256 * 16;
  float *out, *opnd; unsigned long long* clk;
  hipMalloc(&out, blocks_max * 256 * 4); hipMalloc(&opnd, n * 4); hipMalloc(&clk, blocks_max * 16);
  float* h = (float*)malloc(n * 4);
  unsigned long long* hc = (unsigned long long*)malloc(blocks_max * 16);
  const char* names[4] = {"zeros", "constant 1.5", "random [-.5,.5)", "random bits (finite)"};
  for (int blocks : {256, 512}) {
    for (int mode = 0; mode < 4; ++mode) {
      for (int i = 0; i < n; ++i) {
        if (mode == 0) h[i] = 0.f;
        else if (mode == 1) h[i] = 1.5f;
        else if (mode == 2) h[i] = (float)rand() / RAND_MAX - 0.5f;
        else { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); u = (u & 0x807fffffu) | ((unsigned)(96 + rand() % 48) << 23); __builtin_memcpy(&h[i], &u, 4); }
      }
      hipMemcpy(opnd, h, n * 4, hipMemcpyHostToDevice);
      const int steps = 20000;                   // 48 MFMAs per step per wave
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, opnd, steps / 10, clk);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, opnd, steps, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost);
      double rc = 0, rw = 0;
      for (int i = 0; i < blocks; ++i) { rc += (double)hc[2 * i]; rw += (double)hc[2 * i + 1]; }
      const double flops = (double)blocks * 4 * steps * 48 * 4096.0;
      // waves per SIMD: blocks / 256; cycles per MFMA per SIMD if the pipe never idles: 64
      const double mfma_per_simd = (double)blocks / 256 * steps * 48;
      printf("%d workgroups (%d waves/SIMD), operands %-22s: %8.1f us  %6.1f TFLOP/s  -> %.0f MHz if issue-bound at 64 clk/MFMA;  "
             "readcyclecounter/wall_clock64 = %.3f\n", blocks, blocks / 256, names[mode], ms * 1e3, flops / ms / 1e9,
             mfma_per_simd * 64 / (ms * 1e3), rc / rw);
    }
  }
  return 0;
}
