// Micro-benchmark: ceiling of the conv kernel's inner pattern on gfx950.
//  mode 0: 96 x v_mfma_f32_32x32x2_f32 per step from registers only (6 accumulators)
//  mode 1: + the 20 ds_read_b128 fragment loads per step (LDS never written: values irrelevant)
//  mode 2: + one __syncthreads per step
//  mode 3: + 10 ds_write_b128 per step
//  mode 4: + 10 global_load_dwordx4 per step per lane (gathered rows from a 100 MB buffer), written to LDS
//  mode 5: mode 4 with random (not constant) MFMA operand data in LDS
//  mode 6: operands arrive by global_load_lds_dwordx4 (no VGPR staging, no ds_write) into a second LDS
//          stage while the current one is multiplied; unpadded 128-B rows, XOR-swizzled 16-B chunks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void bench(float* out, int steps, const float* src, size_t src_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  f32x16 acc[2][3];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int arow = wm * 64 + (lane & 31), brow = wn * 96 + (lane & 31), koff = (lane >> 5) * 4;
  const float* As = smem; const float* Bs = smem + 128 * 36;
  f32x4 af[2], bf[3];
  for (int i = 0; i < 2; ++i) af[i] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
  for (int i = 0; i < 3; ++i) bf[i] = f32x4{0.5f, 0.25f, lane * 1.f, 1.f};
  f32x4 st = {1.f, 2.f, 3.f, 4.f};
  f32x4 ld[10];
  unsigned base[10];
  const unsigned mask = (unsigned)(src_floats / 2 - 1) & ~3u;   // src_floats/2 is a power of two
  for (int p = 0; p < 10; ++p) base[p] = ((blockIdx.x * 128u + p * 32u + (tid >> 3)) * 192u + (tid & 7) * 4u);
  for (int p = 0; p < 10; ++p) ld[p] = st;
  if (MODE >= 5) { for (int i = tid; i < 2 * (128 + 192) * 36; i += 256) smem[i] = src[(blockIdx.x * 977 + i) % src_floats]; __syncthreads(); }
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE >= 1) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) af[tm] = *reinterpret_cast<const f32x4*>(&As[(arow + tm * 32) * 36 + q * 8 + koff]);
#pragma unroll
        for (int tn = 0; tn < 3; ++tn) bf[tn] = *reinterpret_cast<const f32x4*>(&Bs[(brow + tn * 32) * 36 + q * 8 + koff]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 3; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][r], bf[tn][r], acc[tm][tn], 0, 0, 0);
      if (MODE >= 3 && q == 1) {
#pragma unroll
        for (int p = 0; p < 10; ++p)
          *reinterpret_cast<f32x4*>(&smem[(128 + 192) * 36 + (p * 32 + (tid >> 3)) * 36 + (tid & 7) * 4]) = (MODE >= 4 ? ld[p] : st);
      }
      if (MODE >= 4 && q == 2) {
#pragma unroll
        for (int p = 0; p < 10; ++p) {
          // cheap addressing: per-lane base + uniform per-step offset (wraps with a mask)
          const unsigned off = (base[p] + (unsigned)s * 6151u * 192u) & mask;
          ld[p] = *reinterpret_cast<const f32x4*>(&src[off]);
        }
      }
    }
    if (MODE >= 2) __syncthreads();
    if (MODE == 0) { asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2])); }
  }
  float v = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) v += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = v;
}

// mode 6 kernel: LDS-DMA double buffer
__global__ __launch_bounds__(256, 2) void bench_glds(float* out, int steps, const float* src, size_t src_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  f32x16 acc[2][3];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int arow = wm * 64 + (lane & 31), brow = 128 + wn * 96 + (lane & 31);
  constexpr int STAGE = (128 + 192) * 32;          // floats per stage, rows of 32 floats (128 B), no padding
  const unsigned mask = (unsigned)(src_floats / 2 - 1) & ~3u;
  // each wave fills rows [wid*80, wid*80+80): 10 instructions x 8 rows; lane -> row (l>>3), slot (l&7)
  unsigned base[10];
  for (int p = 0; p < 10; ++p) {
    const int row = wid * 80 + p * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);           // swizzle on the SOURCE side
    base[p] = ((blockIdx.x * 320u + row) * 192u + chunk * 4u);
  }
  auto issue = [&](int s, int stage) {
#pragma unroll
    for (int p = 0; p < 10; ++p) {
      const unsigned off = (base[p] + (unsigned)s * 6151u * 192u) & mask;
      __builtin_amdgcn_global_load_lds(src + off, (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (wid * 80 + p * 8) * 32), 16, 0, 0);
    }
  };
  issue(0, 0);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const float* St = smem + (s & 1) * STAGE;
    issue(s + 1, (s + 1) & 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 af[2], bf[3];
      const int c = 2 * q + (lane >> 5);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) { const int r = arow + tm * 32; af[tm] = *reinterpret_cast<const f32x4*>(&St[r * 32 + ((c ^ ((r >> 1) & 7)) * 4)]); }
#pragma unroll
      for (int tn = 0; tn < 3; ++tn) { const int r = brow + tn * 32; bf[tn] = *reinterpret_cast<const f32x4*>(&St[r * 32 + ((c ^ ((r >> 1) & 7)) * 4)]); }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 3; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][r], bf[tn][r], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }
  float v = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) v += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = v;
}

void run_glds(int blocks, int steps, float* out, const float* src, size_t nsrc) {
  const size_t lds = 2 * (128 + 192) * 32 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&bench_glds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(bench_glds, dim3(blocks), dim3(256), lds, 0, out, steps, src, nsrc);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(bench_glds, dim3(blocks), dim3(256), lds, 0, out, steps, src, nsrc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double flops = (double)blocks * 4 * steps * 96 * 4096.0;
  printf("mode 6 (glds, 2 WG/CU) blocks %d steps %d: %.1f us  %.1f TFLOP/s\n", blocks, steps, ms * 1e3, flops / ms / 1e9);
}

template <int MODE>
void run(int blocks, int steps, float* out, const float* src, size_t nsrc) {
  const size_t lds = 2 * (128 + 192) * 36 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), lds, 0, out, steps, src, nsrc);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), lds, 0, out, steps, src, nsrc);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double flops = (double)blocks * 4 * steps * 96 * 4096.0;
  printf("mode %d blocks %d steps %d: %.1f us  %.1f TFLOP/s\n", MODE, blocks, steps, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  const size_t nsrc = 32u << 20; float* src; hipMalloc(&src, nsrc * 4);
  { float* hsrc = (float*)malloc(nsrc * 4); for (size_t i = 0; i < nsrc; ++i) hsrc[i] = (float)rand() / RAND_MAX - 0.5f; hipMemcpy(src, hsrc, nsrc * 4, hipMemcpyHostToDevice); free(hsrc); }
  for (int blocks : {256, 1024}) {
    run<0>(blocks, 150, out, src, nsrc); run<2>(blocks, 150, out, src, nsrc); run<3>(blocks, 150, out, src, nsrc); run<4>(blocks, 150, out, src, nsrc); run<5>(blocks, 150, out, src, nsrc);
    run_glds(blocks, 150, out, src, nsrc);
  }
  return 0;
}
