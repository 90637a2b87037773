// Stand-alone repro attempt (no libsga_hip): does an elementwise kernel full of transcendental instructions (v_exp / v_rcp /
// v_log through tanhf, expf, logf -- the arithmetic of k_factorized, csrc/elementwise.hip) compute the SAME values when its waves
// share SIMDs with waves issuing dense bf16 MFMAs?  Found in round 4 (DESIGN_EXPERIMENTS.md A.8): inside the bf16x3 two-stream
// graph the rate gradient of z (k_factorized's output) deviated in single 64-byte chunks -- always lanes 48..63 of a wave -- in
// a few per cent of identical runs, never with f32 MFMA kernels beside it, never single-stream.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/trans_mfma_repro.hip -o scripts/trans_mfma_repro.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int STRIDE = 44;
__device__ __forceinline__ float sigm(float t) { return 1.0f / (1.0f + expf(-t)); }
__device__ __forceinline__ float sgn(float t) { return (t > 0.f) - (t < 0.f); }
__device__ __forceinline__ void logit(const float* __restrict__ P, float x, float& out, float& dout) {
  float h[3], d[3];
  for (int r = 0; r < 3; ++r) {
    h[r] = P[r] * x + P[3 + r]; d[r] = P[r];
    const float t = tanhf(h[r]); const float f = P[6 + r];
    h[r] += f * t; d[r] *= 1.0f + f * (1.0f - t * t);
  }
  for (int layer = 0; layer < 2; ++layer) {
    const float* M = P + 9 + layer * 15;
    float h2[3], d2[3];
    for (int r = 0; r < 3; ++r) {
      float acc = 0.f, dacc = 0.f;
      for (int c = 0; c < 3; ++c) { acc += M[r * 3 + c] * h[c]; dacc += M[r * 3 + c] * d[c]; }
      acc += M[9 + r];
      const float t = tanhf(acc); const float f = M[12 + r];
      h2[r] = acc + f * t; d2[r] = dacc * (1.0f + f * (1.0f - t * t));
    }
    for (int r = 0; r < 3; ++r) { h[r] = h2[r]; d[r] = d2[r]; }
  }
  const float* M3 = P + 39;
  float acc = 0.f, dacc = 0.f;
  for (int c = 0; c < 3; ++c) { acc += M3[c] * h[c]; dacc += M3[c] * d[c]; }
  out = acc + M3[3]; dout = dacc;
}
template <int V>      // 0: as k_factorized; 1: without the f64 sum; 2: the network only (no sigmoid / log); 3: one network call, no f64
__global__ void k_trans(const float* __restrict__ z, const float* __restrict__ P, int n, int C, float scale, float* __restrict__ g,
                        double* __restrict__ sum) {
  __shared__ double sh[4];
  double nats = 0.0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const float* Pc = P + (size_t)(e % C) * STRIDE;
    float lo, dlo, up, dup;
    logit(Pc, z[e] - 0.5f, lo, dlo);
    if constexpr (V == 3) { g[e] = lo * dlo; continue; }
    logit(Pc, z[e] + 0.5f, up, dup);
    if constexpr (V == 2) { g[e] = lo * dup - up * dlo; continue; }
    const float sg = -sgn(lo + up);
    const float su = sigm(sg * up), sl = sigm(sg * lo);
    const float diff = su - sl;
    const float p = fabsf(diff);
    const float dp = sgn(diff) * sg * (su * (1.0f - su) * dup - sl * (1.0f - sl) * dlo);
    const float pb = fmaxf(p, 1e-9f);
    if constexpr (V == 0) nats += (double)(-logf(pb));
    g[e] = (p >= 1e-9f ? -scale / pb : 0.f) * dp;
  }
  if constexpr (V == 0) {
    for (int o = 32; o > 0; o >>= 1) nats += __shfl_down(nats, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = nats;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, sh[0] + sh[1] + sh[2] + sh[3]);
  }
}

// One arithmetic family at a time (which instruction is it?): 64 dependent steps per element
template <int OP>
__global__ void k_op(const float* __restrict__ z, int n, float* __restrict__ g) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    float x = z[e] * 0.3f + 0.1f * (float)(e & 7);
    double xd = (double)x;
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
      if constexpr (OP == 1) x = __builtin_amdgcn_exp2f(x * 0.25f) - 0.5f;                  // v_exp_f32
      else if constexpr (OP == 2) x = __builtin_amdgcn_rcpf(x + 2.5f) + 0.3f;               // v_rcp_f32
      else if constexpr (OP == 3) x = __builtin_amdgcn_logf(x * x + 1.5f) * 0.7f;           // v_log_f32
      else if constexpr (OP == 4) x = tanhf(x) * 1.7f + 0.05f;                              // ocml tanh (exp + rcp + branches)
      else if constexpr (OP == 5) x = __builtin_amdgcn_sqrtf(x * x + 0.5f) - 0.4f;          // v_sqrt_f32
      else if constexpr (OP == 6) x = fmaf(x, 0.9f, 0.1f) * fmaf(x, 0.01f, 1.0f);           // plain VALU
      else if constexpr (OP == 7) { xd = xd * 0.9 + 0.1 / (1.0 + xd * xd); x = (float)xd; } // f64 VALU (+ v_rcp_f64)
      else if constexpr (OP == 8) x = expf(x * 0.5f) * 0.3f;                                 // ocml exp (v_exp_f32 + v_ldexp + cndmask)
      else if constexpr (OP == 9) x = x / (fabsf(x) * 0.5f + 1.3f) + 0.2f;                   // f32 division (v_div_scale / v_rcp / v_div_fmas / v_div_fixup)
      else if constexpr (OP == 11) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {x, x + 0.25f}; v = v * f2{0.9f, 0.8f}; v = v + f2{0.1f, 0.05f}; v = v * v; x = v.x - v.y * 0.5f; }   // v_pk_mul_f32 / v_pk_add_f32
      else if constexpr (OP == 12) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {x, x + 0.25f}; v = __builtin_elementwise_fma(v, f2{0.9f, 0.8f}, f2{0.1f, 0.05f}); x = v.x - v.y * 0.5f; }   // v_pk_fma_f32
      else if constexpr (OP == 10) x = (float)__shfl_xor((int)__float_as_int(x) & 0x3fffffff, 1 + (k & 31), 64) * 1e-9f + x * 0.5f;   // ds_bpermute
    }
    g[e] = x;
  }
}

template <int KIND>      // 1: bf16 32x32x16 + operand re-splitting VALU work, 2: f32 32x32x2, 3: VALU only
__global__ __launch_bounds__(256, 2) void k_busy(float* out, const unsigned* opnd, int steps) {
  const int tid = threadIdx.x;
  f32x16 acc[6];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  u32x4 a[4], b[4];
  for (int k = 0; k < 4; ++k) {
    a[k] = *reinterpret_cast<const u32x4*>(opnd + ((size_t)(blockIdx.x * 256 + tid) * 8 + k) * 4);
    b[k] = *reinterpret_cast<const u32x4*>(opnd + ((size_t)(blockIdx.x * 256 + tid) * 8 + 4 + k) * 4);
  }
  float v = 1.0f + tid * 1e-3f;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if constexpr (KIND == 1)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + j) & 3]), __builtin_bit_cast(bf16x8, b[k]), acc[j], 0, 0, 0);
        else if constexpr (KIND == 2)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[(k + j) & 3].x), __builtin_bit_cast(float, b[k].x), acc[j], 0, 0, 0);
        v = v * 1.0000001f + 1e-7f;                      // some VALU work between the MFMAs (the split of the bf16x3 loop)
      }
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  float r = v;
  for (int j = 0; j < 6; ++j) for (int q = 0; q < 16; ++q) r += acc[j][q];
  out[blockIdx.x * 256 + tid] = r;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 400;
  const int C = 192, n = 96 * C;                        // z of one Kodak image: 8 x 12 x 192
  std::vector<float> hz(n), hP((size_t)C * STRIDE);
  srand(1);
  for (auto& x : hz) x = 6.f * ((float)rand() / RAND_MAX - 0.5f);
  for (auto& x : hP) x = 0.3f + 0.7f * (float)rand() / RAND_MAX;
  for (int c = 0; c < C; ++c) for (int k : {3, 4, 5, 18, 19, 20, 33, 34, 35, 42}) hP[(size_t)c * STRIDE + k] = (float)rand() / RAND_MAX - 0.5f;
  for (int c = 0; c < C; ++c) for (int k : {6, 7, 8, 21, 22, 23, 36, 37, 38}) hP[(size_t)c * STRIDE + k] = 0.6f * ((float)rand() / RAND_MAX - 0.5f);
  float *z, *P, *g, *busy_out; double* sum; unsigned* opnd;
  CHK(hipMalloc(&z, n * 4)); CHK(hipMalloc(&P, hP.size() * 4)); CHK(hipMalloc(&g, n * 4)); CHK(hipMalloc(&sum, 8));
  CHK(hipMalloc(&busy_out, 512 * 256 * 4)); CHK(hipMalloc(&opnd, (size_t)512 * 256 * 32 * 4));
  CHK(hipMemcpy(z, hz.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(P, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
  std::vector<unsigned> ho((size_t)512 * 256 * 32);
  for (auto& x : ho) { unsigned lo = (rand() & 0x807f) | ((120 + rand() % 8) << 7), hi = (rand() & 0x807f) | ((120 + rand() % 8) << 7); x = lo | (hi << 16); }
  CHK(hipMemcpy(opnd, ho.data(), ho.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  std::vector<float> ref(n), got(n);
  CHK(hipMemset(sum, 0, 8));
  hipLaunchKernelGGL(k_trans<0>, dim3(12), dim3(256), 0, s2, z, P, n, C, 3.7e-6f, g, sum);
  CHK(hipStreamSynchronize(s2));
  CHK(hipMemcpy(ref.data(), g, n * 4, hipMemcpyDeviceToHost));
  const char* names[4] = {"alone", "beside bf16 MFMA waves", "beside f32 MFMA waves", "beside VALU-only waves"};
  for (int kind = 0; kind < 4; ++kind) {
    long long bad_runs = 0, bad_elems = 0, lane_hist[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
      const int blocks = 256 + 256 * (r & 1);            // one or two busy waves per SIMD
      if (kind == 1) hipLaunchKernelGGL(k_busy<1>, dim3(blocks), dim3(256), 0, s1, busy_out, opnd, 3000);
      if (kind == 2) hipLaunchKernelGGL(k_busy<2>, dim3(blocks), dim3(256), 0, s1, busy_out, opnd, 800);
      if (kind == 3) hipLaunchKernelGGL(k_busy<3>, dim3(blocks), dim3(256), 0, s1, busy_out, opnd, 20000);
      for (int q = 0; q < 8; ++q) {                      // eight launches of the elementwise kernel land inside the busy kernel
        CHK(hipMemsetAsync(g, 0, n * 4, s2));
        hipLaunchKernelGGL(k_trans<0>, dim3(12), dim3(256), 0, s2, z, P, n, C, 3.7e-6f, g, sum);
        CHK(hipMemcpyAsync(got.data(), g, n * 4, hipMemcpyDeviceToHost, s2));
        CHK(hipStreamSynchronize(s2));
        int nb = 0;
        for (int i = 0; i < n; ++i) if (memcmp(&got[i], &ref[i], 4)) { ++nb; ++lane_hist[(i & 63) >> 4]; }
        bad_elems += nb; bad_runs += nb > 0;
      }
      CHK(hipStreamSynchronize(s1));
    }
    printf("%-26s: %d launches, %lld with deviating elements (%lld elements; by quarter-wave lanes 0-15 / 16-31 / 32-47 / 48-63: %lld %lld %lld %lld)\n",
           names[kind], reps * 8, bad_runs, bad_elems, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
  }
  // ---- which part of the kernel?  variants 1..3 beside bf16 MFMA waves ----
  for (int v = 1; v <= 3; ++v) {
    auto lv = [&]() {
      if (v == 1) hipLaunchKernelGGL(k_trans<1>, dim3(12), dim3(256), 0, s2, z, P, n, C, 3.7e-6f, g, sum);
      if (v == 2) hipLaunchKernelGGL(k_trans<2>, dim3(12), dim3(256), 0, s2, z, P, n, C, 3.7e-6f, g, sum);
      if (v == 3) hipLaunchKernelGGL(k_trans<3>, dim3(12), dim3(256), 0, s2, z, P, n, C, 3.7e-6f, g, sum);
    };
    lv(); CHK(hipStreamSynchronize(s2));
    CHK(hipMemcpy(ref.data(), g, n * 4, hipMemcpyDeviceToHost));
    long long bad_runs = 0, bad_elems = 0, lane_hist[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps / 4; ++r) {
      hipLaunchKernelGGL(k_busy<1>, dim3(256 + 256 * (r & 1)), dim3(256), 0, s1, busy_out, opnd, 3000);
      for (int q = 0; q < 8; ++q) {
        CHK(hipMemsetAsync(g, 0, n * 4, s2));
        lv();
        CHK(hipMemcpyAsync(got.data(), g, n * 4, hipMemcpyDeviceToHost, s2));
        CHK(hipStreamSynchronize(s2));
        int nb = 0;
        for (int i = 0; i < n; ++i) if (memcmp(&got[i], &ref[i], 4)) { ++nb; ++lane_hist[(i & 63) >> 4]; }
        bad_elems += nb; bad_runs += nb > 0;
      }
      CHK(hipStreamSynchronize(s1));
    }
    printf("variant %d beside bf16 MFMA waves: %d launches, %lld deviating (%lld elements; lanes 0-15 / 16-31 / 32-47 / 48-63: %lld %lld %lld %lld)\n",
           v, reps / 4 * 8, bad_runs, bad_elems, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
  }
  // ---- which instruction family?  each k_op<OP> alone (reference) and beside bf16 MFMA waves ----
  const char* ops[13] = {"", "v_exp_f32", "v_rcp_f32", "v_log_f32", "tanhf (ocml)", "v_sqrt_f32", "plain f32 VALU", "f64 VALU", "expf (ocml)",
                         "f32 division", "ds_bpermute", "v_pk_mul/add_f32", "v_pk_fma_f32"};
  auto launch_op = [&](int op) {
    switch (op) {
      case 1: hipLaunchKernelGGL(k_op<1>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 2: hipLaunchKernelGGL(k_op<2>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 3: hipLaunchKernelGGL(k_op<3>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 4: hipLaunchKernelGGL(k_op<4>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 5: hipLaunchKernelGGL(k_op<5>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 6: hipLaunchKernelGGL(k_op<6>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 7: hipLaunchKernelGGL(k_op<7>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 8: hipLaunchKernelGGL(k_op<8>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 9: hipLaunchKernelGGL(k_op<9>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 10: hipLaunchKernelGGL(k_op<10>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 11: hipLaunchKernelGGL(k_op<11>, dim3(12), dim3(256), 0, s2, z, n, g); break;
      case 12: hipLaunchKernelGGL(k_op<12>, dim3(12), dim3(256), 0, s2, z, n, g); break;
    }
  };
  for (int op = (argc > 2 ? atoi(argv[2]) : 1); op <= 12; ++op) {
    launch_op(op);
    CHK(hipStreamSynchronize(s2));
    CHK(hipMemcpy(ref.data(), g, n * 4, hipMemcpyDeviceToHost));
    long long bad_runs = 0, bad_elems = 0, lane_hist[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps / 4; ++r) {
      hipLaunchKernelGGL(k_busy<1>, dim3(256 + 256 * (r & 1)), dim3(256), 0, s1, busy_out, opnd, 3000);
      for (int q = 0; q < 8; ++q) {
        CHK(hipMemsetAsync(g, 0, n * 4, s2));
        launch_op(op);
        CHK(hipMemcpyAsync(got.data(), g, n * 4, hipMemcpyDeviceToHost, s2));
        CHK(hipStreamSynchronize(s2));
        int nb = 0;
        for (int i = 0; i < n; ++i) if (memcmp(&got[i], &ref[i], 4)) { ++nb; ++lane_hist[(i & 63) >> 4]; }
        bad_elems += nb; bad_runs += nb > 0;
      }
      CHK(hipStreamSynchronize(s1));
    }
    printf("op %-16s beside bf16 MFMA waves: %d launches, %lld deviating (%lld elements; lanes 0-15 / 16-31 / 32-47 / 48-63: %lld %lld %lld %lld)\n",
           ops[op], reps / 4 * 8, bad_runs, bad_elems, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3]);
  }
  return 0;
}
