// C -> 3 transposed 5x5/2 convolution (last synthesis layer, nn_models.py:60-63) as a PLAIN GEMM + col2im.
//
//   out[2i + ky - 2, 2j + kx - 2, c] += sum_ci in[i, j, ci] * W[ky, kx, ci, c]
//
// Step 1 (deconv3_gemm_kernel): P[pixel, (ky, kx, c)] = in[pixel, :] . W[ky, kx, :, c] -- every input pixel
//   against all 25 x 3 = 75 kernel columns (padded to 80 = 5 MFMA column blocks).  No halo, no zero columns: 80
//   column slots per input pixel, where the halo-tiled form (deconv3.hip) spends 9 taps x 16 = 144 and a
//   halo-recomputing tile GEMM ~105.  The A rows go from global memory STRAIGHT into the 16x16x4 MFMA's A layout
//   (lane = row + 16 * k-group: one 16-byte load = the lane's k-slices of four consecutive MFMAs), the whole weight
//   matrix (80 x C) sits in LDS for the life of the workgroup.  P (80 floats per pixel) is written once.
// Step 2 (deconv3_col2im_kernel): each output pixel sums its 9 / 6 / 6 / 4 products in a fixed (ky, kx) order, adds
//   the bias, and -- in a step -- does the distortion kernel's work on the value still in a register (k_mse:
//   sga.py:150, 170-173 and the gradient image of sga.py:161).
// Bound: HBM (read in 101 MB + write / re-read P 2 x 42 MB at the bench shape) ~ MFMA (4.0 GFLOP incl. padding).
#include <atomic>

#include "sga_common.h"
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NCOL = 80;                  // 75 product columns + 5 of padding (zero weights)
constexpr int NCB = NCOL / 16;            // MFMA column blocks

// grid-stride over 16-pixel row blocks; a wave owns one row block at a time.  8 waves share one copy of the weights
// (64 KB of LDS: 2 workgroups = 16 waves per CU, 4 per SIMD): while one wave multiplies (240 MFMAs = 3.2 us per row
// block) the other three wait for their A rows, so no register double-buffering is needed (<= 128 VGPRs).
constexpr int GNT = 512, GNW = GNT / 64;
template <int CQ /* C / 16 */>
__global__ __launch_bounds__(GNT, 4) void deconv3_gemm_kernel(const float* __restrict__ in, const float* __restrict__ w /*[80][C]*/,
                                                              float* __restrict__ P, long long M) {
  constexpr int C = CQ * 16, PITCH = C + 8;        // LDS row pitch = 8 mod 64 banks: conflict-free b128 fragment reads
  extern __shared__ __attribute__((aligned(16))) float Bs[];      // [80][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int f = tid; f < NCOL * (C / 4); f += GNT) {
    const int n = f / (C / 4), c4 = f - n * (C / 4);
    *reinterpret_cast<f32x4*>(&Bs[n * PITCH + c4 * 4]) = *reinterpret_cast<const f32x4*>(w + (size_t)n * C + c4 * 4);
  }
  __syncthreads();
  const int li = lane & 15, g = lane >> 4;
  const long long nrb = (M + 15) / 16;
  const long long stride = (long long)gridDim.x * GNW;
  long long rb = (long long)blockIdx.x * GNW + wid;
  f32x4 a_cur[CQ];
  auto load_rows = [&](long long r, f32x4 (&dst)[CQ]) {
    long long row = r * 16 + li;
    if (row >= M) row = M - 1;                     // clamped re-read, discarded at the store
    const float* src = in + (size_t)row * C + g * 4;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
#if SGA_NT & 16
      dst[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + q * 16));
#else
      dst[q] = *reinterpret_cast<const f32x4*>(src + q * 16);
#endif
    }
  };
  for (; rb < nrb; rb += stride) {
    load_rows(rb, a_cur);
    // the weight fragments are the same for every row block: without this the compiler hoists all 12 x 5 of them out
    // of the loop (240 VGPRs, 132 of them spilled); they are re-read from LDS per row block instead
    asm volatile("" ::: "memory");
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      f32x4 bf[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        bf[cb] = *reinterpret_cast<const f32x4*>(&Bs[(cb * 16 + li) * PITCH + q * 16 + g * 4]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cb][r], a_cur[q][r], acc[cb], 0, 0, 0);   // D = W . in^T
    }
    // The weights are the MFMA's A operand and the pixels its B operand, so that D[row = 4 * (lane >> 4) + reg][column =
    // lane & 15] = P[pixel lane & 15][column block * 16 + 4 * (lane >> 4) + reg]: a lane's 4 registers are 4 consecutive
    // product columns of ONE pixel -> one 16-byte store per column block (5 per row block instead of 20 4-byte ones).
    {
      const long long row = rb * 16 + li;
      if (row < M) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          *reinterpret_cast<f32x4*>(P + (size_t)row * NCOL + cb * 16 + 4 * g) = acc[cb];
      }
    }
  }
}

struct Col2imMse {
  const float* x; const StepCtx* ctx; ImgSums* sums; float* gpad; int Hp, Wp;
};

// A workgroup owns CT_H x CT_W input positions = a (2 CT_H) x (2 CT_W) block of output pixels: the P rows of the
// positions and their 1-wide halo are read ONCE, whole rows in 16-byte pieces, into LDS ([100][80] floats = 32 KB: 5
// workgroups per CU cover each other's load latency), and every thread sums the 9 / 6 / 6 / 4 products of its output
// pixel out of LDS.  (Summing straight from
// global memory -- 3 floats from each of up to 9 rows 320 bytes apart per thread -- ran at 47 us: 27 load
// instructions per wave, each touching 32 cache lines.)
constexpr int CT_H = 8, CT_W = 8, CH_H = CT_H + 2, CH_W = CT_W + 2, CPOS = CH_H * CH_W;
// LDS pitch of a position's products.  Lanes of one parity class read the SAME column offset of positions hx, hx + 1, ...:
// at a pitch of 80 floats (= 16 banks mod 32) positions two apart collide -- 38.7 % LDS bank conflicts, 73 % wait
// (profiles/r04_pmc_kernels.txt).  Only 75 of the 80 columns carry products: 19 of the 20 row pieces are staged at a pitch of
// 76 floats = 12 mod 32, which puts eight consecutive positions on eight different bank groups (0, 12, 24, 4, 16, 28, 8, 20),
// keeps the 16-byte alignment of the pieces and the 5 workgroups per CU (30.4 KB)
#ifndef SGA_COL2IM_PITCH
#define SGA_COL2IM_PITCH 76
#endif
constexpr int CPITCH = SGA_COL2IM_PITCH, CPIECES = CPITCH < NCOL ? CPITCH / 4 : NCOL / 4;

template <bool MSE>
__global__ __launch_bounds__(256) void deconv3_col2im_kernel(const float* __restrict__ P, const float* __restrict__ bias,
                                                             float* __restrict__ out, int Hi, int Wi, int Ho, int Wo,
                                                             int tiles_x, Col2imMse ms) {
  __shared__ __attribute__((aligned(16))) float Ps[CPOS * CPITCH];
  __shared__ double red[8];
  const int b = blockIdx.y;
  const int ty0 = (blockIdx.x / tiles_x) * CT_H, tx0 = (blockIdx.x % tiles_x) * CT_W;
  for (int f = threadIdx.x; f < CPOS * CPIECES; f += 256) {
    const int pos = f / CPIECES, c4 = f - pos * CPIECES;
    const int i = ty0 + pos / CH_W - 1, j = tx0 + pos % CH_W - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)i < (unsigned)Hi && (unsigned)j < (unsigned)Wi)
      v = *reinterpret_cast<const f32x4*>(P + ((size_t)(b * Hi + i) * Wi + j) * NCOL + c4 * 4);
    *reinterpret_cast<f32x4*>(&Ps[pos * CPITCH + c4 * 4]) = v;
  }
  __syncthreads();
  float a0 = 0.f, a1 = 0.f;
  float coef = 0.f;
  if constexpr (MSE) {
    if (ms.ctx->lambda > 0.f) coef = ms.ctx->lambda * 2.0f * 65025.0f * ms.ctx->loss_scale / (float)(Ho * Wo * 3);
  }
  const float bz[3] = {bias ? bias[0] : 0.f, bias ? bias[1] : 0.f, bias ? bias[2] : 0.f};
#pragma unroll
  for (int k = 0; k < (4 * CT_H * CT_W) / 256; ++k) {
    const int q = threadIdx.x + 256 * k;
    const int ly = q / (2 * CT_W), lx = q - ly * (2 * CT_W);          // output pixel within the block
    const int oy = 2 * ty0 + ly, ox = 2 * tx0 + lx;
    if (oy >= Ho || ox >= Wo) continue;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    // ky = oy (mod 2): input row i = (oy - ky + 2) / 2, i.e. halo row (ly - ky + 2) / 2 + 1; likewise kx.
    // Fixed order: ky ascending, kx ascending.  Positions outside the image hold zeros.
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ky = (ly & 1) + 2 * a;
      if (ky > 4) continue;
      const int hy = ((ly - ky + 2) >> 1) + 1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int kx = (lx & 1) + 2 * c;
        if (kx > 4) continue;
        const int hx = ((lx - kx + 2) >> 1) + 1;
        const float* t = &Ps[(hy * CH_W + hx) * CPITCH + (ky * 5 + kx) * 3];
        s0 += t[0]; s1 += t[1]; s2 += t[2];
      }
    }
    const float tv[3] = {s0 + bz[0], s1 + bz[1], s2 + bz[2]};
    const size_t idx = (((size_t)b * Ho + oy) * Wo + ox) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      out[idx + c] = tv[c];
      if constexpr (MSE) {
        const float xv = ms.x[idx + c];
        const float d = xv - tv[c];
        a0 += d * d;
        const float qv = rintf(fminf(fmaxf(tv[c], 0.f), 1.f) * 255.0f);
        const float dq = xv * 255.0f - qv;
        a1 += dq * dq;
        ms.gpad[((size_t)(b * ms.Hp + oy + 2) * ms.Wp + ox + 2) * 3 + c] = coef * (tv[c] - xv);
      }
    }
  }
  if constexpr (MSE) {
    double d0 = (double)a0, d1 = (double)a1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      d0 += __shfl_down(d0, o, 64);
      d1 += __shfl_down(d1, o, 64);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid] = d0; red[4 + wid] = d1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&ms.sums[b].sq_p[blockIdx.x % kSqSlots], red[0] + red[1] + red[2] + red[3]);
      atomicAdd(&ms.sums[b].sqq_p[blockIdx.x % kSqSlots], red[4] + red[5] + red[6] + red[7]);
    }
  }
}

template <int CQ>
int launch_gemm(const float* in, const float* w, float* P, long long M, hipStream_t s) {
  constexpr int C = CQ * 16;
  const size_t lds = (size_t)NCOL * (C + 8) * sizeof(float);
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&deconv3_gemm_kernel<CQ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  const long long nrb = (M + 15) / 16;
  long long grid = (nrb + GNW - 1) / GNW;
  if (grid > 512) grid = 512;                      // 2 workgroups per CU, each loads the weights once
  hipLaunchKernelGGL(deconv3_gemm_kernel<CQ>, dim3((unsigned)grid), dim3(GNT), lds, s, in, w, P, M);
  return (int)hipGetLastError();
}

}  // namespace

// in [B,Hi,Wi,C] -> P [B*Hi*Wi][80];  w80: [80][C] (row n = (ky*5+kx)*3 + c, rows 75..79 zero)
int launch_deconv3_gemm(const float* in, const float* w80, float* P, int B, int Hi, int Wi, int C, hipStream_t s) {
  const long long M = (long long)B * Hi * Wi;
  switch (C / 16) {
    case 4: return launch_gemm<4>(in, w80, P, M, s);
    case 8: return launch_gemm<8>(in, w80, P, M, s);
    case 12: return launch_gemm<12>(in, w80, P, M, s);
    case 16: return launch_gemm<16>(in, w80, P, M, s);
    case 20: return launch_gemm<20>(in, w80, P, M, s);
    case 24: return launch_gemm<24>(in, w80, P, M, s);
  }
  return (int)hipErrorInvalidValue;
}

int launch_deconv3_col2im(const float* P, const float* bias, float* out, int B, int Hi, int Wi, int Ho, int Wo,
                          const float* x, const StepCtx* ctx, ImgSums* sums, float* gpad, int Hp, int Wp, hipStream_t s) {
  const int tiles_x = (Wi + CT_W - 1) / CT_W, tiles_y = (Hi + CT_H - 1) / CT_H;
  if (x)
    hipLaunchKernelGGL(deconv3_col2im_kernel<true>, dim3(tiles_x * tiles_y, B), dim3(256), 0, s, P, bias, out, Hi, Wi, Ho,
                       Wo, tiles_x, Col2imMse{x, ctx, sums, gpad, Hp, Wp});
  else
    hipLaunchKernelGGL(deconv3_col2im_kernel<false>, dim3(tiles_x * tiles_y, B), dim3(256), 0, s, P, bias, out, Hi, Wi, Ho,
                       Wo, tiles_x, Col2imMse{});
  return (int)hipGetLastError();
}
