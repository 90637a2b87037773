// C -> 3 transposed 5x5/2 convolution (last synthesis layer, nn_models.py:60-63) as a halo-tiled
// implicit GEMM.  In the generic gather-GEMM each of the 9 (dy,dx) taps re-reads its 128-pixel A
// tile from global memory (876 MB per launch at B=8, 256^2: HBM-bound, PMC profile r01).  Here a
// workgroup owns a 4 x 16 tile of input positions, stages the (4+2) x (16+2) halo of one
// 32-channel chunk in LDS once, and all 9 taps read their A fragments from it at shifted
// positions.  4 x 16 (not 8 x 16): 40 KB of LDS -> 4 workgroups per CU, and at the bench shape
// 2048 workgroups = exactly two full rounds of the chip (8 x 16: 1024 workgroups on 768 slots,
// a third of the chip idle in the second round; 80 -> 72 us).  N = 4 phases x 3 channels = 12 columns, padded to 16 and
// multiplied with v_mfma_f32_16x16x4_f32 (the 32-wide MFMA would waste 62 % of the tile).
#include "sga_common.h"
#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TH = 4, TW = 16;            // input positions per workgroup (4 waves x RPW rows x 16)
constexpr int RPW = TH / 4;               // tile rows per wave
constexpr int HH = TH + 2, HW = TW + 2;   // halo
constexpr int NPX = HH * HW;              // 108 halo pixels
constexpr int PIT = 40;                   // LDS row pitch (32 + 8 floats): with the 16x16x4 fragment pattern (16 pixels x 4 k-slots per
                                          // b128 lane group) 36 gives 40 % bank conflicts (PMC), 40 none (slot = 10*pixel + kslot mod 16)
constexpr int NB = 9 * 16;                // weight rows per chunk: 9 taps x 16 columns
constexpr int HALO_F4 = NPX * 8;          // float4 per halo chunk
constexpr int B_F4 = NB * 8;
constexpr int PH = (HALO_F4 + 255) / 256;
constexpr int PBW = (B_F4 + 255) / 256;

// MSE = true: the distortion kernel (elementwise.hip k_mse; sga.py:150, 170-173 and the gradient of :161
// w.r.t. x_tilde) runs in the epilogue on the values still in registers: per-image f64 sums of
// (x - x_tilde)^2 and (255 x - round(255 clip(x_tilde)))^2, and the zero-bordered gradient image
// gpad = coef * (x_tilde - x) that igdn2.bwd's 3-channel prologue reads.  Same expressions as k_mse.
struct Deconv3Mse {
  const float* x; const StepCtx* ctx; ImgSums* sums; float* gpad; int Hp, Wp; int prio;
};

template <bool MSE>
__global__ __launch_bounds__(256) void deconv3_halo_kernel(
    const float* __restrict__ in, const float* __restrict__ w /*[C/32][9][16][32]*/,
    const float* __restrict__ bias, float* __restrict__ out, int B, int Hi, int Wi, int C, int Ho,
    int Wo, int tiles_x, int tiles_y, Deconv3Mse ms) {
  __shared__ __attribute__((aligned(16))) float Hs[NPX * PIT];
  __shared__ __attribute__((aligned(16))) float Bs[NB * PIT];
  if (ms.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (ms.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (ms.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int bid = blockIdx.x;
  const int tx0 = (bid % tiles_x) * TW; bid /= tiles_x;
  const int ty0 = (bid % tiles_y) * TH;
  const int b = bid / tiles_y;
  const int nchunk = C / 32;

  // halo gather metadata (loop-invariant): byte offset of the pixel or -1 (zero padding)
  long long hoff[PH];
#pragma unroll
  for (int k = 0; k < PH; ++k) {
    const int f = tid + 256 * k;
    hoff[k] = -1;
    if (f < HALO_F4) {
      const int px = f >> 3, c4 = f & 7;
      const int hy = px / HW, hx = px - hy * HW;
      const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
      if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi)
        hoff[k] = ((long long)(b * Hi + iy) * Wi + ix) * C + c4 * 4;
    }
  }
  f32x4 rh[PH], rw[PBW];
  auto gload = [&](int c) {
#pragma unroll
    for (int k = 0; k < PH; ++k)
      rh[k] = hoff[k] >= 0 ? *reinterpret_cast<const f32x4*>(in + hoff[k] + c * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PBW; ++k) {
      const int f = tid + 256 * k;
      if (f < B_F4) rw[k] = *reinterpret_cast<const f32x4*>(w + (size_t)c * NB * 32 + f * 4);
    }
  };

  f32x4 acc[RPW];
#pragma unroll
  for (int s = 0; s < RPW; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, g = lane >> 4;
  gload(0);
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int k = 0; k < PH; ++k) {
      const int f = tid + 256 * k;
      if (f < HALO_F4) *reinterpret_cast<f32x4*>(&Hs[(f >> 3) * PIT + (f & 7) * 4]) = rh[k];
    }
#pragma unroll
    for (int k = 0; k < PBW; ++k) {
      const int f = tid + 256 * k;
      if (f < B_F4) *reinterpret_cast<f32x4*>(&Bs[(f >> 3) * PIT + (f & 7) * 4]) = rw[k];
    }
    __syncthreads();
    if (c + 1 < nchunk) gload(c + 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x4 bf = *reinterpret_cast<const f32x4*>(&Bs[(t * 16 + li) * PIT + q * 16 + g * 4]);
        f32x4 af[RPW];
#pragma unroll
        for (int s = 0; s < RPW; ++s) {
          const int hy = RPW * wid + s + dy + 1, hx = li + dx + 1;
          af[s] = *reinterpret_cast<const f32x4*>(&Hs[(hy * HW + hx) * PIT + q * 16 + g * 4]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int s = 0; s < RPW; ++s)
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][r], bf[r], acc[s], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D layout of 16x16x4: column (n) = lane & 15, row (position within the 16-wide tile row) = 4*(lane>>4) + reg
  const int n = li;
  float a0 = 0.f, a1 = 0.f;
  float coef = 0.f;
  if constexpr (MSE) {
    if (ms.ctx->lambda > 0.f)
      coef = ms.ctx->lambda * 2.0f * 65025.0f * ms.ctx->loss_scale / (float)(Ho * Wo * 3);
  }
  if (n < 12) {
    const int pp = n / 3, ch = n - pp * 3;
    const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
    for (int s = 0; s < RPW; ++s) {
      const int iy = ty0 + RPW * wid + s;
      if (iy >= Hi) continue;
      const int oy = 2 * iy + (pp >> 1);
      if (oy >= Ho) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ix = tx0 + 4 * g + reg;
        const int ox = 2 * ix + (pp & 1);
        if (ix < Wi && ox < Wo) {
          const size_t idx = ((size_t)(b * Ho + oy) * Wo + ox) * 3 + ch;
          const float tv = acc[s][reg] + bv;
          out[idx] = tv;
          if constexpr (MSE) {
            const float xv = ms.x[idx];
            const float d = xv - tv;
            a0 += d * d;
            const float q = rintf(fminf(fmaxf(tv, 0.f), 1.f) * 255.0f);
            const float dq = xv * 255.0f - q;
            a1 += dq * dq;
            ms.gpad[((size_t)(b * ms.Hp + oy + 2) * ms.Wp + ox + 2) * 3 + ch] = coef * (tv - xv);
          }
        }
      }
    }
  }
  if constexpr (MSE) {
    // f32 partials of <= 4 values per lane, f64 across the workgroup, one atomic pair per workgroup
    double d0 = (double)a0, d1 = (double)a1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      d0 += __shfl_down(d0, o, 64);
      d1 += __shfl_down(d1, o, 64);
    }
    double* red = reinterpret_cast<double*>(Hs);     // the K loop ended with a barrier: Hs is free
    if (lane == 0) { red[wid] = d0; red[4 + wid] = d1; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&ms.sums[b].sq_p[blockIdx.x % kSqSlots], red[0] + red[1] + red[2] + red[3]);
      atomicAdd(&ms.sums[b].sqq_p[blockIdx.x % kSqSlots], red[4] + red[5] + red[6] + red[7]);
    }
  }
}

}  // namespace

int g_deconv3_prio = 0;      // experiment (SGA_MAIN_WAVE_PRIO), set by sga_create

int launch_deconv3_halo(const float* in, const float* w, const float* bias, float* out, int B,
                        int Hi, int Wi, int C, int Ho, int Wo, hipStream_t stream) {
  const int tiles_x = (Wi + TW - 1) / TW, tiles_y = (Hi + TH - 1) / TH;
  hipLaunchKernelGGL(deconv3_halo_kernel<false>, dim3(B * tiles_x * tiles_y), dim3(256), 0, stream, in, w,
                     bias, out, B, Hi, Wi, C, Ho, Wo, tiles_x, tiles_y, Deconv3Mse{});
  return (int)hipGetLastError();
}

int launch_deconv3_halo_mse(const float* in, const float* w, const float* bias, float* out, int B,
                            int Hi, int Wi, int C, int Ho, int Wo, const float* x, const StepCtx* ctx,
                            ImgSums* sums, float* gpad, int Hp, int Wp, hipStream_t stream) {
  const int tiles_x = (Wi + TW - 1) / TW, tiles_y = (Hi + TH - 1) / TH;
  hipLaunchKernelGGL(deconv3_halo_kernel<true>, dim3(B * tiles_x * tiles_y), dim3(256), 0, stream, in, w,
                     bias, out, B, Hi, Wi, C, Ho, Wo, tiles_x, tiles_y, Deconv3Mse{x, ctx, sums, gpad, Hp, Wp, g_deconv3_prio});
  return (int)hipGetLastError();
}
