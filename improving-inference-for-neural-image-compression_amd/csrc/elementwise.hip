// HBM-bound kernels of the SGA step: the Gumbel-softmax floor/ceil relaxation, both entropy
// models with their analytic gradients, the distortion, the fused chain-rule + Adam update and
// the per-image reductions.  Each is a single coalesced pass over a latent- or image-sized
// array (<= a few MB per image-step, SURVEY.md 8(d) "algorithmic bytes"); the arithmetic the
// reference expresses as TF graph ops + tf.gradients (sga.py:86-164) is written out in closed
// form here.  Compiled with -ffp-contract=off so the f32 arithmetic matches an unfused
// restatement (Adam is bit-exact vs. the f32-pinned adam.py restatement).
#include <cstdlib>

#include "kernels.h"

namespace {

constexpr float kEps = 1e-5f;          // sga.py:30
constexpr float kLikBound = 1e-9f;     // sga.py:28 / tfc likelihood_bound
constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.3989422804014327f;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of up to 4 doubles; result valid in thread 0
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* sh /* [N][16] */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    v[k] = wave_sum(v[k]);
    if (lane == 0) sh[k * 16 + wid] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      double s = 0.0;
      for (int w = 0; w < nw; ++w) s += sh[k * 16 + w];
      v[k] = s;
    }
  }
}

__device__ __forceinline__ float sigmoidf(float t) { return 1.0f / (1.0f + expf(-t)); }
__device__ __forceinline__ float sgnf(float t) { return (t > 0.f) - (t < 0.f); }

// ------------------------------------------------------------------------------------------
// SGA relaxation (sga.py:86-98 for z, :111-121 for y) + its Jacobian d v_tilde / d v.
//   logits = [-atanh(clip(v-floor)) / T, -atanh(clip(ceil-v)) / T]
//   s = softmax((logits + gumbel) / T)          (tfp RelaxedOneHotCategorical.sample)
//   v_tilde = floor*s0 + ceil*s1
// floor/ceil carry no gradient; clip passes gradient inside [-1+eps, 1-eps].
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sample_one(float x, int64_t idx, const float* __restrict__ u, float T, int it,
                                           unsigned k0, unsigned k1, int stream_id, int mode,
                                           const int4* __restrict__ img_ids, int64_t per_img,
                                           float& vt_out, float& dvt_out) {
  float u0, u1;
  if (u) {
    u0 = u[2 * idx];
    u1 = u[2 * idx + 1];
  } else {
    uint32_t r[4];
    // counter = element index within the REFERENCE batch: image img_ids[b] of it (sga_set_image_ids),
    // so a shard draws the noise its images would have drawn in the un-sharded batch
    // img_ids[b] = {position in the reference batch, that batch's seed (lo, hi), seed valid}: images pooled from several
    // reference batches into one launch each keep the key of their own batch (sga_set_image_seeds)
    int64_t ctr = idx;
    if (img_ids) {
      const int64_t b = idx / per_img;
      const int4 ik = img_ids[b];
      ctr = (int64_t)ik.x * per_img + (idx - b * per_img);
      if (ik.w) { k0 = (unsigned)ik.y; k1 = (unsigned)ik.z; }
    }
    philox4x32_10((uint32_t)ctr, (uint32_t)((uint64_t)ctr >> 32), (uint32_t)it,
                  (uint32_t)stream_id, k0, k1, r);
    u0 = bits_to_uniform(r[0]);
    u1 = bits_to_uniform(r[1]);
  }
  if (mode >= 2) {   // unoise.py:76 / ste.py:78 (identity STE) / map.py: unit Jacobian
    vt_out = mode == 2 ? x + (u0 - 0.5f) : (mode == 3 ? rintf(x) : x);
    dvt_out = 1.0f;
    return;
  }
  const float fl = floorf(x), ce = ceilf(x);
  const float rdn = x - fl, rup = ce - x;
  const float lo = -1.0f + kEps, hi = 1.0f - kEps;
  const float ddn = fminf(fmaxf(rdn, lo), hi);
  const float dup = fminf(fmaxf(rup, lo), hi);
  const float ldn = -atanhf(ddn) / T;
  const float lup = -atanhf(dup) / T;
  const float g0 = -logf(-logf(u0));
  const float g1 = -logf(-logf(u1));
  // danneal.py:83-84: plain softmax of the logits (no noise, no second division by T)
  const float a0 = mode == 1 ? ldn : (ldn + g0) / T, a1 = mode == 1 ? lup : (lup + g1) / T;
  const float mx = fmaxf(a0, a1);
  const float e0 = expf(a0 - mx), e1 = expf(a1 - mx);
  const float den = e0 + e1;
  const float s0 = e0 / den, s1 = e1 / den;
  vt_out = fl * s0 + ce * s1;
  const float mdn = (rdn >= lo && rdn <= hi) ? 1.0f : 0.0f;
  const float mup = (rup >= lo && rup <= hi) ? 1.0f : 0.0f;
  // d(a1 - a0)/dv = (1/T^2) * ( mup/(1-dup^2) + mdn/(1-ddn^2) )
  const float dd = (mup / (1.0f - dup * dup) + mdn / (1.0f - ddn * ddn)) / (mode == 1 ? T : T * T);
  dvt_out = (ce - fl) * s0 * s1 * dd;
}

__device__ __forceinline__ void sample_range(const float* __restrict__ v, const float* __restrict__ u,
                         const StepCtx* __restrict__ ctx, int stream_id, float* __restrict__ vt,
                         float* __restrict__ dvt, int64_t n, int mode,
                         const int4* __restrict__ img_ids, int64_t per_img, int bid, int nblk) {
  const float T = ctx->T;
  const int it = ctx->it;
  const unsigned k0 = ctx->seed_lo, k1 = ctx->seed_hi;
  for (int64_t idx = (int64_t)bid * blockDim.x + threadIdx.x; idx < n;
       idx += (int64_t)nblk * blockDim.x) {
    float a, d;
    sample_one(v[idx], idx, u, T, it, k0, k1, stream_id, mode, img_ids, per_img, a, d);
    vt[idx] = a;
    if (dvt) dvt[idx] = d;
  }
}

__global__ void k_sample(const float* __restrict__ v, const float* __restrict__ u,
                         const StepCtx* __restrict__ ctx, int stream_id, float* __restrict__ vt,
                         float* __restrict__ dvt, int64_t n, int mode,
                         const int4* __restrict__ img_ids, int64_t per_img) {
  sample_range(v, u, ctx, stream_id, vt, dvt, n, mode, img_ids, per_img, blockIdx.x, gridDim.x);
}

// y (stream 0) and z (stream 1) in one launch: blocks [0, gy) relax y, the rest z
__global__ void k_sample_yz(const float* __restrict__ y, float* __restrict__ yt, float* __restrict__ dyt,
                            int64_t ny, const float* __restrict__ z, float* __restrict__ zt,
                            float* __restrict__ dzt, int64_t nz, const StepCtx* __restrict__ ctx, int mode,
                            const int4* __restrict__ img_ids, int B, int gy) {
  if ((int)blockIdx.x < gy)
    sample_range(y, nullptr, ctx, 0, yt, dyt, ny, mode, img_ids, ny / B, blockIdx.x, gy);
  else
    sample_range(z, nullptr, ctx, 1, zt, dzt, nz, mode, img_ids, nz / B, blockIdx.x - gy, gridDim.x - gy);
}

// ------------------------------------------------------------------------------------------
// Factorized prior: per-channel monotone 1->3->3->3->1 network giving the CDF logit
// (learned_prior.py:96-121) and its derivative w.r.t. the input (forward mode).
// Packed per channel (EB_STRIDE floats): m0[3] b0[3] f0[3] m1[9] b1[3] f1[3] m2[9] b2[3] f2[3] m3[3] b3
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void eb_logit(const float* __restrict__ P, float x, float& out,
                                         float& dout) {
  float h[3], d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    h[r] = P[r] * x + P[3 + r];
    d[r] = P[r];
    const float t = tanhf(h[r]);
    const float f = P[6 + r];
    h[r] += f * t;
    d[r] *= 1.0f + f * (1.0f - t * t);
  }
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
    const float* M = P + 9 + layer * 15;
    float h2[3], d2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float acc = 0.f, dacc = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc += M[r * 3 + c] * h[c];
        dacc += M[r * 3 + c] * d[c];
      }
      acc += M[9 + r];
      const float t = tanhf(acc);
      const float f = M[12 + r];
      h2[r] = acc + f * t;
      d2[r] = dacc * (1.0f + f * (1.0f - t * t));
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { h[r] = h2[r]; d[r] = d2[r]; }
  }
  const float* M3 = P + 39;
  float acc = 0.f, dacc = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    acc += M3[c] * h[c];
    dacc += M3[c] * d[c];
  }
  out = acc + M3[3];
  dout = dacc;
}

// tfc EntropyBottleneck._likelihood (sga.py:101): box mass with the sign trick, and dp/dv
__device__ __forceinline__ void eb_mass(const float* __restrict__ P, float x, float& p,
                                        float& dp) {
  float lo, dlo, up, dup_;
  eb_logit(P, x - 0.5f, lo, dlo);
  eb_logit(P, x + 0.5f, up, dup_);
  const float sg = -sgnf(lo + up);
  const float su = sigmoidf(sg * up), sl = sigmoidf(sg * lo);
  const float diff = su - sl;
  p = fabsf(diff);
  dp = sgnf(diff) * sg * (su * (1.0f - su) * dup_ - sl * (1.0f - sl) * dlo);
}

// Second-order forward mode through the same network: logit L, L' and L'' w.r.t. the input
// (learned_prior.py:296-347 gives the first derivative in closed form; the gradient of the
// density needs one more).
__device__ __forceinline__ void eb_logit2(const float* __restrict__ P, float x, float& L,
                                          float& L1, float& L2) {
  float h[3], d[3], dd[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float pre = P[r] * x + P[3 + r];
    const float pre1 = P[r];
    const float t = tanhf(pre);
    const float f = P[6 + r];
    const float g1 = 1.0f + f * (1.0f - t * t);
    const float g2 = f * (-2.0f * t * (1.0f - t * t));
    h[r] = pre + f * t;
    d[r] = pre1 * g1;
    dd[r] = pre1 * pre1 * g2;
  }
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
    const float* M = P + 9 + layer * 15;
    float h2[3], d2[3], e2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float pre = 0.f, pre1 = 0.f, pre2 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        pre += M[r * 3 + c] * h[c];
        pre1 += M[r * 3 + c] * d[c];
        pre2 += M[r * 3 + c] * dd[c];
      }
      pre += M[9 + r];
      const float t = tanhf(pre);
      const float f = M[12 + r];
      const float g1 = 1.0f + f * (1.0f - t * t);
      const float g2 = f * (-2.0f * t * (1.0f - t * t));
      h2[r] = pre + f * t;
      d2[r] = pre1 * g1;
      e2[r] = pre2 * g1 + pre1 * pre1 * g2;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { h[r] = h2[r]; d[r] = d2[r]; dd[r] = e2[r]; }
  }
  const float* M3 = P + 39;
  float a = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    a += M3[c] * h[c];
    a1 += M3[c] * d[c];
    a2 += M3[c] * dd[c];
  }
  L = a + M3[3];
  L1 = a1;
  L2 = a2;
}

// math_ops.py:63-76: gradient of lower_bound passes if x >= bound or the gradient is negative
__device__ __forceinline__ float lower_bound_grad(float x, float bound, float g) {
  return (x >= bound || g < 0.f) ? g : 0.f;
}

__global__ void k_factorized(const float* __restrict__ zt, const float* __restrict__ ebp,
                             const StepCtx* __restrict__ ctx, int n_per_img, int C,
                             float inv_ln2_hw, ImgSums* __restrict__ sums,
                             float* __restrict__ g_zt, float* __restrict__ p_out,
                             float* __restrict__ dp_out) {
#if SGA_SIDE_ELEM_PRIO
  __builtin_amdgcn_s_setprio(SGA_SIDE_ELEM_PRIO);
#endif
  __shared__ double sh[16];
  const int b = blockIdx.y;
  const float ls = ctx ? ctx->loss_scale : 1.0f;
  double nats[1] = {0.0};
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_per_img; e += gridDim.x * blockDim.x) {
    const size_t idx = (size_t)b * n_per_img + e;
    const int c = e % C;
    float p, dp;
    eb_mass(ebp + (size_t)c * EB_STRIDE, zt[idx], p, dp);
    if (p_out) p_out[idx] = p;
    if (dp_out) dp_out[idx] = dp;
    const float pb = fmaxf(p, kLikBound);
    nats[0] += (double)(-logf(pb));
    if (g_zt) {
      // d rd_loss / d p_bounded = -loss_scale / (ln2 * HW * p_bounded)
      const float gp = lower_bound_grad(p, kLikBound, -ls * inv_ln2_hw / pb);
      g_zt[idx] = gp * dp;
    }
  }
  block_sum<1>(nats, sh);
  if (threadIdx.x == 0 && sums) atomicAdd(&sums[b].z_nats, nats[0]);
}

__global__ void k_factorized_pdf(const float* __restrict__ zt, const float* __restrict__ ebp,
                                 const StepCtx* __restrict__ ctx, int n_per_img, int C,
                                 float inv_ln2_hw, ImgSums* __restrict__ sums,
                                 float* __restrict__ g_zt, float* __restrict__ p_out,
                                 float* __restrict__ dp_out) {
  __shared__ double sh[16];
  const int b = blockIdx.y;
  const float ls = ctx ? ctx->loss_scale : 1.0f;
  double nats[1] = {0.0};
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_per_img; e += gridDim.x * blockDim.x) {
    const size_t idx = (size_t)b * n_per_img + e;
    const int c = e % C;
    float L, L1, L2;
    eb_logit2(ebp + (size_t)c * EB_STRIDE, zt[idx], L, L1, L2);
    const float sg = sigmoidf(L);
    const float sp = sg * (1.0f - sg);
    const float p = sp * L1;                                        // density
    const float dp = sp * (1.0f - 2.0f * sg) * L1 * L1 + sp * L2;   // its derivative
    if (p_out) p_out[idx] = p;
    if (dp_out) dp_out[idx] = dp;
    const float pb = fmaxf(p, kLikBound);
    nats[0] += (double)(-logf(pb));
    if (g_zt) g_zt[idx] = lower_bound_grad(p, kLikBound, -ls * inv_ln2_hw / pb) * dp;
  }
  block_sum<1>(nats, sh);
  if (threadIdx.x == 0 && sums) atomicAdd(&sums[b].z_nats, nats[0]);
}

// z_tilde = eps * exp(.5 logvar) + mean (bb_sga.py:99-100); ln q via utils.py:72-77
__global__ void k_bb_sample_z(const float* __restrict__ zml, const float* __restrict__ eps_in,
                              const StepCtx* __restrict__ ctx, int stream_id, int n_per_img, int C,
                              float* __restrict__ zt, float* __restrict__ jac_lv,
                              ImgSums* __restrict__ sums, const int4* __restrict__ img_ids) {
  __shared__ double sh[16];
  const int b = blockIdx.y;
  const int4 ik = img_ids ? img_ids[b] : int4{b, 0, 0, 0};
  const size_t ctr_base = (size_t)ik.x * n_per_img;
  const int it = ctx->it;
  const unsigned k0 = ik.w ? (unsigned)ik.y : ctx->seed_lo, k1 = ik.w ? (unsigned)ik.z : ctx->seed_hi;
  double acc[1] = {0.0};
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_per_img; e += gridDim.x * blockDim.x) {
    const size_t idx = (size_t)b * n_per_img + e;
    const int c = e % C;
    const size_t pix = idx / C;
    const float mean = zml[pix * 2 * C + c], lv = zml[pix * 2 * C + C + c];
    float ep;
    if (eps_in) {
      ep = eps_in[idx];
    } else {
      uint32_t r[4];
      const size_t ctr = ctr_base + e;
      philox4x32_10((uint32_t)ctr, (uint32_t)((uint64_t)ctr >> 32), (uint32_t)it,
                    (uint32_t)stream_id, k0, k1, r);
      const float u1 = bits_to_uniform(r[2]), u2 = bits_to_uniform(r[3]);
      ep = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795864769f * u2);   // Box-Muller
    }
    const float sd = expf(0.5f * lv);
    const float z = ep * sd + mean;
    zt[idx] = z;
    if (jac_lv) jac_lv[idx] = 0.5f * ep * sd;
    const float dz = z - mean;
    acc[0] += (double)(-0.5f * (dz * dz * expf(-lv) + lv + 1.8378770664093453f));
  }
  block_sum<1>(acc, sh);
  if (threadIdx.x == 0 && sums) atomicAdd(&sums[b].q_ln, acc[0]);
}

__global__ void k_bb_zgrad(const float* __restrict__ ga, const float* __restrict__ gb,
                           const float* __restrict__ jac_lv, const StepCtx* __restrict__ ctx,
                           float inv_ln2_hw, int64_t n, int C, float* __restrict__ g_zml) {
  const float cq = -0.5f * ctx->loss_scale * inv_ln2_hw;     // d(-bpp_back)/d logvar, bb_sga.py:129-139
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    float g = ga[i];
    if (gb) g += gb[i];
    g_zml[pix * 2 * C + c] = g;
    g_zml[pix * 2 * C + C + c] = g * jac_lv[i] + cq;
  }
}

// ------------------------------------------------------------------------------------------
// Gaussian conditional, box-convolved (utils.py:80-102 == tfc GaussianConditional._likelihood):
//   sigma = max(exp(sraw), smin);  v = |y - mu|
//   p = Phi((.5 - v)/sigma) - Phi((-.5 - v)/sigma),  Phi(t) = erfc(-t/sqrt2)/2
// smin = the handle's scale_bound (sga_config.scale_bound / sga_set_scale_bound): 0 = raw sigma, what sga.py:130-133
// evaluates (the tfc layer is never built there); 0.11 = tfc's default bound scale_table[0] of a BUILT layer
// (mbt2018.py:77-80).  exp(.) > 0, so smin = 0 needs no branch: max(sigma, 0) = sigma and the lower_bound gradient
// rule (math_ops.py:63-76) passes everything.
// ------------------------------------------------------------------------------------------
struct GaussOut { float p, dp_dy, dp_dmu, dp_dsig_b, sigma; };

__device__ __forceinline__ GaussOut gauss_mass(float y, float mu, float sraw, float smin) {
  GaussOut o;
  const float sigma = expf(sraw);
  const float sb = fmaxf(sigma, smin);
  const float d = y - mu;
  const float v = fabsf(d);
  const float up = (0.5f - v) / sb, lo = (-0.5f - v) / sb;
  const float cu = 0.5f * erfcf(-kInvSqrt2 * up);
  const float cl = 0.5f * erfcf(-kInvSqrt2 * lo);
  o.p = cu - cl;
  const float phu = kInvSqrt2Pi * expf(-0.5f * up * up);
  const float phl = kInvSqrt2Pi * expf(-0.5f * lo * lo);
  const float dp_dv = (phl - phu) / sb;
  const float sg = sgnf(d);
  o.dp_dy = sg * dp_dv;
  o.dp_dmu = -sg * dp_dv;
  o.dp_dsig_b = (phl * lo - phu * up) / sb;
  o.sigma = sigma;
  return o;
}

__global__ void k_gaussian(const float* __restrict__ yt, const float* __restrict__ ms,
                           const StepCtx* __restrict__ ctx, int h, int w, int hs, int ws, int C,
                           float inv_ln2_hw, float smin, ImgSums* __restrict__ sums,
                           float* __restrict__ g_yt, float* __restrict__ g_ms) {
#if SGA_SIDE_ELEM_PRIO
  __builtin_amdgcn_s_setprio(SGA_SIDE_ELEM_PRIO);
#endif
  __shared__ double sh[16];
  const int b = blockIdx.y;
  const float ls = ctx->loss_scale;
  const int n_per_img = hs * ws * C;
  double nats[1] = {0.0};
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_per_img; e += gridDim.x * blockDim.x) {
    const int c = e % C;
    const int pix = e / C;
    const int j = pix % ws, i = pix / ws;
    const size_t mo = ((size_t)(b * hs + i) * ws + j) * (2 * C) + c;
    if (i < h && j < w) {
      const size_t yo = ((size_t)(b * h + i) * w + j) * C + c;
      const GaussOut o = gauss_mass(yt[yo], ms[mo], ms[mo + C], smin);
      const float pb = fmaxf(o.p, kLikBound);
      nats[0] += (double)(-logf(pb));
      const float gp = lower_bound_grad(o.p, kLikBound, -ls * inv_ln2_hw / pb);
      if (g_yt) g_yt[yo] = gp * o.dp_dy;
      if (g_ms) {
        g_ms[mo] = gp * o.dp_dmu;
        const float gsb = lower_bound_grad(o.sigma, smin, gp * o.dp_dsig_b);
        g_ms[mo + C] = gsb * o.sigma;       // d sigma / d sraw = sigma
      }
    } else if (g_ms) {
      g_ms[mo] = 0.f;
      g_ms[mo + C] = 0.f;
    }
  }
  block_sum<1>(nats, sh);
  if (threadIdx.x == 0) atomicAdd(&sums[b].y_nats, nats[0]);
}

__global__ void k_gaussian_op(const float* __restrict__ y, const float* __restrict__ mu,
                              const float* __restrict__ sraw, int64_t n, float smin, float* __restrict__ p,
                              float* __restrict__ dy, float* __restrict__ dmu,
                              float* __restrict__ dsr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const GaussOut o = gauss_mass(y[i], mu[i], sraw[i], smin);
    if (p) p[i] = o.p;
    if (dy) dy[i] = o.dp_dy;
    if (dmu) dmu[i] = o.dp_dmu;
    // unit form: straight derivative of max(exp(sraw), bound) (no upstream sign available)
    if (dsr) dsr[i] = (o.sigma >= smin) ? o.dp_dsig_b * o.sigma : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// Distortion (sga.py:150-154, 170-174)
// ------------------------------------------------------------------------------------------
__global__ void k_mse(const float* __restrict__ x, const float* __restrict__ xt,
                      const StepCtx* __restrict__ ctx, int H, int W, int Hp, int Wp,
                      ImgSums* __restrict__ sums, float* __restrict__ gpad,
                      float* __restrict__ xq_out) {
  // one thread per pixel (3 channels), grid-stride over the image; f32 partials per thread,
  // f64 across the block, one atomic pair per block
  __shared__ double sh[32];
  const int b = blockIdx.y;
  const int npix = H * W;
  const int n_per_img = npix * 3;
  float coef = 0.f;
  if (ctx && ctx->lambda > 0.f)
    coef = ctx->lambda * 2.0f * 65025.0f * ctx->loss_scale / (float)n_per_img;
  float a0 = 0.f, a1 = 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const size_t idx = ((size_t)b * npix + p) * 3;
    const int i = p / W, j = p - i * W;
    float* gp = gpad ? gpad + ((size_t)(b * Hp + i + 2) * Wp + j + 2) * 3 : nullptr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = x[idx + c], tv = xt[idx + c];
      const float d = xv - tv;
      a0 += d * d;
      const float q = rintf(fminf(fmaxf(tv, 0.f), 1.f) * 255.0f);
      if (xq_out) xq_out[idx + c] = q;
      const float dq = xv * 255.0f - q;
      a1 += dq * dq;
      if (gp) gp[c] = coef * (tv - xv);
    }
  }
  double acc[2] = {(double)a0, (double)a1};
  block_sum<2>(acc, sh);
  if (threadIdx.x == 0) {
    atomicAdd(&sums[b].sq_p[blockIdx.x % kSqSlots], acc[0]);
    atomicAdd(&sums[b].sqq_p[blockIdx.x % kSqSlots], acc[1]);
  }
}

__global__ void k_pad_image(const float* __restrict__ x, int H, int W, int Hp, int Wp,
                            float* __restrict__ xp) {
  const int b = blockIdx.y;
  const int n = H * W * 3;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int c = e % 3, pix = e / 3;
    const int j = pix % W, i = pix / W;
    xp[((size_t)(b * Hp + i + 2) * Wp + j + 2) * 3 + c] = x[(size_t)b * n + e];
  }
}

// ------------------------------------------------------------------------------------------
// chain rule through the sampler + Adam (adam.py:40-56), float32 as numpy 1.17 computed it
// ------------------------------------------------------------------------------------------
__global__ void k_combine(const float* __restrict__ ga, const float* __restrict__ gb,
                          const float* __restrict__ jac, float* __restrict__ g, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = ga[i];
    if (gb) s += gb[i];
    g[i] = s * jac[i];
  }
}

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr_t,
                                            float b1, float omb1, float b2, float omb2,
                                            float eps) {
  const float m_t = (b1 * m) + omb1 * g;
  const float v_t = (b2 * v) + omb2 * (g * g);
  p = p - lr_t * m_t / (sqrtf(v_t) + eps);
  m = m_t;
  v = v_t;
}

__global__ void k_adam_latent(float* __restrict__ p, const float* __restrict__ ga,
                              const float* __restrict__ gb, const float* __restrict__ jac,
                              float* __restrict__ m, float* __restrict__ v, int64_t n,
                              const StepCtx* __restrict__ ctx) {
  const float lr_t = ctx->lr_t;
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = ga[i];
    if (gb) s += gb[i];
    const float g = s * jac[i];
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, g, mm, vv, lr_t, b1, omb1, b2, omb2, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

// y and z in one launch: blocks [0, gy) update y, the rest z (same arithmetic as k_adam_latent)
__global__ void k_adam_latent_yz(float* __restrict__ py, const float* __restrict__ gay, const float* __restrict__ gby,
                                 const float* __restrict__ jy, float* __restrict__ my, float* __restrict__ vy,
                                 int64_t ny, float* __restrict__ pz, const float* __restrict__ gaz,
                                 const float* __restrict__ gbz, const float* __restrict__ jz,
                                 float* __restrict__ mz, float* __restrict__ vz, int64_t nz,
                                 const StepCtx* __restrict__ ctx, int gy) {
  const float lr_t = ctx->lr_t;
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999);
  const bool isy = (int)blockIdx.x < gy;
  float* p = isy ? py : pz; const float* ga = isy ? gay : gaz; const float* gb = isy ? gby : gbz;
  const float* jac = isy ? jy : jz; float* m = isy ? my : mz; float* v = isy ? vy : vz;
  const int64_t n = isy ? ny : nz;
  const int bid = isy ? blockIdx.x : blockIdx.x - gy, nblk = isy ? gy : gridDim.x - gy;
  for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < n; i += (int64_t)nblk * blockDim.x) {
    float s = ga[i];
    if (gb) s += gb[i];
    const float g = s * jac[i];
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, g, mm, vv, lr_t, b1, omb1, b2, omb2, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, float lr_t, float b1, float omb1,
                       float b2, float omb2, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, g[i], mm, vv, lr_t, b1, omb1, b2, omb2, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

__global__ void k_adam_ctx(float* __restrict__ p, const float* __restrict__ g,
                           float* __restrict__ m, float* __restrict__ v, int64_t n,
                           const StepCtx* __restrict__ ctx) {
  const float lr_t = ctx->lr_t;
  const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, g[i], mm, vv, lr_t, 0.9f, omb1, 0.999f, omb2, 1e-8f);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

__global__ void k_set_ctx(StepCtx* ctx, int it, int its, float T, float lr_t, float lambda,
                          float loss_scale, unsigned s_lo, unsigned s_hi) {
  ctx->it = it; ctx->its = its; ctx->T = T; ctx->lr_t = lr_t; ctx->lambda = lambda;
  ctx->loss_scale = loss_scale; ctx->seed_lo = s_lo; ctx->seed_hi = s_hi;
}

__global__ void k_advance_ctx(StepCtx* ctx, const float* __restrict__ Ttab,
                              const float* __restrict__ lrtab) {
  const int it = ctx->it + 1;
  ctx->it = it;
  ctx->T = Ttab[it];
  ctx->lr_t = lrtab[it];
}

// debug (SGA_DEBUG_FORK=1): is the side stream really behind the main stream's k_advance_ctx?
__global__ void k_check_iter(const StepCtx* __restrict__ ctx, int expected_it, int* __restrict__ bad) {
  if (ctx->it != expected_it) atomicAdd(bad, 1);
}

// One-wave kernel placed on either side of a cross-stream event (see rd_forward_backward).
__global__ void k_fence() { __threadfence_system(); }

// debug (SGA_DEBUG_DUMP): order-independent 64-bit checksum of a float buffer
// ctx != null: the slot row is chosen ON THE DEVICE -- out + (ctx->it - 1) * 16 (the context was already advanced by the
// iteration's finalize) -- so that the launch can live in a replayed graph
__global__ void k_checksum(const unsigned* __restrict__ p, int64_t n, unsigned long long* out, const StepCtx* __restrict__ ctx) {
  if (ctx) out += (size_t)(ctx->it > 0 ? ctx->it - 1 : 0) * 16;
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)p[i] * (2654435761ull * (unsigned long long)(i + 1) + 0x9E3779B97F4A7C15ull);
  atomicAdd(out, acc);
}

// debug (SGA_DEBUG_DUMP, A.8): ordering probe.  k_mark (main chain, right after the relaxation kernel) stores the wall clock in
// slot 15 of the iteration's row; k_probe (FIRST kernel of the hyper branch) copies what it sees there into slot 14 and its own
// wall clock into slot 13: a branch kernel that ran before its fork dependency sees 0 (the row is zeroed at run begin).
__global__ void k_mark(unsigned long long* out, const StepCtx* __restrict__ ctx) {
  out[(size_t)ctx->it * 16 + 15] = wall_clock64();
}
__global__ void k_probe(unsigned long long* out, const StepCtx* __restrict__ ctx) {
  const size_t row = (size_t)ctx->it * 16;
  out[row + 14] = __hip_atomic_load(&out[row + 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  out[row + 13] = wall_clock64();
}

// debug: hold a stream for `ticks` of the 100 MHz wall clock (SGA_DEBUG_DELAY_US)
__global__ void k_spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ void k_set_int(int* p, int v) { *p = v; }
__global__ void k_check_int(const int* __restrict__ p, int expected, int* __restrict__ bad) {
  if (*p != expected) atomicAdd(bad, 1);
}

__device__ __forceinline__ void finalize_step_body(ImgSums* sums, StepCtx* __restrict__ ctx, int B, int H,
                                                   int W, float* scalars, float* psnr, float* trace,
                                                   const float* __restrict__ Ttab, const float* __restrict__ lrtab, int lane);

__global__ void k_finalize_step(ImgSums* sums, StepCtx* __restrict__ ctx, int B, int H,
                                int W, float* scalars, float* psnr, float* trace,
                                const float* __restrict__ Ttab, const float* __restrict__ lrtab) {
  if (threadIdx.x >= 64) return;          // one wave, all 64 lanes: the distortion sub-accumulators are folded cooperatively
  finalize_step_body(sums, ctx, B, H, W, scalars, psnr, trace, Ttab, lrtab, threadIdx.x);
}

// The boundary between two SGA iterations in ONE launch: Adam on (y, z) with this iteration's gradients
// (adam.py:40-56), the relaxation of the updated latents for the NEXT iteration (sga.py:86-98, 111-121: its
// temperature Ttab[it + 1] and Philox counter it + 1 are read from the tables, not from the context), and --
// by the workgroup that finishes last (ticket counter), after every workgroup has read the context -- the
// per-iteration scalars and the context of the next iteration (k_finalize_step).  Same arithmetic as
// k_adam_latent_yz / k_sample_yz / k_finalize_step launched one after the other: bit-identical.
__global__ void k_step_boundary(float* __restrict__ py, const float* __restrict__ gay, const float* __restrict__ gby,
                                float* __restrict__ jy, float* __restrict__ my, float* __restrict__ vy,
                                float* __restrict__ yt, int64_t ny, float* __restrict__ pz,
                                const float* __restrict__ gaz, const float* __restrict__ gbz,
                                float* __restrict__ jz, float* __restrict__ mz, float* __restrict__ vz,
                                float* __restrict__ zt, int64_t nz, StepCtx* __restrict__ ctx, int gy, int mode,
                                const int4* __restrict__ img_ids, int B, int H, int W, ImgSums* sums,
                                float* trace, const float* __restrict__ Ttab, const float* __restrict__ lrtab,
                                unsigned* __restrict__ ticket) {
  __shared__ int last;
  {
    const float lr_t = ctx->lr_t;
    const int it_n = ctx->it + 1;
    const float T_n = it_n < ctx->its ? Ttab[it_n] : ctx->T;
    const unsigned k0 = ctx->seed_lo, k1 = ctx->seed_hi;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999);
    const bool isy = (int)blockIdx.x < gy;
    float* p = isy ? py : pz; const float* ga = isy ? gay : gaz; const float* gb = isy ? gby : gbz;
    float* jac = isy ? jy : jz; float* m = isy ? my : mz; float* v = isy ? vy : vz; float* vt = isy ? yt : zt;
    const int64_t n = isy ? ny : nz;
    const int bid = isy ? blockIdx.x : blockIdx.x - gy, nblk = isy ? gy : gridDim.x - gy;
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < n; i += (int64_t)nblk * blockDim.x) {
      float s = ga[i];
      if (gb) s += gb[i];
      const float g = s * jac[i];
      float pp = p[i], mm = m[i], vv = v[i];
      adam_update(pp, g, mm, vv, lr_t, b1, omb1, b2, omb2, eps);
      p[i] = pp; m[i] = mm; v[i] = vv;
      float a, d;
      sample_one(pp, i, nullptr, T_n, it_n, k0, k1, isy ? 0 : 1, mode, img_ids, n / B, a, d);
      vt[i] = a;
      jac[i] = d;
    }
  }
  // No fences: what must be ordered is every workgroup's READ of the context before the last one's WRITE of it.
  // A workgroup takes its ticket after the barrier that follows its loop, i.e. after its context values have
  // been consumed; the relaxed device-scope atomic orders the tickets; the writes below only have to be visible
  // to the next kernel.  (A device-scope release per workgroup = an L2 write-back each: 50 us for this launch.)
  // Two-level ticket: 512 workgroups taking ONE counter serialise at the L2 atomic unit (~45 ns per same-address
  // atomic: the kernel ran 21 us for 5 us of work); 16 group counters (workgroup index mod 16, one cache line each)
  // and a top counter taken by each group's last workgroup cut the longest same-address chain to 32 + 16.
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 15u;
    const unsigned gsize = (gridDim.x >> 4) + ((gridDim.x & 15u) > g ? 1u : 0u);
    const unsigned ngroups = gridDim.x < 16u ? gridDim.x : 16u;
    int lastv = 0;
    if (__hip_atomic_fetch_add(ticket + 32 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      ticket[32 * (1 + g)] = 0;
      if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1) lastv = 1;
    }
    last = lastv;
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    finalize_step_body(sums, ctx, B, H, W, nullptr, nullptr, trace, Ttab, lrtab, threadIdx.x);
    if (threadIdx.x == 0) *ticket = 0;
  }
}

// Called by the 64 lanes of ONE wave (uniform control flow).  8 images per pass, 8 lanes per image: every lane loads two of
// the image's distortion sub-accumulators (ImgSums::sq_p / sqq_p), the image's first lane also its other sums -- ALL
// loads of the pass are in flight together (one memory round trip; the one-thread version paid one per image, because
// each image's loads queued behind the previous image's zeroing stores: 21 -> 14 us for the boundary kernel) -- the
// lanes fold by shuffles, lane 0 adds the images up in index order (the f64 order of the one-thread version), and the
// zeroing stores go out together at the end.
__device__ __forceinline__ void finalize_step_body(ImgSums* sums, StepCtx* __restrict__ ctx, int B, int H,
                                                   int W, float* scalars, float* psnr, float* trace,
                                                   const float* __restrict__ Ttab, const float* __restrict__ lrtab, int lane) {
  static_assert(kSqSlots == 16, "two sub-accumulators per lane, 8 lanes per image");
  const bool l0 = lane == 0;
  const double npx = (double)H * W;
  const double ls = ctx->loss_scale;
  const float lam = ctx->lambda;
  const int it0 = ctx->it, its = ctx->its;
  double sq = 0.0, nats = 0.0, ps = 0.0;
  for (int b0 = 0; b0 < B; b0 += 8) {
    const int b = b0 + (lane >> 3), k = lane & 7;
    const bool live = b < B, first = live && k == 0;
    double pa = 0.0, pc = 0.0, o_sq = 0.0, o_sqq = 0.0, o_nats = 0.0;
    if (live) {
      pa = sums[b].sq_p[k] + sums[b].sq_p[k + 8];
      pc = sums[b].sqq_p[k] + sums[b].sqq_p[k + 8];
    }
    if (first) {
      o_sq = sums[b].sq; o_sqq = sums[b].sq_q;
      o_nats = sums[b].y_nats + sums[b].z_nats + sums[b].q_ln;     // q_ln = 0 outside bits-back
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      pa += __shfl_down(pa, o, 64);
      pc += __shfl_down(pc, o, 64);
    }
    // the image's first lane: its totals and PSNR (sga.py:170-174)
    const double tsq = o_sq + pa, tsqq = o_sqq + pc;
    float pb = 0.f;
    if (first) {
      pb = (float)(10.0 * log10(65025.0 / (tsqq / (npx * 3.0))));
      if (psnr) psnr[b] = pb;
    }
    for (int j = 0; j < 8 && b0 + j < B; ++j) {      // index order, every lane the same values
      sq += __shfl(tsq, 8 * j, 64);
      nats += __shfl(o_nats, 8 * j, 64);
      ps += (double)__shfl(pb, 8 * j, 64);
    }
    if (live) { sums[b].sq_p[k] = 0.0; sums[b].sq_p[k + 8] = 0.0; sums[b].sqq_p[k] = 0.0; sums[b].sqq_p[k + 8] = 0.0; }
    if (first) {
      sums[b].sq = 0.0; sums[b].sq_q = 0.0; sums[b].y_nats = 0.0; sums[b].z_nats = 0.0; sums[b].q_ln = 0.0;
    }
  }
  const float train_mse = (float)(sq * ls / (npx * 3.0) * 65025.0);
  const float train_bpp = (float)(nats * ls / (0.6931471805599453 * npx));
  const float loss = lam > 0.f ? lam * train_mse + train_bpp : train_bpp;
  if (!l0) return;
  if (scalars) { scalars[0] = loss; scalars[1] = train_mse; scalars[2] = train_bpp; }
  if (trace) {
    float* row = trace + (size_t)it0 * 4;
    row[0] = loss; row[1] = train_mse; row[2] = train_bpp; row[3] = (float)(ps / B);
  }
  // last kernel of the iteration: set the step context of the next one (it, T = Ttab[it], lr_t =
  // lrtab[it]; sga.py:211, adam.py:40-42), so that no separate launch has to do it
  if (Ttab) {
    const int it = it0 + 1;
    ctx->it = it;
    if (it < its) { ctx->T = Ttab[it]; ctx->lr_t = lrtab[it]; }
  }
}

// the evaluation kernels (once per run): thread b folds image b's sub-accumulators in slot order
__device__ __forceinline__ void fold_sq(ImgSums& s) {
  double a = s.sq, c = s.sq_q;
  for (int k = 0; k < kSqSlots; ++k) { a += s.sq_p[k]; c += s.sqq_p[k]; s.sq_p[k] = 0.0; s.sqq_p[k] = 0.0; }
  s.sq = a; s.sq_q = c;
}

__global__ void k_finalize_eval(ImgSums* sums, int B, int H, int W, float* metrics) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  fold_sq(sums[b]);
  const double npx = (double)H * W;
  const double mse = sums[b].sq_q / (npx * 3.0);
  const double ybpp = sums[b].y_nats / (0.6931471805599453 * npx);
  const double zbpp = sums[b].z_nats / (0.6931471805599453 * npx);
  float* m = metrics + (size_t)b * 7;
  m[0] = (float)mse;
  m[1] = (float)(10.0 * log10(65025.0 / mse));
  m[2] = __builtin_nanf("");      // msssim: filled by the MS-SSIM stage when enabled
  m[3] = __builtin_nanf("");
  m[4] = (float)(ybpp + zbpp);
  m[5] = (float)ybpp;
  m[6] = (float)zbpp;
  sums[b].sq = 0.0; sums[b].sq_q = 0.0; sums[b].y_nats = 0.0; sums[b].z_nats = 0.0;
  sums[b].q_ln = 0.0;
}

// bb_sga.py:181-182: 8 fields, est_bpp = y_bpp + z_bpp - bpp_back
__global__ void k_finalize_eval_bb(ImgSums* sums, int B, int H, int W, float* metrics) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  fold_sq(sums[b]);
  const double npx = (double)H * W, c = 0.6931471805599453 * npx;
  const double mse = sums[b].sq_q / (npx * 3.0);
  const double ybpp = sums[b].y_nats / c, zbpp = sums[b].z_nats / c, back = -sums[b].q_ln / c;
  float* m = metrics + (size_t)b * 8;
  m[0] = (float)mse;
  m[1] = (float)(10.0 * log10(65025.0 / mse));
  m[2] = __builtin_nanf("");
  m[3] = __builtin_nanf("");
  m[4] = (float)(ybpp + zbpp - back);
  m[5] = (float)ybpp;
  m[6] = (float)zbpp;
  m[7] = (float)back;
  sums[b].sq = 0.0; sums[b].sq_q = 0.0; sums[b].y_nats = 0.0; sums[b].z_nats = 0.0;
  sums[b].q_ln = 0.0;
}

__global__ void k_round(const float* __restrict__ v, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = rintf(v[i]);      // round-half-to-even == np.round (sga.py:240-241)
}

__global__ void k_round_centered(const float* __restrict__ y, const float* __restrict__ ms, int B,
                                 int h, int w, int hs, int ws, int C, float* __restrict__ out) {
  const int64_t n = (int64_t)B * h * w * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    int64_t pix = e / C;
    const int j = (int)(pix % w); pix /= w;
    const int i = (int)(pix % h);
    const int b = (int)(pix / h);
    const float mu = ms[((size_t)(b * hs + i) * ws + j) * (2 * C) + c];
    out[e] = rintf(y[e] - mu) + mu;
  }
}

__global__ void k_round_median(const float* __restrict__ z, const float* __restrict__ med,
                               int64_t n, int C, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float m = med ? med[i % C] : 0.f;
    out[i] = rintf(z[i] - m) + m;
  }
}

__global__ void k_fill(float* __restrict__ p, float val, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    p[i] = val;
}

// device-to-device copy as a kernel on the caller's stream (see launch_copy)
__global__ void k_copy(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

__global__ void k_relu_mask(const float* __restrict__ g, const float* __restrict__ act,
                            float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = act[i] > 0.f ? g[i] : 0.f;
}

inline int grid_for(int64_t n, int block = 256, int cap = 2048) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

#define LAUNCH_RET() return (int)hipGetLastError()

int launch_sample(const float* v, const float* u, const StepCtx* ctx, int stream_id, float* vt,
                  float* dvt, int64_t n, hipStream_t s, int mode, const int4* img_ids, int64_t per_img) {
  hipLaunchKernelGGL(k_sample, dim3(grid_for(n)), dim3(256), 0, s, v, u, ctx, stream_id, vt, dvt, n,
                     mode, img_ids, per_img > 0 ? per_img : n);
  LAUNCH_RET();
}

int launch_factorized(const float* zt, const float* eb_packed, const StepCtx* ctx, int B, int npix,
                      int C, float inv_ln2_hw, ImgSums* sums, float* g_zt, float* p_out,
                      float* dp_out, hipStream_t s) {
  const int n_per_img = npix * C;
  hipLaunchKernelGGL(k_factorized, dim3(grid_for(n_per_img, 256, 256), B), dim3(256), 0, s, zt,
                     eb_packed, ctx, n_per_img, C, inv_ln2_hw, sums, g_zt, p_out, dp_out);
  LAUNCH_RET();
}

int launch_gaussian(const float* yt, const float* ms, const StepCtx* ctx, int B, int h, int w,
                    int hs, int ws, int C, float inv_ln2_hw, float scale_bound, ImgSums* sums, float* g_yt,
                    float* g_ms, hipStream_t s) {
  const int n_per_img = hs * ws * C;
  // <= 32 workgroups per image: each ends in one f64 atomic on the image's y_nats, and same-address atomics
  // serialise at ~45 ns (384 per image cost 17 us)
  hipLaunchKernelGGL(k_gaussian, dim3(grid_for(n_per_img, 256, 32), B), dim3(256), 0, s, yt, ms,
                     ctx, h, w, hs, ws, C, inv_ln2_hw, scale_bound, sums, g_yt, g_ms);
  LAUNCH_RET();
}

int launch_gaussian_op(const float* y, const float* mu, const float* sraw, int64_t n, float scale_bound,
                       float* p, float* dp_dy, float* dp_dmu, float* dp_dsraw, hipStream_t s) {
  hipLaunchKernelGGL(k_gaussian_op, dim3(grid_for(n)), dim3(256), 0, s, y, mu, sraw, n, scale_bound, p, dp_dy,
                     dp_dmu, dp_dsraw);
  LAUNCH_RET();
}

int launch_mse(const float* x, const float* xt, const StepCtx* ctx, int B, int H, int W, int Hp,
               int Wp, ImgSums* sums, float* gpad, float* xq_out, hipStream_t s) {
  hipLaunchKernelGGL(k_mse, dim3(grid_for((int64_t)H * W, 256, 64), B), dim3(256), 0, s, x, xt, ctx,
                     H, W, Hp, Wp, sums, gpad, xq_out);
  LAUNCH_RET();
}

int launch_pad_image(const float* x, int B, int H, int W, int Hp, int Wp, float* xp,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_pad_image, dim3(grid_for((int64_t)H * W * 3, 256, 512), B), dim3(256), 0, s,
                     x, H, W, Hp, Wp, xp);
  LAUNCH_RET();
}

int launch_combine_grad(const float* ga, const float* gb, const float* jac, float* g, int64_t n,
                        hipStream_t s) {
  hipLaunchKernelGGL(k_combine, dim3(grid_for(n)), dim3(256), 0, s, ga, gb, jac, g, n);
  LAUNCH_RET();
}

int launch_adam_latent(float* p, const float* ga, const float* gb, const float* jac, float* m,
                       float* v, int64_t n, const StepCtx* ctx, hipStream_t s) {
  hipLaunchKernelGGL(k_adam_latent, dim3(grid_for(n)), dim3(256), 0, s, p, ga, gb, jac, m, v, n,
                     ctx);
  LAUNCH_RET();
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, double b1,
                double b2, double eps, hipStream_t s) {
  // adam.py:44-45: (1. - beta) is formed in double, then cast to the array dtype (float32)
  const float omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2);
  hipLaunchKernelGGL(k_adam, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, n, lr_t, (float)b1,
                     omb1, (float)b2, omb2, (float)eps);
  LAUNCH_RET();
}

int launch_set_ctx(StepCtx* ctx, int it, int its, float T, float lr_t, float lambda,
                   float loss_scale, uint64_t seed, hipStream_t s) {
  hipLaunchKernelGGL(k_set_ctx, dim3(1), dim3(1), 0, s, ctx, it, its, T, lr_t, lambda, loss_scale,
                     (unsigned)(seed & 0xFFFFFFFFu), (unsigned)(seed >> 32));
  LAUNCH_RET();
}

int launch_advance_ctx(StepCtx* ctx, const float* Ttab, const float* lrtab, hipStream_t s) {
  hipLaunchKernelGGL(k_advance_ctx, dim3(1), dim3(1), 0, s, ctx, Ttab, lrtab);
  LAUNCH_RET();
}

int launch_check_iter(const StepCtx* ctx, int expected_it, int* bad, hipStream_t s) {
  hipLaunchKernelGGL(k_check_iter, dim3(1), dim3(1), 0, s, ctx, expected_it, bad);
  LAUNCH_RET();
}

int launch_fence(hipStream_t s) {
  hipLaunchKernelGGL(k_fence, dim3(64), dim3(64), 0, s);   // >= 1 workgroup per XCD (8 XCDs, round-robin)
  LAUNCH_RET();
}
int launch_checksum(const float* p, int64_t n, unsigned long long* out, hipStream_t s, const StepCtx* ctx) {
  hipLaunchKernelGGL(k_checksum, dim3(grid_for(n, 256, 256)), dim3(256), 0, s, (const unsigned*)p, n, out, ctx);
  LAUNCH_RET();
}
int launch_mark(unsigned long long* out, const StepCtx* ctx, hipStream_t s) {
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, s, out, ctx);
  LAUNCH_RET();
}
int launch_probe(unsigned long long* out, const StepCtx* ctx, hipStream_t s) {
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(1), 0, s, out, ctx);
  LAUNCH_RET();
}
int launch_spin(int us, hipStream_t s) {
  hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, (long long)us * 100);
  LAUNCH_RET();
}
int launch_set_int(int* p, int v, hipStream_t s) {
  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(1), 0, s, p, v);
  LAUNCH_RET();
}
int launch_check_int(const int* p, int expected, int* bad, hipStream_t s) {
  hipLaunchKernelGGL(k_check_int, dim3(1), dim3(1), 0, s, p, expected, bad);
  LAUNCH_RET();
}

int launch_finalize_step(ImgSums* sums, StepCtx* ctx, int B, int H, int W, float* scalars,
                         float* psnr, float* trace, hipStream_t s, const float* Ttab, const float* lrtab) {
  hipLaunchKernelGGL(k_finalize_step, dim3(1), dim3(64), 0, s, sums, ctx, B, H, W, scalars, psnr,
                     trace, Ttab, lrtab);
  LAUNCH_RET();
}

int launch_step_boundary(float* py, const float* gay, const float* gby, float* jy, float* my, float* vy, float* yt,
                         int64_t ny, float* pz, const float* gaz, const float* gbz, float* jz, float* mz, float* vz,
                         float* zt, int64_t nz, StepCtx* ctx, int mode, const int4* img_ids, int B, int H, int W,
                         ImgSums* sums, float* trace, const float* Ttab, const float* lrtab, unsigned* ticket,
                         hipStream_t s) {
  // at most 1024 workgroups (measured 19.7 / 16.7 / 17.3 us at 512 / 1024 / 2048 with the two-level ticket; with ONE
  // counter 1632 workgroups serialised to 25 us); fewer than ~400 leave the relaxation's long dependent chains (Philox,
  // log, atanh, exp) too little parallelism
  int gy = grid_for(ny), gz = grid_for(nz);
  static const int cap = LAB_ENV("SGA_BOUNDARY_CAP") ? atoi(LAB_ENV("SGA_BOUNDARY_CAP")) : 1024;
  if (gy + gz > cap) {
    gz = gz > cap / 8 ? cap / 8 : gz;
    gy = gy > cap - gz ? cap - gz : gy;
  }
  hipLaunchKernelGGL(k_step_boundary, dim3(gy + gz), dim3(256), 0, s, py, gay, gby, jy, my, vy, yt, ny, pz, gaz, gbz,
                     jz, mz, vz, zt, nz, ctx, gy, mode, img_ids, B, H, W, sums, trace, Ttab, lrtab, ticket);
  LAUNCH_RET();
}

int launch_sample_yz(const float* y, float* yt, float* dyt, int64_t ny, const float* z, float* zt, float* dzt,
                     int64_t nz, const StepCtx* ctx, int mode, const int4* img_ids, int B, hipStream_t s) {
  const int gy = grid_for(ny), gz = grid_for(nz);
  hipLaunchKernelGGL(k_sample_yz, dim3(gy + gz), dim3(256), 0, s, y, yt, dyt, ny, z, zt, dzt, nz, ctx, mode,
                     img_ids, B, gy);
  LAUNCH_RET();
}

int launch_adam_latent_yz(float* py, const float* gay, const float* gby, const float* jy, float* my, float* vy,
                          int64_t ny, float* pz, const float* gaz, const float* gbz, const float* jz,
                          float* mz, float* vz, int64_t nz, const StepCtx* ctx, hipStream_t s) {
  const int gy = grid_for(ny), gz = grid_for(nz);
  hipLaunchKernelGGL(k_adam_latent_yz, dim3(gy + gz), dim3(256), 0, s, py, gay, gby, jy, my, vy, ny, pz, gaz,
                     gbz, jz, mz, vz, nz, ctx, gy);
  LAUNCH_RET();
}

int launch_finalize_eval(ImgSums* sums, int B, int H, int W, float* metrics, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize_eval, dim3((B + 63) / 64), dim3(64), 0, s, sums, B, H, W, metrics);
  LAUNCH_RET();
}

int launch_round(const float* v, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_round, dim3(grid_for(n)), dim3(256), 0, s, v, out, n);
  LAUNCH_RET();
}

int launch_round_centered(const float* y, const float* ms, int B, int h, int w, int hs, int ws,
                          int C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_round_centered, dim3(grid_for((int64_t)B * h * w * C)), dim3(256), 0, s, y,
                     ms, B, h, w, hs, ws, C, out);
  LAUNCH_RET();
}

int launch_round_median(const float* z, const float* med, int64_t n, int C, float* out,
                        hipStream_t s) {
  hipLaunchKernelGGL(k_round_median, dim3(grid_for(n)), dim3(256), 0, s, z, med, n, C, out);
  LAUNCH_RET();
}

int launch_fill(float* p, float val, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(256), 0, s, p, val, n);
  LAUNCH_RET();
}

// Copies and clears inside the step sequences are kernels on the caller's compute queue rather
// than hipMemcpyAsync/hipMemsetAsync (which the runtime may route to a copy engine or a blit
// kernel of its own): one ordering domain, and capturable into the step graph as plain kernel nodes.
int launch_copy(float* dst, const float* src, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy, dim3(grid_for(n)), dim3(256), 0, s, dst, src, n);
  LAUNCH_RET();
}

int launch_relu_mask(const float* g, const float* act, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_relu_mask, dim3(grid_for(n)), dim3(256), 0, s, g, act, out, n);
  LAUNCH_RET();
}

int launch_bb_sample_z(const float* zml, const float* eps_in, const StepCtx* ctx, int stream_id,
                       int B, int npix, int C, float* zt, float* jac_lv, ImgSums* sums,
                       hipStream_t s, const int4* img_ids) {
  const int n_per_img = npix * C;
  hipLaunchKernelGGL(k_bb_sample_z, dim3(grid_for(n_per_img, 256, 256), B), dim3(256), 0, s, zml,
                     eps_in, ctx, stream_id, n_per_img, C, zt, jac_lv, sums, img_ids);
  LAUNCH_RET();
}

int launch_factorized_pdf(const float* zt, const float* eb_packed, const StepCtx* ctx, int B,
                          int npix, int C, float inv_ln2_hw, ImgSums* sums, float* g_zt,
                          float* p_out, float* dp_out, hipStream_t s) {
  const int n_per_img = npix * C;
  hipLaunchKernelGGL(k_factorized_pdf, dim3(grid_for(n_per_img, 256, 256), B), dim3(256), 0, s, zt,
                     eb_packed, ctx, n_per_img, C, inv_ln2_hw, sums, g_zt, p_out, dp_out);
  LAUNCH_RET();
}

int launch_bb_zgrad(const float* ga, const float* gb, const float* jac_lv, const StepCtx* ctx,
                    float inv_ln2_hw, int64_t npix_total, int C, float* g_zml, hipStream_t s) {
  const int64_t n = npix_total * C;
  hipLaunchKernelGGL(k_bb_zgrad, dim3(grid_for(n)), dim3(256), 0, s, ga, gb, jac_lv, ctx,
                     inv_ln2_hw, n, C, g_zml);
  LAUNCH_RET();
}

int launch_adam_ctx(float* p, const float* g, float* m, float* v, int64_t n, const StepCtx* ctx,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_adam_ctx, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, n, ctx);
  LAUNCH_RET();
}

int launch_finalize_eval_bb(ImgSums* sums, int B, int H, int W, float* metrics8, hipStream_t s) {
  hipLaunchKernelGGL(k_finalize_eval_bb, dim3((B + 63) / 64), dim3(64), 0, s, sums, B, H, W,
                     metrics8);
  LAUNCH_RET();
}
