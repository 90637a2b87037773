// MS-SSIM of the final evaluation (sga.py:175 `tf.image.ssim_multiscale(x_tilde, x, 255)`), TF 1.15
// defaults: 5 scales, 11x11 Gaussian (sigma 1.5) VALID windows, k1 = 0.01, k2 = 0.03, 2x2 average
// pooling with symmetric end-padding, relu, weighted geometric mean, mean over channels.
// Runs once per batch (not on the 2000-step loop): plain coalesced kernels, f64 reductions.
#include "kernels.h"

namespace {

__constant__ float kGauss[11];   // normalised 1-D Gaussian; the 2-D window is its outer product

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// per (image b, channel c): sums over the VALID window positions of luminance*cs and cs
__global__ void k_ssim_stats(const float* __restrict__ a, const float* __restrict__ b2, float sa,
                             float sb, int h, int w, double* __restrict__ stats /*[B][3][2]*/) {
  __shared__ double sh[6][4];
  const int b = blockIdx.y;
  const int hv = h - 10, wv = w - 10;
  const int n = hv * wv * 3;
  const float c1 = 6.5025f, c2 = 58.5225f;     // (0.01*255)^2, (0.03*255)^2
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int c = e % 3;
    const int p = e / 3;
    const int ox = p % wv, oy = p / wv;
    float m0 = 0.f, m1 = 0.f, sxy = 0.f, sq = 0.f;
    for (int i = 0; i < 11; ++i) {
      const size_t row = ((size_t)(b * h + oy + i) * w + ox) * 3 + c;
      float r0 = 0.f, r1 = 0.f, rxy = 0.f, rq = 0.f;
#pragma unroll
      for (int j = 0; j < 11; ++j) {
        const float x = a[row + j * 3] * sa, y = b2[row + j * 3] * sb;
        const float wj = kGauss[j];
        r0 += wj * x; r1 += wj * y; rxy += wj * (x * y); rq += wj * (x * x + y * y);
      }
      const float wi = kGauss[i];
      m0 += wi * r0; m1 += wi * r1; sxy += wi * rxy; sq += wi * rq;
    }
    const float num0 = m0 * m1 * 2.0f, den0 = m0 * m0 + m1 * m1;
    const float lum = (num0 + c1) / (den0 + c1);
    const float cs = (sxy * 2.0f - num0 + c2) / (sq - den0 + c2);
    acc[c * 2] += (double)(lum * cs);
    acc[c * 2 + 1] += (double)cs;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double s = wave_sum_d(acc[k]);
    if (lane == 0) sh[k][wid] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double s = 0.0;
    for (int wv_ = 0; wv_ < (int)(blockDim.x >> 6); ++wv_) s += sh[threadIdx.x][wv_];
    atomicAdd(&stats[(size_t)b * 6 + threadIdx.x], s);
  }
}

// 2x2 average pool, stride 2, after SYMMETRIC end-padding of odd sizes (= clamp)
__global__ void k_down2(const float* __restrict__ in, float s, int h, int w, int ho, int wo,
                        float* __restrict__ out) {
  const int b = blockIdx.y;
  const int n = ho * wo * 3;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int c = e % 3, p = e / 3;
    const int j = p % wo, i = p / wo;
    const int i0 = 2 * i, i1 = min(2 * i + 1, h - 1), j0 = 2 * j, j1 = min(2 * j + 1, w - 1);
    const float* base = in + (size_t)b * h * w * 3 + c;
    const float v = base[((size_t)i0 * w + j0) * 3] + base[((size_t)i0 * w + j1) * 3] +
                    base[((size_t)i1 * w + j0) * 3] + base[((size_t)i1 * w + j1) * 3];
    out[(size_t)b * n + e] = 0.25f * v * s;
  }
}

__global__ void k_msssim_final(const double* __restrict__ stats /*[5][B][3][2]*/, int B,
                               const int* __restrict__ counts /*[5]*/, float* __restrict__ metrics,
                               int mstride) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double pw[5] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
  double mean = 0.0;
  for (int c = 0; c < 3; ++c) {
    double prod = 1.0;
    for (int k = 0; k < 5; ++k) {
      const double* s = stats + ((size_t)(k * B + b) * 3 + c) * 2;
      const double v = (k < 4 ? s[1] : s[0]) / counts[k];     // cs for scales 0..3, ssim for the last
      prod *= pow(v > 0.0 ? v : 0.0, pw[k]);
    }
    mean += prod / 3.0;
  }
  metrics[(size_t)b * mstride + 2] = (float)mean;
  metrics[(size_t)b * mstride + 3] = (float)(-10.0 * log10(1.0 - mean));
}

}  // namespace

int msssim_init() {
  float w[11];
  double sum = 0.0;
  for (int i = 0; i < 11; ++i) { w[i] = (float)exp(-0.5 * (i - 5.0) * (i - 5.0) / (1.5 * 1.5)); sum += w[i]; }
  for (int i = 0; i < 11; ++i) w[i] = (float)(w[i] / sum);
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(kGauss), w, sizeof(w));
}

// xq: reconstruction already rounded to 0..255; x: original in [0,1].  lvl[k] (k = 1..4) are
// pairs of scratch images; stats: 5*B*6 doubles; counts: 5 ints (device).  metrics[b][2..3].
int launch_msssim(const float* xq, const float* x, int B, int H, int W, float* const* lvlA,
                  float* const* lvlB, double* stats, int* counts_dev, float* metrics, int mstride,
                  hipStream_t s) {
  int hs[5], ws[5], cnt[5];
  hs[0] = H; ws[0] = W;
  for (int k = 1; k < 5; ++k) { hs[k] = (hs[k - 1] + 1) / 2; ws[k] = (ws[k - 1] + 1) / 2; }
  for (int k = 0; k < 5; ++k) cnt[k] = (hs[k] - 10) * (ws[k] - 10);
  if (hs[4] < 11 || ws[4] < 11) return (int)hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 5 * B * 6, s);
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(counts_dev, cnt, sizeof(cnt), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return (int)e;
  const float* a = xq; const float* b = x;
  float sa = 1.f, sb = 255.f;
  for (int k = 0; k < 5; ++k) {
    if (k > 0) {
      const int n = hs[k] * ws[k] * 3;
      int g = (n + 255) / 256; if (g > 1024) g = 1024;
      hipLaunchKernelGGL(k_down2, dim3(g, B), dim3(256), 0, s, a, sa, hs[k - 1], ws[k - 1], hs[k], ws[k], lvlA[k]);
      hipLaunchKernelGGL(k_down2, dim3(g, B), dim3(256), 0, s, b, sb, hs[k - 1], ws[k - 1], hs[k], ws[k], lvlB[k]);
      a = lvlA[k]; b = lvlB[k]; sa = 1.f; sb = 1.f;
    }
    const int n = cnt[k] * 3;
    int g = (n + 255) / 256; if (g > 512) g = 512;
    hipLaunchKernelGGL(k_ssim_stats, dim3(g, B), dim3(256), 0, s, a, b, sa, sb, hs[k], ws[k],
                       stats + (size_t)k * B * 6);
  }
  hipLaunchKernelGGL(k_msssim_final, dim3((B + 63) / 64), dim3(64), 0, s, stats, B, counts_dev, metrics, mstride);
  return (int)hipGetLastError();
}
