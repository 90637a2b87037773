// Entropy coding of the quantised latents on the device (mbt2018.py:84-85, 211-222: the reference
// calls tfc's C++ range-coder ops there).  The coder is the rANS of csrc_cpu/rans.c (32-bit state, byte
// renormalisation, 16-bit probabilities, escape symbol + two raw 16-bit chunks) over BLOCKED streams: symbols are cut
// into blocks of `block`, each an independent stream, and one lane encodes / decodes one block -- byte for byte what
// rans_encode_blocked / rans_decode_blocked produce on the host (tests/test_gpu_entropy.py).  The work is tiny (one
// 256^2 image: 49 152 + 3 072 symbols) and serial per block; what matters is that (mu, sigma) -> table index ->
// bytes stays on the device next to the kernels that produced (mu, sigma), and that the result is bit-exact.
#include <stdint.h>

#include "../../include/sga_hip.h"
#include "sga_common.h"

namespace {

constexpr unsigned kScaleBits = 16;
constexpr unsigned kRansL = 1u << 23;

__device__ __forceinline__ uint8_t* rans_put(unsigned& x, uint8_t* p, unsigned start, unsigned freq) {
  const unsigned x_max = ((kRansL >> kScaleBits) << 8) * freq;
  unsigned v = x;
  while (v >= x_max) { *--p = (uint8_t)(v & 0xff); v >>= 8; }
  x = ((v / freq) << kScaleBits) + (v % freq) + start;
  return p;
}

// y_hat - round(mu) and the table of every element (entropy_coding.EntropyCoder._y_symbols): scale level = number of
// table scales below max(sigma, scale_table[0]) (np.searchsorted 'left'), mean bin = floor((mu - rint(mu) + .5) * bins);
// comparisons in double like the numpy original.  y == null: table / r0 only (decoder side).
__global__ void k_y_symbols(const float* __restrict__ y, const float* __restrict__ mu, const float* __restrict__ sigma,
                            int64_t n, const double* __restrict__ scales, int levels, int bins, int tab0,
                            int* __restrict__ sym, int* __restrict__ tab, int* __restrict__ r0_out, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = mu[i];
    const float r0 = rintf(m);
    const double frac = (double)(m - r0);
    int j = (int)floor((frac + 0.5) * bins);
    j = j < 0 ? 0 : (j > bins - 1 ? bins - 1 : j);
    double sg = (double)sigma[i];
    sg = sg > scales[0] ? sg : scales[0];
    int lo = 0, hi = levels;                       // first index with scales[idx] >= sg
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (scales[mid] < sg) lo = mid + 1; else hi = mid; }
    const int lvl = lo > levels - 1 ? levels - 1 : lo;
    tab[i] = tab0 + lvl * bins + j;
    if (r0_out) r0_out[i] = (int)r0;
    if (y) {
      const float yv = y[i];
      if (rintf(yv) != yv && bad) atomicAdd(bad, 1);            // the coder codes integers (sga.py:240-241)
      sym[i] = (int)rintf(yv) - (int)r0;
    }
  }
}

// The same for MEAN-CENTRED latents (mbt2018.py:80 through tfc's conditional bottleneck: y_hat = round(y - mu) + mu): the symbol
// is round(y_hat - mu), coded under the zero-offset table of its scale level (one table per level: tab0 + level).  `bad` counts
// elements whose y_hat - mu is not an integer to 1e-3 (float32 noise of the + mu is ~1e-6 at |y| < 100).
__global__ void k_y_symbols_centred(const float* __restrict__ y, const float* __restrict__ mu, const float* __restrict__ sigma,
                                    int64_t n, const double* __restrict__ scales, int levels, int tab0, int* __restrict__ sym,
                                    int* __restrict__ tab, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double sg = (double)sigma[i];
    sg = sg > scales[0] ? sg : scales[0];
    int lo = 0, hi = levels;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (scales[mid] < sg) lo = mid + 1; else hi = mid; }
    tab[i] = tab0 + (lo > levels - 1 ? levels - 1 : lo);
    if (y) {
      const float d = y[i] - mu[i];
      const float r = rintf(d);
      if (fabsf(d - r) > 1e-3f && bad) atomicAdd(bad, 1);
      sym[i] = (int)r;
    }
  }
}

// ... and z_hat = round(z - median_c) + median_c (mbt2018.py:69 through tfc's EntropyBottleneck): symbol round(z_hat - median_c)
__global__ void k_z_symbols_centred(const float* __restrict__ z, const float* __restrict__ med, int64_t n, int C,
                                    int* __restrict__ sym, int* __restrict__ tab, int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    tab[i] = c;
    if (z) {
      const float d = z[i] - med[c];
      const float r = rintf(d);
      if (fabsf(d - r) > 1e-3f && bad) atomicAdd(bad, 1);
      sym[i] = (int)r;
    }
  }
}

// z: integer latent [.., C] with one table per channel
__global__ void k_z_symbols(const float* __restrict__ z, int64_t n, int C, int* __restrict__ sym, int* __restrict__ tab,
                            int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    tab[i] = (int)(i % C);
    if (z) {
      const float v = z[i];
      if (rintf(v) != v && bad) atomicAdd(bad, 1);
      sym[i] = (int)rintf(v);
    }
  }
}

// one lane = one block of symbols, written backwards into its own slot [slot_cap] (as rans_encode writes at the end
// of its buffer); block_bytes[b] = bytes used (0 = overflow)
__global__ void k_rans_encode(const int* __restrict__ sym, const int* __restrict__ tab, int64_t n, int block,
                              const unsigned* __restrict__ cdf, const int* __restrict__ lens,
                              const int* __restrict__ offs, int stride, uint8_t* __restrict__ slots, int slot_cap,
                              unsigned* __restrict__ block_bytes, int nblocks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const int64_t s0 = (int64_t)b * block;
  const int m = (int)(n - s0 < block ? n - s0 : block);
  uint8_t* const out = slots + (size_t)b * slot_cap;
  uint8_t* p = out + slot_cap;
  unsigned x = kRansL;
  bool ok = true;
  for (int k = m; k-- > 0;) {
    if (p - out < 16) { ok = false; break; }
    const int t = tab[s0 + k];
    const unsigned* c = cdf + (size_t)t * stride;
    const int len = lens[t];
    const int sv = sym[s0 + k];
    const int idx = sv - offs[t];
    if (idx >= 0 && idx < len - 1) {
      p = rans_put(x, p, c[idx], c[idx + 1] - c[idx]);
    } else {
      const unsigned z = ((unsigned)sv << 1) ^ (unsigned)(sv >> 31);
      p = rans_put(x, p, z >> 16, 1);
      p = rans_put(x, p, z & 0xffff, 1);
      p = rans_put(x, p, c[len - 1], c[len] - c[len - 1]);
    }
  }
  if (ok) {
    p -= 4;
    p[0] = (uint8_t)(x >> 24); p[1] = (uint8_t)(x >> 16); p[2] = (uint8_t)(x >> 8); p[3] = (uint8_t)x;
  }
  block_bytes[b] = ok ? (unsigned)(out + slot_cap - p) : 0u;
}

// slots -> one contiguous stream: block b's bytes go to out + block_off[b]
__global__ void k_rans_compact(const uint8_t* __restrict__ slots, int slot_cap, const unsigned* __restrict__ block_bytes,
                               const unsigned long long* __restrict__ block_off, uint8_t* __restrict__ out) {
  const int b = blockIdx.x;
  const unsigned nb = block_bytes[b];
  const uint8_t* src = slots + (size_t)b * slot_cap + (slot_cap - nb);
  uint8_t* dst = out + block_off[b];
  for (unsigned i = threadIdx.x; i < nb; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ unsigned rans_get_raw(unsigned& x, const uint8_t*& p, const uint8_t* end) {
  const unsigned s = x & 0xffff;
  unsigned v = x >> kScaleBits;
  while (v < kRansL && p < end) v = (v << 8) | *p++;
  x = v;
  return s;
}

__global__ void k_rans_decode(const uint8_t* __restrict__ in, const unsigned long long* __restrict__ block_off,
                              const unsigned* __restrict__ block_bytes, int nblocks, const int* __restrict__ tab,
                              int64_t n, int block, const unsigned* __restrict__ cdf, const int* __restrict__ lens,
                              const int* __restrict__ offs, int stride, int* __restrict__ sym, int* __restrict__ bad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const int64_t s0 = (int64_t)b * block;
  const int m = (int)(n - s0 < block ? n - s0 : block);
  const uint8_t* src = in + block_off[b];
  const unsigned len_b = block_bytes[b];
  if (len_b < 4) { atomicAdd(bad, 1); return; }
  const uint8_t* p = src + 4;
  const uint8_t* end = src + len_b;
  unsigned x = ((unsigned)src[0] << 24) | ((unsigned)src[1] << 16) | ((unsigned)src[2] << 8) | src[3];
  for (int k = 0; k < m; ++k) {
    const int t = tab[s0 + k];
    const unsigned* c = cdf + (size_t)t * stride;
    const int len = lens[t];
    const unsigned s = x & 0xffff;
    int lo = 0, hi = len;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c[mid] <= s) lo = mid; else hi = mid; }
    const unsigned start = c[lo], freq = c[lo + 1] - c[lo];
    if (freq == 0) { atomicAdd(bad, 1); return; }
    x = freq * (x >> kScaleBits) + s - start;
    while (x < kRansL && p < end) x = (x << 8) | *p++;
    if (lo < len - 1) {
      sym[s0 + k] = offs[t] + lo;
    } else {
      const unsigned zl = rans_get_raw(x, p, end);
      const unsigned zh = rans_get_raw(x, p, end);
      const unsigned z = (zh << 16) | zl;
      sym[s0 + k] = (int)((z >> 1) ^ (unsigned)(-(int)(z & 1)));
    }
  }
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" {

int sga_ec_y_symbols(const float* y_hat, const float* mu, const float* sigma, int64_t n, const double* scale_table,
                     int levels, int mean_bins, int y_tab0, int32_t* sym, int32_t* tab, int32_t* r0, int32_t* bad,
                     void* stream) {
  if (!mu || !sigma || !scale_table || !tab || n <= 0 || levels <= 0 || mean_bins <= 0 || (y_hat && !sym)) return SGA_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_y_symbols, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y_hat, mu, sigma, n,
                     scale_table, levels, mean_bins, y_tab0, sym, tab, r0, bad);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

int sga_ec_z_symbols(const float* z_hat, int64_t n, int num_filters, int32_t* sym, int32_t* tab, int32_t* bad,
                     void* stream) {
  if (!tab || n <= 0 || num_filters <= 0 || (z_hat && !sym)) return SGA_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_z_symbols, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z_hat, n, num_filters, sym, tab, bad);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

int sga_ec_y_symbols_centred(const float* y_hat, const float* mu, const float* sigma, int64_t n, const double* scale_table,
                             int levels, int y_tab0, int32_t* sym, int32_t* tab, int32_t* bad, void* stream) {
  if (!sigma || !scale_table || !tab || n <= 0 || levels <= 0 || (y_hat && (!sym || !mu))) return SGA_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_y_symbols_centred, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y_hat, mu, sigma, n,
                     scale_table, levels, y_tab0, sym, tab, bad);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

int sga_ec_z_symbols_centred(const float* z_hat, const float* medians, int64_t n, int num_filters, int32_t* sym, int32_t* tab,
                             int32_t* bad, void* stream) {
  if (!tab || n <= 0 || num_filters <= 0 || (z_hat && (!sym || !medians))) return SGA_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_z_symbols_centred, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, z_hat, medians, n,
                     num_filters, sym, tab, bad);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

int sga_ec_encode(const int32_t* sym, const int32_t* tab, int64_t n, int block, const uint32_t* cdf, const int32_t* lens,
                  const int32_t* offs, int stride, uint8_t* slots, int slot_cap, uint32_t* block_bytes, void* stream) {
  if (!sym || !tab || !cdf || !lens || !offs || !slots || !block_bytes || n <= 0 || block <= 0 || slot_cap < 16)
    return SGA_ERR_BAD_ARG;
  const int nblocks = (int)((n + block - 1) / block);
  hipLaunchKernelGGL(k_rans_encode, dim3((nblocks + 63) / 64), dim3(64), 0, (hipStream_t)stream, sym, tab, n, block,
                     cdf, lens, offs, stride, slots, slot_cap, block_bytes, nblocks);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

int sga_ec_compact(const uint8_t* slots, int slot_cap, const uint32_t* block_bytes, const uint64_t* block_off,
                   int nblocks, uint8_t* out, void* stream) {
  if (!slots || !block_bytes || !block_off || !out || nblocks <= 0) return SGA_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_rans_compact, dim3(nblocks), dim3(64), 0, (hipStream_t)stream, slots, slot_cap, block_bytes,
                     (const unsigned long long*)block_off, out);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

int sga_ec_decode(const uint8_t* in, const uint64_t* block_off, const uint32_t* block_bytes, int nblocks,
                  const int32_t* tab, int64_t n, int block, const uint32_t* cdf, const int32_t* lens, const int32_t* offs,
                  int stride, int32_t* sym, int32_t* bad, void* stream) {
  if (!in || !block_off || !block_bytes || !tab || !cdf || !lens || !offs || !sym || !bad || n <= 0 || block <= 0 ||
      nblocks != (int)((n + block - 1) / block))
    return SGA_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_rans_decode, dim3((nblocks + 63) / 64), dim3(64), 0, (hipStream_t)stream, in,
                     (const unsigned long long*)block_off, block_bytes, nblocks, tab, n, block, cdf, lens, offs, stride,
                     sym, bad);
  return hipGetLastError() == hipSuccess ? SGA_OK : SGA_ERR_HIP;
}

}  // extern "C"
