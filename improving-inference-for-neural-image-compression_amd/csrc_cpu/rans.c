/* rANS entropy coder core (32-bit state, byte renormalisation, 16-bit probabilities) for the
 * quantised latents (y_hat, z_hat).  Counterpart of the C++ range-coder ops inside
 * tensorflow-compression that mbt2018.py:84-85 calls (`entropy_bottleneck.compress`,
 * `conditional_bottleneck.compress`); the container is this build's own, not .tfci.
 *
 * Each symbol i is coded with table tab[i]: cdf row (prefix sums, total 1<<16), `len` entries
 * (= len-1 regular symbols for values off .. off+len-2, plus a final ESCAPE symbol); an escaped
 * value follows as two raw 16-bit chunks of its zig-zag code.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define SCALE_BITS 16
#define RANS_L (1u << 23)

static inline uint8_t* put(uint32_t* x, uint8_t* p, uint32_t start, uint32_t freq) {
  const uint32_t x_max = ((RANS_L >> SCALE_BITS) << 8) * freq;
  uint32_t v = *x;
  while (v >= x_max) { *--p = (uint8_t)(v & 0xff); v >>= 8; }
  *x = ((v / freq) << SCALE_BITS) + (v % freq) + start;
  return p;
}

/* returns number of bytes written at the END of `out` (out + cap - n .. out + cap), or 0 on overflow */
size_t rans_encode(const int32_t* sym, const int32_t* tab, size_t n, const uint32_t* cdf,
                   const int32_t* lens, const int32_t* offs, int stride, uint8_t* out, size_t cap) {
  if (cap < 16) return 0;
  uint8_t* p = out + cap;
  uint32_t x = RANS_L;
  for (size_t k = n; k-- > 0;) {
    if ((size_t)(p - out) < 16) return 0;
    const int t = tab[k];
    const uint32_t* c = cdf + (size_t)t * stride;
    const int len = lens[t];
    const int32_t idx = sym[k] - offs[t];
    if (idx >= 0 && idx < len - 1) {
      p = put(&x, p, c[idx], c[idx + 1] - c[idx]);
    } else {                                     /* escape: raw zig-zag value AFTER the escape symbol */
      const uint32_t z = ((uint32_t)sym[k] << 1) ^ (uint32_t)(sym[k] >> 31);
      p = put(&x, p, z >> 16, 1);                /* decoded second  */
      p = put(&x, p, z & 0xffff, 1);             /* decoded first   */
      p = put(&x, p, c[len - 1], c[len] - c[len - 1]);
    }
  }
  p -= 4;
  p[0] = (uint8_t)(x >> 24); p[1] = (uint8_t)(x >> 16); p[2] = (uint8_t)(x >> 8); p[3] = (uint8_t)x;
  return (size_t)(out + cap - p);
}

static inline uint32_t get_raw(uint32_t* x, const uint8_t** p, const uint8_t* end) {
  const uint32_t s = *x & 0xffff;
  uint32_t v = (*x >> SCALE_BITS) + 0 * s;       /* freq 1, start s: x = 1*(x>>16) + s - s */
  while (v < RANS_L && *p < end) v = (v << 8) | *(*p)++;
  *x = v;
  return s;
}

/* returns 0 on success, -1 on a corrupt / truncated stream */
int rans_decode(const uint8_t* in, size_t in_len, const int32_t* tab, size_t n, const uint32_t* cdf,
                const int32_t* lens, const int32_t* offs, int stride, int32_t* sym) {
  if (in_len < 4) return -1;
  const uint8_t* p = in + 4;
  const uint8_t* end = in + in_len;
  uint32_t x = ((uint32_t)in[0] << 24) | ((uint32_t)in[1] << 16) | ((uint32_t)in[2] << 8) | in[3];
  for (size_t k = 0; k < n; ++k) {
    const int t = tab[k];
    const uint32_t* c = cdf + (size_t)t * stride;
    const int len = lens[t];
    const uint32_t s = x & 0xffff;
    int lo = 0, hi = len;                         /* largest idx with c[idx] <= s */
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c[mid] <= s) lo = mid; else hi = mid; }
    const uint32_t start = c[lo], freq = c[lo + 1] - c[lo];
    if (freq == 0) return -1;
    x = freq * (x >> SCALE_BITS) + s - start;
    while (x < RANS_L && p < end) x = (x << 8) | *p++;
    if (lo < len - 1) {
      sym[k] = offs[t] + lo;
    } else {
      const uint32_t zl = get_raw(&x, &p, end);
      const uint32_t zh = get_raw(&x, &p, end);
      const uint32_t z = (zh << 16) | zl;
      sym[k] = (int32_t)((z >> 1) ^ (uint32_t)(-(int32_t)(z & 1)));
    }
  }
  return 0;
}

/* ---- blocked streams ---------------------------------------------------------------------------
 * A sequence of n symbols is cut into blocks of `block` symbols, each an independent rANS stream as above (own
 * state, own flush).  Blocks are what the device coder (csrc/rans.hip: one lane per block) runs in parallel, and
 * byte for byte what this function writes.  Output: the blocks' streams back to back in `out`; block_bytes[b] =
 * length of block b.  Returns the total number of bytes, or 0 on overflow. */
size_t rans_encode_blocked(const int32_t* sym, const int32_t* tab, size_t n, size_t block, const uint32_t* cdf,
                           const int32_t* lens, const int32_t* offs, int stride, uint8_t* out, size_t cap,
                           uint32_t* block_bytes, uint8_t* scratch, size_t scratch_cap) {
  size_t total = 0, b = 0;
  for (size_t s = 0; s < n; s += block, ++b) {
    const size_t m = n - s < block ? n - s : block;
    const size_t k = rans_encode(sym + s, tab + s, m, cdf, lens, offs, stride, scratch, scratch_cap);
    if (k == 0 || total + k > cap) return 0;
    memcpy(out + total, scratch + scratch_cap - k, k);
    block_bytes[b] = (uint32_t)k;
    total += k;
  }
  return total;
}

int rans_decode_blocked(const uint8_t* in, const uint32_t* block_bytes, size_t nblocks, const int32_t* tab, size_t n,
                        size_t block, const uint32_t* cdf, const int32_t* lens, const int32_t* offs, int stride,
                        int32_t* sym) {
  size_t off = 0, b = 0;
  for (size_t s = 0; s < n; s += block, ++b) {
    if (b >= nblocks) return -1;
    const size_t m = n - s < block ? n - s : block;
    if (rans_decode(in + off, block_bytes[b], tab + s, m, cdf, lens, offs, stride, sym + s) != 0) return -1;
    off += block_bytes[b];
  }
  return 0;
}

/* ---- stack coder for bits-back (bb_sga.py:133-139 estimates a refund; this is the coder that earns it) ----------
 * rANS is a STACK: what `put` pushes, `get` pops, exactly inverse.  Bits-back coding interleaves the two on one stack:
 * pop z under q(z | y) (the bits "got back"), push y under p(y | z), push z under the prior.  State x and a byte stack
 * whose top is buf[*len - 1].  Symbols / tables as in rans_encode; push takes the symbols in REVERSE so that a later
 * pop with the same tables returns them in forward order. */
static inline int spush(uint32_t* x, uint8_t* buf, size_t cap, size_t* len, uint32_t start, uint32_t freq) {
  const uint32_t x_max = ((RANS_L >> SCALE_BITS) << 8) * freq;
  uint32_t v = *x;
  while (v >= x_max) {
    if (*len >= cap) return -1;
    buf[(*len)++] = (uint8_t)(v & 0xff);
    v >>= 8;
  }
  *x = ((v / freq) << SCALE_BITS) + (v % freq) + start;
  return 0;
}

int rans_stack_push(uint32_t* x, uint8_t* buf, size_t cap, size_t* len, const int32_t* sym, const int32_t* tab, size_t n,
                    const uint32_t* cdf, const int32_t* lens, const int32_t* offs, int stride) {
  for (size_t k = n; k-- > 0;) {
    const int t = tab[k];
    const uint32_t* c = cdf + (size_t)t * stride;
    const int len_t = lens[t];
    const int32_t idx = sym[k] - offs[t];
    if (idx >= 0 && idx < len_t - 1) {
      if (spush(x, buf, cap, len, c[idx], c[idx + 1] - c[idx])) return -1;
    } else {
      const uint32_t z = ((uint32_t)sym[k] << 1) ^ (uint32_t)(sym[k] >> 31);
      if (spush(x, buf, cap, len, z >> 16, 1) || spush(x, buf, cap, len, z & 0xffff, 1) ||
          spush(x, buf, cap, len, c[len_t - 1], c[len_t] - c[len_t - 1]))
        return -1;
    }
  }
  return 0;
}

static inline void srenorm(uint32_t* x, const uint8_t* buf, size_t* len) {
  while (*x < RANS_L && *len > 0) *x = (*x << 8) | buf[--(*len)];
}

/* returns 0, or -1 on a corrupt table / -2 when the stack runs dry (not enough initial bits to sample from) */
int rans_stack_pop(uint32_t* x, const uint8_t* buf, size_t* len, const int32_t* tab, size_t n, const uint32_t* cdf,
                   const int32_t* lens, const int32_t* offs, int stride, int32_t* sym) {
  for (size_t k = 0; k < n; ++k) {
    const int t = tab[k];
    const uint32_t* c = cdf + (size_t)t * stride;
    const int len_t = lens[t];
    if (*x < RANS_L) return -2;
    const uint32_t s = *x & 0xffff;
    int lo = 0, hi = len_t;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c[mid] <= s) lo = mid; else hi = mid; }
    const uint32_t start = c[lo], freq = c[lo + 1] - c[lo];
    if (freq == 0) return -1;
    *x = freq * (*x >> SCALE_BITS) + s - start;
    srenorm(x, buf, len);
    if (lo < len_t - 1) {
      sym[k] = offs[t] + lo;
    } else {
      if (*x < RANS_L) return -2;
      const uint32_t zl = *x & 0xffff; *x >>= SCALE_BITS; srenorm(x, buf, len);
      if (*x < RANS_L) return -2;
      const uint32_t zh = *x & 0xffff; *x >>= SCALE_BITS; srenorm(x, buf, len);
      const uint32_t z = (zh << 16) | zl;
      sym[k] = (int32_t)((z >> 1) ^ (uint32_t)(-(int32_t)(z & 1)));
    }
  }
  return 0;
}
