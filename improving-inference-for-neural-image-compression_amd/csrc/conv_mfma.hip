// Gather-GEMM convolution on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// Every convolution on the SGA hot path -- the stride-2 5x5 convs of the analysis side,
// the stride-2 transposed 5x5 convs of the synthesis side (as 4 sub-pixel phases), their
// data-gradients, the 3x3 stride-1 convs, and the C x C channel contractions of GDN/IGDN
// and their backward (tfc.SignalConv2D / tfc.GDN at nn_models.py:14-163) -- is one
// implicit GEMM:
//     M = output pixels of one sub-pixel phase,  N = output channels,  K = taps x C_in
// with both operands stored K-major:
//     A row = C_in contiguous floats of one NHWC input pixel (gathered per tap),
//     B row = C_in contiguous floats of the pre-packed weight slab [tap][n][ci].
// A workgroup owns a (BM x BN) output tile, BN = all (or a large slice of) the output
// channels so that each gathered activation row is read once; K is walked in 32-wide
// steps: global -> registers (prefetch of step k+1 overlaps the MFMAs of step k) ->
// LDS [row][36] (pad 4: conflict-free ds_read_b128) -> per-lane float4 fragments that
// feed four back-to-back 32x32x2 MFMAs each.
//
// f32 MFMA is a bitwise f32 fmaf chain (cdna_hip_programming.md s3), so results match an
// f32 reference to summation-order rounding; no reduced precision anywhere.
#include <atomic>
#include <cstdio>

#include <type_traits>

#include "sga_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;   // K-step
constexpr int LDK = 36;  // LDS row pitch in floats (BK + 4)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }

// ---- precision mode "bf16x3" -------------------------------------------------------------------
// An f32 value is the exact sum of three bf16 values (3 x 8 mantissa bits = the 24-bit mantissa):
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m).  A product a*b is then the
// sum of 9 bf16 x bf16 products (each exact in f32); dropping the three of order 2^-24 and below
// (m*l, l*m, l*l) leaves 6 products whose f32-accumulated sum has the accuracy of an f32 FMA chain
// (measured: 2.4e-7 max rel. error at K = 4800, vs 4.4e-7 for an f32 GEMM).  The bf16 matrix pipe
// is 16x the f32 one, so 6 MFMAs per product are 2.67x faster than one f32 MFMA.
constexpr int X3_PITCH = 80;     // bytes per LDS row per plane: 32 bf16 + 16 B pad (conflict-free b128 reads)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));   // [15:0] = bf16(lo), [31:16] = bf16(hi), RNE
  return r;
}
__device__ __forceinline__ float bf16_lo(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// 4 floats -> three planes of 4 packed bf16 (8 bytes each)
__device__ __forceinline__ void split3(const f32x4 v, u32x2& h, u32x2& m, u32x2& l) {
  h.x = cvt_pk_bf16(v.x, v.y);
  h.y = cvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - bf16_lo(h.x), r1 = v.y - bf16_hi(h.x);
  const float r2 = v.z - bf16_lo(h.y), r3 = v.w - bf16_hi(h.y);
  m.x = cvt_pk_bf16(r0, r1);
  m.y = cvt_pk_bf16(r2, r3);
  l.x = cvt_pk_bf16(r0 - bf16_lo(m.x), r1 - bf16_hi(m.x));
  l.y = cvt_pk_bf16(r2 - bf16_lo(m.y), r3 - bf16_hi(m.y));
}

// 4 floats -> the two upper planes only (precision mode bf16x2: x ~ h + m, |x - h - m| <= 2^-17 |x|)
__device__ __forceinline__ void split2(const f32x4 v, u32x2& h, u32x2& m) {
  h.x = cvt_pk_bf16(v.x, v.y);
  h.y = cvt_pk_bf16(v.z, v.w);
  m.x = cvt_pk_bf16(v.x - bf16_lo(h.x), v.y - bf16_hi(h.x));
  m.y = cvt_pk_bf16(v.z - bf16_lo(h.y), v.w - bf16_hi(h.y));
}

// Instances whose K loop takes its operands by LDS-DMA (global_load_lds_dwordx4): see the GLDS block in conv_tile
constexpr bool glds_instance(int TM, int TN, int WM, int WN, int PRO, bool SMALLC, bool X3, int POST) {
  // 64-row (2 x 32 KB stages, still 2 workgroups/CU) and 256-row (2 x 56 KB, 1/CU either way).  The 128-row 4-wave
  // instances would drop to one workgroup per CU with two stages and keep the register-staged loop; so does the
  // 2-wave BN = 96 instance of the hyper branch (same speed alone, but with 56 KB per workgroup it gets in the
  // main chain's way: the iteration measured 1868 against 1827 us)
  return !X3 && !SMALLC && PRO == PRO_NONE &&
         ((POST <= 1 && TM == 1 && TN == 3 && WM == 2 && WN == 2) || (TM == 2 && TN == 3 && WM == 4 && WN == 2) ||
          (TM == 2 && TN == 4 && WM == 4 && WN == 2));
}

// 16 bytes per lane, global memory -> LDS at `lds_dst` (wave-uniform) + lane * 16, no registers in between.
// (The builtin only exists in the device pass; the host pass needs the kernel template to stay instantiable.)
__device__ __forceinline__ void dma16_to_lds(const float* src, float* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// POST = 1 (256-row tile, BN = C = 192 or 256): the IGDN that follows the transposed convolution
// (nn_models.py:48-59) runs as a post-phase of the SAME launch, on the tile while it is on chip:
// u = acc + bias goes to LDS, n = gamma . u^2 is a second MFMA contraction out of LDS, s = sqrt(n + beta),
// v = u * s; (u,) s and v leave in whole 16-byte row pieces.  See the block after the K loop.
// The tile goes through LDS in parts of post_rows(BM, BN) rows, sized so that a part + two gamma K-chunks fit:
// C = 192: 128 x 196 + 2 x 192 x 36 floats = 152 KB;  C = 256 (README.md:58-60, cfg 4): 64 x 260 + 2 x 256 x 36 = 137 KB.
// The 64-row 4-wave instance (two workgroups per CU; layer 1 at cfg 2) holds its whole tile and ONE gamma chunk:
// 64 x 196 + 192 x 36 floats = 78 KB, the footprint of gdn_tile_kernel, whose launch it replaces.
// bf16x3, PRO_NONE instances: the K loop in 16-wide stages through TWO LDS stages of unpadded 32-byte rows per plane (see the
// X3P block in the kernel); the prologue-transform instances keep the single-stage 32-wide loop
constexpr bool x3_pipelined(int PRO, bool SMALLC, bool X3) { return X3 && !SMALLC && PRO == PRO_NONE; }
constexpr int x3_main_floats(int BM, int BN, bool pipelined) {
  return pipelined ? 2 * 3 * ((BM + BN) * 32 + 128) / 4 : 3 * (BM + BN) * X3_PITCH / 4;      // (+ 128: the weight planes' bank offset)
}
constexpr int post_rows(int BM, int BN) { return BM < 128 ? BM : (BN <= 192 ? 128 : 64); }
constexpr int post_qbufs(int BM) { return BM < 128 ? 1 : 2; }
// X3: 0 = f32 MFMA; 1 = bf16x3 (3 planes, 6 products); 2 = bf16x2 (the two upper planes, 3 products: operands rounded to 16
// mantissa bits; only the pipelined PRO_NONE loop and the post-phase of such an instance have this form)
template <int TM, int TN, int WM, int WN, int PRO, bool SMALLC, int X3, int POST = 0>
__global__ __launch_bounds__(WM * WN * 64, (X3 && (TN >= 4 || TM >= 4 || WM * WN >= 8)) ? 1 : 2) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RPP = NT / 8;                 // tile rows covered by one pass of the loaders
  constexpr int PA = BM / RPP, PB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/loader mismatch");
  constexpr int PX = (PRO == PRO_IGDN_BWD) ? PA : 1;

  constexpr int CPITCH = TN * 32 + 4;         // epilogue staging pitch (floats) per wave row
  constexpr bool GLDS = glds_instance(TM, TN, WM, WN, PRO, SMALLC, X3, POST);
  constexpr bool HASPOST = POST == 1;
  constexpr bool X3P = x3_pipelined(PRO, SMALLC, X3);
  constexpr int MAIN_FLOATS = X3 ? x3_main_floats(BM, BN, X3P) : (GLDS ? 2 * (BM + BN) * 32 : (BM + BN) * LDK);
  constexpr int EPI_FLOATS = WM * WN * 32 * CPITCH;
  constexpr int POST_FLOATS = HASPOST ? (post_rows(BM, BN) * (BN + 4) + post_qbufs(BM) * BN * LDK) : 0;
  constexpr int LDS_FLOATS0 = MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS;
  constexpr int LDS_FLOATS = LDS_FLOATS0 > POST_FLOATS ? LDS_FLOATS0 : POST_FLOATS;
  static_assert(!POST || (((BM == 256 && (BN == 192 || BN == 256) && NT == 512 && TM == 2 && HASPOST) ||
                           (BM == 64 && BN == 192 && NT == 256 && TM == 1 && POST == 1)) && !SMALLC && PRO == PRO_NONE),
                "post-phase instance");
  constexpr int PBX = X3 ? (BN * 12 + NT - 1) / NT : 1;     // 16-byte pieces of the 3-plane weight tile per thread
  constexpr bool PBX_TAIL = X3 && (BN * 12) % NT != 0;       // ... the last round covers part of the threads (8 waves x BN = 192)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + BM * LDK;
  long long* rowpix = reinterpret_cast<long long*>(smem + LDS_FLOATS);   // [BM] output pixel index or -1

  if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int chunk = tid & 7, lrow = tid >> 3;
  // measurement (sga_profile_graph_begin; null otherwise): earliest workgroup entry / latest workgroup exit of this
  // launch on the 100 MHz wall clock -> the launch's duration inside a hipGraph replay, where no event can bracket it
  if (a.stamp && tid == 0) atomicMin(&a.stamp[0], wall_clock64());
#define SGA_STAMP_END() do { if (a.stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); \
                                             if (tid == 0) atomicMax(&a.stamp[1], wall_clock64()); } } while (0)
#ifdef SGA_CLOCK_PROBE
  const unsigned long long wallE = a.clk ? wall_clock64() : 0;
#define SGA_PROBE_END() do { SGA_STAMP_END(); if (a.clk && tid == 0) a.clk[6 * blockIdx.x + 5] = wall_clock64(); } while (0)
#else
#define SGA_PROBE_END() SGA_STAMP_END()
#endif

  int bid = blockIdx.x;
  int split = 0, nsplit = 1, phase, mt, nt;
  if (a.ksplit > 1) {
    phase = 0;
    while (phase + 1 < a.nphase && bid >= a.blk_begin[phase + 1]) ++phase;
    bid -= a.blk_begin[phase];
    const int tiles = a.tiles_per_phase * a.ntiles_n;
    nsplit = a.nsplit[phase];
    split = bid / tiles;
    bid -= split * tiles;
    nt = bid % a.ntiles_n;
    mt = bid / a.ntiles_n;
  } else {
    nt = bid % a.ntiles_n;
    bid /= a.ntiles_n;
    if (a.xcd_remap) {
      // blocks b, b + 8, b + 16, ... run on XCD b % 8 (observed dispatch rule; speed only): give that XCD the M tiles
      // [x * T/8, (x + 1) * T/8) of every phase -- at cfg 2 exactly one image -- so that the 25 tap gathers of an input
      // region are served by one L2 instead of eight
      const int x = bid & 7, idx = bid >> 3, tpx = a.tiles_per_phase >> 3;
      phase = idx / tpx;
      mt = x * tpx + (idx - phase * tpx);
    } else {
      phase = bid / a.tiles_per_phase;
      mt = bid - phase * a.tiles_per_phase;
    }
    // the 4 sub-pixel phases of a transposed conv have 9/6/6/4 taps: walk them in the order 9,6,4,6 so that
    // the workgroups that land on one CU together (block b and b + #CUs when the whole grid is resident at
    // once) are a heavy and a light phase (78 / 72 K-steps per CU instead of 90 / 60).  Only then: a grid
    // that runs in rounds is balanced by the dispatcher and wants heaviest-first.
    if (a.pair_phases && phase >= 2) phase = 5 - phase;
  }
  const ConvPhase ph = a.ph[phase];
  const int Mtot = a.B * a.Hg * a.Wg;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread gather metadata for the PA rows this thread stages -----------------
  int a_iy[PA], a_ix[PA], a_base[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int m = m0 + p * RPP + lrow;
    if (m < Mtot) {
      const int j = m % a.Wg;
      const int t = m / a.Wg;
      const int i = t % a.Hg;
      const int b = t / a.Hg;
      a_iy[p] = i * a.s_in;
      a_ix[p] = j * a.s_in;
      a_base[p] = b * a.Hin * a.Win;
    } else {
      a_iy[p] = -(1 << 20);
      a_ix[p] = 0;
      a_base[p] = 0;
    }
  }

  // output pixel of every tile row (epilogue), -1 if the row is past M or outside the output
  for (int r = tid; r < BM; r += NT) {
    const int m = m0 + r;
    long long px = -1;
    if (m < Mtot) {
      const int j = m % a.Wg;
      const int t = m / a.Wg;
      const int i = t % a.Hg;
      const int b = t / a.Hg;
      const int oy = i * a.s_out + ph.py, ox = j * a.s_out + ph.px;
      if (oy < a.Hout && ox < a.Wout) px = ((long long)b * a.Hout + oy) * a.Wout + ox;
    }
    rowpix[r] = px;
  }

  f32x4 ra[PA], rb[X3 ? 1 : PB], ra1[PX], ra2[PX];
  u32x4 rbx[PBX];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};

  // Per-tap gather state of the generic path: element offset of this thread's 16-byte piece in each of
  // its PA rows (channel chunk 0) and whether the tap lands inside the image; the weight rows likewise.
  // Recomputed when the K walk moves to the next tap (every Cin/32 steps); inside a tap a K-step only
  // adds ci0 (the index arithmetic was ~50 64-bit VALU instructions per K-step before).
  long long a_off[PA];
  bool a_ok[PA];
  long long b_off = 0;
  auto set_tap = [&](int tapi) {
    const ConvTap tp = a.taps[ph.tap_begin + tapi];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int iy = a_iy[p] + tp.dy, ix = a_ix[p] + tp.dx;
      a_ok[p] = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      a_off[p] = (long long)((size_t)(a_base[p] + iy * a.Win + ix) * a.in_cs + a.in_coff + chunk * 4);
    }
    b_off = (long long)(((size_t)tp.slab * a.Npad + n0 + lrow) * a.Cin + chunk * 4);
  };
  auto gload = [&](int tapi, int ci0) {
    if constexpr (!SMALLC && !X3) {
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const size_t off = (size_t)(a_off[p] + ci0);
        ra[p] = a_ok[p] ? ld4(a.in + off) : zero4;
        if constexpr (PRO == PRO_IGDN_BWD) {
          ra1[p] = a_ok[p] ? ld4(a.aux1 + off) : one4;
          ra2[p] = a_ok[p] ? ld4(a.aux2 + off) : zero4;
        }
      }
#pragma unroll
      for (int p = 0; p < PB; ++p) rb[p] = ld4(a.w + (size_t)(b_off + ci0) + (size_t)p * RPP * a.Cin);
      return;
    }
    const ConvTap tp = a.taps[ph.tap_begin + tapi];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      if constexpr (SMALLC) {
        // `in` is a zero-padded 3-channel image; one K-step = two kernel rows (ky, ky+1),
        // each 16 contiguous floats (5 taps x 3 channels + 1 float of slack, weight 0).
        int ky = tp.dy + (chunk >> 2);
        ky = ky > 4 ? 4 : ky;
        const bool ok = a_iy[p] >= 0;
        const size_t off =
            ((size_t)(a_base[p] + (a_iy[p] + ky) * a.Win + a_ix[p])) * 3 + (chunk & 3) * 4;
        f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
        if (ok) {
          lo = ld2(a.in + off);
          hi = ld2(a.in + off + 2);
        }
        ra[p] = f32x4{lo.x, lo.y, hi.x, hi.y};
      } else {
        const int iy = a_iy[p] + tp.dy, ix = a_ix[p] + tp.dx;
        const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        const size_t off =
            (size_t)(a_base[p] + iy * a.Win + ix) * a.in_cs + a.in_coff + ci0 + chunk * 4;
        ra[p] = ok ? ld4(a.in + off) : zero4;
        if constexpr (PRO == PRO_IGDN_BWD) {
          ra1[p] = ok ? ld4(a.aux1 + off) : one4;
          ra2[p] = ok ? ld4(a.aux2 + off) : zero4;
        }
      }
    }
    if constexpr (X3) {
      // weight tile: BN rows x (3 planes x 32 bf16) = 192 contiguous bytes per row and K-step
      const unsigned short* wb = a.w3 + (((size_t)tp.slab * a.Npad + n0) * (a.Cin / BK) + ci0 / BK) * 96;
#pragma unroll
      for (int k = 0; k < PBX; ++k) {
        const int f = tid + NT * k;
        const int nl = f / 12, piece = f - nl * 12;
        if (!PBX_TAIL || k + 1 < PBX || f < BN * 12)
          rbx[k] = *reinterpret_cast<const u32x4*>(wb + (size_t)nl * (a.Cin / BK) * 96 + piece * 8);
      }
    } else {
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const int n = n0 + p * RPP + lrow;
        const size_t off = ((size_t)tp.slab * a.Npad + n) * a.Cin + ci0 + chunk * 4;
        rb[p] = ld4(a.w + off);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int nchunk = a.Cin / BK;
  const int nsteps_all = ph.ntaps * nchunk;
  int k_begin = 0, k_end = nsteps_all;
  if (nsplit > 1) {
    k_begin = (int)((long long)split * nsteps_all / nsplit);
    k_end = (int)((long long)(split + 1) * nsteps_all / nsplit);
  }
  int tapi = k_begin / nchunk, ci0 = (k_begin - tapi * nchunk) * BK;
  if (k_begin < k_end && !GLDS && !X3P) {
    if constexpr (!SMALLC && !X3) set_tap(tapi);
    gload(tapi, ci0);
  }

  const int arow = (wm * TM) * 32 + (lane & 31);
  const int brow = (wn * TN) * 32 + (lane & 31);
  const int koff = (lane >> 5) * 4;
#ifdef SGA_CLOCK_PROBE
  unsigned long long clk0 = 0, wall0 = 0;
  if (a.clk) { clk0 = __builtin_readcyclecounter(); wall0 = wall_clock64(); }
#endif

  // ---- X3P (bf16x3, PRO_NONE): software-pipelined K loop in 16-wide stages ----------------------------------------------------
  // The single-stage loop (below; round 1) alternates "split + write LDS" and "read LDS + multiply" between two barriers per
  // 32-wide K-step: while the operands of the next step are split (22 VALU instructions per 4 floats) the matrix pipe idles,
  // and it ran at half of the bf16 MFMA rate (gs2.fwd 393 us for 150 us of MFMA work).  Here a stage is 16 k's -- exactly one
  // v_mfma_f32_32x32x16_bf16 deep -- in unpadded 32-byte rows per plane (8 consecutive lanes x 16 bytes cover all 64 banks:
  // conflict-free ds_read_b128), and there are TWO stages: while a wave's 6 x TM x TN MFMAs of stage s execute, the same
  // wave splits the registers of stage s + 1 into the other LDS stage and requests stage s + 2 from global memory; one
  // barrier per stage.  Same operands, same plane order, same k order per accumulator as the single-stage loop: bit-identical.
  if constexpr (X3P) {
    constexpr int ROWB = 32;                                   // bytes per row per plane per stage: 16 bf16
    constexpr int NPL = X3 == 2 ? 2 : 3;                       // operand planes in use (the weights are stored as 3 either way)
    static_assert(X3 == 1 || X3 == 2, "precision mode");
    // weight planes 96 / 128 bytes further apart than their size: the loader's 16 lanes of a ds_write_b128 pass hold the same row of
    // all NPL planes, which would otherwise land on the same banks (plane size = 0 mod 256 bytes: 3- / 2-way conflicts)
    constexpr int A_PL = BM * ROWB, B_PL = BN * ROWB + (NPL == 3 ? 96 : 128), STAGE_B = NPL * (A_PL + B_PL);
    constexpr int PA2 = BM * 4 / NT;                           // 16-byte f32 pieces (4 k's) of the A stage per thread
    static_assert((BM * 4) % NT == 0 && NT % 4 == 0, "x3 pipelined loader mismatch");
    constexpr int RP = 2 * NPL;                                // 16-byte pieces per weight row and stage (NPL planes x 32 B)
    constexpr int NBP = BN * RP, PB2 = (NBP + NT - 1) / NT;    // 16-byte pieces of the pre-split weight stage
    char* const sm = reinterpret_cast<char*>(smem);
    const int c4 = tid & 3, prow = tid >> 2;                   // piece (row prow + p * NT / 4, floats c4 * 4 .. + 3 of the stage)
    int q_iy[PA2], q_ix[PA2], q_base[PA2];
#pragma unroll
    for (int p = 0; p < PA2; ++p) {
      const int m = m0 + prow + p * (NT / 4);
      if (m < Mtot) {
        const int j = m % a.Wg;
        const int t = m / a.Wg;
        q_iy[p] = (t % a.Hg) * a.s_in; q_ix[p] = j * a.s_in; q_base[p] = (t / a.Hg) * a.Hin * a.Win;
      } else {
        q_iy[p] = -(1 << 20); q_ix[p] = 0; q_base[p] = 0;
      }
    }
    long long q_off[PA2];
    bool q_ok[PA2];
    size_t w_off = 0;
    auto tap2 = [&](int t) {
      const ConvTap tp = a.taps[ph.tap_begin + t];
#pragma unroll
      for (int p = 0; p < PA2; ++p) {
        const int iy = q_iy[p] + tp.dy, ix = q_ix[p] + tp.dx;
        q_ok[p] = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        q_off[p] = (long long)((size_t)(q_base[p] + iy * a.Win + ix) * a.in_cs + a.in_coff + c4 * 4);
      }
      w_off = ((size_t)tp.slab * a.Npad + n0) * (size_t)(a.Cin / BK) * 96;
    };
    f32x4 qa[PA2];
    u32x4 qb[PB2];
    auto gload2 = [&](int ci) {                                // ci: first input channel of the stage (multiple of 16)
#pragma unroll
      for (int p = 0; p < PA2; ++p) qa[p] = q_ok[p] ? ld4(a.in + (size_t)(q_off[p] + ci)) : zero4;
      const unsigned short* wb = a.w3 + w_off + (size_t)(ci >> 5) * 96 + ((ci >> 4) & 1) * 16;
#pragma unroll
      for (int k = 0; k < PB2; ++k) {
        const int f = tid + NT * k;
        const int nl = f / RP, r6 = f - nl * RP;               // row nl: plane r6 >> 1, 16-byte half r6 & 1
        if (NBP % NT == 0 || k + 1 < PB2 || f < NBP)
          qb[k] = *reinterpret_cast<const u32x4*>(wb + (size_t)nl * (a.Cin / BK) * 96 + (r6 >> 1) * 32 + (r6 & 1) * 8);
      }
    };
    auto stash = [&](int buf) {                                // split the staged activations, write both operands to stage `buf`
      char* const st = sm + buf * STAGE_B;
#pragma unroll
      for (int p = 0; p < PA2; ++p) {
        u32x2 h, m, l;
        if constexpr (NPL == 3) split3(qa[p], h, m, l);
        else split2(qa[p], h, m);
        // the two 16-byte halves of a row are swapped in rows 8..15 (mod 16): a ds_read_b128 is served 16 lanes at a time, and
        // 16 rows x 16 bytes at a 32-byte pitch would hit every bank twice (45 % bank conflicts measured, r04_pmc_kernels.txt)
        const int row = prow + p * (NT / 4);
        char* dst = st + row * ROWB + (((c4 >> 1) ^ ((row >> 3) & 1)) * 16) + (c4 & 1) * 8;
        *reinterpret_cast<u32x2*>(dst) = h;
        *reinterpret_cast<u32x2*>(dst + A_PL) = m;
        if constexpr (NPL == 3) *reinterpret_cast<u32x2*>(dst + 2 * A_PL) = l;
      }
#pragma unroll
      for (int k = 0; k < PB2; ++k) {
        const int f = tid + NT * k;
        const int nl = f / RP, r6 = f - nl * RP;
        if (NBP % NT == 0 || k + 1 < PB2 || f < NBP)
          *reinterpret_cast<u32x4*>(st + NPL * A_PL + (r6 >> 1) * B_PL + nl * ROWB + (((r6 & 1) ^ ((nl >> 3) & 1)) * 16)) = qb[k];
      }
    };
    const int nchunk2 = a.Cin / 16;
    const int s_begin = 2 * k_begin, s_end = 2 * k_end;
    int tap_l = s_begin / nchunk2, ci_l = (s_begin - tap_l * nchunk2) * 16;      // the stage the loader is at
    auto advance = [&]() {
      ci_l += 16;
      if (ci_l >= a.Cin) { ci_l = 0; ++tap_l; }
    };
    if (s_begin < s_end) {
      tap2(tap_l);
      gload2(ci_l);
      stash(0);
      advance();
      if (s_begin + 1 < s_end) {
        if (ci_l == 0) tap2(tap_l);
        gload2(ci_l);
      }
    }
    __syncthreads();
    const int hsw = ((lane >> 5) ^ ((lane >> 3) & 1)) * 16;      // (arow, brow and the + 32 * t rows all have row bits 3 = lane bit 3)
    const int fa = arow * ROWB + hsw;
    const int fb = NPL * A_PL + brow * ROWB + hsw;
    for (int s2 = s_begin; s2 < s_end; ++s2) {
      const int cur = (s2 - s_begin) & 1;
      const char* const st = sm + cur * STAGE_B;
      bf16x8 af3[NPL][TM], bf3[NPL][TN];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          af3[pl][tm] = *reinterpret_cast<const bf16x8*>(st + pl * A_PL + fa + tm * 32 * ROWB);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          bf3[pl][tn] = *reinterpret_cast<const bf16x8*>(st + pl * B_PL + fb + tn * 32 * ROWB);
      }
      if (s2 + 1 < s_end) {
        stash(cur ^ 1);                                        // stage s2 + 1 (its registers were requested one stage ago)
        advance();
        if (s2 + 2 < s_end) {
          if (ci_l == 0) tap2(tap_l);
          gload2(ci_l);                                        // stage s2 + 2
        }
      }
      // bf16x3: A plane l, h, m, m, h, h x B plane h, l, m, h, m, h (smallest products first); bf16x2: m.h, h.m, h.h
      constexpr int NPR = NPL == 3 ? 6 : 3;
      constexpr int PA6[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0};
      constexpr int PB6[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
      for (int c = 0; c < NPR; ++c)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[PA6[c]][tm], bf3[PB6[c]][tn], acc[tm][tn], 0, 0, 0);
      __syncthreads();
    }
  } else
  // ---- GLDS (f32 instances of glds_instance): operands by LDS-DMA ---------------------------------------------------------
  // `global_load_lds_dwordx4` moves 16 bytes per lane from global memory straight into LDS: no staging
  // registers, no ds_write, and with two LDS stages one barrier per K-step (the loads of step k+1 are issued at
  // the top of step k and have the whole step to land; `__syncthreads` waits for them).  A DMA instruction
  // writes its 64 lanes' pieces contiguously, so a stage is unpadded 128-byte rows; the bank conflicts that the
  // 144-byte pitch avoided are avoided by an XOR swizzle instead: lane slot j of row r loads K-chunk
  // j ^ ((r >> 1) & 7), and the fragment reads apply the same XOR.  Taps outside the image read a zero page.
  // Same operands, same MFMA order: bit-identical to the register-staged loop (micro-benchmark: +4 %).
  if constexpr (GLDS) {
    constexpr int STAGE = (BM + BN) * 32;                  // floats per stage
    constexpr int NW = WM * WN;
    constexpr int IA = BM / (NW * 8), IB = BN / (NW * 8);  // DMA instructions per wave: 8 rows x 8 slots each
    static_assert(BM % (NW * 8) == 0 && BN % (NW * 8) == 0, "LDS-DMA row split");
    const int slot = lane & 7, r8 = lane >> 3;
    // A rows of this lane: tile row wid*(BM/NW) + p*8 + r8
    int g_iy[IA], g_ix[IA], g_base[IA], g_chunk[IA];
#pragma unroll
    for (int p = 0; p < IA; ++p) {
      const int r = wid * (BM / NW) + p * 8 + r8;
      const int m = m0 + r;
      g_chunk[p] = (slot ^ ((r >> 1) & 7)) * 4;
      if (m < Mtot) {
        const int j = m % a.Wg;
        const int t = m / a.Wg;
        g_iy[p] = (t % a.Hg) * a.s_in; g_ix[p] = j * a.s_in; g_base[p] = (t / a.Hg) * a.Hin * a.Win;
      } else {
        g_iy[p] = -(1 << 20); g_ix[p] = 0; g_base[p] = 0;
      }
    }
    int gb_chunk[IB];
#pragma unroll
    for (int p = 0; p < IB; ++p) {
      const int r = BM + wid * (BN / NW) + p * 8 + r8;     // stage row of this weight row
      gb_chunk[p] = (slot ^ ((r >> 1) & 7)) * 4;
    }
    const float* ga_src[IA];                               // per tap: address of the row's piece at channel 0
    const float* gb_src[IB];
    auto tap_setup = [&](int t) {
      const ConvTap tp = a.taps[ph.tap_begin + t];
#pragma unroll
      for (int p = 0; p < IA; ++p) {
        const int iy = g_iy[p] + tp.dy, ix = g_ix[p] + tp.dx;
        const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        ga_src[p] = ok ? a.in + ((size_t)(g_base[p] + iy * a.Win + ix) * a.in_cs + a.in_coff + g_chunk[p]) : nullptr;
      }
#pragma unroll
      for (int p = 0; p < IB; ++p)
        gb_src[p] = a.w + (((size_t)tp.slab * a.Npad + n0 + wid * (BN / NW) + p * 8 + r8) * a.Cin + gb_chunk[p]);
    };
    const int lds_a = __builtin_amdgcn_readfirstlane(wid * (BM / NW) * 32);
    const int lds_b = __builtin_amdgcn_readfirstlane((BM + wid * (BN / NW)) * 32);
    auto issue = [&](int stage, int ci) {
      float* st = smem + stage * STAGE;
#pragma unroll
      for (int p = 0; p < IA; ++p)
        dma16_to_lds(ga_src[p] ? ga_src[p] + ci : a.zeros, st + lds_a + p * 256);
#pragma unroll
      for (int p = 0; p < IB; ++p)
        dma16_to_lds(gb_src[p] + ci, st + lds_b + p * 256);
    };
    if (k_begin < k_end) {
      tap_setup(tapi);
      issue(0, ci0);
    }
    __syncthreads();
    const int ra_ = (wm * TM) * 32 + (lane & 31);
    const int rb_ = BM + (wn * TN) * 32 + (lane & 31);
    const int hlf = lane >> 5;
    for (int ks = k_begin; ks < k_end; ++ks) {
      const int cur = (ks - k_begin) & 1;
      if (ks + 1 < k_end) {
        ci0 += BK;
        if (ci0 >= a.Cin) { ci0 = 0; ++tapi; tap_setup(tapi); }
        issue(cur ^ 1, ci0);
      }
      const float* St = smem + cur * STAGE;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 2 * q + hlf;
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int r = ra_ + tm * 32;
          af[tm] = *reinterpret_cast<const f32x4*>(&St[r * 32 + ((c ^ ((r >> 1) & 7)) * 4)]);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int r = rb_ + tn * 32;
          bf[tn] = *reinterpret_cast<const f32x4*>(&St[r * 32 + ((c ^ ((r >> 1) & 7)) * 4)]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][r], bf[tn][r], acc[tm][tn], 0, 0, 0);
      }
      __syncthreads();
    }
  } else
  for (int ks = k_begin; ks < k_end; ++ks) {
    // ---- staged registers -> LDS (prologue transform fused here) ----------------------
    char* const smem_b = reinterpret_cast<char*>(smem);
    constexpr int A_PLANE = BM * X3_PITCH, B_PLANE = BN * X3_PITCH, B3_BASE = 3 * A_PLANE;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      f32x4 v = ra[p];
      if constexpr (PRO == PRO_SQUARE) v = v * v;
      if constexpr (PRO == PRO_IGDN_BWD) v = v * ra2[p] / ra1[p];   // g * u / s
      if constexpr (X3) {
        u32x2 h, m, l;
        split3(v, h, m, l);
        char* dst = smem_b + (p * RPP + lrow) * X3_PITCH + chunk * 8;
        *reinterpret_cast<u32x2*>(dst) = h;
        *reinterpret_cast<u32x2*>(dst + A_PLANE) = m;
        *reinterpret_cast<u32x2*>(dst + 2 * A_PLANE) = l;
      } else {
        *reinterpret_cast<f32x4*>(&As[(p * RPP + lrow) * LDK + chunk * 4]) = v;
      }
    }
    if constexpr (X3) {
#pragma unroll
      for (int k = 0; k < PBX; ++k) {
        const int f = tid + NT * k;
        const int nl = f / 12, piece = f - nl * 12;
        const int plane = piece >> 2, j = piece & 3;
        if (!PBX_TAIL || k + 1 < PBX || f < BN * 12)
          *reinterpret_cast<u32x4*>(smem_b + B3_BASE + plane * B_PLANE + nl * X3_PITCH + j * 16) = rbx[k];
      }
    } else {
#pragma unroll
      for (int p = 0; p < PB; ++p)
        *reinterpret_cast<f32x4*>(&Bs[(p * RPP + lrow) * LDK + chunk * 4]) = rb[p];
    }
    __syncthreads();

    // ---- prefetch the next K-step while this one is multiplied ------------------------
    ci0 += BK;
    if (ks + 1 < k_end) {
      if (ci0 >= a.Cin) {
        ci0 = 0; ++tapi;
        if constexpr (!SMALLC && !X3) set_tap(tapi);
      }
      gload(tapi, ci0);
    }

    if constexpr (X3) {
      // ---- 32 k's = 2 x (16-deep bf16 MFMA) x 6 plane pairs, smallest products first ----------
      const int fa = arow * X3_PITCH + (lane >> 5) * 16;
      const int fb = B3_BASE + brow * X3_PITCH + (lane >> 5) * 16;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16x8 af3[3][TM], bf3[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
            af3[pl][tm] = *reinterpret_cast<const bf16x8*>(smem_b + pl * A_PLANE + fa + tm * 32 * X3_PITCH + q * 32);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            bf3[pl][tn] = *reinterpret_cast<const bf16x8*>(smem_b + pl * B_PLANE + fb + tn * 32 * X3_PITCH + q * 32);
        }
        constexpr int PA6[6] = {2, 0, 1, 1, 0, 0};   // A plane: l, h, m, m, h, h
        constexpr int PB6[6] = {0, 2, 1, 0, 1, 0};   // B plane: h, l, m, h, m, h
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af3[PA6[c]][tm], bf3[PB6[c]][tn],
                                                                  acc[tm][tn], 0, 0, 0);
      }
    } else {
    // ---- 32 k's = 4 x (float4 fragment -> 4 MFMAs per output sub-tile) -----------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        af[tm] = *reinterpret_cast<const f32x4*>(&As[(arow + tm * 32) * LDK + q * 8 + koff]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        bf[tn] = *reinterpret_cast<const f32x4*>(&Bs[(brow + tn * 32) * LDK + q * 8 + koff]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][r], bf[tn][r], acc[tm][tn], 0, 0, 0);
    }
    }
    __syncthreads();
  }

#ifdef SGA_CLOCK_PROBE
  if (a.clk && tid == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.clk[6 * blockIdx.x] = __builtin_readcyclecounter() - clk0;
    a.clk[6 * blockIdx.x + 1] = wall_clock64() - wall0;
    a.clk[6 * blockIdx.x + 2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
    a.clk[6 * blockIdx.x + 3] = wall0;
    a.clk[6 * blockIdx.x + 4] = wallE;
  }
#endif
  // ---- epilogue -------------------------------------------------------------------------------
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  const int half = lane >> 5, col = lane & 31;
  if constexpr (HASPOST) {
    // ---- fused IGDN (the tile in parts of HR rows; all 8 waves multiply, one barrier per K-step) ----
    constexpr int C = BN, TP = C + 4;
    constexpr int NWV = WM * WN;                         // 8 waves (256-row tiles) or 4 (64-row tile)
    constexpr int HR = post_rows(BM, BN), NPART = BM / HR;   // 128 x 2 (C = 192), 64 x 4 (C = 256), 64 x 1 (64-row tile)
    constexpr int PM = HR / 32, PN = NWV / PM;           // post-phase wave grid: PM (rows) x PN (cols) blocks of 32 x (PTN * 32)
    constexpr int PTN = (C / 32) / PN;                   // 3 (32 x 96 per wave) or 2 (32 x 64)
    constexpr int WR = TM * 32;                          // tile rows of one main-loop wave row
    constexpr int OW = HR / WR;                          // main-loop wave rows per part
    constexpr int QBUF = post_qbufs(BM);                 // gamma K-chunk buffers: 2, or 1 where two workgroups share a CU
    static_assert(PM * PN == NWV && PTN * PN * 32 == C && OW * WR == HR, "post-phase wave grid");
    float* const Tt = smem;                          // [HR][TP]: u of the current part
    float* const Bq = smem + HR * TP;                // [QBUF][C][LDK]: gamma K-chunks
    const int m4 = wid / PN, n2 = wid % PN;
    float bias_c[TN], beta_c[PTN];                   // per-lane columns: before any store is in flight
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bias_c[tn] = a.bias ? a.bias[(wn * TN + tn) * 32 + col] : 0.f;
#pragma unroll
    for (int tn = 0; tn < PTN; ++tn) beta_c[tn] = a.post_beta[(n2 * PTN + tn) * 32 + col];
    // gamma chunk loader: C rows x 32 floats = C*8 float4 over 512 threads
    constexpr int QB = C * 8 / NT;                   // 3 or 4
    f32x4 rq[QB];
    auto load_q = [&](int kc) {
#pragma unroll
      for (int p = 0; p < QB; ++p) rq[p] = ld4(a.post_w + (size_t)(p * RPP + lrow) * C + kc * 32 + chunk * 4);
    };
    auto store_q = [&](int buf) {
#pragma unroll
      for (int p = 0; p < QB; ++p)
        *reinterpret_cast<f32x4*>(&Bq[buf * (C * LDK) + (p * RPP + lrow) * LDK + chunk * 4]) = rq[p];
    };
    // bf16x3 form of the contraction (X3 instances): n = gamma . u^2 on v_mfma_f32_32x32x16_bf16 in 16-wide stages -- u^2 is
    // split into its three bf16 planes as the fragments leave the tile (8 floats per lane and stage), gamma comes pre-split
    // (a.post_wx3, the planes sga_create packs for every weight) through the gamma buffers as [plane][C][32 bytes]
    // (bf16x2 instances, X3 == 2: the two upper planes of both operands, three products, like the K loop)
    constexpr int QPL = X3 == 2 ? 2 : 3, QRP = 2 * QPL;        // planes in use; 16-byte pieces per gamma row and stage
    constexpr int QPLB = C * 32 + (QPL == 3 ? 96 : 128);       // plane pitch: + 96 / 128 bytes against write bank conflicts (see the K loop)
    constexpr int QSTG = QPL * QPLB;                 // bytes per staged gamma stage (QPL planes x C rows x 16 bf16)
    static_assert(QSTG <= C * LDK * 4, "gamma stage must fit the f32 chunk buffer");
    constexpr int NQX = X3 ? (C * QRP + NT - 1) / NT : 1;
    u32x4 qx[NQX];
    char* const Bqb = reinterpret_cast<char*>(Bq);
    auto load_qx = [&](int st) {                     // stage st = k's [16 st, 16 st + 16)
      const unsigned short* wx = a.post_wx3 + (size_t)(st >> 1) * 96 + (st & 1) * 16;
#pragma unroll
      for (int k = 0; k < NQX; ++k) {
        const int f = tid + NT * k;
        const int nl = f / QRP, r6 = f - nl * QRP;
        if ((C * QRP) % NT == 0 || k + 1 < NQX || f < C * QRP)
          qx[k] = *reinterpret_cast<const u32x4*>(wx + (size_t)nl * (C / 32) * 96 + (r6 >> 1) * 32 + (r6 & 1) * 8);
      }
    };
    auto store_qx = [&](int buf) {
#pragma unroll
      for (int k = 0; k < NQX; ++k) {
        const int f = tid + NT * k;
        const int nl = f / QRP, r6 = f - nl * QRP;
        if ((C * QRP) % NT == 0 || k + 1 < NQX || f < C * QRP)
          *reinterpret_cast<u32x4*>(Bqb + buf * QSTG + (r6 >> 1) * QPLB + nl * 32 + (((r6 & 1) ^ ((nl >> 3) & 1)) * 16)) = qx[k];   // halves swapped in rows 8..15 mod 16 (bank conflicts, see the K loop)
      }
    };
    lds_barrier();                                   // main-loop LDS is dead from here on
#pragma unroll
    for (int h = 0; h < NPART; ++h) {
      if constexpr (X3) load_qx(0); else load_q(0);
      if (wm / OW == h) {                            // the waves that own these HR rows: u = acc + bias
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
              Tt[((wm % OW) * WR + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * TP + (wn * TN + tn) * 32 + col] =
                  acc[tm][tn][reg] + bias_c[tn];
      }
      if constexpr (X3) { store_qx(0); load_qx(1); } else { store_q(0); load_q(1); }
      lds_barrier();
      f32x16 acc2[PTN];
#pragma unroll
      for (int tn = 0; tn < PTN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[tn][r] = 0.f;
      if constexpr (X3) {
        constexpr int NSTG = C / 16;
#pragma unroll 2
        for (int st = 0; st < NSTG; ++st) {
          if constexpr (QBUF == 2) {
            if (st + 1 < NSTG) store_qx((st + 1) & 1);
            if (st + 2 < NSTG) load_qx(st + 2);
          } else if (st > 0) {                       // one buffer: stage st replaces stage st - 1 between two barriers
            store_qx(0);
            if (st + 1 < NSTG) load_qx(st + 1);
            lds_barrier();
          }
          const char* Bs3 = Bqb + (QBUF == 2 ? (st & 1) : 0) * QSTG;
          const float* ap = &Tt[(m4 * 32 + col) * TP + st * 16 + half * 8];
          f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
          a0 = a0 * a0; a1 = a1 * a1;
          u32x2 h0, m0, l0, h1, m1, l1;
          if constexpr (QPL == 3) { split3(a0, h0, m0, l0); split3(a1, h1, m1, l1); }
          else { split2(a0, h0, m0); split2(a1, h1, m1); }
          bf16x8 ax[QPL];
          ax[0] = __builtin_bit_cast(bf16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
          ax[1] = __builtin_bit_cast(bf16x8, u32x4{m0.x, m0.y, m1.x, m1.y});
          if constexpr (QPL == 3) ax[2] = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
          bf16x8 bx[QPL][PTN];
#pragma unroll
          for (int pl = 0; pl < QPL; ++pl)
#pragma unroll
            for (int tn = 0; tn < PTN; ++tn)
              bx[pl][tn] = *reinterpret_cast<const bf16x8*>(Bs3 + pl * QPLB + ((n2 * PTN + tn) * 32 + col) * 32 + ((half ^ ((col >> 3) & 1)) * 16));
          constexpr int NQP = QPL == 3 ? 6 : 3;        // three planes: A l, h, m, m, h, h x B h, l, m, h, m, h; two: m.h, h.m, h.h
          constexpr int PA6[6] = {QPL == 3 ? 2 : 1, 0, QPL == 3 ? 1 : 0, 1, 0, 0};
          constexpr int PB6[6] = {0, QPL == 3 ? 2 : 1, QPL == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
          for (int c = 0; c < NQP; ++c)
#pragma unroll
            for (int tn = 0; tn < PTN; ++tn)
              acc2[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[PA6[c]], bx[PB6[c]][tn], acc2[tn], 0, 0, 0);
          lds_barrier();
        }
      } else {
#pragma unroll
      for (int kc = 0; kc < C / 32; ++kc) {
        if constexpr (QBUF == 2) {
          if (kc + 1 < C / 32) store_q((kc + 1) & 1);
          if (kc + 2 < C / 32) load_q(kc + 2);
        } else if (kc > 0) {                         // one buffer: chunk kc replaces chunk kc - 1 between two barriers
          store_q(0);
          if (kc + 1 < C / 32) load_q(kc + 1);
          lds_barrier();
        }
        const float* Bs2 = Bq + (QBUF == 2 ? (kc & 1) : 0) * (C * LDK);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 af = *reinterpret_cast<const f32x4*>(&Tt[(m4 * 32 + col) * TP + kc * 32 + q * 8 + koff]);
          af = af * af;
          f32x4 bf2[PTN];
#pragma unroll
          for (int tn = 0; tn < PTN; ++tn)
            bf2[tn] = *reinterpret_cast<const f32x4*>(&Bs2[((n2 * PTN + tn) * 32 + col) * LDK + q * 8 + koff]);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tn = 0; tn < PTN; ++tn)
              acc2[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r], bf2[tn][r], acc2[tn], 0, 0, 0);
        }
        lds_barrier();
      }
      }
      // ---- epilogue of the part: this wave's private 32 x (PTN * 32) block of the tile ---------------
      float* const cb = Tt + (m4 * 32 + 4 * half) * TP + n2 * PTN * 32 + col;     // C-layout base
      const int er = lane >> 3, ec = lane & 7;                                    // row-major: 8 lanes per row
      const float* const rb0 = Tt + (m4 * 32 + er) * TP + n2 * PTN * 32 + ec * 4;
      long long px[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) px[k] = rowpix[h * HR + m4 * 32 + er + 8 * k];
      auto emit = [&](float* dst, auto nt_c) {       // the block, row-major, 16 bytes per lane
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < PTN; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rb0 + k * 8 * TP + j * 32);
            if (px[k] >= 0) {
              f32x4* const q = reinterpret_cast<f32x4*>(dst + (size_t)px[k] * a.out_cs + a.out_coff + n2 * PTN * 32 + j * 32 + ec * 4);
              if constexpr (decltype(nt_c)::value) __builtin_nontemporal_store(v, q);
              else *q = v;
            }
          }
      };
      if (a.out) {                                   // u (null: the backward pass forms it as v / s)
        emit(a.out, std::false_type{});
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int tn = 0; tn < PTN; ++tn)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float* e = cb + ((reg & 3) + 8 * (reg >> 2)) * TP + tn * 32;
          const float sv = sqrtf(acc2[tn][reg] + beta_c[tn]);
          acc2[tn][reg] = sv;                                   // s replaces n in the accumulator registers
          *e = *e * sv;                                         // v = u * s, in place
        }
      __builtin_amdgcn_wave_barrier();
      emit(a.post_v, std::integral_constant<bool, (SGA_NT & 8) != 0>{});      // v
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int tn = 0; tn < PTN; ++tn)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          cb[((reg & 3) + 8 * (reg >> 2)) * TP + tn * 32] = acc2[tn][reg];   // s (kept where n was)
      __builtin_amdgcn_wave_barrier();
      emit(a.post_s, std::integral_constant<bool, (SGA_NT & 4) != 0>{});      // s (not read again before the backward pass)
      lds_barrier();                                 // the tile is rewritten by the next part
    }
    SGA_PROBE_END();
    return;
  }
  if (a.epi == EPI_SHUFFLE3 && a.ksplit <= 1) {
    // columns n = (py*2+px)*3 + c of a combined-phase C->3 transposed conv (scalar scatter)
    if (TN == 1 && wn == 0) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
          const int m = m0 + (wm * TM + tm) * 32 + row;
          const int n = n0 + col;
          if (m >= Mtot || n >= 12) continue;
          const int j = m % a.Wg;
          const int t = m / a.Wg;
          const int i = t % a.Hg;
          const int b = t / a.Hg;
          const int pp = n / 3, c = n - pp * 3;
          const int oy = 2 * i + (pp >> 1), ox = 2 * j + (pp & 1);
          if (oy < a.Hout && ox < a.Wout)
            a.out[((size_t)(b * a.Hout + oy) * a.Wout + ox) * 3 + c] =
                acc[tm][0][reg] + (a.bias ? a.bias[c] : 0.f);
        }
      }
    }
    return;
  }
  // Stage each wave's 32 x (TN*32) accumulator slab through LDS and emit whole 16-byte pieces of
  // output rows (coalesced float4 stores and aux loads instead of 4-byte column scatters).
  const int epi = a.ksplit > 1 ? -1 : a.epi;      // split-K: raw partial sums, epilogue in the reduce
  float* const outp = a.ksplit > 1 ? a.part + (size_t)split * a.slab : a.out;
  float* Cs = smem + wid * (32 * CPITCH);
  constexpr int F4_PER_ROW = TN * 8;
  constexpr int F4_ITERS = (32 * F4_PER_ROW) / 64;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        Cs[((reg & 3) + 8 * (reg >> 2) + 4 * half) * CPITCH + tn * 32 + col] = acc[tm][tn][reg];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < F4_ITERS; ++k) {
      const int f = lane + 64 * k;
      const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
      const long long px = rowpix[(wm * TM + tm) * 32 + row];
      const int n = n0 + (wn * TN) * 32 + c4 * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CPITCH + c4 * 4]);
      if (px < 0 || n >= a.Cout) continue;
      const size_t o = (size_t)px * a.out_cs + a.out_coff + n;
      switch (epi) {
        case EPI_BIAS:
          if (a.bias) v += ld4(a.bias + n);
          break;
        case EPI_BIAS_RELU: {
          if (a.bias) v += ld4(a.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } break;
        case EPI_IGDN: {
          const f32x4 u = ld4(a.aux0 + o);
          const f32x4 nb = v + ld4(a.bias + n);
          f32x4 sq;
#pragma unroll
          for (int e = 0; e < 4; ++e) sq[e] = sqrtf(nb[e]);
          *reinterpret_cast<f32x4*>(a.aux_out + o) = sq;
          v = u * sq;
        } break;
        case EPI_GDN: {
          const f32x4 u = ld4(a.aux0 + o);
          const f32x4 nb = v + ld4(a.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = u[e] / sqrtf(nb[e]);
        } break;
        case EPI_IGDN_BWD:
          v = ld4(a.in + o) * ld4(a.aux1 + o) + ld4(a.aux2 + o) * v;
          break;
        case EPI_RELU_MASK: {
          const f32x4 mk = ld4(a.aux0 + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
        } break;
        default:
          break;
      }
      *reinterpret_cast<f32x4*>(outp + o) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  SGA_PROBE_END();
}

struct ReduceArgs {
  const float* part; long long slab; long long n4;
  int cout, epi, hout, wout, s_out;
  int nsplit[4];
  const float* bias; const float* aux0; float* out;
  int side;      // the launch belongs to the hyper branch (SGA_SIDE_ELEM_PRIO)
};

// BATCH: the loads of 8 slabs are issued together and only the adds are serial (same order, same result).  A
// 75-slab reduce is otherwise a chain of 75 dependent L2 round trips (20 us against 3).  Used on the main
// chain; the hyper branch on the second stream keeps the serial form: at cfg 2 the faster reduce there makes
// `hs2.bwd` ready exactly when `gs2.fwd` starts and the ITERATION 70 us slower.
template <bool BATCH>
__global__ void splitk_reduce_kernel(const ReduceArgs r) {
#if SGA_SIDE_ELEM_PRIO
  if (r.side) __builtin_amdgcn_s_setprio(SGA_SIDE_ELEM_PRIO);
#endif
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < r.n4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int c = (int)(e % r.cout);
    int S = r.nsplit[0];
    if (r.s_out == 2) {      // transposed conv: the split factor depends on the sub-pixel phase
      const long long pix = e / r.cout;
      const int ox = (int)(pix % r.wout), oy = (int)((pix / r.wout) % r.hout);
      S = r.nsplit[(oy & 1) * 2 + (ox & 1)];
    }
    f32x4 acc = ld4(r.part + e);
    if constexpr (BATCH) {
      for (int s = 1; s < S; s += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int sk = s + k < S ? s + k : S - 1;
          v[k] = ld4(r.part + (size_t)sk * r.slab + e);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s + k < S) acc += v[k];                                            // fixed order
      }
    } else {
      for (int s = 1; s < S; ++s) acc += ld4(r.part + (size_t)s * r.slab + e);   // fixed order
    }
    if ((r.epi == EPI_BIAS || r.epi == EPI_BIAS_RELU) && r.bias) acc += ld4(r.bias + c);
    if (r.epi == EPI_BIAS_RELU) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = fmaxf(acc[k], 0.f);
    } else if (r.epi == EPI_RELU_MASK) {
      const f32x4 m = ld4(r.aux0 + e);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = m[k] > 0.f ? acc[k] : 0.f;
    }
    *reinterpret_cast<f32x4*>(r.out + e) = acc;
  }
}

template <int TM, int TN, int WM, int WN, int PRO, bool SMALLC, int X3 = 0, int POST = 0>
int launch_inst(const ConvArgs& a, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr bool GLDS = glds_instance(TM, TN, WM, WN, PRO, SMALLC, X3, POST);
  constexpr int MAIN_FLOATS = X3 ? x3_main_floats(BM, BN, x3_pipelined(PRO, SMALLC, X3))
                                 : (GLDS ? 2 * (BM + BN) * 32 : (BM + BN) * LDK);
  constexpr int EPI_FLOATS = WM * WN * 32 * (TN * 32 + 4);
  constexpr int POST_FLOATS = POST == 1 ? (post_rows(BM, BN) * (BN + 4) + post_qbufs(BM) * BN * LDK) : 0;
  constexpr int F0 = MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS;
  const size_t lds = (size_t)(F0 > POST_FLOATS ? F0 : POST_FLOATS) * sizeof(float) + BM * sizeof(long long);
  // once per (instance, device): a process may drive several GPUs, and handles may live on several threads
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<TM, TN, WM, WN, PRO, SMALLC, X3, POST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  int grid = a.nphase * a.tiles_per_phase * a.ntiles_n;
  if (a.ksplit > 1) {
    grid = 0;
    for (int p = 0; p < a.nphase; ++p) grid += a.tiles_per_phase * a.ntiles_n * a.nsplit[p];
  }
  if (grid <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((conv_mfma_kernel<TM, TN, WM, WN, PRO, SMALLC, X3, POST>), dim3(grid), dim3(NT), lds, stream, a);
  return (int)hipGetLastError();
}

// bf16-pipe PRO_NONE instance: a.x3 == 2 -> the two-plane loop (bf16x2), else three planes (bf16x3)
template <int TM, int TN, int WM, int WN, int POST = 0>
int launch_xp(const ConvArgs& a, hipStream_t s) {
  if (a.x3 == 2) return launch_inst<TM, TN, WM, WN, PRO_NONE, false, 2, POST>(a, s);
  return launch_inst<TM, TN, WM, WN, PRO_NONE, false, 1, POST>(a, s);
}

template <int TM, int TN, int WM, int WN>
int launch_pro(const ConvArgs& a, hipStream_t s) {
  if (a.smallc) return launch_inst<TM, TN, WM, WN, PRO_NONE, true>(a, s);
  if (a.x3) {
    switch (a.pro) {
      case PRO_NONE: return launch_xp<TM, TN, WM, WN>(a, s);
      case PRO_SQUARE: return launch_inst<TM, TN, WM, WN, PRO_SQUARE, false, 1>(a, s);
      case PRO_IGDN_BWD: return launch_inst<TM, TN, WM, WN, PRO_IGDN_BWD, false, 1>(a, s);
    }
  }
  switch (a.pro) {
    case PRO_NONE: return launch_inst<TM, TN, WM, WN, PRO_NONE, false>(a, s);
    case PRO_SQUARE: return launch_inst<TM, TN, WM, WN, PRO_SQUARE, false>(a, s);
    case PRO_IGDN_BWD: return launch_inst<TM, TN, WM, WN, PRO_IGDN_BWD, false>(a, s);
  }
  return (int)hipErrorInvalidValue;
}

}  // namespace

int conv_pick_bn(int cout, int epi) {
  if (epi == EPI_SHUFFLE3) return 32;
  const int cand[4] = {256, 192, 96, 64};
  int best = 0, best_cost = 1 << 30;
  for (int k = 0; k < 4; ++k) {
    const int bn = cand[k];
    const int cost = (cout + bn - 1) / bn * bn;
    if (cost < best_cost) { best_cost = cost; best = bn; }   // ties keep the larger (earlier) BN
  }
  return best;
}

void conv_kernel_name(const ConvArgs& a, char* out, int len) {
  const int bn = a.Npad / a.ntiles_n;
  int tn = 0, wm = 2, wn = 2, tm = 2;
  switch (bn) {
    case 192: tn = 3; break;
    case 256: tn = 4; break;
    case 64: tn = 1; break;
    case 96: tn = 3; wn = 1; break;
    case 32: tn = 1; tm = 1; wm = 4; wn = 1; break;
  }
  if (a.bm == 256) wm = 4;
  if (a.bm == 64) tm = 1;
  // the symbol as rocprofv3 prints it, spaces removed (profiles/*_kernel_stats.csv, *_pmc_traffic.json); the 7th argument is the
  // precision: 0 = f32 MFMA, 1 = bf16x3, 2 = bf16x2
  snprintf(out, len, "conv_mfma_kernel<%d,%d,%d,%d,%d,%s,%d,%d>", tm, tn, wm, wn, a.smallc ? 0 : a.pro,
           a.smallc ? "true" : "false",
           (a.x3 && !a.smallc && bn != 32) ? ((a.x3 == 2 && a.pro == PRO_NONE) ? 2 : 1) : 0,
           a.post ? 1 : 0);
}

int launch_splitk_reduce(const ConvArgs& a, long long n, hipStream_t stream) {
  ReduceArgs r;
  r.part = a.part; r.slab = a.slab; r.n4 = n / 4;
  r.cout = a.Cout; r.epi = a.epi; r.hout = a.Hout; r.wout = a.Wout; r.s_out = a.s_out;
  for (int p = 0; p < 4; ++p) r.nsplit[p] = a.nsplit[p] > 0 ? a.nsplit[p] : 1;
  // the phase table is ordered heaviest-first == (py,px) = (0,0),(0,1),(1,0),(1,1): index py*2+px
  r.bias = a.bias; r.aux0 = a.aux0; r.out = a.out; r.side = a.side;
  long long g = (r.n4 + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  if (a.reduce_batch) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((unsigned)g), dim3(256), 0, stream, r);
  else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((unsigned)g), dim3(256), 0, stream, r);
  return (int)hipGetLastError();
}

int launch_conv(const ConvArgs& a, hipStream_t stream) {
  const int bn = a.Npad / a.ntiles_n;
  if ((a.bm == 64 || a.bm == 256) && !a.x3 && !a.zeros) return (int)hipErrorInvalidValue;   // LDS-DMA instances read the zero page
  switch (bn) {
    case 192:
      if (a.bm == 64) {
        if (a.smallc || a.pro != PRO_NONE) return (int)hipErrorInvalidValue;
        if (a.post) {
          if (a.ksplit > 1 || a.epi != EPI_BIAS || a.Cout != 192 || a.out_coff != 0 || a.out_cs != 192)
            return (int)hipErrorInvalidValue;
          if (a.x3) return launch_xp<1, 3, 2, 2, 1>(a, stream);
          return launch_inst<1, 3, 2, 2, PRO_NONE, false, false, 1>(a, stream);
        }
        if (a.x3) return launch_xp<1, 3, 2, 2>(a, stream);
        return launch_inst<1, 3, 2, 2, PRO_NONE, false>(a, stream);
      }
      if (a.bm == 256) {
        if (a.smallc || a.pro != PRO_NONE) return (int)hipErrorInvalidValue;
        if (a.post) {
          if (a.ksplit > 1 || a.epi != EPI_BIAS || a.Cout != 192 || a.out_coff != 0 || a.out_cs != 192)
            return (int)hipErrorInvalidValue;
          if (a.x3) return launch_xp<2, 3, 4, 2, 1>(a, stream);
          return launch_inst<2, 3, 4, 2, PRO_NONE, false, false, 1>(a, stream);
        }
        if (a.x3) return launch_xp<2, 3, 4, 2>(a, stream);
        return launch_inst<2, 3, 4, 2, PRO_NONE, false>(a, stream);
      }
      return launch_pro<2, 3, 2, 2>(a, stream);
    case 256:
      if (a.bm == 256) {      // C = 256: 8 waves, 256 x 256 tile, LDS-DMA loop
        if (a.smallc || a.pro != PRO_NONE || a.x3) return (int)hipErrorInvalidValue;
        if (a.post) {
          if (a.ksplit > 1 || a.epi != EPI_BIAS || a.Cout != 256 || a.out_coff != 0 || a.out_cs != 256)
            return (int)hipErrorInvalidValue;
          return launch_inst<2, 4, 4, 2, PRO_NONE, false, false, 1>(a, stream);
        }
        return launch_inst<2, 4, 4, 2, PRO_NONE, false>(a, stream);
      }
      return launch_pro<2, 4, 2, 2>(a, stream);
    case 64: return launch_pro<2, 1, 2, 2>(a, stream);
    case 96:
      if (a.smallc || a.pro != PRO_NONE) return (int)hipErrorInvalidValue;
      if (a.x3) return launch_xp<2, 3, 2, 1>(a, stream);
      return launch_inst<2, 3, 2, 1, PRO_NONE, false>(a, stream);
    case 32:
      if (a.smallc || a.pro != PRO_NONE) return (int)hipErrorInvalidValue;
      return launch_inst<1, 1, 4, 1, PRO_NONE, false>(a, stream);
  }
  return (int)hipErrorInvalidValue;
}
