// GDN / IGDN (tfc.GDN, nn_models.py:17-25,51-59) and the IGDN data-gradient as ONE tile kernel that
// touches every HBM byte once and absorbs the work of its neighbours:
//
//   prologue   T [BM pixels x C] =  sum of the producing convolution's split-K partial slabs (+ bias)
//                                   (what splitk_reduce_kernel + a re-read used to do), or
//                                   a fresh 5x5/2 convolution of the 3-channel gradient image
//                                   (the data-gradient of the C->3 synthesis layer, nn_models.py:63),
//                                   computed on the matrix pipe straight into the tile;
//   operand    A = T^2 (forward)  or  g*u/s (backward), kept RESIDENT in LDS for the whole C x C
//              contraction with gamma: no re-staging per K-step, no second read from HBM;
//   epilogue   forward : n = A.gamma + beta, s = sqrt(n), v = T*s (IGDN) | T/s (GDN)   -> s, v (+ T)
//              backward: g_u = g*s + u*(A.gamma)                                        -> g_u
//              from per-thread registers that were filled by the SAME coalesced loads as the tile.
//
// A tile is BM consecutive NHWC pixels = one contiguous block of BM*C floats, so all global traffic
// is linear 16-byte accesses.  The contraction runs on v_mfma_f32_32x32x2_f32 with the K order of
// conv_mfma.hip (bitwise the same f32 fmaf chain as the stand-alone kernels it replaces).
//
// LDS: tile [BM][C+4] (pitch = 4 mod 64 banks: conflict-free ds_read_b128 fragments, conflict-free
// C-layout scatter) + one K-step of gamma [C][36]; BM = 64, C = 192: 78 KB -> 2 workgroups per CU,
// so one workgroup's loads/stores overlap the other's MFMAs.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "sga_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

// exact split of 4 floats into three planes of 4 packed bf16 (conv_mfma.hip: precision mode bf16x3)
__device__ __forceinline__ unsigned gcvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void gsplit3(const f32x4 v, u32x2& h, u32x2& m, u32x2& l) {
  h.x = gcvt_pk_bf16(v.x, v.y);
  h.y = gcvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(h.x << 16), r1 = v.y - __uint_as_float(h.x & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(h.y << 16), r3 = v.w - __uint_as_float(h.y & 0xffff0000u);
  m.x = gcvt_pk_bf16(r0, r1);
  m.y = gcvt_pk_bf16(r2, r3);
  l.x = gcvt_pk_bf16(r0 - __uint_as_float(m.x << 16), r1 - __uint_as_float(m.x & 0xffff0000u));
  l.y = gcvt_pk_bf16(r2 - __uint_as_float(m.y << 16), r3 - __uint_as_float(m.y & 0xffff0000u));
}


constexpr int LDK = 36;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x2 ld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }

template <int NC, int WM, int WN, int MODE, int PRO, bool X3 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gdn_tile_kernel(const GdnArgs a) {
  constexpr int C = NC * 32, NT = WM * WN * 64, BM = WM * 32, TN = NC / WN;
  constexpr int TP = C + 4;                      // tile pitch (floats)
  constexpr int C4 = C / 4;                      // float4 per tile row
  constexpr int NF = BM * C4 / NT;               // float4 pieces owned by a thread
  // piece (kr, kc) of thread (pr, pc): tile row pr + R*kr, float4 column pc + CW*kc.  Every address is
  // then one per-thread base + a compile-time offset (no per-piece address registers)
  constexpr int CW = (WN == NC) ? C4 : 16, R = NT / CW, KR = BM / R, KC = C4 / CW;
  static_assert(NT % CW == 0 && BM % R == 0 && C4 % CW == 0 && KR * KC == NF, "piece mapping");
  constexpr int RPP = NT / 8;                    // rows covered by one pass of the K-step loaders
  constexpr int PB = C / RPP;
  static_assert(NC % WN == 0 && (BM * C4) % NT == 0 && C % RPP == 0, "tile/loader mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Tt = smem;                              // [BM][TP]
  float* Bs = smem + BM * TP;                    // [C][LDK]

  if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int chunk = tid & 7, lrow = tid >> 3;
  const int half = lane >> 5, col = lane & 31, koff = half * 4;
  const float* __restrict__ gw = a.w;
  const long long m0 = (long long)blockIdx.x * BM;
#ifdef SGA_CLOCK_PROBE
#define GDN_STAMP(k) do { if (a.clk && tid == 0) a.clk[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
  if (a.clk && tid == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.clk[8 * blockIdx.x + 5] = (unsigned long long)hwid | ((unsigned long long)xcc << 32);
  }
#else
#define GDN_STAMP(k) do { } while (0)
#endif
  GDN_STAMP(0);

  f32x4 rb[PB];
  auto load_b = [&](const float* w, int kpitch, int k0) {
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = ld4(w + (size_t)(p * RPP + lrow) * kpitch + k0 + chunk * 4);
  };
  auto store_b = [&]() {
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<f32x4*>(&Bs[(p * RPP + lrow) * LDK + chunk * 4]) = rb[p];
  };

  f32x16 acc[TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
  };
  // 32 k's of A (rows of `A`, pitch `ap`, starting at float offset k0) against the staged Bs
  auto mfma_step = [&](const float* A, int ap, int k0) {
    const int arow = wm * 32 + col;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // backward keeps 96 epilogue registers alive through this loop: stop the scheduler from hoisting
      // all four q's fragment reads (64 VGPRs) above the MFMAs, which would spill
      if (MODE == GDN_IGDN_BWD && TN >= 3 && (q & 1) == 0 && q) __builtin_amdgcn_sched_barrier(0);
      const f32x4 af = *reinterpret_cast<const f32x4*>(&A[arow * ap + k0 + q * 8 + koff]);
      f32x4 bf[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        bf[tn] = *reinterpret_cast<const f32x4*>(&Bs[((wn * TN + tn) * 32 + col) * LDK + q * 8 + koff]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r], bf[tn][r], acc[tn], 0, 0, 0);
    }
  };
  // accumulators -> tile (C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3)+8*(reg>>2)+4*half)
  auto acc_to_tile = [&]() {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        Tt[(wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half) * TP + (wn * TN + tn) * 32 + col] = acc[tn][reg];
  };

  // ---- prologue (b): g = 5x5/2 convolution of the zero-bordered 3-channel image ---------------
  if constexpr (PRO == GDN_PRO_CONV3) {
    constexpr int PA = (BM + RPP - 1) / RPP;     // (32-row tile x 6 waves: one pass of the loader covers 48 rows, 32 are real)
    constexpr bool PA_TAIL = BM % RPP != 0;
    float* As = Tt;                              // [BM][LDK], aliases the (not yet filled) tile
    int a_off[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const long long m = m0 + p * RPP + lrow;
      a_off[p] = -1;
      if (m < a.M && (!PA_TAIL || p * RPP + lrow < BM)) {
        const int j = (int)(m % a.Wg);
        const long long t = m / a.Wg;
        const int i = (int)(t % a.Hg), b = (int)(t / a.Hg);
        a_off[p] = ((b * a.Hp + 2 * i) * a.Wp + 2 * j) * 3;
      }
    }
    zero_acc();
    // All three K-steps' operands are requested up front (registers are free here: the fill's streams and the
    // epilogue registers are not live yet).  Step by step -- load, LDS, barrier, multiply -- the prologue was three
    // dependent round trips to L2: 11.1 us per workgroup for 3.9 us of MFMA work (in-kernel stamps,
    // profiles/r02_clock_probe.txt).
    // (the 32-row shape -- three workgroups per CU, 96 VGPRs -- takes the steps one at a time: its latency is hidden by the
    // other resident workgroups, and 60 staging registers would cost it the third one)
    constexpr int NS = PA_TAIL ? 1 : 3;
    f32x4 ra[NS][PA], rbs[NS][PB];
    auto load_step = [&](int step, int slot) {
      // one K-step = kernel rows (2*step, 2*step+1), 16 floats each (5 taps x 3 channels + 1 of slack
      // that meets a zero weight); row 5 does not exist: its weights are zero, re-read row 4
      int ky = 2 * step + (chunk >> 2);
      ky = ky > 4 ? 4 : ky;
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const bool okp = a_off[p] >= 0;
        const float* src = a.pad + (size_t)(okp ? a_off[p] : 0) + (size_t)ky * a.Wp * 3 + (chunk & 3) * 4;
        const f32x2 lo = ld2(src), hi = ld2(src + 2);
        ra[slot][p] = okp ? f32x4{lo.x, lo.y, hi.x, hi.y} : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int p = 0; p < PB; ++p) rbs[slot][p] = ld4(a.wc + (size_t)step * C * 32 + (size_t)(p * RPP + lrow) * 32 + chunk * 4);
    };
    if constexpr (NS == 3) {
#pragma unroll
      for (int step = 0; step < 3; ++step) load_step(step, step);
    }
#pragma unroll
    for (int step = 0; step < 3; ++step) {
      const int slot = NS == 3 ? step : 0;
      if constexpr (NS == 1) load_step(step, 0);
#pragma unroll
      for (int p = 0; p < PA; ++p)
        if (!PA_TAIL || p * RPP + lrow < BM) *reinterpret_cast<f32x4*>(&As[(p * RPP + lrow) * LDK + chunk * 4]) = ra[slot][p];
#pragma unroll
      for (int p = 0; p < PB; ++p) *reinterpret_cast<f32x4*>(&Bs[(p * RPP + lrow) * LDK + chunk * 4]) = rbs[slot][p];
      __syncthreads();
      mfma_step(As, LDK, 0);
      __syncthreads();
    }
    acc_to_tile();                               // g in the tile (As is dead: every wave passed the barrier)
    __syncthreads();
  }

  GDN_STAMP(1);
  // first K-step of gamma: in flight during the fill (backward: after it -- its three input streams
  // leave no registers for it)
  // (bf16x3 form of the contraction, X3: gamma comes pre-split into its bf16 planes -- a.wx3, the planes sga_create packs
  //  for every weight -- one 16-wide stage at a time as [plane][C][32 bytes]; see the contraction below)
  constexpr int GPLB = C * 32 + 96;              // plane pitch of the staged gamma: + 96 bytes against write bank conflicts (conv_mfma.hip)
  static_assert(3 * GPLB <= C * LDK * 4, "gamma stage must fit Bs");
  constexpr int NBX = X3 ? (C * 6 + NT - 1) / NT : 1;
  u32x4 bxr[NBX];
  char* const Bsb = reinterpret_cast<char*>(Bs);
  auto load_bx = [&](int st) {
    const unsigned short* wx = a.wx3 + (size_t)(st >> 1) * 96 + (st & 1) * 16;
#pragma unroll
    for (int k = 0; k < NBX; ++k) {
      const int f = tid + NT * k;
      const int nl = f / 6, r6 = f - nl * 6;
      if ((C * 6) % NT == 0 || k + 1 < NBX || f < C * 6)
        bxr[k] = *reinterpret_cast<const u32x4*>(wx + (size_t)nl * (C / 32) * 96 + (r6 >> 1) * 32 + (r6 & 1) * 8);
    }
  };
  auto store_bx = [&]() {
#pragma unroll
    for (int k = 0; k < NBX; ++k) {
      const int f = tid + NT * k;
      const int nl = f / 6, r6 = f - nl * 6;
      if ((C * 6) % NT == 0 || k + 1 < NBX || f < C * 6)
        // (the two 16-byte halves swapped in rows 8..15 mod 16: a ds_read_b128 is served 16 lanes at a time, and 16 rows x 16 bytes at
        //  a 32-byte pitch would hit every bank twice -- 41 % bank conflicts measured, profiles/r04_pmc_kernels.txt)
        *reinterpret_cast<u32x4*>(Bsb + (r6 >> 1) * GPLB + nl * 32 + (((r6 & 1) ^ ((nl >> 3) & 1)) * 16)) = bxr[k];
    }
  };
  if constexpr (MODE != GDN_IGDN_BWD) { if constexpr (X3) load_bx(0); else load_b(gw, C, 0); }

  // ---- fill: one coalesced pass over the tile's inputs ----------------------------------------
  // All loads are unconditional (rows past M re-read row m0 and are discarded by a select) and issued
  // one tile-row group (KC pieces per stream) at a time.
  const int pr = tid / CW, pc = tid - pr * CW;
  const long long rows_left = a.M - m0;
  const int rvalid = rows_left >= BM ? BM : (int)rows_left;                // rows r < rvalid are real
  const size_t tbase = (size_t)m0 * C + (size_t)pc * 4;                    // + row * C + kc * CW * 4
  float* const tp = Tt + pr * TP + pc * 4;                                 // + kr * R * TP + kc * CW * 4
  int smax = a.nsplit[0];
  if (a.s_out == 2) smax = max(max(a.nsplit[0], a.nsplit[1]), max(a.nsplit[2], a.nsplit[3]));
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // beta and the conv bias depend on the piece's column only: KC float4 each, loaded ONCE up front.  A load
  // inside the epilogue would make every piece wait (vmcnt is in-order and counts stores) for the stores
  // of the pieces before it.
  f32x4 betav[KC], biasv[KC];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    betav[kc] = (MODE != GDN_IGDN_BWD) ? ld4(a.beta + pc * 4 + kc * CW * 4) : zero4;
    biasv[kc] = (PRO != GDN_PRO_CONV3 && a.bias) ? ld4(a.bias + pc * 4 + kc * CW * 4) : zero4;
  }
  f32x4 e1[NF], e2[MODE == GDN_IGDN_BWD ? NF : 1];
  // Row groups are software-pipelined: group kr+1's loads are issued before group kr is consumed, so two
  // groups are in flight (not all four: the backward instance has three input streams and 96 epilogue
  // registers to keep; a compiler barrier stops hipcc from hoisting every load to the top and spilling).
  f32x4 t[2][KC], uu[MODE == GDN_IGDN_BWD ? 2 : 1][KC], ss[MODE == GDN_IGDN_BWD ? 2 : 1][KC];
  auto issue = [&](int kr, int slot) {
    const int row = pr + R * kr;
    const size_t ro = tbase + (size_t)(row < rvalid ? row : 0) * C;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if constexpr (PRO == GDN_PRO_CONV3) t[slot][kc] = *reinterpret_cast<const f32x4*>(tp + kr * R * TP + kc * CW * 4);
      else t[slot][kc] = ld4(a.src + ro + kc * CW * 4);
      if constexpr (MODE == GDN_IGDN_BWD) {
#if SGA_IGDN_NT & 1      // read once by this launch, dead afterwards (round 5)
        uu[slot][kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((a.v ? a.v : a.u) + ro + kc * CW * 4));
        ss[slot][kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.s + ro + kc * CW * 4));
#else
        uu[slot][kc] = ld4((a.v ? a.v : a.u) + ro + kc * CW * 4);
        ss[slot][kc] = ld4(a.s + ro + kc * CW * 4);
#endif
      }
    }
  };
  issue(0, 0);
#pragma unroll
  for (int kr = 0; kr < KR; ++kr) {
    const int slot = kr & 1;
    const int row = pr + R * kr;
    const bool ok = row < rvalid;
    const size_t ro = tbase + (size_t)(ok ? row : 0) * C;
    if (kr + 1 < KR) issue(kr + 1, slot ^ 1);
    if constexpr (MODE == GDN_IGDN_BWD) asm volatile("" ::: "memory");
    if constexpr (PRO != GDN_PRO_CONV3) {
      if (smax > 1) {
        int srow = a.nsplit[0];
        if (a.s_out == 2) {                      // transposed conv: the split factor depends on the phase
          const long long m = m0 + row;
          const int ox = (int)(m % a.wout), oy = (int)((m / a.wout) % a.hout);
          const int ph = (oy & 1) * 2 + (ox & 1);
          srow = ph == 0 ? a.nsplit[0] : (ph == 1 ? a.nsplit[1] : (ph == 2 ? a.nsplit[2] : a.nsplit[3]));
        }
#pragma unroll 4
        for (int s = 1; s < smax; ++s) {         // fixed order (= splitk_reduce_kernel), uniform trip count
          const float* __restrict__ ps = a.src + (size_t)s * a.slab + ro;
          f32x4 x[KC];
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) x[kc] = ld4(ps + kc * CW * 4);
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) t[slot][kc] = s < srow ? t[slot][kc] + x[kc] : t[slot][kc];
        }
      }
      if (a.bias) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) t[slot][kc] += biasv[kc];
      }
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      f32x4 av;
      if constexpr (MODE == GDN_IGDN_BWD) {
        f32x4 uv = uu[slot][kc];
        if (a.v) {
          // the forward pass stored v = u*s and s only (u is a third of the IGDN's write traffic): u = v/s, with
          // v_rcp_f32 (1 ulp; the data-gradient is compared with float64 autograd at 1e-4)
          f32x4 ri;
#pragma unroll
          for (int x = 0; x < 4; ++x) ri[x] = __builtin_amdgcn_rcpf(ss[slot][kc][x]);
          uv = uv * ri;
          av = t[slot][kc] * uv * ri;                        // operand of the contraction: g*u/s
        } else {
          av = t[slot][kc] * uv / ss[slot][kc];
        }
        e1[kr * KC + kc] = t[slot][kc] * ss[slot][kc];       // g*s
        e2[kr * KC + kc] = uv;
      } else {
        av = t[slot][kc] * t[slot][kc];
        e1[kr * KC + kc] = t[slot][kc];
      }
      *reinterpret_cast<f32x4*>(tp + kr * R * TP + kc * CW * 4) = ok ? av : zero4;
    }
    if constexpr (MODE == GDN_IGDN_BWD) asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  GDN_STAMP(2);
  // ---- C x C contraction with gamma, A resident in the tile ------------------------------------
  if constexpr (MODE == GDN_IGDN_BWD) { if constexpr (X3) load_bx(0); else load_b(gw, C, 0); }
  zero_acc();
  if constexpr (X3) {
    // 16-wide stages on v_mfma_f32_32x32x16_bf16: the resident f32 operand is split into its three bf16 planes as the
    // fragments leave the tile (8 floats per lane and stage); six plane products per stage, smallest first (conv_mfma.hip)
#pragma unroll 1
    for (int st = 0; st < C / 16; ++st) {
      store_bx();
      __syncthreads();                           // (st = 0: also publishes the tile)
      if (st + 1 < C / 16) load_bx(st + 1);
      const float* ap = &Tt[(wm * 32 + col) * TP + st * 16 + half * 8];
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
      u32x2 h0, m0, l0, h1, m1, l1;
      gsplit3(a0, h0, m0, l0);
      gsplit3(a1, h1, m1, l1);
      bf16x8 ax[3];
      ax[0] = __builtin_bit_cast(bf16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
      ax[1] = __builtin_bit_cast(bf16x8, u32x4{m0.x, m0.y, m1.x, m1.y});
      ax[2] = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
      constexpr int PA6[6] = {2, 0, 1, 1, 0, 0};
      constexpr int PB6[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {          // one output block at a time: 12 fragment registers live, not 12 x TN
        bf16x8 bx[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bx[pl] = *reinterpret_cast<const bf16x8*>(Bsb + pl * GPLB + ((wn * TN + tn) * 32 + col) * 32 + ((half ^ ((col >> 3) & 1)) * 16));
#pragma unroll
        for (int c = 0; c < 6; ++c)
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[PA6[c]], bx[PB6[c]], acc[tn], 0, 0, 0);
      }
      __syncthreads();
    }
  } else {
#pragma unroll 1
  for (int kc = 0; kc < NC; ++kc) {
    store_b();
    __syncthreads();                             // (kc = 0: also publishes the tile)
    if (kc + 1 < NC) load_b(gw, C, (kc + 1) * 32);
    mfma_step(Tt, TP, kc * 32);
    __syncthreads();
  }
  }
  acc_to_tile();
  __syncthreads();
  GDN_STAMP(3);

  // ---- epilogue: row-major 16-byte pieces, operands from registers ----------------------------
#pragma unroll
  for (int kr = 0; kr < KR; ++kr) {
    const int row = pr + R * kr;
    if (row < rvalid) {
      const size_t ro = tbase + (size_t)row * C;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int k = kr * KC + kc;
        const size_t e = ro + kc * CW * 4;
        const f32x4 n = *reinterpret_cast<const f32x4*>(tp + kr * R * TP + kc * CW * 4);
        if constexpr (MODE == GDN_IGDN_BWD) {
#if SGA_IGDN_NT & 2
          __builtin_nontemporal_store(e1[k] + e2[k] * n, reinterpret_cast<f32x4*>(a.out + e));
#else
          *reinterpret_cast<f32x4*>(a.out + e) = e1[k] + e2[k] * n;
#endif
        } else {
          const f32x4 nb = n + betav[kc];
          f32x4 sq, v;
#pragma unroll
          for (int x = 0; x < 4; ++x) sq[x] = sqrtf(nb[x]);
          if constexpr (MODE == GDN_IGDN_FWD) {
            v = e1[k] * sq;
            if (a.s_out_p) {
#if SGA_NT & 32
              __builtin_nontemporal_store(sq, reinterpret_cast<f32x4*>(a.s_out_p + e));
#else
              *reinterpret_cast<f32x4*>(a.s_out_p + e) = sq;
#endif
            }
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = e1[k][x] / sq[x];
          }
          if (a.u_out) *reinterpret_cast<f32x4*>(a.u_out + e) = e1[k];
#if SGA_NT & 64
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + e));
#else
          *reinterpret_cast<f32x4*>(a.out + e) = v;
#endif
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  GDN_STAMP(4);
}

template <int NC, int WM, int WN, int MODE, int PRO, bool X3 = false>
int launch_inst(const GdnArgs& a, hipStream_t stream) {
  constexpr int C = NC * 32, BM = WM * 32, NT = WM * WN * 64;
  const size_t lds = (size_t)(BM * (C + 4) + C * LDK) * sizeof(float);
  // once per (instance, device): a process may drive several GPUs, and handles may live on several threads
  static std::atomic<unsigned long long> attr_devs{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gdn_tile_kernel<NC, WM, WN, MODE, PRO, X3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  const long long grid = (a.M + BM - 1) / BM;
  if (grid <= 0 || grid > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((gdn_tile_kernel<NC, WM, WN, MODE, PRO, X3>), dim3((unsigned)grid), dim3(NT), lds, stream, a);
  return (int)hipGetLastError();
}

template <int NC, int WM, int WN>
int launch_mode(const GdnArgs& a, hipStream_t s) {
  if (a.pro == GDN_PRO_CONV3) {
    if (a.mode != GDN_IGDN_BWD) return (int)hipErrorInvalidValue;
    if (a.x3 && a.wx3) return launch_inst<NC, WM, WN, GDN_IGDN_BWD, GDN_PRO_CONV3, true>(a, s);
    return launch_inst<NC, WM, WN, GDN_IGDN_BWD, GDN_PRO_CONV3>(a, s);
  }
  if (a.x3 && a.wx3) {
    switch (a.mode) {
      case GDN_IGDN_FWD: return launch_inst<NC, WM, WN, GDN_IGDN_FWD, GDN_PRO_LOAD, true>(a, s);
      case GDN_GDN_FWD: return launch_inst<NC, WM, WN, GDN_GDN_FWD, GDN_PRO_LOAD, true>(a, s);
      case GDN_IGDN_BWD: return launch_inst<NC, WM, WN, GDN_IGDN_BWD, GDN_PRO_LOAD, true>(a, s);
    }
  }
  switch (a.mode) {
    case GDN_IGDN_FWD: return launch_inst<NC, WM, WN, GDN_IGDN_FWD, GDN_PRO_LOAD>(a, s);
    case GDN_GDN_FWD: return launch_inst<NC, WM, WN, GDN_GDN_FWD, GDN_PRO_LOAD>(a, s);
    case GDN_IGDN_BWD: return launch_inst<NC, WM, WN, GDN_IGDN_BWD, GDN_PRO_LOAD>(a, s);
  }
  return (int)hipErrorInvalidValue;
}

// tile shapes: "general" = 2 workgroups per CU of 4 waves; "small" = 32-row tiles with one 32x32
// block per wave, for launches that would otherwise leave most of the 256 CUs idle
void pick_shape(int C, long long M, bool conv3, int& wm, int& wn) {
  const int nc = C / 32;
  if (nc == 8) { wm = 1; wn = 4; } else { wm = 2; wn = 2; }
  const long long tiles = (M + wm * 32 - 1) / (wm * 32);
  static const int force = [] { const char* e = LAB_ENV("SGA_GDN_SHAPE"); return e ? atoi(e) : 0; }();   // experiments
  if ((((tiles < 256 && force != 2) || force == 1) && !conv3) || force == 3) { wm = 1; wn = nc; }      // 3: also the conv3 prologue
}

}  // namespace

int gdn_tile_rows(int C, long long M, int pro) {
  int wm, wn;
  pick_shape(C, M, pro == GDN_PRO_CONV3, wm, wn);
  return wm * 32;
}

void gdn_kernel_name(const GdnArgs& a, char* out, int len) {
  int wm, wn;
  pick_shape(a.C, a.M, a.pro == GDN_PRO_CONV3, wm, wn);
  // the symbol as rocprofv3 prints it, spaces removed (profiles/*_kernel_stats.csv, *_pmc_traffic.json)
  snprintf(out, len, "gdn_tile_kernel<%d,%d,%d,%d,%d,%s>", a.C / 32, wm, wn, a.mode, a.pro, (a.x3 && a.wx3) ? "true" : "false");
}

int launch_gdn_tile(const GdnArgs& a, hipStream_t s) {
  int wm, wn;
  pick_shape(a.C, a.M, a.pro == GDN_PRO_CONV3, wm, wn);
  switch (a.C / 32) {
    case 2: return (wm == 1) ? launch_mode<2, 1, 2>(a, s) : launch_mode<2, 2, 2>(a, s);
    case 4: return (wm == 1) ? launch_mode<4, 1, 4>(a, s) : launch_mode<4, 2, 2>(a, s);
    case 6: return (wm == 1) ? launch_mode<6, 1, 6>(a, s) : launch_mode<6, 2, 2>(a, s);
    case 8: return (wn == 8) ? launch_mode<8, 1, 8>(a, s) : launch_mode<8, 1, 4>(a, s);
  }
  return (int)hipErrorInvalidValue;
}
