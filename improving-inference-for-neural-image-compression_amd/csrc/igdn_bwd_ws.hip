// The IGDN data-gradient (tfc.GDN inverse=True, nn_models.py:51-59, its part of sga.py:164) of the LAST synthesis stage --
// `igdn2.bwd`, which also contains the data-gradient of the C -> 3 layer (nn_models.py:60-63) as a 5x5/2 convolution of
// the 3-channel gradient image -- as ONE PERSISTENT, WAVE-SPECIALISED workgroup per CU (round 5).
//
// gdn_tile_kernel (gdn_fused.hip) runs the four phases of a tile -- gradient convolution, fill, C x C contraction with
// gamma, store -- one after the other inside a 4-wave workgroup and hopes that the two workgroups of a CU de-phase: the
// matrix pipe was busy 59 % of the launch (profiles/r04_pmc_kernels.txt), 149 us for 95 us of MFMA work.  Here the
// phases of CONSECUTIVE tiles overlap by construction:
//
//   waves 0..3   (one per SIMD)  "matrix" waves: nothing but MFMAs.  Their B operands -- gamma and the 3-channel
//                kernel -- come straight from L2 into registers as ready-made fragments (packed once by sga_create in
//                fragment order: a wave-load is 1 KB contiguous), prefetched PF steps ahead, so their K loops contain
//                no barrier and no LDS write; the A operand of the contraction is resident in LDS, the A operand of the
//                gradient convolution is gathered from the (L2-resident) gradient image.
//   waves 4..11  "memory" waves (WS_LW_N = 8; 4 with 256 registers per wave measured slower): every HBM byte.  They turn the matrix waves' g tile into the contraction operand
//                g u / s IN PLACE in LDS (reading v, s once, keeping g s and u = v / s in registers) and later combine
//                those registers with the contraction result into g_u = g s + u (A . gamma) and store it.
//
//   phase p of a workgroup (tiles i = 0, 1, ... taken from a shared counter; three rotating LDS buffers X[i % 3]):
//     matrix:  conv3(i = p + 2) -> acc | Ba | acc -> X[(p+2)%3] ;  contract(i = p) from X[p%3] -> acc | Bb | acc -> X[p%3] | Bc
//     memory:  epilogue(i = p - 1) from X[(p-1)%3] | Ba | request v, s of tile p + 1; fill(i = p + 1) in X[(p+1)%3] | Bb | Bc
//   ((p+2) % 3 == (p-1) % 3: the gradient tile of i = p + 2 lands in the buffer the epilogue of i = p - 1 has just left.)
//
// LABORATORY BUILD ONLY (csrc/Makefile): measured 144.5 us alone against 148.3 for the tile kernel and +20 us inside the iteration
// (one 150-KB workgroup per CU leaves no LDS for the hyper branch's kernels); what its in-kernel stamps say about L2 latency under
// load and about the two roles sharing one vector-memory pipe: DESIGN_EXPERIMENTS.md A.11.  Harness: scripts/r05/igdn_ws_bench.hip.
//
// Arithmetic and summation order are those of gdn_tile_kernel<NC,2,2,GDN_IGDN_BWD,GDN_PRO_CONV3> (same 32 x 96 blocks per
// wave, same K order, same elementwise expressions): results are bit-identical (tests/test_gpu_fused.py).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "sga_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t FragBuf;

namespace {

constexpr int WS_BM = 64;          // rows (pixels) per tile
constexpr int WS_MW = 4;           // matrix waves
#ifndef WS_LW_N
#define WS_LW_N 8
#endif
#ifndef WS_PF_N
#define WS_PF_N 4
#endif
#ifndef WS_AQ_N
#define WS_AQ_N 2
#endif
#ifndef WS_B_AUX
#define WS_B_AUX 0                 // cache policy bits of the fragment loads (16 = sc1: bypass the vector L1)
#endif
#ifndef WS_THROTTLE
#define WS_THROTTLE 0              // 1: the memory waves request one row group at a time
#endif
constexpr int WS_LW = WS_LW_N;     // memory waves
constexpr int WS_NT = (WS_MW + WS_LW) * 64;
constexpr int WS_PF = WS_PF_N;     // gamma / kernel fragments requested this many 8-wide K steps ahead (L2 under this load: ~1.3 us = 4 steps)
constexpr int WS_AQ = WS_AQ_N;     // ... and the gradient image's
constexpr int WS_QLEN = 8;         // tile-id ring

__device__ __forceinline__ f32x4 wld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x2 wld2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
// LDS-only barrier: outstanding GLOBAL loads (the matrix waves' fragment prefetch) stay in flight across it
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NC>
__global__ __launch_bounds__(WS_NT) void igdn_bwd_ws_kernel(const GdnArgs a) {
  constexpr int C = NC * 32, TN = NC / 2, NB = NC;      // matrix waves: 2 (rows) x 2 (columns), TN 32-wide blocks each
  constexpr int TP = C + 4;                             // tile pitch (floats): conflict-free fragments and scatter
  constexpr int C4 = C / 4;
  constexpr int LT = WS_LW * 64;                        // memory threads
  constexpr int CW = 16, R = LT / CW, KR = WS_BM / R, KC = C4 / CW;   // piece (kr, kc) of a memory thread: row pr + R kr, float4 column pc + CW kc
  static_assert(NC % 2 == 0 && C4 % CW == 0 && WS_BM % R == 0 && WS_PF <= 8 && 10 % WS_AQ == 0, "shape");
  constexpr int NQG = C / 8;                            // 8-wide K steps of the contraction
  constexpr int NQC = 10;                               // ... of the gradient convolution: 5 kernel rows x 16 floats (15 + 1 zero-weight)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const X = smem;                                // [3][WS_BM][TP]
  int* const ring = reinterpret_cast<int*>(smem + 3 * WS_BM * TP);   // [WS_QLEN] tile ids (-1: none)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const long long ntiles = (a.M + WS_BM - 1) / WS_BM;
  // ---- tile ids: a shared counter (late-starting workgroups take fewer tiles), or b, b + G, ... -------------------
  auto next_tile = [&](int i) -> int {                  // one lane
    long long t;
    if (a.sched) t = (long long)atomicAdd(a.sched, 1u);
    else t = (long long)blockIdx.x + (long long)i * gridDim.x;
    return t < ntiles ? (int)t : -1;
  };
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) ring[i] = next_tile(i);
  }
  __syncthreads();
#ifdef SGA_CLOCK_PROBE      // measurement build: [workgroup][phase + 2 (0, 1: prologue)][16] wall-clock stamps; 0..7 matrix wave 0, 8..15 memory wave 0
#define WS_STAMP(ph, k) do { if (a.clk && lane == 0 && (wid == 0 || wid == WS_MW) && (ph) < 40) { const size_t i_ = ((size_t)blockIdx.x * 40 + (ph)) * 16 + (k); a.clk[i_] = wall_clock64(); a.clk[i_ + 256 * 40 * 16] = __builtin_readcyclecounter(); } } while (0)
#else
#define WS_STAMP(ph, k) do { } while (0)
#endif

  if (wid < WS_MW) {
    // =============================== matrix waves ===============================================
#ifdef WS_MATRIX_PRIO
    __builtin_amdgcn_s_setprio(WS_MATRIX_PRIO);
#endif
    const int wm = wid >> 1, wn = wid & 1;
    const int half = lane >> 5, col = lane & 31, koff = half * 4;
    const int arow = wm * 32 + col;
    // fragment loads through buffer descriptors: address = descriptor base (SGPRs) + one 32-bit lane offset (the same
    // register for every load) + a constant in the instruction's scalar offset -- no per-load 64-bit address registers
    // (flat pointers: the compiler hoists 100 of them out of the loops and spills)
    const unsigned lob = (unsigned)((wn * TN) * 256 + lane * 4) * 4u;
    const FragBuf fg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wf), 0, NQG * NB * 1024, 0x00020000);   // + (Q * NB + tn) * 1024 + lob
    const FragBuf fc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wcf), 0, 12 * NB * 1024, 0x00020000);
    f32x16 acc[TN];
    // B fragments: steps Q >= WS_PF of the running stage rotate through bq; the first WS_PF steps of the NEXT stage wait in
    // hq (requested while the running stage consumes its own heads), so a stage starts without a round trip to L2
    f32x4 bq[WS_PF][TN], hq[WS_PF][TN];
    auto zero_acc = [&]() {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
    };
    auto load_frag = [&](f32x4 (&dst)[TN], const FragBuf f, int Q) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        dst[tn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(f, (int)lob, (Q * NB + tn) * 1024, WS_B_AUX));
    };
    auto mfma8 = [&](const f32x4 af, const f32x4 (&b)[TN]) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[tn][r], af[r], acc[tn], 0, 0, 0);      // D[channel][pixel]
    };
    // one 8-wide K step Q of a stage of N steps whose fragments are `fthis`; `fnext`: the stage that follows
    auto kstep = [&](int Q, int N, const f32x4 af, const FragBuf fthis, const FragBuf fnext) {
      if (Q < WS_PF) {
        mfma8(af, hq[Q]);
        load_frag(hq[Q], fnext, Q);
      } else {
        mfma8(af, bq[Q % WS_PF]);
      }
      if (Q + WS_PF < N) load_frag(bq[Q % WS_PF], fthis, Q + WS_PF);
    };
    auto heads = [&](const FragBuf f) {
#pragma unroll
      for (int s = 0; s < WS_PF; ++s) load_frag(hq[s], f, s);
    };
    // accumulators -> tile.  The MFMA's first operand is the gamma / kernel fragment, so D is [channel][pixel]: a lane holds
    // ONE pixel (lane & 31) and, in registers 4j .. 4j+3, the four consecutive channels 8j + 4 half + (0..3) of its block --
    // 16-byte stores (the products and their order per output element are those of the [pixel][channel] form: bit-identical)
    auto acc_to_tile = [&](float* T) {
      float* const row = T + arow * TP + wn * TN * 32 + 4 * half;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(row + tn * 32 + 8 * j) = f32x4{acc[tn][4 * j], acc[tn][4 * j + 1], acc[tn][4 * j + 2], acc[tn][4 * j + 3]};
    };
    // g = 5x5/2 convolution of the zero-bordered 3-channel gradient image, rows of tile `t`; hq: the kernel's heads on
    // entry, gamma's on exit.  Its pixel operand is gathered from the gradient image (L1 / L2 hits: neighbouring pixels'
    // windows overlap); the first WS_AQ steps' fragments are requested a whole stage ahead (preload_a) -- at the start of a
    // stage nothing else hides the round trip (measured: 12 700 cycles per stage for 7 680 of MFMA without it)
    f32x4 aq[WS_AQ];
    auto a_src = [&](int t) -> const float* {
      const long long m = (long long)t * WS_BM + arow;
      int off = 0;
      if (m < a.M) {
        const int j = (int)(m % a.Wg);
        const long long tt = m / a.Wg;
        const int i = (int)(tt % a.Hg), b = (int)(tt / a.Hg);
        off = ((b * a.Hp + 2 * i) * a.Wp + 2 * j) * 3;
      }
      return a.pad + (size_t)off + koff;                         // + ky * Wp * 3 + (Q & 1) * 8
    };
    auto load_a = [&](const float* src, int Q) -> f32x4 {
      const float* p = src + (size_t)(Q >> 1) * a.Wp * 3 + (Q & 1) * 8;
      const f32x2 lo = wld2(p), hi = wld2(p + 2);
      return f32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto preload_a = [&](int t) {
      const float* const src = a_src(t);
#pragma unroll
      for (int Q = 0; Q < WS_AQ; ++Q) aq[Q] = load_a(src, Q);
    };
    auto conv3 = [&](int t) {
      const float* const src = a_src(t);
      zero_acc();
#pragma unroll
      for (int Q = 0; Q < NQC; ++Q) {
        const f32x4 af = aq[Q % WS_AQ];
        if (Q + WS_AQ < NQC) aq[Q % WS_AQ] = load_a(src, Q + WS_AQ);
        kstep(Q, NQC, af, fc, fg);
      }
    };
    // n = A . gamma with A resident in T; hq: gamma's heads on entry, the 3-channel kernel's on exit
    auto contract = [&](const float* T) {
      const float* const ap = T + arow * TP + koff;
      zero_acc();
      f32x4 an = wld4(ap);
#pragma unroll
      for (int Q = 0; Q < NQG; ++Q) {
        const f32x4 af = an;
        if (Q + 1 < NQG) an = wld4(ap + (Q + 1) * 8);
        kstep(Q, NQG, af, fg, fc);
      }
    };

    // prologue: g(0) -> X[0], g(1) -> X[1]
    const int t0 = ring[0], t1 = ring[1], t2 = ring[2];
    heads(fc);
    if (t0 >= 0) { preload_a(t0); conv3(t0); }
    if (t1 >= 0) preload_a(t1);
    if (t0 >= 0) acc_to_tile(X);
    ws_barrier();                                        // P1: g(0) visible (the memory waves start fill(0))
    if (t1 >= 0) { heads(fc); conv3(t1); }
    if (t2 >= 0) preload_a(t2);
    if (t1 >= 0) acc_to_tile(X + WS_BM * TP);
    heads(fc);
    ws_barrier();                                        // P2: A(0), g(1) visible
    for (int p = 0;; ++p) {                              // hq: the 3-channel kernel's heads
      const int tcur = ring[p % WS_QLEN], tprev = p > 0 ? ring[(p - 1) % WS_QLEN] : -1;
      if (p == 0 ? tcur < 0 : tprev < 0) break;
      const int tn2 = ring[(p + 2) % WS_QLEN];
      float* const Xa = X + (p % 3) * (WS_BM * TP);
      float* const Xc = X + ((p + 2) % 3) * (WS_BM * TP);
      WS_STAMP(p, 0);
      if (tn2 >= 0) conv3(tn2);
      else heads(fg);
      { const int tn3 = ring[(p + 3) % WS_QLEN]; if (tn3 >= 0) preload_a(tn3); }      // the next phase's gradient tile
      WS_STAMP(p, 1);
      ws_barrier();                                      // Ba: epilogue(p-1) has left Xc
      WS_STAMP(p, 2);
      if (tn2 >= 0) acc_to_tile(Xc);
      WS_STAMP(p, 3);
      if (tcur >= 0) contract(Xa);
      else heads(fc);
      WS_STAMP(p, 4);
      ws_barrier();                                      // Bb: every matrix wave is done with A(p); A(p+1) is complete
      WS_STAMP(p, 5);
      if (tcur >= 0) acc_to_tile(Xa);
      WS_STAMP(p, 6);
      ws_barrier();                                      // Bc: n(p) and g(p+2) visible
      WS_STAMP(p, 7);
    }
  } else {
    // =============================== memory waves ===============================================
    const int mt = tid - WS_MW * 64;
    const int pr = mt / CW, pc = mt - pr * CW;
    const float* const vsrc = a.v ? a.v : a.u;
    // slot = tile & 1: s and v as requested, then (fill, in place) g s and u, until the tile's epilogue
    f32x4 e1[2][KR * KC], e2[2][KR * KC];
    auto request = [&](int t, auto slot_c, int kr0 = 0, int kr1 = KR) {     // v, s of tile t -> registers
      constexpr int SL = decltype(slot_c)::value;
      const long long m0 = (long long)t * WS_BM;
      const long long left = a.M - m0;
      const int rvalid = left >= WS_BM ? WS_BM : (int)left;
      const size_t tb = (size_t)m0 * C + (size_t)pc * 4;
#pragma unroll
      for (int kr = 0; kr < KR; ++kr) {
        if (kr < kr0 || kr >= kr1) continue;
        const int row = pr + R * kr;
        const size_t ro = tb + (size_t)(row < rvalid ? row : 0) * C;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#if SGA_IGDN_NT & 1      // read once, dead afterwards: leave the caches to the operands that are re-read
          e2[SL][kr * KC + kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vsrc + ro + kc * CW * 4));
          e1[SL][kr * KC + kc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.s + ro + kc * CW * 4));
#else
          e2[SL][kr * KC + kc] = wld4(vsrc + ro + kc * CW * 4);
          e1[SL][kr * KC + kc] = wld4(a.s + ro + kc * CW * 4);
#endif
        }
      }
    };
    auto fill = [&](float* T, auto slot_c, int kr0 = 0, int kr1 = KR) {      // g (in T) -> g u / s (in T); e1 = g s, e2 = u
      constexpr int SL = decltype(slot_c)::value;
      float* const tp = T + pr * TP + pc * 4;
#pragma unroll
      for (int kr = 0; kr < KR; ++kr)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          if (kr < kr0 || kr >= kr1) continue;
          const int k = kr * KC + kc;
          float* const q = tp + kr * R * TP + kc * CW * 4;
          const f32x4 g = *reinterpret_cast<const f32x4*>(q);
          const f32x4 sv = e1[SL][k];
          f32x4 uv = e2[SL][k], av;
          if (a.v) {
            // the forward pass stored v = u s and s only: u = v / s with v_rcp_f32 (gdn_fused.hip)
            f32x4 ri;
#pragma unroll
            for (int x = 0; x < 4; ++x) ri[x] = __builtin_amdgcn_rcpf(sv[x]);
            uv = uv * ri;
            av = g * uv * ri;
          } else {
            av = g * uv / sv;
          }
          e1[SL][k] = g * sv;
          e2[SL][k] = uv;
          *reinterpret_cast<f32x4*>(q) = av;
        }
    };
    auto epilogue = [&](const float* T, auto slot_c, int t) {
      constexpr int SL = decltype(slot_c)::value;
      const long long m0 = (long long)t * WS_BM;
      const long long left = a.M - m0;
      const int rvalid = left >= WS_BM ? WS_BM : (int)left;
      const size_t tb = (size_t)m0 * C + (size_t)pc * 4;
      const float* const tp = T + pr * TP + pc * 4;
#pragma unroll
      for (int kr = 0; kr < KR; ++kr) {
        const int row = pr + R * kr;
        if (row < rvalid) {
#pragma unroll
          for (int kc = 0; kc < KC; ++kc) {
            const int k = kr * KC + kc;
            const f32x4 n = *reinterpret_cast<const f32x4*>(tp + kr * R * TP + kc * CW * 4);
#if SGA_IGDN_NT & 2
            __builtin_nontemporal_store(e1[SL][k] + e2[SL][k] * n, reinterpret_cast<f32x4*>(a.out + tb + (size_t)row * C + kc * CW * 4));
#else
            *reinterpret_cast<f32x4*>(a.out + tb + (size_t)row * C + kc * CW * 4) = e1[SL][k] + e2[SL][k] * n;
#endif
          }
        }
      }
    };
    // phase p touches the registers of ONE slot, (p + 1) & 1: the epilogue of tile p - 1 frees it, the fill of tile p + 1
    // takes it -- so the loop is unrolled by two with the slot a compile-time constant
    auto phase = [&](int p, auto slot_c) -> bool {
      const int tcur = ring[p % WS_QLEN], tprev = p > 0 ? ring[(p - 1) % WS_QLEN] : -1;
      if (p == 0 ? tcur < 0 : tprev < 0) return false;
      const int tn1 = ring[(p + 1) % WS_QLEN];
      WS_STAMP(p, 8);
      if (tprev >= 0) epilogue(X + ((p + 2) % 3) * (WS_BM * TP), slot_c, tprev);
      asm volatile("" ::: "memory");                     // (the requests stay below the epilogue: they reuse its registers)
      WS_STAMP(p, 9);
      WS_STAMP(p, 10);
      ws_barrier();                                      // Ba
      WS_STAMP(p, 11);
#if WS_THROTTLE
      if (tn1 >= 0) {
#pragma unroll
        for (int kr = 0; kr < KR; ++kr) {
          request(tn1, slot_c, kr, kr + 1);
          fill(X + ((p + 1) % 3) * (WS_BM * TP), slot_c, kr, kr + 1);
          asm volatile("" ::: "memory");
        }
      }
#else
      if (tn1 >= 0) { request(tn1, slot_c); fill(X + ((p + 1) % 3) * (WS_BM * TP), slot_c); }
#endif
      if (mt == 0) ring[(p + 4) % WS_QLEN] = next_tile(p + 4);
      WS_STAMP(p, 12);
      ws_barrier();                                      // Bb
      WS_STAMP(p, 13);
      ws_barrier();                                      // Bc
      WS_STAMP(p, 14);
      return true;
    };

    const int t0 = ring[0];
    if (t0 >= 0) request(t0, std::integral_constant<int, 0>{});
    ws_barrier();                                        // P1
    if (t0 >= 0) fill(X, std::integral_constant<int, 0>{});
    if (mt == 0) ring[3] = next_tile(3);
    ws_barrier();                                        // P2
    for (int p = 0;; p += 2) {
      if (!phase(p, std::integral_constant<int, 1>{})) break;
      if (!phase(p + 1, std::integral_constant<int, 0>{})) break;
    }
  }
  // the shared tile counter is left at zero for the next launch: the last workgroup to leave resets it
  if (a.sched) {
    __syncthreads();
    if (tid == 0) {
      const unsigned done = atomicAdd(a.sched + 1, 1u);
      if (done == gridDim.x - 1) { a.sched[0] = 0u; a.sched[1] = 0u; __threadfence(); }
    }
  }
}

template <int NC>
int launch_ws(const GdnArgs& a, hipStream_t stream) {
  constexpr int C = NC * 32;
  const size_t lds = (size_t)(3 * WS_BM * (C + 4)) * sizeof(float) + WS_QLEN * sizeof(int);
  static std::atomic<unsigned long long> attr_devs{0};
  static std::atomic<int> cus[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_devs.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igdn_bwd_ws_kernel<NC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus[dev & 63].store(n > 0 ? n : 256, std::memory_order_relaxed);
    attr_devs.fetch_or(bit, std::memory_order_release);
  }
  const long long ntiles = (a.M + WS_BM - 1) / WS_BM;
  long long grid = cus[dev & 63].load(std::memory_order_relaxed);
  if (grid > ntiles) grid = ntiles;
  if (grid <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((igdn_bwd_ws_kernel<NC>), dim3((unsigned)grid), dim3(WS_NT), lds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

// C / 32 in {2, 4, 6}; backward with the gradient-convolution prologue; needs the fragment-packed weights
bool igdn_bwd_ws_supported(const GdnArgs& a) {
  const int nc = a.C / 32;
  return a.mode == GDN_IGDN_BWD && a.pro == GDN_PRO_CONV3 && a.wf && a.wcf && a.C % 32 == 0 && (nc == 2 || nc == 4 || nc == 6);
}

int launch_igdn_bwd_ws(const GdnArgs& a, hipStream_t s) {
  if (!igdn_bwd_ws_supported(a)) return (int)hipErrorInvalidValue;
  switch (a.C / 32) {
    case 2: return launch_ws<2>(a, s);
    case 4: return launch_ws<4>(a, s);
    case 6: return launch_ws<6>(a, s);
  }
  return (int)hipErrorInvalidValue;
}
