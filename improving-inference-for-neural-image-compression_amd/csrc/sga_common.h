// Shared device/host definitions of libsga_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Laboratory switches -- ablations that change summation orders, skip work or move launches around (every measured alternative
// of DESIGN_EXPERIMENTS.md) -- exist only in the `make EXPERIMENTS=1` build (libsga_hip_lab.so: tests/test_gpu_fused.py compares
// the shipped kernels with the launches they replace through it).  The product library reads the handful of documented knobs
// of INTEGRATION.md section 6 with plain getenv and contains neither the switches' names nor their template instances
// (tests/test_host.py checks the strings of the built library against that list).
#ifdef SGA_EXPERIMENTS
#define LAB_ENV(name) getenv(name)
#else
#define LAB_ENV(name) (static_cast<const char*>(nullptr))
#endif

// Knobs of the measurement build (make PROBE=1; not an EXPERIMENTS=1 build, so LAB_ENV would never see them -- ADVICE r4)
#ifdef SGA_CLOCK_PROBE
#define PROBE_ENV(name) getenv(name)
#else
#define PROBE_ENV(name) (static_cast<const char*>(nullptr))
#endif

// Per-step scalars live in DEVICE memory so that one captured hipGraph of the step
// sequence can be replayed for every SGA iteration (sga.py:210-215): a 1-thread kernel
// advances `it` and refreshes T / lr_t from tables at the head of each replay.
struct StepCtx {
  int it;             // SGA iteration index (sga.py:210)
  int its;            // total iterations of this run
  float T;            // temperature, utils.py:166-180
  float lr_t;         // bias-corrected Adam step size, adam.py:40-42
  float lambda;       // sga.py:161
  float loss_scale;   // 1/B_ref: the batch means of sga.py:147,150
  unsigned seed_lo, seed_hi;
};

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  Bit-identical restatement in oracle/philox.py.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ inline float bits_to_uniform(uint32_t b) {
  // ((b >> 9) + 0.5) * 2^-23: 23 random bits so that +0.5 is exactly representable in f32
  // (a 24-bit mantissa): u in [2^-24, 1 - 2^-24], never 0 or 1 (log(-log u) stays finite)
  return ((float)(b >> 9) + 0.5f) * 1.1920928955078125e-07f;
}

// ---------------------------------------------------------------------------------------
// Convolution launcher (conv_mfma.hip)
// ---------------------------------------------------------------------------------------
enum ConvPrologue { PRO_NONE = 0, PRO_SQUARE = 1, PRO_IGDN_BWD = 2 };
enum ConvEpilogue {
  EPI_BIAS = 0,       // out = acc + bias (bias may be null)
  EPI_BIAS_RELU = 1,  // out = max(acc + bias, 0)
  EPI_IGDN = 2,       // n = acc + beta; s = sqrt(n); aux_out = s; out = aux0(u) * s
  EPI_GDN = 3,        // out = aux0(u) / sqrt(acc + beta)
  EPI_IGDN_BWD = 4,   // out = in(g) * aux1(s) + aux2(u) * acc
  EPI_RELU_MASK = 5,  // out = aux0 > 0 ? acc : 0
  EPI_SHUFFLE3 = 6    // 12 columns = 4 sub-pixel phases x 3 channels of a C->3 deconv
};

struct ConvTap { int dy, dx, slab; };
struct ConvPhase { int py, px, tap_begin, ntaps; };

// One gather-GEMM convolution:  out[pix(m), n] = epi( sum_{tap, ci} pro(in[pix_in(m,tap), ci]) * w[slab(tap)][n][ci] )
// over a pixel grid m = (b, i, j), i < Hg, j < Wg, per sub-pixel phase:
//   input  pixel (s_in*i + dy, s_in*j + dx)   (zero outside [0,Hin) x [0,Win))
//   output pixel (s_out*i + py, s_out*j + px)
struct ConvArgs {
  const float* in;  const float* w;  const float* bias;  float* out;
  // precision mode "bf16x3": weights pre-split into three bf16 planes, [slab][Npad][Cin/32][3][32]
  const unsigned short* w3;  int x3;
  const float* aux0; const float* aux1; const float* aux2; float* aux_out;
  int in_cs, in_coff;      // channel stride / offset of `in` (and aux1/aux2 in PRO_IGDN_BWD)
  int out_cs, out_coff;    // channel stride / offset of `out` (and aux0/aux_out)
  int B, Hg, Wg;
  int Hin, Win, Cin;       // Cin % 32 == 0
  int Hout, Wout, Cout, Npad;
  int s_in, s_out;
  int nphase, tiles_per_phase, ntiles_n;
  int smallc;              // 1: `in` is a zero-padded 3-channel image [B,Hin,Win,3], 5x5/2 conv
  int pro, epi;
  double flops;            // algorithmic flops of this launch (profiling only)
  const char* tag;         // call-site label, e.g. "gs2.fwd" (profiling only)
  // split-K: grid = tiles x ksplit; split s multiplies K-steps [s*n/S, (s+1)*n/S) of its phase and
  // writes the raw accumulators to part + s*slab (output geometry); a fixed-order reduce kernel
  // then sums the slabs and applies the epilogue (deterministic, no atomics).
  // fused IGDN post-phase (256-row unsplit C = 192 launches): out = u, post_s = sqrt(n), post_v = u * sqrt(n)
  int post; const float* post_w; const float* post_beta; float* post_s; float* post_v;
  const unsigned short* post_wx3;      // gamma pre-split into three bf16 planes (X3 instances), layout as `w3`
  const float* zeros;      // >= 256 bytes of zeros: what taps outside the image load (LDS-DMA instance)
#ifdef SGA_CLOCK_PROBE     // measurement build only (make PROBE=1): keeps the production kernels and their arguments unchanged
  unsigned long long* clk; // measurement only (SGA_CLOCK_PROBE=1, else null): per workgroup, shader-clock and 100 MHz
                           //   wall-clock ticks spent in the K loop -> the clock the chip sustains under this kernel
#endif
  unsigned long long* stamp;   // measurement only (sga_profile_graph_begin), else null: [0] = min over workgroups of the 100 MHz
                           //   wall clock at entry, [1] = max at exit
  int reduce_batch;        // split-K reduce: issue the slab loads 8 at a time (main chain) or one by one
  int prio;                // wave priority (s_setprio) for the whole launch: experiment, SGA_MAIN_WAVE_PRIO / SGA_SIDE_WAVE_PRIO
  int side;                // the launch belongs to the hyper branch (its split-K reduce: SGA_SIDE_ELEM_PRIO)
  int xcd_remap;           // unsplit launch, tiles_per_phase % 8 == 0: XCD x (blocks b % 8 == x) walks a contiguous eighth of every
                           //   phase's M tiles, so that the taps' re-gathers of one input region meet in ONE 4 MiB L2
  int pair_phases;         // 4-phase launch whose whole grid is resident at once: walk the phases as 9,6,4,6 taps
  int bm;                  // rows per tile: 128 (default, 4 waves) or 256 (8 waves, big unsplit layers)
  int ksplit;              // max over phases of nsplit[] (1 = no split)
  int nsplit[4];           // per-phase split factor: phases differ in tap count (9/6/6/4), so each
  int blk_begin[4];        //   gets splits in proportion and all workgroups walk ~equal K
  long long slab;
  float* part;
  ConvPhase ph[4];
  ConvTap taps[28];
};

// returns hipError_t
int launch_conv(const ConvArgs& a, hipStream_t stream);
// out[i] = epi(sum_s part[s*slab + i]) over n floats, channel = i % cout (n, cout multiples of 4)
int launch_splitk_reduce(const ConvArgs& a, long long n, hipStream_t stream);
// tile sizes chosen for an output width (host-side, also used to size Npad when packing)
int conv_pick_bn(int cout, int epi);
// kernel symbol (as rocprofv3 prints it) that launch_conv will use for these args
void conv_kernel_name(const ConvArgs& a, char* out, int len);

// ---------------------------------------------------------------------------------------
// GDN / IGDN tile kernel (gdn_fused.hip): prologue (slab sum | 3-channel conv) + resident-operand
// C x C contraction + epilogue in one launch
// ---------------------------------------------------------------------------------------
// Cache policy of the big streams (compile time: behind a run-time flag the compiler merges the two forms into plain accesses).
// Non-temporal = the line is not kept in L1 / L2 for this access: right for data that is read once by a launch, or written and not
// read again for a long time.  Bits: 1 igdn*.bwd read v (u), s | 2 igdn*.bwd write g_u | 4 IGDN post-phase of a convolution writes s
// | 8 ... writes v | 16 the C -> 3 GEMM reads its input | 32 gdn_tile_kernel forward writes s | 64 ... writes v
// Measured in the cfg-2 iteration (scripts/r05/s09_nt_masks.sh, profiles/r05_nt_masks.txt): 15 = bits 1|2|4|8 -5 us (1784 -> 1778),
// the others within the noise; 15 ships.
#ifndef SGA_NT
#define SGA_NT 15
#endif
#define SGA_IGDN_NT (SGA_NT & 3)
// Wave priority of the hyper branch's small kernels (k_factorized, k_gaussian, the split-K reduces): they run beside the main chain's
// MFMA kernels, whose older waves win the issue arbitration -- k_gaussian takes 97 us in the graph against 9 alone.  0: off
#ifndef SGA_SIDE_ELEM_PRIO
#define SGA_SIDE_ELEM_PRIO 0
#endif
enum GdnMode { GDN_IGDN_FWD = 0, GDN_GDN_FWD = 1, GDN_IGDN_BWD = 2 };
enum GdnPrologue { GDN_PRO_LOAD = 0, GDN_PRO_CONV3 = 1 };
struct GdnArgs {
  int C, mode, pro;
  long long M;             // pixels (rows); row m is the contiguous block [m*C, (m+1)*C) of every tensor
  // GDN_PRO_LOAD: T = sum_{s < S} src[s*slab + .] (+ bias);  S = nsplit[0], or per sub-pixel phase of a
  // stride-2 transposed conv (s_out == 2): nsplit[(oy&1)*2 + (ox&1)] with (oy, ox) from (hout, wout)
  const float* src; long long slab; int nsplit[4]; int s_out, hout, wout;
  const float* bias;
  // GDN_PRO_CONV3: T = 5x5/2 conv of the zero-bordered image pad [B,Hp,Wp,3] over the Hg x Wg grid,
  // weights wc [3][C][32] (pack_smallc)
  const float* pad; const float* wc; int Hg, Wg, Hp, Wp;
  const float* w;          // gamma, [C][C]: row = output channel, K contiguous (pack_gdn)
  const unsigned short* wx3; int x3;   // precision mode bf16x3: gamma pre-split into three bf16 planes (layout of ConvArgs::w3)
  const float* beta;       // forward
  const float* u; const float* s;     // backward: pre-IGDN activation and sqrt(n) of the forward pass
  const float* v;                     // backward, when the forward pass did not store u: v = u*s (then u = v/s; `u` unused)
  float* out;              // forward: v;  backward: g_u
  float* s_out_p;          // forward IGDN: sqrt(n) (or null)
  float* u_out;            // forward: T written back (needed when T was assembled here), or null
  double flops;            // algorithmic flops (profiling only)
  int prio;                // wave priority (experiment)
#ifdef SGA_CLOCK_PROBE
  unsigned long long* clk; // measurement build: per workgroup 8 x u64 = wall clock (100 MHz) at entry, after the prologue,
                           //   after the fill, after the contraction, at exit, hw_id | xcc_id << 32
#endif
};
int launch_gdn_tile(const GdnArgs& a, hipStream_t stream);
int gdn_tile_rows(int C, long long M, int pro);
void gdn_kernel_name(const GdnArgs& a, char* out, int len);

// C -> 3 transposed 5x5/2 conv, halo-tiled (deconv3.hip); w packed [C/32][9 taps][16][32]
int launch_deconv3_halo(const float* in, const float* w, const float* bias, float* out, int B,
                        int Hi, int Wi, int C, int Ho, int Wo, hipStream_t stream);
