// C ABI of libsga_hip (include/sga_hip.h): handle, weight pre-packing, layer descriptors,
// the SGA step sequence and its hipGraph replay.  No torch types, no exceptions across the ABI.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <algorithm>
#include <vector>

#include "../../include/sga_hip.h"
#include "kernels.h"
#include "sga_common.h"

namespace {

constexpr int kMaxIts = 16384;
constexpr int BM_TILE = 128;

struct PackedConv {
  float* w = nullptr;   // device [nslab][Npad][Kc]
  unsigned short* w3 = nullptr;   // device bf16x3 planes [nslab][Npad][Kc/32][3][32] (precision mode bf16x3)
  int Kc = 0;           // contiguous K per slab row (C_in of the GEMM)
  int N = 0;            // real output channels
  int Npad = 0, bn = 0;
  int nslab = 0;
};

struct Buf {
  float* p = nullptr;
  size_t cap = 0;   // floats
};

struct Geom {
  int B = 0, H = 0, W = 0;
  int eh[5], ew[5];       // analysis pyramid: eh[0]=H ... eh[4]=yh   (ceil halves)
  int yh, yw, zh1, zw1, zh, zw;
  int hsh, hsw;           // hyper-synthesis output spatial = 4*zh, 4*zw
  int Hp, Wp;             // zero-bordered gradient image of the synthesis output
  int xHp, xWp;           // zero-bordered input image for the first analysis conv
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

Geom make_geom(int B, int H, int W) {
  Geom g;
  g.B = B; g.H = H; g.W = W;
  g.eh[0] = H; g.ew[0] = W;
  for (int k = 1; k <= 4; ++k) { g.eh[k] = cdiv(g.eh[k - 1], 2); g.ew[k] = cdiv(g.ew[k - 1], 2); }
  g.yh = g.eh[4]; g.yw = g.ew[4];
  g.zh1 = cdiv(g.yh, 2); g.zw1 = cdiv(g.yw, 2);
  g.zh = cdiv(g.zh1, 2); g.zw = cdiv(g.zw1, 2);
  g.hsh = 4 * g.zh; g.hsw = 4 * g.zw;
  g.Hp = 16 * g.yh + 4; g.Wp = 16 * g.yw + 4;
  g.xHp = 2 * g.eh[1] + 4; g.xWp = 2 * g.ew[1] + 4;
  return g;
}

}  // namespace

struct sga_handle {
  sga_config cfg;
  int C = 0, C15 = 0, C2 = 0, haN = 0;
  float scale_bound = 0.f;         // lower bound on the conditional's sigma (sga_config.scale_bound, sga_set_scale_bound); 0 = none
  int last_hip_error = 0;
  char last_msg[256] = {0};

  // ---- packed weights (device) ----
  PackedConv ga_f[4];            // analysis forward (ga_f[0] small-C)
  PackedConv ga_gdn[3];
  PackedConv gs_f[4];            // synthesis forward (gs_f[3] = combined-phase C->3)
  PackedConv gs_b[4];            // synthesis data-gradient (gs_b[3] small-C)
  PackedConv gs_gdn_f[3], gs_gdn_b[3];
  PackedConv ha_f[3];
  PackedConv hs_f[3], hs_b[3];
  float* ga_bias[4] = {nullptr}; float* ga_beta[3] = {nullptr};
  float* gs_bias[4] = {nullptr}; float* gs_beta[3] = {nullptr};
  float* ha_bias[3] = {nullptr}; float* hs_bias[3] = {nullptr};
  float* eb_packed = nullptr;
  float* zeros = nullptr;          // 256 zero bytes (ConvArgs::zeros)
  unsigned long long* clk_probe = nullptr;   // SGA_CLOCK_PROBE=1|2 (measurement): [slot][16384][2] ticks, see ConvArgs::clk
  int clk_mode = 0;                          // 1: per-layer profiling runs (host reads after every launch);
                                             // 2: launches captured into the step graph, read once at sga_destroy
  struct ClkSlot { char name[96]; int grid; };
  std::vector<ClkSlot> clk_slots;
  unsigned* ticket = nullptr;      // k_step_boundary's last-workgroup counter (zero between launches)
  bool fused_boundary = true;      // SGA_FUSED_BOUNDARY=0: Adam, relaxation and finalize as three launches
  float* gs3_halo_w = nullptr;   // C->3 layer packed for deconv3.hip: [C/32][9][16][32]
  bool gs3_generic = false;      // SGA_GS3_GENERIC=1: use the generic gather-GEMM for the C->3 layer
  float* gs3_w80 = nullptr;      // the same layer for deconv3_gemm.hip: [80][C], row (ky*5+kx)*3 + c
  Buf p3;                        // its product matrix P [B * 8yh * 8yw][80]
  bool gs3_gemm = false;         // SGA_GS3_GEMM=1: GEMM + col2im (deconv3_gemm.hip) instead of the halo-tiled kernel (deconv3.hip)
  std::vector<void*> owned;      // every hipMalloc'd block

  // ---- workspace ----
  Buf xin, xpad, gpad, xt;
  Buf y, z, my, vy, mz, vz, yt, dyt, zt, dzt;
  Buf hs0, hs1, ms, g_ms, g_hs1, g_hs0, g_zt_hs, g_zt_eb;
  Buf u[3], s[3], v[3];
  Buf gA, gB, g_yt_dist, g_yt_rate;
  Buf zml, mzml, vzml, g_zml, jac_lv, lrtab2;   // bits-back variant only
  Buf xq;                        // reconstruction rounded to 0..255 (eval)
  float* msA[5] = {nullptr}; float* msB[5] = {nullptr};   // MS-SSIM pyramid scratch
  double* ms_stats = nullptr; int* ms_counts = nullptr;
  Buf scratch;                   // scalars[4] + psnr[max_batch] + metrics[max_batch*7]
  Buf trace, Ttab, lrtab;
  Buf part, partB;               // split-K partial slabs (one per concurrently running branch)
  Buf* cur_part = nullptr;
  // hyper-prior branch runs on its own stream, forked/joined with events (also inside the graph)
  hipStream_t sB = nullptr;
  // The stream every step GRAPH is captured and launched on: the library's own, LOW priority (round 6).  Not for speed -- for the
  // HIP runtime's Graph::UpdateStreams, which reads past the end of a graph's internal stream list when ALL of them share the
  // LAUNCH stream's hardware queue (host SIGSEGV inside hipGraphLaunch: "defect (a)" of rounds 3-5, root cause and stand-alone
  // reproducer in DESIGN_EXPERIMENTS.md A.13 / scripts/r06/graph_stream_collision_repro.hip).  The internal streams are created with
  // NORMAL priority, and hardware queues are pooled per priority class: a launch stream of another class can never share one
  // with them, whatever streams the process has created and destroyed before.  The caller's stream is bridged in and out
  // with one event pair per call (LaunchStream below); null: graphs run on the caller's stream (laboratory: SGA_LAUNCH_STREAM=caller)
  hipStream_t sG = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;           // eager launches
  const char* dump_path = nullptr; unsigned long long* dump = nullptr; int dump_run = 0;   // SGA_DEBUG_DUMP
  bool dump_probe = false;         // SGA_DEBUG_PROBE=1: ordering probe kernels (k_mark / k_probe) with the dump
  int run_B = 0, run_H = 0, run_W = 0, run_its = -1, run_it = 0;   // sga_run_begin/steps state
  float run_lambda = 0.f, run_loss_scale = 1.f; uint64_t run_seed = 0;
  bool split256 = true;            // split-K also for a single-phase launch of exactly 256 tiles (SGA_SPLIT256=0: off)
  bool plan_tiles = true;          // tile height / extra split by estimated grid efficiency (SGA_PLAN_TILES=0: round-2 rules only)
  bool fused_mse = true;           // distortion sums + gradient image in gs3.fwd's epilogue (SGA_FUSED_MSE=0: separate k_mse)
  bool fused_post = true;          // IGDN as the post-phase of the producing convolution launch (SGA_FUSED_POST=0: off)
  bool fused_gdn = true;           // gdn_fused.hip instead of the stand-alone GDN launches (SGA_FUSED_GDN=0: off)
  bool keep_u = false;             // SGA_KEEP_U=1: the synthesis IGDNs also store their input u and the backward pass reads it
                                   //   (bit-equal to the stand-alone launches); default: s and v only, u = v / s in igdn*.bwd
  int bm64_max = 256;              // 64-row tiles when the 128-row grid has at most this many blocks (SGA_BM64_MAX; 0 = off)
  bool bm256 = true;               // 256-row 8-wave tile for big unsplit f32 launches (SGA_BM256=0: off)
  bool bm256_split = true;         // ... and, split in two, for single-phase launches of 128 such tiles (SGA_BM256_SPLIT=0: off)
  int fork_at = 0;                 // main-chain launch index at which the hyper branch is forked (SGA_FORK_AT)
  const char* fork_name = nullptr; // ... or the main-chain launch right before which it is forked (null: fork_at decides)
  const char* fork2_name = nullptr;// captured graph: the branch's backward half also waits for this main-chain launch (null: no split)
  hipEvent_t ev_fork2_cap = nullptr;
  bool fork_auto = true;           // the fork point is chosen per geometry by timing the candidates (off when SGA_FORK_AT is set)
  bool graph_tuned = false;        // the cached step graph was built with a timed fork point
  bool bb_graph_tuned = false;     // ... and the bits-back stage-1 graph
  bool x3_variants = true;         // bf16x3 mode: 64- / 256-row tiles and the IGDN post-phase as in f32 mode (SGA_X3_VARIANTS=0: 128-row only)
  bool x3_fork = true;             // bf16x3 mode with the hyper branch on the second stream (round 4: on again, never forked at the
                                   //   graph's root -- synth_branch tick(); SGA_X3_FORK=0: single-stream as in rounds 2-3)
  // ---- keyed cache of LIVE executable graphs (round 5): one captured iteration per (kind, geometry, relaxation, sigma bound,
  // stamped) with its timed fork point.  A change of geometry / relaxation / bound SELECTS an entry -- nothing is dropped,
  // re-captured or re-timed when a shape comes back (a ragged last batch, a service alternating two sizes) -- and entries
  // live until sga_destroy (the least recently used one is retired only when the cache holds kMaxGraphs).
  struct GraphKey {
    int kind;                      // 0: one SGA iteration; 1 / 2: one iteration of bits-back stage 1 / stage 2
    int B, H, W, relax;
    float scale_bound;             // crosses to k_gaussian by value, i.e. is baked into the captured launch
    int stamped;                   // the measurement graph of sga_profile_graph_begin (one kernel carries a stamp pointer)
    bool operator==(const GraphKey& o) const {
      return kind == o.kind && B == o.B && H == o.H && W == o.W && relax == o.relax && scale_bound == o.scale_bound && stamped == o.stamped;
    }
  };
  struct GraphEntry { GraphKey key; hipGraphExec_t exec; bool tuned; const char* fork_name; bool stamp_in_graph; unsigned long long used; };
  std::vector<GraphEntry> graphs;
  unsigned long long graph_clock = 0;
  long long n_captures = 0;        // stream captures + instantiations so far (sga_debug_counter)
  long long n_graph_evictions = 0;
  long long n_dropped = 0;         // executable graphs destroyed in mid-life so far (losing fork-point candidates, evictions, stamped graphs)
  int tuned_B = 0, tuned_H = 0, tuned_W = 0;   // geometry the last timed choice (tuned_name) was made for: graphs of that
  const char* tuned_name = nullptr;            // geometry built without timing (short runs, the stamped graph) reuse it
  int dbg_it = -1;                 // iteration being enqueued (SGA_DEBUG_DUMP)
  hipEvent_t ev_fork_cap = nullptr, ev_join_cap = nullptr;   // while `st` is being captured
  bool overlap = true;             // SGA_NO_OVERLAP=1 disables
  int main_wave_prio = 0, side_wave_prio = 0;   // experiment: s_setprio of the main chain's / the hyper branch's MFMA kernels
  bool fused_post64 = true;        // IGDN post-phase also in the 64-row convolution instance (C = 192); SGA_FUSED_POST64=0: off
  bool side_hybrid = false;        // SGA_HYBRID=1 (experiment): hybrid replay without a CU mask
  bool side_masked = false;        // sB was created with a CU mask (hipExtStreamCreateWithCUMask)
  int branch_only = 0;             // rd_forward_backward: 0 both branches (fork / join), 1 synthesis branch only, 2 hyper branch only
  hipGraphExec_t graph_main = nullptr;   // hybrid replay: the synthesis branch of one iteration (the hyper branch is launched eagerly on the masked stream)
  int gmain_B = 0, gmain_H = 0, gmain_W = 0;
  ImgSums* sums = nullptr;
  StepCtx* ctx = nullptr;
  int4* img_ids = nullptr;       // [max_batch] {position of the image in its reference batch (sga_set_image_ids), its batch's seed
                                 //   lo / hi, seed valid (sga_set_image_seeds)}
  std::vector<int4> img_keys;    // host copy
  std::vector<float> hT, hLr;    // host tables (kept alive across the async upload)

  Geom geom_zeroed;              // geometry for which xpad/gpad borders are known zero
  bool borders_valid = false;

  int fork_delay_us = 0;           // SGA_FORK_DELAY_US (experiment)
  bool in_hyper = false;           // the launches being enqueued belong to the hyper branch
  int side_target = 256;           // split-K target (workgroups per launch) of the hyper branch when it runs BESIDE the synthesis chain
                                   //   (SGA_SIDE_TARGET; 0: the main chain's 512).  Round 6: 256, was 384 -- with the main chain on the
                                   //   low-priority launch stream the branch's kernels win the dispatch arbitration, and one workgroup per CU
                                   //   disturbs igdn2.bwd / gs2.bwd / igdn1.bwd less: cfg 2 -7.6 us per iteration (5 of 5 A/B rounds), B = 1 +1.5 %,
                                   //   B = 32 +0.6 %, Kodak / Tecnick sizes within +-0.2 % (profiles/r06_side_target.txt)
  int side_target_alone = 384;     // ... and when the branch is ALL that runs (bits-back stage 2, bb_sga.py:239-261: 0.495 against 0.570 ms)
  bool hyper_alone = false;        // rd_forward_backward without the synthesis chain
  bool side_last = true;           // graph capture: create the hyper branch's nodes after the main chain's (SGA_SIDE_LAST=0: before)
  int relax = 0, sched = 0;        // sga_set_relaxation
  int use_graph = 1;

  // ---- per-kernel profiling (sga_profile_begin/end) ----
  struct ProfRec { hipEvent_t a, b; double flops; char name[64]; };
  bool profiling = false;
  bool no_splitk = false;          // SGA_NO_SPLITK=1
  bool x3 = false;                 // precision mode bf16x3 (sga_config.reserved[0] == 2 or SGA_PRECISION=bf16x3)
  bool x2 = false;                 // precision mode bf16x2 (implies x3: same instances and weight planes; the convolution K loops use two planes)
  bool profile_by_layer = false;   // SGA_PROFILE_BY_LAYER=1: aggregate by call site instead of symbol
  const char* cur_tag = "";
  std::vector<ProfRec> prof;
  // ---- one kernel symbol timed INSIDE the graph replay (sga_profile_graph_begin/end).  HIP cannot take the elapsed
  // time of events recorded by graph nodes (hipEventElapsedTime: invalid resource handle), so the first launch of
  // that symbol in the captured iteration gets a stamp pointer: its workgroups record the earliest entry and the
  // latest exit on the 100 MHz wall clock (ConvArgs::stamp); the graph is otherwise the production one.
  bool gprof = false, gprof_in_graph = false;
  char gprof_name[64] = {0};
  unsigned long long* gstamp = nullptr;      // device [2]
  double gprof_ms = 0.0, gprof_flops = 0.0, gprof_flops_launch = 0.0;
  long long gprof_n = 0;
};

extern int g_deconv3_prio;

#ifdef SGA_EXPERIMENTS
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <ucontext.h>
#include <unistd.h>
// debugging aid (SGA_DEBUG_SEGV=1): native frames of a crash inside the process.  Runs on its OWN stack (sigaltstack): a
// handler on the faulting stack cannot run when the fault IS the stack (round 6: the handler of rounds 4-5 printed nothing and
// the process died with rc 139 -- which is what a stack overflow looks like).  Prints the faulting address, the stack pointer,
// the depth of the call stack, its innermost and its outermost frames.
static void* g_segv_frames[1 << 20];
static int g_segv_fd = 2;      // SGA_DEBUG_SEGV=<path>: the report goes to that file (pytest captures fd 2: a report written there is lost with the process)
static void sga_segv_handler(int sig, siginfo_t* si, void* uc_) {
  const ucontext_t* uc = (const ucontext_t*)uc_;
  const greg_t* r = uc->uc_mcontext.gregs;
  char buf[1024];
  const int fd = g_segv_fd;
  const int n = backtrace(g_segv_frames, 1 << 20);
  int len = snprintf(buf, sizeof(buf),
                     "sga: fatal signal %d code %d, fault address %p, rip %p rsp %p rbp %p, %d frames\n"
                     " rax %llx rbx %llx rcx %llx rdx %llx rsi %llx rdi %llx\n r8 %llx r9 %llx r10 %llx r11 %llx r12 %llx r13 %llx r14 %llx r15 %llx\n",
                     sig, si->si_code, si->si_addr, (void*)r[REG_RIP], (void*)r[REG_RSP], (void*)r[REG_RBP], n,
                     (unsigned long long)r[REG_RAX], (unsigned long long)r[REG_RBX], (unsigned long long)r[REG_RCX], (unsigned long long)r[REG_RDX],
                     (unsigned long long)r[REG_RSI], (unsigned long long)r[REG_RDI], (unsigned long long)r[REG_R8], (unsigned long long)r[REG_R9],
                     (unsigned long long)r[REG_R10], (unsigned long long)r[REG_R11], (unsigned long long)r[REG_R12], (unsigned long long)r[REG_R13],
                     (unsigned long long)r[REG_R14], (unsigned long long)r[REG_R15]);
  (void)!write(fd, buf, (size_t)len);
  if ((void*)r[REG_RIP] != si->si_addr && r[REG_RIP] > 4096) {      // the faulting instruction's bytes (not when the fault IS the jump target)
    const unsigned char* ip = (const unsigned char*)r[REG_RIP];
    len = snprintf(buf, sizeof(buf), " code at rip:");
    for (int i = 0; i < 24; ++i) len += snprintf(buf + len, sizeof(buf) - (size_t)len, " %02x", ip[i]);
    len += snprintf(buf + len, sizeof(buf) - (size_t)len, "\n");
    (void)!write(fd, buf, (size_t)len);
  }
  const char m1[] = "sga: innermost frames:\n";
  (void)!write(fd, m1, sizeof(m1) - 1);
  backtrace_symbols_fd(g_segv_frames, n < 64 ? n : 64, fd);
  if (n > 128) {
    const char msg[] = "sga: ... outermost frames:\n";
    (void)!write(fd, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(g_segv_frames + n - 48, 48, fd);
  } else if (n > 64) {
    backtrace_symbols_fd(g_segv_frames + 64, n - 64, fd);
  }
  const char m2[] = "sga: /proc/self/maps:\n";
  (void)!write(fd, m2, sizeof(m2) - 1);
  const int mf = open("/proc/self/maps", O_RDONLY);
  if (mf >= 0) {
    for (;;) {
      const ssize_t k = read(mf, buf, sizeof(buf));
      if (k <= 0) break;
      (void)!write(fd, buf, (size_t)k);
    }
    close(mf);
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
static void sga_install_segv_handler(const char* where) {
  static char alt[1 << 18];
  if (where && where[0] == '/' && g_segv_fd == 2) {
    const int f = open(where, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (f >= 0) g_segv_fd = f;
  }
  stack_t ss;
  ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0;
  (void)sigaltstack(&ss, nullptr);
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = sga_segv_handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigemptyset(&sa.sa_mask);
  (void)sigaction(SIGSEGV, &sa, nullptr);
  (void)sigaction(SIGBUS, &sa, nullptr);
}
#endif

namespace {

#define HIPCHK(h, expr)                                                               \
  do {                                                                                \
    const int _e = (int)(expr);                                                       \
    if (_e != 0) {                                                                    \
      (h)->last_hip_error = _e;                                                       \
      snprintf((h)->last_msg, sizeof((h)->last_msg), "%s:%d: %s -> %s", __FILE__,     \
               __LINE__, #expr, hipGetErrorString((hipError_t)_e));                   \
      return SGA_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

#define SGACHK(expr)                 \
  do {                               \
    const int _s = (expr);           \
    if (_s != SGA_OK) return _s;     \
  } while (0)

// Estimated fraction of the MFMA peak a launch of `n` equal workgroups of `bm`-row tiles reaches: resident
// workgroups per CU (3 / 2 / 1 for 64 / 128 / 256 rows), the rate a CU sustains with that many (measured at
// cfg 2: 64-row 2/CU 0.79-0.80, 128-row 2/CU 0.80, 256-row 8-wave 0.86-0.90 with the LDS-DMA loop; one 4-wave workgroup
// alone 0.55-0.62),
// and the quantisation of n into rounds of 256 x resident.  Kodak, 3 images: gs2.bwd as 576 128-row tiles
// measured 0.51 (estimate 0.45), as 1152 64-row tiles the estimate is 0.63.
double grid_efficiency(int bm, long long n) {
  const int per_cu = bm == 64 ? 3 : (bm == 128 ? 2 : 1);
  static const double R64[4] = {0, 0.55, 0.80, 0.84}, R128[3] = {0, 0.62, 0.80}, R256[2] = {0, 0.88};
  long long c = (n + 255) / 256;
  if (c > per_cu) c = per_cu;
  if (c < 1) c = 1;
  const double r = bm == 64 ? R64[c] : (bm == 128 ? R128[c] : R256[c]);
  const long long slots = 256 * c;
  const long long rounds = (n + slots - 1) / slots;
  return r * (double)n / (double)(rounds * slots);
}

// Best split multiple S <= 8 of a grid of `blocks` tiles by the same estimate, each extra slab priced at 3.5 %
// (S = 1 unless a split wins by 5 %); returns the estimate.
double best_split(int bm, long long blocks, int* S_out) {
  int best = 1;
  double eb = grid_efficiency(bm, blocks);
  for (int S = 2; S <= 8; ++S) {
    const double e = grid_efficiency(bm, blocks * S) / (1.0 + 0.035 * (S - 1));
    if (e > eb * 1.05) { eb = e; best = S; }
  }
  if (S_out) *S_out = best;
  return eb;
}

// Split-K factor for a launch whose tile grid under-fills the chip (256 CUs): spread the K walk
// of each tile over S workgroups so that ~2 workgroups per CU are resident and the serial
// K-chain per workgroup is S times shorter (deterministic slab reduce afterwards).
int pick_ksplit(const sga_handle* h, ConvArgs& a) {
  for (int p = 0; p < 4; ++p) { a.nsplit[p] = 1; a.blk_begin[p] = 0; }
  if (h->no_splitk || a.smallc || a.pro != PRO_NONE) return 1;
  if (a.epi != EPI_BIAS && a.epi != EPI_BIAS_RELU && a.epi != EPI_RELU_MASK) return 1;
  if (a.out_coff != 0 || a.out_cs != a.Cout || (a.Cout & 3)) return 1;
  const int tiles = a.tiles_per_phase * a.ntiles_n;
  const int blocks = a.nphase * tiles;
  static const int main_target = LAB_ENV("SGA_MAIN_TARGET") ? atoi(LAB_ENV("SGA_MAIN_TARGET")) : 512;   // experiments
  int target = a.bm == 256 ? 256 : main_target;      // ~512 workgroups of (nearly) equal K length (tuned at cfg 2); 256-row: one per CU
  // Small grids (<= 64 tiles before the split: the 16^2 ... 32^2 stages of a single 256^2 image) gain nothing from a 512-way
  // grid -- their K walk is already 2-3 steps per workgroup and every extra slab is summed by the consumer -- so they aim at one
  // workgroup per CU (measured per layer at B = 1: gs0.* 19 -> 17 / 30 -> 26 us, igdn0.* 28 -> 24 / 27 -> 21; the larger layers
  // lose with that target and keep 512; profiles/r04_b1_split_targets_by_layer.txt)
  static const int small_tiles = LAB_ENV("SGA_SMALL_TILES") ? atoi(LAB_ENV("SGA_SMALL_TILES")) : 64;      // 0: rule off (experiments)
  if (a.bm != 256 && blocks <= small_tiles) target = 256;
  if (h->in_hyper && h->side_target > 0) target = h->hyper_alone ? h->side_target_alone : h->side_target;   // hyper branch (whichever stream it runs on: the
                                                                     // split decides the summation order, i.e. result bits)
  const int bn = a.Npad / a.ntiles_n;
  const bool big = blocks > 256 || (blocks == 256 && a.nphase == 1 && !h->split256);
  // Where the cfg-2 rule does not reach -- grids of more than 256 blocks that still quantise badly into
  // rounds of resident workgroups (gs2.bwd of one Kodak image: 384 64-row tiles for 768 slots; of one
  // Tecnick image: 704 128-row tiles = 1.4 rounds of 512), and 128-row launches in general (176 tiles x 3 =
  // 528 workgroups is two rounds) -- the split is the multiple S of the grid with the best estimated
  // efficiency, each extra slab priced at 3.5 %.
  const bool search = h->plan_tiles && ((a.bm == 64 && big) || (a.bm == 128 && blocks > 128 && (bn == 192 || bn == 256)) ||
                                        (a.bm == 256 && a.nphase == 1 && blocks > 128 && blocks != 256 && h->bm256_split));
  if (search) {
    int best = 1;
    (void)best_split(a.bm, blocks, &best);
    if (best == 1) return 1;
    target = blocks * best;
  } else if (big) {
    return 1;
  }
  long long total_steps = 0;
  int steps[4] = {0, 0, 0, 0};
  for (int p = 0; p < a.nphase; ++p) {
    steps[p] = a.ph[p].ntaps * (a.Cin / 32);
    total_steps += (long long)steps[p] * tiles;
  }
  int T = (int)((total_steps + target - 1) / target);      // K-steps per workgroup
  if (T < 2) T = 2;
  const long long n_out = (long long)a.B * a.Hout * a.Wout * a.Cout;
  const long long cap = (long long)h->cur_part->cap / n_out;
  int smax = 1, begin = 0;
  for (int p = 0; p < a.nphase; ++p) {
    int S = (steps[p] + T / 2) / T;
    if (S > steps[p] / 2) S = steps[p] / 2;
    if (S > cap) S = (int)cap;
    if (S < 1) S = 1;
    a.nsplit[p] = S;
    a.blk_begin[p] = begin;
    begin += tiles * S;
    if (S > smax) smax = S;
  }
  return smax;
}

// A convolution whose split-K partial slabs are left un-reduced for the kernel that consumes its
// output (gdn_tile_kernel sums them in its prologue): what that kernel needs to know.
struct Deferred {
  bool active = false;       // false: the convolution wrote its complete output (bias included)
  const float* part = nullptr; long long slab = 0; int nsplit[4] = {1, 1, 1, 1};
  int s_out = 1, hout = 0, wout = 0;
  const float* bias = nullptr;
};

// An IGDN to run as the post-phase of the convolution that produces its input (conv_mfma.hip, POST = 1):
// possible when the launch is an unsplit 256-row-tile one with all C = 192 channels in a tile.
struct PostGdn {
  const float* gamma_w = nullptr; const float* beta = nullptr; float* s_out = nullptr; float* v_out = nullptr;
  const unsigned short* gamma_w3 = nullptr;      // the same gamma pre-split into bf16 planes (bf16x3 post-phase)
  bool drop_u = false;       // in: the fused launch need not write u (the backward pass uses v / s)
  bool fused = false;        // out: the convolution launch did the IGDN as well
};

// every MFMA convolution goes through here (so it can be timed)
int conv_launch(sga_handle* h, ConvArgs& a, hipStream_t st, Deferred* defer = nullptr, PostGdn* post = nullptr) {
  // Tile height.  128 rows (4 waves, 2 workgroups/CU) is the general instance; f32 launches with all 192
  // output channels in one tile can also run 256-row tiles (8 waves, 1/CU: each weight byte feeds twice the
  // MFMAs, and the IGDN can run as its post-phase) or 64-row tiles (3/CU: twice the tiles for grids that
  // under-fill the chip).  Rules tuned at cfg 2: >= 1024 128-row tiles -> 256 rows, <= 256 -> 64 rows.  In
  // between (Kodak / Tecnick shapes) the height whose grid quantises best into rounds of resident
  // workgroups wins (grid_efficiency): e.g. 576 128-row tiles = 1.1 rounds, as 1152 64-row tiles 1.5.
  a.bm = 128;
  // (bf16x3 mode: the same tile heights on the register-staged X3 instances, unless SGA_X3_VARIANTS=0)
  const bool variants = (!h->x3 || h->x3_variants) && !a.smallc && a.pro == PRO_NONE && a.Npad / a.ntiles_n == 192 &&
                        (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU || a.epi == EPI_RELU_MASK);
  if (variants) {
    const long long rows = (long long)a.B * a.Hg * a.Wg;
    const long long n128 = (long long)a.nphase * cdiv(rows, 128) * a.ntiles_n;
    const long long n64 = (long long)a.nphase * cdiv(rows, 64) * a.ntiles_n;
    const long long n256 = (long long)a.nphase * cdiv(rows, 256) * a.ntiles_n;
    int bm = 128;
    // A single-phase launch of 128 (or 256) 256-row tiles: split in two (or not at all) it is exactly one
    // 8-wave workgroup per CU, and that K loop runs at 0.90 of the MFMA peak against 0.84 for two 64-row
    // workgroups per CU (in-kernel probe, profiles/r02_clock_probe.txt: 5.70 us per 256 x 192 x 32 step) --
    // more than the second slab costs: gs2.bwd at cfg 2 467 -> 448 us, iteration 1834 -> 1823 us.
    // SGA_BM256_MIN (experiment): also single-phase launches of 32 / 64 such tiles, split 8 / 4 ways
    static const int bm256_min = LAB_ENV("SGA_BM256_MIN") ? atoi(LAB_ENV("SGA_BM256_MIN")) : 128;
    const bool one_per_cu = h->bm256_split && a.nphase == 1 &&
                            (n256 == 128 || n256 == 256 || (n256 >= bm256_min && (n256 == 32 || n256 == 64)));
    if (one_per_cu) bm = 256;
    else if (h->bm64_max > 0 && n128 <= h->bm64_max) bm = 64;
    else if (h->bm256 && n128 >= 1024 && !h->plan_tiles) bm = 256;
    else if (h->plan_tiles) {
      double e = best_split(128, n128, nullptr);
      if (h->bm256 && n128 >= 512) {
        // a 256-row launch that also does the IGDN (POST) saves the IGDN launch: ~17 % of the pair's time
        const bool with_post = post && h->fused_post && a.Cout == 192 && a.Npad == 192 && a.epi == EPI_BIAS &&
                               a.out_coff == 0 && a.out_cs == 192 && post->s_out && post->v_out;
        const double e256 = grid_efficiency(256, n256) * (with_post ? 1.17 : 1.0);
        if (e256 >= e) { e = e256; bm = 256; }
      }
      if (h->bm256_split && a.nphase == 1 && n256 > 128) {      // 256-row tiles split along K (no IGDN post-phase then)
        const double e256s = best_split(256, n256, nullptr);
        if (e256s > e) { e = e256s; bm = 256; }
      }
      if (h->bm64_max > 0) {
        const double e64 = n64 > 256 ? best_split(64, n64, nullptr) : grid_efficiency(64, n64);
        if (e64 > e * 1.03) { e = e64; bm = 64; }
      }
    }
    a.bm = bm;
    a.tiles_per_phase = (int)cdiv(rows, bm);
  }
  // C = 256 (BN = 256): 128-row tiles (register-staged loop) or 256-row tiles (8 waves, LDS-DMA loop)
  if (h->plan_tiles && h->bm256 && !h->x3 && !a.smallc && a.pro == PRO_NONE && a.Npad / a.ntiles_n == 256 &&
      (a.epi == EPI_BIAS || a.epi == EPI_BIAS_RELU || a.epi == EPI_RELU_MASK)) {
    const long long rows = (long long)a.B * a.Hg * a.Wg;
    const long long n128 = (long long)a.nphase * cdiv(rows, 128) * a.ntiles_n;
    const long long n256 = (long long)a.nphase * cdiv(rows, 256) * a.ntiles_n;
    const double bonus = 1.04;      // the LDS-DMA loop against the register-staged 128-row one
    // unsplit, the 256-row launch can also do the IGDN that follows (POST): saves that launch, measured at the
    // Tecnick shape (profiles/r03_configs.txt)
    const bool with_post = post && h->fused_post && a.Cout == 256 && a.Npad == 256 && a.epi == EPI_BIAS &&
                           a.out_coff == 0 && a.out_cs == 256 && post->s_out && post->v_out;
    double e256 = grid_efficiency(256, n256) * (with_post ? 1.12 : 1.0);
    if (h->bm256_split && a.nphase == 1) e256 = std::max(e256, best_split(256, n256, nullptr));
    if (n128 >= 512 && e256 * bonus >= best_split(128, n128, nullptr)) {
      a.bm = 256;
      a.tiles_per_phase = (int)cdiv(rows, 256);
    }
  }
  a.ksplit = pick_ksplit(h, a);
  a.zeros = h->zeros;
  a.prio = (h->cur_part == &h->partB) ? h->side_wave_prio : h->main_wave_prio;
  a.side = h->in_hyper ? 1 : 0;
#ifdef SGA_CLOCK_PROBE
  a.clk = (h->clk_mode == 1 && h->profiling && h->profile_by_layer) ? h->clk_probe : nullptr;
  if (h->clk_mode == 2 && h->clk_slots.size() < 40) {
    hipStreamCaptureStatus ccs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &ccs);
    if (ccs == hipStreamCaptureStatusActive) a.clk = h->clk_probe + h->clk_slots.size() * (size_t)(6 * 16384);
  }
#endif
  { static const int rb = LAB_ENV("SGA_REDUCE_BATCH") ? atoi(LAB_ENV("SGA_REDUCE_BATCH")) : 2;
    a.reduce_batch = rb == 2 ? 1 : (rb == 1 ? (h->cur_part == &h->part) : 0); }
  {
    const long long blocks = (long long)a.nphase * a.tiles_per_phase * a.ntiles_n;
    const int per_cu = a.bm == 64 ? 3 : (a.bm == 128 ? 2 : 1);
    a.pair_phases = (a.nphase == 4 && a.ksplit <= 1 && blocks > 256 && blocks <= 256LL * per_cu) ? 1 : 0;
    static const int xr = LAB_ENV("SGA_XCD_REMAP") ? atoi(LAB_ENV("SGA_XCD_REMAP")) : 1;
    a.xcd_remap = (xr && a.ksplit <= 1 && a.ntiles_n == 1 && a.tiles_per_phase % 8 == 0 && blocks >= 256) ? 1 : 0;
  }
  if (defer && a.epi != EPI_BIAS) return SGA_ERR_BAD_ARG;   // the consumer applies "+ bias" only
  if (post) {
    // 256-row tiles (8 waves, C = 192 / 256) and the 64-row 4-wave instance at C = 192 (gs1.fwd + igdn1.fwd 170 -> 156 us
    // alone at cfg 2; inside the iteration it pays only with the later fork point of the hyper branch: the joint sweep of
    // DESIGN_EXPERIMENTS.md A.7; SGA_FUSED_POST64=0 turns it off)
    const bool post_tile = a.bm == 256 || (a.bm == 64 && a.Cout == 192 && h->fused_post64);
    post->fused = h->fused_post && post_tile && a.ksplit <= 1 && (a.Cout == 192 || a.Cout == 256) && a.Npad == a.Cout &&
                  a.epi == EPI_BIAS && a.out_coff == 0 && a.out_cs == a.Cout && post->s_out && post->v_out;
    if (post->fused) {
      a.post = 1; a.post_w = post->gamma_w; a.post_wx3 = post->gamma_w3; a.post_beta = post->beta; a.post_s = post->s_out; a.post_v = post->v_out;
      if (post->drop_u) a.out = nullptr;
      a.flops += 2.0 * a.B * a.Hout * a.Wout * (double)a.Cout * a.Cout;
    }
  }
  // bf16x3 where it is faster: the IGDN-backward prologue (3 prefetched operands) and the 2-wave
  // BN=96 tile spill to scratch in that mode and measured slower than their f32 instances (193 vs
  // 177 us, 96 vs 88 us), so those launches stay on the f32 MFMA kernel (2.21 -> 2.30 img/s).
  a.x3 = (h->x3 && a.w3 && !a.smallc && a.epi != EPI_SHUFFLE3 && a.pro != PRO_IGDN_BWD &&
          a.Npad / a.ntiles_n != 96) ? 1 : 0;
  if (a.x3 && h->x2 && a.pro == PRO_NONE) a.x3 = 2;      // bf16x2: the two-plane K loop (PRO_NONE instances only)
  { // laboratory (A.8): bf16x3 arithmetic in only one of the two branches, to isolate the two-stream nondeterminism
    static const int only = LAB_ENV("SGA_X3_ONLY") ? atoi(LAB_ENV("SGA_X3_ONLY")) : 0;      // 1: main chain only, 2: hyper branch only
    if ((only == 1 && h->in_hyper) || (only == 2 && !h->in_hyper)) a.x3 = 0; }
  const long long n_out = (long long)a.B * a.Hout * a.Wout * a.Cout;
  if (a.ksplit > 1) { a.part = h->cur_part->p; a.slab = n_out; }
  bool gprof_here = false;
  if (h->gprof && !h->gprof_in_graph) {
    hipStreamCaptureStatus gcs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &gcs);
    char kn[64];
    conv_kernel_name(a, kn, sizeof(kn));
    gprof_here = gcs == hipStreamCaptureStatusActive && strncmp(kn, h->gprof_name, sizeof(kn)) == 0;
    if (gprof_here) {
      a.stamp = h->gstamp;
      h->gprof_flops_launch = a.flops;
      h->gprof_in_graph = true;
    }
  }
  sga_handle::ProfRec r;
  if (h->profiling) {
    r.flops = a.flops;
    if (h->profile_by_layer) {
      char kn[64];
      conv_kernel_name(a, kn, sizeof(kn));
      snprintf(r.name, sizeof(r.name), "%s %s k%d", h->cur_tag, kn + 16, a.ksplit);
    } else {
      conv_kernel_name(a, r.name, sizeof(r.name));
    }
    HIPCHK(h, hipEventCreate(&r.a));
    HIPCHK(h, hipEventCreate(&r.b));
    HIPCHK(h, hipEventRecord(r.a, st));
  }
  HIPCHK(h, launch_conv(a, st));
#ifdef SGA_CLOCK_PROBE
  if (a.clk && h->clk_mode == 2) {
    sga_handle::ClkSlot cs;
    int grid = 0;
    for (int p = 0; p < a.nphase; ++p) grid += a.tiles_per_phase * a.ntiles_n * (a.ksplit > 1 ? a.nsplit[p] : 1);
    cs.grid = grid > 16384 ? 16384 : grid;
    char kn[64];
    conv_kernel_name(a, kn, sizeof(kn));
    snprintf(cs.name, sizeof(cs.name), "%s %s k%d", h->cur_tag, kn + 16, a.ksplit);
    h->clk_slots.push_back(cs);
  } else if (a.clk) {      // measurement: shader clock sustained inside this launch's K loops (wall_clock64 ticks at 100 MHz)
    int grid = 0;
    for (int p = 0; p < a.nphase; ++p) grid += a.tiles_per_phase * a.ntiles_n * (a.ksplit > 1 ? a.nsplit[p] : 1);
    if (grid > 16384) grid = 16384;
    std::vector<unsigned long long> t(6 * (size_t)grid);
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipMemcpy(t.data(), h->clk_probe, t.size() * sizeof(t[0]), hipMemcpyDeviceToHost));
    double c = 0, w = 0;
    for (int i = 0; i < grid; ++i) { c += (double)t[6 * i]; w += (double)t[6 * i + 1]; }
    char kn[64];
    conv_kernel_name(a, kn, sizeof(kn));
    fprintf(stderr, "clock_probe %s %s k%d grid %d: %.0f MHz in the K loop\n", h->cur_tag, kn + 16, a.ksplit, grid,
            w > 0 ? 100.0 * c / w : 0.0);
  }
#endif
  if (h->profiling && !h->profile_by_layer) {   // symbol-level stats: the conv kernel alone
    HIPCHK(h, hipEventRecord(r.b, st));
    h->prof.push_back(r);
  }
  if (defer) {
    defer->active = a.ksplit > 1;
    if (defer->active) {
      defer->part = a.part; defer->slab = a.slab; defer->bias = a.bias;
      defer->s_out = a.s_out; defer->hout = a.Hout; defer->wout = a.Wout;
      for (int p = 0; p < 4; ++p) defer->nsplit[p] = a.nsplit[p] > 0 ? a.nsplit[p] : 1;
      if (a.s_out != 2) for (int p = 1; p < 4; ++p) defer->nsplit[p] = defer->nsplit[0];
    }
  }
  if (a.ksplit > 1 && !defer) HIPCHK(h, launch_splitk_reduce(a, n_out, st));
  if (h->profiling && h->profile_by_layer) {    // layer-level stats: conv + its reduce
    HIPCHK(h, hipEventRecord(r.b, st));
    h->prof.push_back(r);
  }
  return SGA_OK;
}

int dev_alloc(sga_handle* h, void** p, size_t bytes) {
  if (bytes == 0) bytes = 256;
  const hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    h->last_hip_error = (int)e;
    snprintf(h->last_msg, sizeof(h->last_msg), "hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
    return SGA_ERR_NOMEM;
  }
  h->owned.push_back(*p);
  return SGA_OK;
}

int alloc_buf(sga_handle* h, Buf& b, size_t floats) {
  b.cap = floats;
  void* p = nullptr;
  SGACHK(dev_alloc(h, &p, floats * sizeof(float) + 256));   // +256 B slack for vector over-reads
  b.p = (float*)p;
  return SGA_OK;
}

int upload(sga_handle* h, float** dst, const float* src, size_t n) {
  void* p = nullptr;
  SGACHK(dev_alloc(h, &p, n * sizeof(float)));
  HIPCHK(h, hipMemcpy(p, src, n * sizeof(float), hipMemcpyHostToDevice));
  *dst = (float*)p;
  return SGA_OK;
}

// round-to-nearest-even f32 -> bf16 (as v_cvt_pk_bf16_f32 does on the device)
inline unsigned short bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float bf16_to_f32(unsigned short b) {
  const unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int upload_packed(sga_handle* h, PackedConv& pc, const std::vector<float>& host, bool x3 = true) {
  SGACHK(upload(h, &pc.w, host.data(), host.size()));
  if (!x3 || pc.Kc % 32 != 0) return SGA_OK;
  // bf16x3: w = h + m + l exactly; layout [slab][n][Kc/32][plane][32] so that one K-step of one
  // output row is 192 contiguous bytes
  const size_t rows = host.size() / pc.Kc;
  const int nck = pc.Kc / 32;
  std::vector<unsigned short> planes(host.size() * 3);
  for (size_t r = 0; r < rows; ++r)
    for (int c = 0; c < nck; ++c)
      for (int k = 0; k < 32; ++k) {
        const float x = host[r * pc.Kc + c * 32 + k];
        const unsigned short hi = bf16_rne(x);
        const float r1 = x - bf16_to_f32(hi);
        const unsigned short mi = bf16_rne(r1);
        const unsigned short lo = bf16_rne(r1 - bf16_to_f32(mi));
        unsigned short* dst = &planes[((r * nck + c) * 3) * 32 + k];
        dst[0] = hi; dst[32] = mi; dst[64] = lo;
      }
  void* p = nullptr;
  SGACHK(dev_alloc(h, &p, planes.size() * sizeof(unsigned short)));
  HIPCHK(h, hipMemcpy(p, planes.data(), planes.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  pc.w3 = (unsigned short*)p;
  return SGA_OK;
}

// ------------------------------------------------------------------------------------------
// weight packing (host).  K is HWIO [kh][kw][ci][co]  (tfc.SignalConv2D)
// ------------------------------------------------------------------------------------------
// The hyper branch's N = 288 layers (hs1.fwd, hs2.bwd) as two 192-wide tiles (a quarter of the columns padding, but the
// 64-row LDS-DMA instance instead of the 2-wave register-staged BN = 96 one).  Round 3: on by default in f32 mode -- alone
// hs2.bwd 86 -> 62 us; inside the iteration only together with the other "faster alone" variants and a later fork point
// (joint sweep, DESIGN_EXPERIMENTS.md A.7).  SGA_BN96_AS_192=0 restores the BN = 96 instance.
bool bn96_as_192(const sga_handle* h) {
  const char* e = LAB_ENV("SGA_BN96_AS_192");
  return (!h->x3 || h->x3_variants) && !(e && e[0] == '0');
}

// GEMM with N = co, K = ci:  w[t][co][ci] = K[t][ci][co]   (forward of any conv)
int pack_fwd(sga_handle* h, PackedConv& pc, const float* K, int taps, int ci, int co, int epi) {
  pc.Kc = ci; pc.N = co; pc.nslab = taps;
  pc.bn = conv_pick_bn(co, epi);
  if (pc.bn == 96 && bn96_as_192(h)) pc.bn = 192;   // see pack_bwd
  pc.Npad = cdiv(co, pc.bn) * pc.bn;
  std::vector<float> w((size_t)taps * pc.Npad * ci, 0.f);
  for (int t = 0; t < taps; ++t)
    for (int i = 0; i < ci; ++i)
      for (int o = 0; o < co; ++o)
        w[((size_t)t * pc.Npad + o) * ci + i] = K[((size_t)t * ci + i) * co + o];
  return upload_packed(h, pc, w);
}

// GEMM with N = ci, K = co:  w[t][ci][co] = K[t][ci][co]   (data-gradient of any conv)
int pack_bwd(sga_handle* h, PackedConv& pc, const float* K, int taps, int ci, int co) {
  pc.Kc = co; pc.N = ci; pc.nslab = taps;
  pc.bn = conv_pick_bn(ci, EPI_BIAS);
  // experiment (SGA_BN96_AS_192=1): N = 288 (1.5 C at C = 192) as two 192-wide tiles (a quarter of them padding) on the
  // 64-row LDS-DMA instance instead of three 96-wide tiles on the 2-wave register-staged one
  if (pc.bn == 96 && bn96_as_192(h)) pc.bn = 192;
  pc.Npad = cdiv(ci, pc.bn) * pc.bn;
  std::vector<float> w((size_t)taps * pc.Npad * co, 0.f);
  for (int t = 0; t < taps; ++t)
    for (int i = 0; i < ci; ++i)
      memcpy(&w[((size_t)t * pc.Npad + i) * co], &K[((size_t)t * ci + i) * co], co * sizeof(float));
  return upload_packed(h, pc, w);
}

// GDN: n_i = beta_i + sum_j gamma[j][i] x_j^2  ->  forward rows i, K = j: w[i][j] = gamma[j][i]
//      backward acc_i = sum_k gamma[i][k] t_k   ->  rows i, K = k:          w[i][k] = gamma[i][k]
int pack_gdn(sga_handle* h, PackedConv& pc, const float* gamma, int C, bool backward) {
  pc.Kc = C; pc.N = C; pc.nslab = 1;
  pc.bn = conv_pick_bn(C, EPI_BIAS);
  pc.Npad = cdiv(C, pc.bn) * pc.bn;
  std::vector<float> w((size_t)pc.Npad * C, 0.f);
  for (int i = 0; i < C; ++i)
    for (int k = 0; k < C; ++k)
      w[(size_t)i * C + k] = backward ? gamma[(size_t)i * C + k] : gamma[(size_t)k * C + i];
  return upload_packed(h, pc, w);
}

// C->3 transposed 5x5/2 conv as ONE GEMM over the 3x3 input neighbourhood with
// N = 4 phases x 3 channels (padded to 32): tap (dy,dx) feeds phase (py,px) through kernel
// element ky = py + 2 - 2dy, kx = px + 2 - 2dx when that lies inside the 5x5 support.
int pack_shuffle3(sga_handle* h, PackedConv& pc, const float* K /*[5][5][C][3]*/, int C) {
  pc.Kc = C; pc.N = 12; pc.nslab = 9; pc.bn = 32; pc.Npad = 32;
  std::vector<float> w((size_t)9 * 32 * C, 0.f);
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int t = (dy + 1) * 3 + (dx + 1);
      for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
          const int ky = py + 2 - 2 * dy, kx = px + 2 - 2 * dx;
          if (ky < 0 || ky > 4 || kx < 0 || kx > 4) continue;
          for (int c = 0; c < 3; ++c) {
            const int n = (py * 2 + px) * 3 + c;
            for (int ci = 0; ci < C; ++ci)
              w[((size_t)t * 32 + n) * C + ci] = K[(((size_t)ky * 5 + kx) * C + ci) * 3 + c];
          }
        }
    }
  return upload_packed(h, pc, w, false);
}

// 5x5/2 conv over a zero-bordered 3-channel image: 3 K-steps of 32 = two kernel rows x 16
// floats (kx*3+ch for 15, +1 slack).  fwd: N = co, element K[ky][kx][ch][co] (K is [5][5][3][C]);
// bwd (gradient of the C->3 transposed conv): N = ci, element K[ky][kx][ci][ch] (K is [5][5][C][3]).
int pack_smallc(sga_handle* h, PackedConv& pc, const float* K, int C, bool bwd) {
  pc.Kc = 32; pc.N = C; pc.nslab = 3;
  pc.bn = conv_pick_bn(C, EPI_BIAS);
  pc.Npad = cdiv(C, pc.bn) * pc.bn;
  std::vector<float> w((size_t)3 * pc.Npad * 32, 0.f);
  for (int s = 0; s < 3; ++s)
    for (int half = 0; half < 2; ++half) {
      const int ky = 2 * s + half;
      if (ky > 4) continue;
      for (int kk = 0; kk < 15; ++kk) {
        const int kx = kk / 3, ch = kk % 3;
        for (int n = 0; n < C; ++n) {
          const float val = bwd ? K[(((size_t)ky * 5 + kx) * C + n) * 3 + ch]
                                : K[(((size_t)ky * 5 + kx) * 3 + ch) * C + n];
          w[((size_t)s * pc.Npad + n) * 32 + half * 16 + kk] = val;
        }
      }
    }
  return upload_packed(h, pc, w, false);
}

// ------------------------------------------------------------------------------------------
// tap tables
// ------------------------------------------------------------------------------------------
void taps_single(ConvArgs& a) {
  a.nphase = 1;
  a.ph[0] = ConvPhase{0, 0, 0, 1};
  a.taps[0] = ConvTap{0, 0, 0};
}

// stride-2 5x5 cross-correlation, pad 2: in(2i+ky-2, 2j+kx-2)
void taps_conv5_s2(ConvArgs& a) {
  a.nphase = 1;
  a.ph[0] = ConvPhase{0, 0, 0, 25};
  for (int ky = 0; ky < 5; ++ky)
    for (int kx = 0; kx < 5; ++kx) a.taps[ky * 5 + kx] = ConvTap{ky - 2, kx - 2, ky * 5 + kx};
}

// stride-2 transposed 5x5 conv: out(2i+py, 2j+px) = sum in(i+dy, j+dx) K[ky][kx],
// dy = (py + 2 - ky)/2 over ky = py (mod 2).  Heaviest phase first.
void taps_deconv5_s2(ConvArgs& a) {
  a.nphase = 4;
  int t = 0;
  for (int p = 0; p < 4; ++p) {
    const int py = p >> 1, px = p & 1;
    a.ph[p].py = py; a.ph[p].px = px; a.ph[p].tap_begin = t;
    for (int ky = py; ky < 5; ky += 2)
      for (int kx = px; kx < 5; kx += 2) a.taps[t++] = ConvTap{(py + 2 - ky) / 2, (px + 2 - kx) / 2, ky * 5 + kx};
    a.ph[p].ntaps = t - a.ph[p].tap_begin;
  }
}

// 3x3 stride 1: corr -> in(i+ky-1); true convolution (corr=False) -> in(i+1-ky)
void taps_conv3(ConvArgs& a, bool true_conv) {
  a.nphase = 1;
  a.ph[0] = ConvPhase{0, 0, 0, 9};
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx)
      a.taps[ky * 3 + kx] = true_conv ? ConvTap{1 - ky, 1 - kx, ky * 3 + kx} : ConvTap{ky - 1, kx - 1, ky * 3 + kx};
}

void taps_shuffle3(ConvArgs& a) {
  a.nphase = 1;
  a.ph[0] = ConvPhase{0, 0, 0, 9};
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int t = (dy + 1) * 3 + (dx + 1);
      a.taps[t] = ConvTap{dy, dx, t};
    }
}

void taps_smallc(ConvArgs& a) {
  a.nphase = 1;
  a.ph[0] = ConvPhase{0, 0, 0, 3};
  for (int s = 0; s < 3; ++s) a.taps[s] = ConvTap{2 * s, 0, s};
}

ConvArgs base_args(const PackedConv& pc, int B, int Hg, int Wg) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.w = pc.w;
  a.w3 = pc.w3;
  a.B = B; a.Hg = Hg; a.Wg = Wg;
  a.Cin = pc.Kc; a.Cout = pc.N; a.Npad = pc.Npad;
  a.ntiles_n = pc.Npad / pc.bn;
  a.tiles_per_phase = cdiv(B * Hg * Wg, BM_TILE);
  a.in_cs = pc.Kc; a.out_cs = pc.N;
  a.s_in = 1; a.s_out = 1;
  a.pro = PRO_NONE; a.epi = EPI_BIAS;
  return a;
}

// ---- layer launchers (all: in/out NHWC device) ---------------------------------------------
// transposed 5x5/2: [B,Hi,Wi,Cin] -> [B,2Hi,2Wi,Cout]
int deconv_fwd(sga_handle* h, const PackedConv& pc, const float* bias, const float* in, int B,
               int Hi, int Wi, float* out, int epi, hipStream_t st, Deferred* defer = nullptr,
               PostGdn* post = nullptr) {
  ConvArgs a = base_args(pc, B, Hi, Wi);
  a.in = in; a.out = out; a.bias = bias;
  a.Hin = Hi; a.Win = Wi; a.Hout = 2 * Hi; a.Wout = 2 * Wi;
  a.s_in = 1; a.s_out = 2; a.epi = epi;
  taps_deconv5_s2(a);
  a.flops = 2.0 * B * Hi * Wi * 25.0 * pc.Kc * pc.N;
  return conv_launch(h, a, st, defer, post);
}

// stride-2 5x5 conv over `in` [B,Hi,Wi,K] -> [B,Ho,Wo,N]; used for analysis forward and for the
// data-gradient of deconv_fwd (then in = g_out, Ho = Hi/2).  aux0: ReLU-mask activation (or null)
int conv5s2(sga_handle* h, const PackedConv& pc, const float* bias, const float* in, int B, int Hi,
            int Wi, int Ho, int Wo, float* out, int epi, const float* aux0, hipStream_t st,
            Deferred* defer = nullptr) {
  ConvArgs a = base_args(pc, B, Ho, Wo);
  a.in = in; a.out = out; a.bias = bias; a.aux0 = aux0;
  a.Hin = Hi; a.Win = Wi; a.Hout = Ho; a.Wout = Wo;
  a.s_in = 2; a.s_out = 1; a.epi = epi;
  taps_conv5_s2(a);
  a.flops = 2.0 * B * Ho * Wo * 25.0 * pc.Kc * pc.N;
  return conv_launch(h, a, st, defer);
}

int conv3(sga_handle* h, const PackedConv& pc, const float* bias, const float* in, int in_cs, int B,
          int Hi, int Wi, float* out, bool true_conv, int epi, const float* aux0, hipStream_t st) {
  ConvArgs a = base_args(pc, B, Hi, Wi);
  a.in = in; a.out = out; a.bias = bias; a.aux0 = aux0; a.in_cs = in_cs;
  a.Hin = Hi; a.Win = Wi; a.Hout = Hi; a.Wout = Wi;
  a.epi = epi;
  taps_conv3(a, true_conv);
  a.flops = 2.0 * B * Hi * Wi * 9.0 * pc.Kc * pc.N;
  return conv_launch(h, a, st);
}

// one launch of the fused GDN tile kernel (gdn_fused.hip), timed like the convolutions
int gdn_launch(sga_handle* h, GdnArgs& g, hipStream_t st) {
  sga_handle::ProfRec r;
  if (h->profiling) {
    r.flops = g.flops;
    char kn[64];
    gdn_kernel_name(g, kn, sizeof(kn));
    if (h->profile_by_layer) snprintf(r.name, sizeof(r.name), "%s %s", h->cur_tag, kn);
    else snprintf(r.name, sizeof(r.name), "%s", kn);
    HIPCHK(h, hipEventCreate(&r.a));
    HIPCHK(h, hipEventCreate(&r.b));
    HIPCHK(h, hipEventRecord(r.a, st));
  }
#ifdef SGA_CLOCK_PROBE
  g.clk = nullptr;
  if (h->clk_mode == 2 && h->clk_slots.size() < 40) {
    hipStreamCaptureStatus ccs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &ccs);
    const long long grid = (g.M + gdn_tile_rows(g.C, g.M, g.pro) - 1) / gdn_tile_rows(g.C, g.M, g.pro);
    if (ccs == hipStreamCaptureStatusActive && grid * 8 <= 6 * 16384) {
      g.clk = h->clk_probe + h->clk_slots.size() * (size_t)(6 * 16384);
      sga_handle::ClkSlot cs;
      char kn[64];
      gdn_kernel_name(g, kn, sizeof(kn));
      snprintf(cs.name, sizeof(cs.name), "gdn:%s %s", h->cur_tag, kn);
      cs.grid = (int)grid;
      h->clk_slots.push_back(cs);
    }
  }
#endif
  g.prio = (h->cur_part == &h->partB) ? h->side_wave_prio : h->main_wave_prio;
  HIPCHK(h, launch_gdn_tile(g, st));
  if (h->profiling) {
    HIPCHK(h, hipEventRecord(r.b, st));
    h->prof.push_back(r);
  }
  return SGA_OK;
}

void gdn_source(GdnArgs& g, const float* tensor, const Deferred* d) {
  for (int p = 0; p < 4; ++p) g.nsplit[p] = 1;
  g.s_out = 1; g.src = tensor; g.slab = 0; g.bias = nullptr;
  if (d && d->active) {
    g.src = d->part; g.slab = d->slab; g.bias = d->bias;
    g.s_out = d->s_out; g.hout = d->hout; g.wout = d->wout;
    for (int p = 0; p < 4; ++p) g.nsplit[p] = d->nsplit[p];
  }
}

// GDN / IGDN forward on u [B,Hh,Ww,C]: out = u * sqrt(n) (inverse) or u / sqrt(n); s_out = sqrt(n).
// d (optional): u has not been assembled yet -- it is the sum of the producing convolution's split-K
// slabs + bias; the kernel assembles it in its prologue and writes it to `u` as well.
int gdn_fwd(sga_handle* h, const PackedConv& pc, const float* beta, float* u, int B, int Hh,
            int Ww, float* s_out, float* out, bool inverse, hipStream_t st, const Deferred* d = nullptr,
            bool write_u = true) {
  if (h->fused_gdn) {
    GdnArgs g;
    memset(&g, 0, sizeof(g));
    g.C = pc.N; g.mode = inverse ? GDN_IGDN_FWD : GDN_GDN_FWD; g.pro = GDN_PRO_LOAD;
    g.M = (long long)B * Hh * Ww;
    gdn_source(g, u, d);
    g.w = pc.w; g.wx3 = pc.w3; g.x3 = (h->x3 && h->x3_variants) ? 1 : 0; g.beta = beta; g.out = out; g.s_out_p = s_out;
    g.u_out = (d && d->active && write_u) ? u : nullptr;
    g.flops = 2.0 * B * Hh * Ww * (double)pc.Kc * pc.N;
    return gdn_launch(h, g, st);
  }
  if (d && d->active) return SGA_ERR_BAD_ARG;
  ConvArgs a = base_args(pc, B, Hh, Ww);
  a.in = u; a.out = out; a.bias = beta; a.aux0 = u; a.aux_out = s_out;
  a.Hin = Hh; a.Win = Ww; a.Hout = Hh; a.Wout = Ww;
  a.pro = PRO_SQUARE; a.epi = inverse ? EPI_IGDN : EPI_GDN;
  taps_single(a);
  a.flops = 2.0 * B * Hh * Ww * (double)pc.Kc * pc.N;
  return conv_launch(h, a, st);
}

// IGDN backward: g_u = g_v * s + u * (gamma . (g_v * u / s)).  g_v is a tensor, or (d) the un-reduced
// split-K slabs of the convolution that produces it, or (gpad != null) the 5x5/2 convolution of the
// zero-bordered 3-channel gradient image with the C->3 layer's kernel, computed in the same launch.
// `v` (optional, fused kernel only): the IGDN's output; when given, u is not read but formed as v / s.
int igdn_bwd(sga_handle* h, const PackedConv& pc, const float* g_v, const float* u, const float* s,
             int B, int Hh, int Ww, float* g_u, hipStream_t st, const Deferred* d = nullptr,
             const float* gpad = nullptr, const PackedConv* pc3 = nullptr, int Hp = 0, int Wp = 0,
             const float* v = nullptr) {
  if (h->fused_gdn) {
    GdnArgs g;
    memset(&g, 0, sizeof(g));
    g.C = pc.N; g.mode = GDN_IGDN_BWD; g.pro = gpad ? GDN_PRO_CONV3 : GDN_PRO_LOAD;
    g.M = (long long)B * Hh * Ww;
    gdn_source(g, g_v, d);
    if (gpad) {
      g.pad = gpad; g.wc = pc3->w; g.Hg = Hh; g.Wg = Ww; g.Hp = Hp; g.Wp = Wp;
    }
    g.w = pc.w; g.wx3 = pc.w3; g.x3 = (h->x3 && h->x3_variants) ? 1 : 0; g.u = u; g.s = s; g.out = g_u; g.v = v;
    g.flops = 2.0 * B * Hh * Ww * (double)pc.Kc * pc.N + (gpad ? 2.0 * B * Hh * Ww * 75.0 * pc.N : 0.0);
    return gdn_launch(h, g, st);
  }
  if ((d && d->active) || gpad) return SGA_ERR_BAD_ARG;
  ConvArgs a = base_args(pc, B, Hh, Ww);
  a.in = g_v; a.aux1 = s; a.aux2 = u; a.out = g_u;
  a.Hin = Hh; a.Win = Ww; a.Hout = Hh; a.Wout = Ww;
  a.pro = PRO_IGDN_BWD; a.epi = EPI_IGDN_BWD;
  taps_single(a);
  a.flops = 2.0 * B * Hh * Ww * (double)pc.Kc * pc.N;
  return conv_launch(h, a, st);
}

// C->3 transposed conv (combined phases): [B,Hi,Wi,C] -> out [B,Ho,Wo,3] cropped to (Ho,Wo)
// `mse_x` != null: the distortion kernel runs in the epilogue (sums, gpad; *mse_done = true) when the halo
// kernel is in use; otherwise the caller launches k_mse
int deconv_to3(sga_handle* h, const PackedConv& pc, const float* bias, const float* in, int B,
               int Hi, int Wi, int Ho, int Wo, float* out, hipStream_t st, const float* mse_x = nullptr,
               int Hp = 0, int Wp = 0, bool* mse_done = nullptr) {
  const bool gemm = h->gs3_gemm && !h->gs3_generic && pc.Kc % 64 == 0 && pc.Kc <= 384 && (size_t)B * Hi * Wi * 80 <= h->p3.cap;
  if (!h->gs3_generic) {
    sga_handle::ProfRec r;
    if (h->profiling) {
      r.flops = 2.0 * B * Hi * Wi * 25.0 * pc.Kc * 3.0;
      const char* kn = gemm ? "deconv3_gemm+col2im" : "deconv3_halo_kernel";
      if (h->profile_by_layer) snprintf(r.name, sizeof(r.name), "%s %s", h->cur_tag, kn);
      else snprintf(r.name, sizeof(r.name), "%s", kn);
      HIPCHK(h, hipEventCreate(&r.a));
      HIPCHK(h, hipEventCreate(&r.b));
      HIPCHK(h, hipEventRecord(r.a, st));
    }
    if (gemm) {      // (the pair as ONE launch -- products of a 16 x 16 tile + halo in LDS -- was built in round 6, bit-equal, and measured
                     //  77.5 us alone against 64.2, +20 us per iteration: git tag lab-r06, profiles/r06_gs3_fused_one_launch.txt)
      const bool fuse = mse_x && h->fused_mse;
      HIPCHK(h, launch_deconv3_gemm(in, h->gs3_w80, h->p3.p, B, Hi, Wi, pc.Kc, st));
      HIPCHK(h, launch_deconv3_col2im(h->p3.p, bias, out, B, Hi, Wi, Ho, Wo, fuse ? mse_x : nullptr, h->ctx, h->sums,
                                      h->gpad.p, Hp, Wp, st));
      if (fuse) *mse_done = true;
    } else if (mse_x && h->fused_mse) {
      HIPCHK(h, launch_deconv3_halo_mse(in, h->gs3_halo_w, bias, out, B, Hi, Wi, pc.Kc, Ho, Wo, mse_x, h->ctx,
                                        h->sums, h->gpad.p, Hp, Wp, st));
      *mse_done = true;
    } else {
      HIPCHK(h, launch_deconv3_halo(in, h->gs3_halo_w, bias, out, B, Hi, Wi, pc.Kc, Ho, Wo, st));
    }
    if (h->profiling) {
      HIPCHK(h, hipEventRecord(r.b, st));
      h->prof.push_back(r);
    }
    return SGA_OK;
  }
  ConvArgs a = base_args(pc, B, Hi, Wi);
  a.in = in; a.out = out; a.bias = bias;
  a.Hin = Hi; a.Win = Wi; a.Hout = Ho; a.Wout = Wo;
  a.Cout = 12; a.out_cs = 3; a.epi = EPI_SHUFFLE3;
  taps_shuffle3(a);
  a.flops = 2.0 * B * Hi * Wi * 25.0 * pc.Kc * 3.0;
  return conv_launch(h, a, st);
}

// 5x5/2 conv over the zero-bordered 3-channel image `pad` [B,Hp,Wp,3] -> [B,Ho,Wo,C]
int conv_smallc(sga_handle* h, const PackedConv& pc, const float* bias, const float* pad, int B,
                int Hp, int Wp, int Ho, int Wo, float* out, hipStream_t st) {
  ConvArgs a = base_args(pc, B, Ho, Wo);
  a.in = pad; a.out = out; a.bias = bias;
  a.Hin = Hp; a.Win = Wp; a.Hout = Ho; a.Wout = Wo;
  a.s_in = 2; a.smallc = 1; a.epi = EPI_BIAS;
  taps_smallc(a);
  a.flops = 2.0 * B * Ho * Wo * 75.0 * pc.N;
  return conv_launch(h, a, st);
}

// ------------------------------------------------------------------------------------------
int check_shape(const sga_handle* h, int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return SGA_ERR_BAD_ARG;
  if (B > h->cfg.max_batch || H > h->cfg.max_height || W > h->cfg.max_width) return SGA_ERR_BAD_SHAPE;
  return SGA_OK;
}

int ensure_borders(sga_handle* h, const Geom& g, hipStream_t st) {
  if (h->borders_valid && h->geom_zeroed.B == g.B && h->geom_zeroed.H == g.H &&
      h->geom_zeroed.W == g.W)
    return SGA_OK;
  HIPCHK(h, launch_fill(h->xpad.p, 0.f, (int64_t)(h->xpad.cap), st));
  HIPCHK(h, launch_fill(h->gpad.p, 0.f, (int64_t)(h->gpad.cap), st));
  h->geom_zeroed = g;
  h->borders_valid = true;
  return SGA_OK;
}

inline float inv_ln2_hw(const Geom& g) { return (float)(1.0 / (0.6931471805599453 * g.H * (double)g.W)); }

// sga.py:77-78: y = g_a(x), z = h_a(y)
int encode_impl(sga_handle* h, const Geom& g, const float* x, float* y, float* z, hipStream_t st) {
  const int B = g.B, C = h->C;
  HIPCHK(h, launch_pad_image(x, B, g.H, g.W, g.xHp, g.xWp, h->xpad.p, st));
  // temporaries borrowed from the synthesis workspace (large enough: 8*yh >= eh[1])
  float* ubuf = h->u[2].p; float* vbuf = h->v[2].p; float* v2buf = h->s[2].p;
  SGACHK(conv_smallc(h, h->ga_f[0], h->ga_bias[0], h->xpad.p, B, g.xHp, g.xWp, g.eh[1], g.ew[1], ubuf, st));
  SGACHK(gdn_fwd(h, h->ga_gdn[0], h->ga_beta[0], ubuf, B, g.eh[1], g.ew[1], nullptr, vbuf, false, st));
  SGACHK(conv5s2(h, h->ga_f[1], h->ga_bias[1], vbuf, B, g.eh[1], g.ew[1], g.eh[2], g.ew[2], ubuf, EPI_BIAS, nullptr, st));
  SGACHK(gdn_fwd(h, h->ga_gdn[1], h->ga_beta[1], ubuf, B, g.eh[2], g.ew[2], nullptr, v2buf, false, st));
  SGACHK(conv5s2(h, h->ga_f[2], h->ga_bias[2], v2buf, B, g.eh[2], g.ew[2], g.eh[3], g.ew[3], ubuf, EPI_BIAS, nullptr, st));
  SGACHK(gdn_fwd(h, h->ga_gdn[2], h->ga_beta[2], ubuf, B, g.eh[3], g.ew[3], nullptr, vbuf, false, st));
  SGACHK(conv5s2(h, h->ga_f[3], h->ga_bias[3], vbuf, B, g.eh[3], g.ew[3], g.yh, g.yw, y, EPI_BIAS, nullptr, st));
  // h_a (nn_models.py:85-96): conv3x3+relu, conv5x5/2+relu, conv5x5/2 (no bias)
  float* t0 = h->hs1.p; float* t1 = h->hs0.p;
  SGACHK(conv3(h, h->ha_f[0], h->ha_bias[0], y, C, B, g.yh, g.yw, t0, false, EPI_BIAS_RELU, nullptr, st));
  SGACHK(conv5s2(h, h->ha_f[1], h->ha_bias[1], t0, B, g.yh, g.yw, g.zh1, g.zw1, t1, EPI_BIAS_RELU, nullptr, st));
  SGACHK(conv5s2(h, h->ha_f[2], nullptr, t1, B, g.zh1, g.zw1, g.zh, g.zw, z, EPI_BIAS, nullptr, st));
  return SGA_OK;
}

// Hyper-prior branch given z_tilde (and y_tilde for the conditional): p(z_tilde), (mu, sigma) =
// h_s(z_tilde), p(y_tilde | z_tilde) and, with_grad, the data-gradients back to z_tilde
// (sga.py:100-108, 126-136 and their part of sga.py:164).  Every launch goes to `st`.
// `part`: 0 = everything, 1 = the forward half only (up to the conditional's likelihood), 2 = the backward half only
int hyper_branch_impl(sga_handle* h, const Geom& g, bool with_grad, hipStream_t st, bool density, int part);
int hyper_branch(sga_handle* h, const Geom& g, bool with_grad, hipStream_t st, bool density = false, int part = 0) {
  h->in_hyper = true;
  const int rc = hyper_branch_impl(h, g, with_grad, st, density, part);
  h->in_hyper = false;
  return rc;
}
int hyper_branch_impl(sga_handle* h, const Geom& g, bool with_grad, hipStream_t st, bool density, int part) {
  const int B = g.B, C = h->C;
  const float il = inv_ln2_hw(g);
  if (part != 2) {
  // experiment (SGA_FORK_DELAY_US): hold the branch back for a fixed time after its fork point -- a fork "inside" a launch
  if (h->fork_delay_us > 0 && st == h->sB) HIPCHK(h, launch_spin(h->fork_delay_us, st));
  if (h->dump && h->dump_probe && with_grad) HIPCHK(h, launch_probe(h->dump, h->ctx, st));
  if (density)   // bits-back: prior DENSITY (bb_sga.py:105-106)
    HIPCHK(h, launch_factorized_pdf(h->zt.p, h->eb_packed, h->ctx, B, g.zh * g.zw, C, il, h->sums,
                                    with_grad ? h->g_zt_eb.p : nullptr, nullptr, nullptr, st));
  else
    HIPCHK(h, launch_factorized(h->zt.p, h->eb_packed, h->ctx, B, g.zh * g.zw, C, il, h->sums,
                                with_grad ? h->g_zt_eb.p : nullptr, nullptr, nullptr, st));
  h->cur_tag = "hs0.fwd";
  SGACHK(deconv_fwd(h, h->hs_f[0], h->hs_bias[0], h->zt.p, B, g.zh, g.zw, h->hs0.p, EPI_BIAS_RELU, st));
  h->cur_tag = "hs1.fwd";
  SGACHK(deconv_fwd(h, h->hs_f[1], h->hs_bias[1], h->hs0.p, B, 2 * g.zh, 2 * g.zw, h->hs1.p, EPI_BIAS_RELU, st));
  h->cur_tag = "hs2.fwd";
  SGACHK(conv3(h, h->hs_f[2], h->hs_bias[2], h->hs1.p, h->C15, B, g.hsh, g.hsw, h->ms.p, true, EPI_BIAS, nullptr, st));
  HIPCHK(h, launch_gaussian(h->yt.p, h->ms.p, h->ctx, B, g.yh, g.yw, g.hsh, g.hsw, C, il, h->scale_bound, h->sums,
                            with_grad ? h->g_yt_rate.p : nullptr, with_grad ? h->g_ms.p : nullptr, st));
  }
  if (!with_grad || part == 1) return SGA_OK;
  h->cur_tag = "hs2.bwd";
  SGACHK(conv3(h, h->hs_b[2], nullptr, h->g_ms.p, 2 * C, B, g.hsh, g.hsw, h->g_hs1.p, false,
               EPI_RELU_MASK, h->hs1.p, st));
  h->cur_tag = "hs1.bwd";
  SGACHK(conv5s2(h, h->hs_b[1], nullptr, h->g_hs1.p, B, g.hsh, g.hsw, 2 * g.zh, 2 * g.zw, h->g_hs0.p,
                 EPI_RELU_MASK, h->hs0.p, st));
  h->cur_tag = "hs0.bwd";
  SGACHK(conv5s2(h, h->hs_b[0], nullptr, h->g_hs0.p, B, 2 * g.zh, 2 * g.zw, g.zh, g.zw, h->g_zt_hs.p,
                 EPI_BIAS, nullptr, st));
  h->cur_tag = "";
  return SGA_OK;
}

// Synthesis branch given y_tilde: x_tilde = g_s(y_tilde), distortion and, with_grad, the
// data-gradients back to y_tilde (sga.py:122-123, 150-154 and their part of sga.py:164).
// `side`: called once, right before main-chain launch number `fork_at` (0 = the first one), to
// enqueue whatever runs concurrently on the second stream.
int synth_branch(sga_handle* h, const Geom& g, const float* x, bool with_grad, hipStream_t st,
                 int fork_at = 0, const std::function<int()>* side = nullptr, const std::function<int()>* side2 = nullptr) {
  int launches = 0;
  bool side_started = false, side2_started = false, fork_pending = false;
  // call before every main-chain launch.  The hyper branch is forked at launch number `fork_at` (SGA_FORK_AT, experiments)
  // or, when the handle names one (fork_name: chosen per geometry by sga_run_steps), right before that launch
  auto tick = [&](const char* name = nullptr) -> int {
    // Never before the FIRST launch: the branch's first kernel would then be a second root node of the captured graph, with
    // no predecessor inside it.  Kept as a structural rule (one root per step graph).  It was introduced while hunting the
    // bf16x3 two-stream nondeterminism and moved the failure rate (3 outcomes in 20 replays -> 0 in 20) without being the
    // cause: the cause was the packed-f32 VALU hazard beside bf16 MFMA waves (csrc/Makefile, DESIGN_EXPERIMENTS.md A.8b),
    // which the root fork merely made likelier by co-scheduling the elementwise kernels with the X3 convolutions.
    // (a fork point that NAMES the first launch -- SGA_FORK_NAME=gs0.fwd -- is therefore taken at the second one: `pending`;
    //  before round 5 it never matched and the step ran serialised)
    const bool named = h->fork_name ? (name && strcmp(name, h->fork_name) == 0) : launches >= fork_at;
    if (named && launches == 0) fork_pending = true;
    const bool here = (named || fork_pending) && launches >= 1;
    if (side && !side_started && here) { side_started = true; SGACHK((*side)()); }
    // second fork point (captured graph only): the branch's BACKWARD half may not start before this launch
    if (side2 && !side2_started && name && h->fork2_name && strcmp(name, h->fork2_name) == 0) { side2_started = true; SGACHK((*side2)()); }
    ++launches;
    return SGA_OK;
  };
  auto finish_side = [&]() -> int {
    if (side && !side_started) { side_started = true; SGACHK((*side)()); }
    return SGA_OK;
  };
  const int B = g.B;
  const float* cur = h->yt.p;
  int hh = g.yh, ww = g.yw;
  static const char* kFwd[3] = {"gs0.fwd", "gs1.fwd", "gs2.fwd"};
  static const char* kIgdn[3] = {"igdn0.fwd", "igdn1.fwd", "igdn2.fwd"};
  static const char* kIgdnB[3] = {"igdn0.bwd", "igdn1.bwd", "igdn2.bwd"};
  static const char* kBwd[3] = {"gs0.bwd", "gs1.bwd", "gs2.bwd"};
  // Layers whose tile grid under-fills the chip run split-K; their partial slabs are summed (+ bias)
  // by the IGDN kernel that follows instead of by a reduce launch (fused_gdn).
  const bool fz = h->fused_gdn;
  // u is needed again only by the IGDN data-gradient, which can form it as v / s: with the tile kernels in use the
  // forward pass stores s and v only (gs2.fwd + IGDN writes 201 instead of 302 MB at cfg 2)
  const bool drop_u = fz && with_grad && !h->keep_u;
  for (int L = 0; L < 3; ++L) {
    Deferred d;
    PostGdn pg;
    pg.gamma_w = h->gs_gdn_f[L].w; pg.gamma_w3 = h->gs_gdn_f[L].w3; pg.beta = h->gs_beta[L]; pg.s_out = h->s[L].p; pg.v_out = h->v[L].p;
    pg.drop_u = drop_u;
    SGACHK(tick(kFwd[L]));
    h->cur_tag = kFwd[L];
    SGACHK(deconv_fwd(h, h->gs_f[L], h->gs_bias[L], cur, B, hh, ww, h->u[L].p, EPI_BIAS, st, fz ? &d : nullptr,
                      fz ? &pg : nullptr));
    hh *= 2; ww *= 2;
    if (!pg.fused) {           // otherwise the IGDN ran as the post-phase of the convolution launch
      SGACHK(tick());
      h->cur_tag = kIgdn[L];
      SGACHK(gdn_fwd(h, h->gs_gdn_f[L], h->gs_beta[L], h->u[L].p, B, hh, ww, h->s[L].p, h->v[L].p, true, st, &d, !drop_u));
    }
    cur = h->v[L].p;
  }
  SGACHK(tick("gs3.fwd"));
  h->cur_tag = "gs3.fwd";
  bool mse_done = false;     // step: the distortion sums and the gradient image come out of gs3.fwd's epilogue
  SGACHK(deconv_to3(h, h->gs_f[3], h->gs_bias[3], cur, B, hh, ww, g.H, g.W, h->xt.p, st, with_grad ? x : nullptr,
                    g.Hp, g.Wp, &mse_done));
  if (!mse_done) {
    SGACHK(tick());
    HIPCHK(h, launch_mse(x, h->xt.p, with_grad ? h->ctx : nullptr, B, g.H, g.W, g.Hp, g.Wp, h->sums,
                         with_grad ? h->gpad.p : nullptr, with_grad ? nullptr : h->xq.p, st));
  }
  if (!with_grad) {
    SGACHK(finish_side());
    return SGA_OK;
  }
  // hh,ww = 8yh,8yw: gradient w.r.t. v[2] from the bordered gradient image
  // gs3.bwd (the 5x5/2 conv of the 3-channel gradient image) is the prologue of igdn2.bwd when fused
  const bool conv3_fused = fz;
  if (!conv3_fused) {
    SGACHK(tick());
    h->cur_tag = "gs3.bwd";
    SGACHK(conv_smallc(h, h->gs_b[3], nullptr, h->gpad.p, B, g.Hp, g.Wp, hh, ww, h->gA.p, st));
  }
  Deferred d;
  for (int L = 2; L >= 0; --L) {
    SGACHK(tick(kIgdnB[L]));
    h->cur_tag = kIgdnB[L];
    if (L == 2 && conv3_fused)
      SGACHK(igdn_bwd(h, h->gs_gdn_b[L], nullptr, h->u[L].p, h->s[L].p, B, hh, ww, h->gB.p, st, nullptr,
                      h->gpad.p, &h->gs_b[3], g.Hp, g.Wp, drop_u ? h->v[L].p : nullptr));
    else
      SGACHK(igdn_bwd(h, h->gs_gdn_b[L], h->gA.p, h->u[L].p, h->s[L].p, B, hh, ww, h->gB.p, st, &d, nullptr, nullptr, 0, 0,
                      drop_u ? h->v[L].p : nullptr));
    float* dst = (L == 0) ? h->g_yt_dist.p : h->gA.p;
    SGACHK(tick(kBwd[L]));
    h->cur_tag = kBwd[L];
    d = Deferred();
    SGACHK(conv5s2(h, h->gs_b[L], nullptr, h->gB.p, B, hh, ww, hh / 2, ww / 2, dst, EPI_BIAS, nullptr, st,
                   (fz && L > 0) ? &d : nullptr));
    hh /= 2; ww /= 2;
  }
  SGACHK(finish_side());
  h->cur_tag = "";
  return SGA_OK;
}

// forward (+ backward) of the rate-distortion graph given (relaxed or rounded) latents in
// h->yt / h->zt.  The two branches are independent once y_tilde and z_tilde exist: the hyper
// branch is forked to the handle's second stream and joined back before returning (the same
// fork/join is recorded into the hipGraph when `st` is being captured).
int rd_forward_backward(sga_handle* h, const Geom& g, const float* x, bool with_grad,
                        hipStream_t st, bool density = false, bool do_synth = true) {
  // bf16x3 mode runs single-stream: with the hyper branch on a second stream its results were not
  // reproducible run to run (the f32 mode is, and is the default)
  if (h->branch_only == 1) {       // hybrid replay: this call enqueues one branch only (sga_run_steps orders them)
    h->cur_part = &h->part;
    return synth_branch(h, g, x, with_grad, st);
  }
  if (h->branch_only == 2) {
    h->cur_part = &h->partB;
    const int rc = hyper_branch(h, g, with_grad, st, density);
    h->cur_part = &h->part;
    return rc;
  }
  h->hyper_alone = !do_synth;
  const bool fork = h->overlap && (!h->x3 || h->x3_fork) && !h->profiling && do_synth;
  if (!fork) {
    h->cur_part = &h->part;
    SGACHK(hyper_branch(h, g, with_grad, st, density));
    return do_synth ? synth_branch(h, g, x, with_grad, st) : SGA_OK;
  }
  // An event recorded during stream capture only exists as a graph dependency; eager launches
  // (evaluation, step_grads) therefore use their own pair and never wait on a captured event.
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  hipEvent_t evf = cs == hipStreamCaptureStatusActive ? h->ev_fork_cap : h->ev_fork;
  hipEvent_t evj = cs == hipStreamCaptureStatusActive ? h->ev_join_cap : h->ev_join;
  bool forked = false;
  // Under capture the branch's launches are recorded AFTER the whole main chain (the edges are the same:
  // fork event at `fork_at`, join at the end).  The graph then puts the main chain on the launching
  // stream's queue and issues its kernels ahead of the branch's whenever both are ready: the branch
  // fills what the chain leaves free instead of competing with it (2009 -> 1917 us per iteration at the
  // bench shape; unrolling several iterations into one graph measured no gain).
  const bool late = h->side_last && cs == hipStreamCaptureStatusActive;
  auto side_body = [&]() -> int {
    h->cur_part = &h->partB;
    const char* tag = h->cur_tag;
    const int rc = hyper_branch(h, g, with_grad, h->sB, density);
    h->cur_part = &h->part;
    h->cur_tag = tag;
    return rc;
  };
  const std::function<int()> side = [&]() -> int {
    HIPCHK(h, hipEventRecord(evf, st));
    HIPCHK(h, hipStreamWaitEvent(h->sB, evf, 0));
    forked = true;
    if (late) return SGA_OK;       // the branch's nodes are created after the main chain's (same edges)
    h->cur_part = &h->partB;
    const char* tag = h->cur_tag;
    const int rc = hyper_branch(h, g, with_grad, h->sB, density);
    h->cur_part = &h->part;
    h->cur_tag = tag;
    return rc;
  };
  // Split fork (captured graph, fork2_name set): the branch's forward half is released at the first fork point, its
  // backward half additionally waits for a second, later main-chain launch.
  bool forked2 = false;
  const bool split = late && with_grad && h->fork2_name != nullptr;
  const std::function<int()> side2 = [&]() -> int {
    HIPCHK(h, hipEventRecord(h->ev_fork2_cap, st));
    forked2 = true;
    return SGA_OK;
  };
  int rc = synth_branch(h, g, x, with_grad, st, h->fork_at, &side, split ? &side2 : nullptr);
  if (forked && late && rc == SGA_OK) {
    if (split && forked2) {
      h->cur_part = &h->partB;
      const char* tag = h->cur_tag;
      rc = hyper_branch(h, g, with_grad, h->sB, density, 1);
      if (rc == SGA_OK && hipStreamWaitEvent(h->sB, h->ev_fork2_cap, 0) != hipSuccess) rc = SGA_ERR_HIP;
      if (rc == SGA_OK) rc = hyper_branch(h, g, with_grad, h->sB, density, 2);
      h->cur_part = &h->part;
      h->cur_tag = tag;
    } else {
      rc = side_body();
    }
  }
  // always join, even on error, so a capture in progress is not left forked
  if (forked) {
    const hipError_t e1 = hipEventRecord(evj, h->sB);
    const hipError_t e2 = hipStreamWaitEvent(st, evj, 0);
    SGACHK(rc);
    HIPCHK(h, e1);
    HIPCHK(h, e2);
  }
  return rc;
}

// one SGA evaluation: sample both latents (sga.py:86-98,111-121) then forward + backward
int sga_step_core(sga_handle* h, const Geom& g, const float* x, const float* y, const float* z,
                  const float* u_y, const float* u_z, hipStream_t st) {
  const int64_t ny = (int64_t)g.B * g.yh * g.yw * h->C, nz = (int64_t)g.B * g.zh * g.zw * h->C;
  if (!u_y && !u_z) {
    HIPCHK(h, launch_sample_yz(y, h->yt.p, h->dyt.p, ny, z, h->zt.p, h->dzt.p, nz, h->ctx, h->relax, h->img_ids,
                               g.B, st));
    if (h->dump && h->dump_probe) HIPCHK(h, launch_mark(h->dump, h->ctx, st));
  } else {
    HIPCHK(h, launch_sample(z, u_z, h->ctx, 1, h->zt.p, h->dzt.p, nz, st, h->relax, h->img_ids, nz / g.B));
    HIPCHK(h, launch_sample(y, u_y, h->ctx, 0, h->yt.p, h->dyt.p, ny, st, h->relax, h->img_ids, ny / g.B));
  }
  return rd_forward_backward(h, g, x, true, st);
}

int eval_impl(sga_handle* h, const Geom& g, const float* x, const float* y_hat, const float* z_hat,
              float* metrics, float* x_hat, hipStream_t st) {
  const int64_t ny = (int64_t)g.B * g.yh * g.yw * h->C, nz = (int64_t)g.B * g.zh * g.zw * h->C;
  HIPCHK(h, launch_copy(h->yt.p, y_hat, (int64_t)(ny), st));
  HIPCHK(h, launch_copy(h->zt.p, z_hat, (int64_t)(nz), st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (g.B), st));
  SGACHK(rd_forward_backward(h, g, x, false, st));
  if (metrics) {
    HIPCHK(h, launch_finalize_eval(h->sums, g.B, g.H, g.W, metrics, st));
    // sga.py:175-176; TF asserts H,W >= 176 for 5 scales: smaller images keep NaN
    if (g.H >= 176 && g.W >= 176)
      HIPCHK(h, launch_msssim(h->xq.p, x, g.B, g.H, g.W, h->msA, h->msB, h->ms_stats, h->ms_counts,
                              metrics, 7, st));
  }
  if (x_hat)
    HIPCHK(h, launch_copy(x_hat, h->xt.p, (int64_t)((size_t)g.B * g.H * g.W * 3), st));
  return SGA_OK;
}

// Disposal of an executable graph that is dropped in mid-life (a losing fork-point candidate, the least recently used entry of a
// full cache, the stamped graph of sga_profile_graph_*): destroyed at once, behind a synchronisation of the streams its replays
// ran on.  Rounds 3-5 RETIRED such graphs until sga_destroy instead, because mid-life destruction was followed -- in long
// processes, never in a short one -- by a host SIGSEGV inside a later hipGraphLaunch.  Round 6 found why (DESIGN_EXPERIMENTS.md
// A.13): not the destruction, but what it does to the hardware queues' reference counts -- the HIP runtime's
// Graph::UpdateStreams overruns a graph's internal stream list when all of them share the LAUNCH stream's hardware queue, and
// destroying streams is what makes two consecutive stream creations land on one queue.  Graphs are launched on a stream of
// another priority class now (sga_handle::sG), which can share no queue with them; the retire list is gone.
// The calls that replay step graphs run on the handle's own launch stream (sga_handle::sG): ordered after everything the caller
// has enqueued on ITS stream so far, and the caller's stream ordered after them on the way out -- also on an error return.
struct LaunchStream {
  sga_handle* h;
  hipStream_t user, run;
  LaunchStream(sga_handle* h_, void* stream) : h(h_), user((hipStream_t)stream), run((hipStream_t)stream) {
    if (h->sG && h->use_graph && hipEventRecord(h->ev_in, user) == hipSuccess && hipStreamWaitEvent(h->sG, h->ev_in, 0) == hipSuccess)
      run = h->sG;
  }
  ~LaunchStream() {
    if (run != user && (hipEventRecord(h->ev_out, run) != hipSuccess || hipStreamWaitEvent(user, h->ev_out, 0) != hipSuccess))
      (void)hipStreamSynchronize(run);      // the ordering must hold even if the event pair failed
  }
};

void drop_graph(sga_handle* h, hipGraphExec_t& ex, hipStream_t st = nullptr) {
  if (!ex) return;
  if (st) (void)hipStreamSynchronize(st);
  if (h->sG) (void)hipStreamSynchronize(h->sG);      // (callers without a stream: sga_profile_graph_begin / _end)
  if (h->sB) (void)hipStreamSynchronize(h->sB);
  (void)hipGraphExecDestroy(ex);
  ++h->n_dropped;
  ex = nullptr;
}

constexpr size_t kMaxGraphs = 16;

sga_handle::GraphEntry* find_graph(sga_handle* h, const sga_handle::GraphKey& k) {
  for (auto& e : h->graphs)
    if (e.key == k) { e.used = ++h->graph_clock; return &e; }
  return nullptr;
}

// the fork point a TIMED graph of this geometry AND KIND was built with (any relaxation / bound: the point depends on the launch
// durations, which these do not change; the bits-back stage-1 step has other launches in its hyper branch -- the density, the
// posterior terms -- so its point is never taken from an SGA graph or the other way round), or null
const char* tuned_fork_for(const sga_handle* h, int kind, int B, int H, int W, bool* found) {
  *found = false;
  for (const auto& e : h->graphs)
    if (e.tuned && e.key.kind == kind && e.key.B == B && e.key.H == H && e.key.W == W) { *found = true; return e.fork_name; }
  return nullptr;
}

sga_handle::GraphEntry* insert_graph(sga_handle* h, const sga_handle::GraphKey& k, hipGraphExec_t ex, bool tuned, const char* fork_name,
                                     bool stamp_in_graph, hipStream_t st) {
  if (h->graphs.size() >= kMaxGraphs) {      // the least recently used entry makes room (retired, or destroyed behind a synchronisation)
    size_t lru = 0;
    for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].used < h->graphs[lru].used) lru = i;
    (void)hipStreamSynchronize(st);
    drop_graph(h, h->graphs[lru].exec, st);
    h->graphs.erase(h->graphs.begin() + (long)lru);
    ++h->n_graph_evictions;
  }
  h->graphs.push_back(sga_handle::GraphEntry{k, ex, tuned, fork_name, stamp_in_graph, ++h->graph_clock});
  return &h->graphs.back();
}

void erase_graph(sga_handle* h, sga_handle::GraphEntry* e, hipStream_t st) {
  (void)hipStreamSynchronize(st);      // replays of it may still be queued
  drop_graph(h, e->exec, st);
  h->graphs.erase(h->graphs.begin() + (e - h->graphs.data()));
}

// The fork point of the hyper branch, chosen by time: captures one step graph per candidate (`capture` records
// one iteration under the current h->fork_name), replays each a few times -- on the caller's live state: the replays are
// iterations of the run and are counted in *done -- and returns the fastest; the others are destroyed (drop_graph).
template <typename Cap>
int timed_fork_choice(sga_handle* h, hipStream_t st, int B, int H, int W, Cap&& capture, hipGraphExec_t* out, int* done) {
  constexpr int kTuneReps = 8;
  static const char* const cands[3] = {nullptr, "gs2.fwd", "gs3.fwd"};
  static const bool verbose = getenv("SGA_FORK_VERBOSE") != nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HIPCHK(h, hipEventCreate(&e0));
  HIPCHK(h, hipEventCreate(&e1));
  float best_ms = 0.f;
  hipGraphExec_t best = nullptr;
  const char* best_name = nullptr;
  for (int c = 0; c < 3; ++c) {
    h->fork_name = cands[c];
    hipGraphExec_t ex = capture();
    if (!ex) continue;
    float ms = 0.f;
    bool ok = true;                                            // replays to settle (the first candidate also absorbs
    const int settle = c == 0 ? 3 : 1;                         // the clock ramp of a fresh run), kTuneReps timed
    for (int r = 0; r < settle && ok; ++r) ok = hipGraphLaunch(ex, st) == hipSuccess;
    ok = ok && hipEventRecord(e0, st) == hipSuccess;
    for (int r = 0; r < kTuneReps && ok; ++r) ok = hipGraphLaunch(ex, st) == hipSuccess;
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (!ok) {      // a failed launch leaves the run in an unknown state: report it
      drop_graph(h, ex, st);
      drop_graph(h, best, st);
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      h->fork_name = nullptr;
      HIPCHK(h, hipErrorUnknown);
    }
    *done += settle + kTuneReps;
    if (verbose)
      fprintf(stderr, "sga fork point %-8s B=%d %dx%d: %.1f us per iteration\n", cands[c] ? cands[c] : "start", B, H, W,
              ms * 1000.f / kTuneReps);
    if (!best || ms < best_ms) {
      drop_graph(h, best, st);
      best = ex; best_ms = ms; best_name = cands[c];
    } else {
      drop_graph(h, ex, st);
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *out = best;
  h->fork_name = best_name;
  if (best) { h->tuned_B = B; h->tuned_H = H; h->tuned_W = W; h->tuned_name = best_name; }
  return SGA_OK;
}

void free_all(sga_handle* h) {
  for (auto& e : h->graphs) if (e.exec) (void)hipGraphExecDestroy(e.exec);
  h->graphs.clear();
  if (h->graph_main) (void)hipGraphExecDestroy(h->graph_main);
  for (auto& r : h->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  if (h->ev_fork2_cap) (void)hipEventDestroy(h->ev_fork2_cap);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_fork_cap) (void)hipEventDestroy(h->ev_fork_cap);
  if (h->ev_join_cap) (void)hipEventDestroy(h->ev_join_cap);
  if (h->sB) (void)hipStreamDestroy(h->sB);
  if (h->sG) (void)hipStreamDestroy(h->sG);
  if (h->ev_in) (void)hipEventDestroy(h->ev_in);
  if (h->ev_out) (void)hipEventDestroy(h->ev_out);
  for (void* p : h->owned) (void)hipFree(p);
  h->owned.clear();
}

}  // namespace

// ============================================================================================
extern "C" {

int sga_abi_version(void) { return SGA_ABI_VERSION; }

int sga_create(sga_handle** out, const sga_config* cfg, const sga_weights* w) {
  if (!out || !cfg || !w) return SGA_ERR_BAD_ARG;
  *out = nullptr;
  if (cfg->num_filters <= 0 || cfg->num_filters % 64 != 0) return SGA_ERR_UNSUPPORTED;
  if (cfg->max_batch <= 0 || cfg->max_height <= 0 || cfg->max_width <= 0) return SGA_ERR_BAD_ARG;
  if (!(cfg->scale_bound >= 0.f) || !(cfg->scale_bound < 1e30f)) return SGA_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return SGA_ERR_NO_DEVICE;
#ifdef SGA_EXPERIMENTS
  if (LAB_ENV("SGA_DEBUG_SEGV")) sga_install_segv_handler(LAB_ENV("SGA_DEBUG_SEGV"));
#endif
  sga_handle* h = new (std::nothrow) sga_handle();
  if (!h) return SGA_ERR_NOMEM;
  h->cfg = *cfg;
  h->scale_bound = cfg->scale_bound;
  const int C = cfg->num_filters;
  h->C = C; h->C15 = (int)(C * 1.5); h->C2 = 2 * C;
  h->haN = cfg->bits_back ? 2 * C : C;
  {
    const char* pe = getenv("SGA_PRECISION");
    h->x2 = cfg->precision == SGA_PRECISION_BF16X2 ||
            (cfg->precision == SGA_PRECISION_DEFAULT && pe && strcmp(pe, "bf16x2") == 0);
    h->x3 = h->x2 || cfg->precision == SGA_PRECISION_BF16X3 ||
            (cfg->precision == SGA_PRECISION_DEFAULT && pe && strcmp(pe, "bf16x3") == 0);
    pe = getenv("SGA_X3_VARIANTS");      // read here: the weight packing below depends on it (bn96_as_192)
    h->x3_variants = !(pe && pe[0] == '0');
  }
  int st = SGA_OK;
  auto fail = [&](int code) { free_all(h); delete h; return code; };
#define TRY(expr) do { st = (expr); if (st != SGA_OK) return fail(st); } while (0)

  for (int i = 0; i < 4; ++i) {
    if (!w->ga_kernel[i] || !w->ga_bias[i] || !w->gs_kernel[i] || !w->gs_bias[i]) return fail(SGA_ERR_BAD_ARG);
  }
  // ---- analysis ----
  TRY(pack_smallc(h, h->ga_f[0], w->ga_kernel[0], C, false));
  for (int i = 1; i < 4; ++i) TRY(pack_fwd(h, h->ga_f[i], w->ga_kernel[i], 25, C, C, EPI_BIAS));
  for (int i = 0; i < 4; ++i) TRY(upload(h, &h->ga_bias[i], w->ga_bias[i], C));
  for (int i = 0; i < 3; ++i) {
    TRY(pack_gdn(h, h->ga_gdn[i], w->ga_gamma[i], C, false));
    TRY(upload(h, &h->ga_beta[i], w->ga_beta[i], C));
  }
  // ---- synthesis ----
  for (int i = 0; i < 3; ++i) {
    TRY(pack_fwd(h, h->gs_f[i], w->gs_kernel[i], 25, C, C, EPI_BIAS));
    TRY(pack_bwd(h, h->gs_b[i], w->gs_kernel[i], 25, C, C));
    TRY(pack_gdn(h, h->gs_gdn_f[i], w->gs_gamma[i], C, false));
    TRY(pack_gdn(h, h->gs_gdn_b[i], w->gs_gamma[i], C, true));
    TRY(upload(h, &h->gs_beta[i], w->gs_beta[i], C));
    TRY(upload(h, &h->gs_bias[i], w->gs_bias[i], C));
  }
  TRY(pack_shuffle3(h, h->gs_f[3], w->gs_kernel[3], C));
  {
    // same tap/column mapping as pack_shuffle3, chunk-major for the halo kernel
    std::vector<float> hw((size_t)(C / 32) * 9 * 16 * 32, 0.f);
    const float* K = w->gs_kernel[3];
    for (int c = 0; c < C / 32; ++c)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int t = (dy + 1) * 3 + (dx + 1);
          for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
              const int ky = py + 2 - 2 * dy, kx = px + 2 - 2 * dx;
              if (ky < 0 || ky > 4 || kx < 0 || kx > 4) continue;
              for (int ch = 0; ch < 3; ++ch) {
                const int n = (py * 2 + px) * 3 + ch;
                for (int k = 0; k < 32; ++k)
                  hw[(((size_t)c * 9 + t) * 16 + n) * 32 + k] =
                      K[(((size_t)ky * 5 + kx) * C + c * 32 + k) * 3 + ch];
              }
            }
        }
    TRY(upload(h, &h->gs3_halo_w, hw.data(), hw.size()));
    {
      std::vector<float> w80((size_t)80 * C, 0.f);
      for (int t = 0; t < 25; ++t)
        for (int ci = 0; ci < C; ++ci)
          for (int ch = 0; ch < 3; ++ch) w80[(size_t)(t * 3 + ch) * C + ci] = K[((size_t)t * C + ci) * 3 + ch];
      TRY(upload(h, &h->gs3_w80, w80.data(), w80.size()));
      // Measured at cfg 2: GEMM 40.7 us + col2im 22.4 us = 63 us alone against 75.6 us for the halo
      // kernel, but inside the iteration the pair is 3 us SLOWER (two launches, 64 KB of LDS per 8-wave workgroup next
      // to the hyper branch): the halo kernel stays the default, SGA_GS3_GEMM=1 selects this path.
      // Round 3: ON by default -- together with the other faster-alone variants and the later fork point of the hyper branch
      // the iteration is 30 us shorter (joint sweep, DESIGN_EXPERIMENTS.md A.7); SGA_GS3_GEMM=0 selects the halo kernel.
      const char* eg = LAB_ENV("SGA_GS3_GEMM");
      h->gs3_gemm = !(eg && eg[0] == '0');
    }
    const char* e3 = LAB_ENV("SGA_GS3_GENERIC");
    h->gs3_generic = e3 && e3[0] == '1';
  }
  TRY(pack_smallc(h, h->gs_b[3], w->gs_kernel[3], C, true));
  TRY(upload(h, &h->gs_bias[3], w->gs_bias[3], 3));
  // ---- hyper-analysis ----
  TRY(pack_fwd(h, h->ha_f[0], w->ha_kernel[0], 9, C, C, EPI_BIAS));
  TRY(pack_fwd(h, h->ha_f[1], w->ha_kernel[1], 25, C, C, EPI_BIAS));
  TRY(pack_fwd(h, h->ha_f[2], w->ha_kernel[2], 25, C, h->haN, EPI_BIAS));
  TRY(upload(h, &h->ha_bias[0], w->ha_bias[0], C));
  TRY(upload(h, &h->ha_bias[1], w->ha_bias[1], C));
  // ---- hyper-synthesis ----
  TRY(pack_fwd(h, h->hs_f[0], w->hs_kernel[0], 25, C, C, EPI_BIAS));
  TRY(pack_fwd(h, h->hs_f[1], w->hs_kernel[1], 25, C, h->C15, EPI_BIAS));
  TRY(pack_fwd(h, h->hs_f[2], w->hs_kernel[2], 9, h->C15, 2 * C, EPI_BIAS));
  TRY(pack_bwd(h, h->hs_b[0], w->hs_kernel[0], 25, C, C));
  TRY(pack_bwd(h, h->hs_b[1], w->hs_kernel[1], 25, C, h->C15));
  TRY(pack_bwd(h, h->hs_b[2], w->hs_kernel[2], 9, h->C15, 2 * C));
  TRY(upload(h, &h->hs_bias[0], w->hs_bias[0], C));
  TRY(upload(h, &h->hs_bias[1], w->hs_bias[1], h->C15));
  TRY(upload(h, &h->hs_bias[2], w->hs_bias[2], 2 * C));
  // ---- factorized prior, packed per channel (elementwise.hip: eb_logit) ----
  {
    std::vector<float> P((size_t)C * EB_STRIDE, 0.f);
    for (int c = 0; c < C; ++c) {
      float* p = &P[(size_t)c * EB_STRIDE];
      for (int r = 0; r < 3; ++r) {
        p[r] = w->eb_matrix[0][c * 3 + r];
        p[3 + r] = w->eb_bias[0][c * 3 + r];
        p[6 + r] = w->eb_factor[0][c * 3 + r];
      }
      for (int layer = 0; layer < 2; ++layer) {
        float* q = p + 9 + layer * 15;
        for (int e = 0; e < 9; ++e) q[e] = w->eb_matrix[1 + layer][c * 9 + e];
        for (int r = 0; r < 3; ++r) {
          q[9 + r] = w->eb_bias[1 + layer][c * 3 + r];
          q[12 + r] = w->eb_factor[1 + layer][c * 3 + r];
        }
      }
      for (int r = 0; r < 3; ++r) p[39 + r] = w->eb_matrix[3][c * 3 + r];
      p[42] = w->eb_bias[3][c];
    }
    TRY(upload(h, &h->eb_packed, P.data(), P.size()));
  }

  // ---- workspace, sized for the largest admissible batch ----
  const Geom g = make_geom(cfg->max_batch, cfg->max_height, cfg->max_width);
  const size_t B = g.B;
  const size_t ny = B * g.yh * g.yw * C, nz = B * g.zh * g.zw * C;
  TRY(alloc_buf(h, h->xin, B * g.H * g.W * 3));
  TRY(alloc_buf(h, h->xt, B * g.H * g.W * 3));
  TRY(alloc_buf(h, h->xpad, B * g.xHp * g.xWp * 3));
  TRY(alloc_buf(h, h->gpad, B * g.Hp * g.Wp * 3));
  Buf* lat_y[] = {&h->y, &h->my, &h->vy, &h->yt, &h->dyt, &h->g_yt_dist, &h->g_yt_rate};
  for (Buf* b : lat_y) TRY(alloc_buf(h, *b, ny));
  Buf* lat_z[] = {&h->z, &h->mz, &h->vz, &h->zt, &h->dzt, &h->g_zt_hs, &h->g_zt_eb};
  for (Buf* b : lat_z) TRY(alloc_buf(h, *b, nz));
  const size_t n_hs0 = B * (2 * g.zh) * (2 * g.zw) * C;
  const size_t n_hs1 = B * g.hsh * g.hsw * h->C15;
  const size_t n_ms = B * g.hsh * g.hsw * 2 * C;
  TRY(alloc_buf(h, h->hs0, n_hs0)); TRY(alloc_buf(h, h->g_hs0, n_hs0));
  TRY(alloc_buf(h, h->hs1, n_hs1)); TRY(alloc_buf(h, h->g_hs1, n_hs1));
  TRY(alloc_buf(h, h->ms, n_ms)); TRY(alloc_buf(h, h->g_ms, n_ms));
  for (int L = 0; L < 3; ++L) {
    const size_t n = B * (size_t)(g.yh << (L + 1)) * (g.yw << (L + 1)) * C;
    TRY(alloc_buf(h, h->u[L], n)); TRY(alloc_buf(h, h->s[L], n)); TRY(alloc_buf(h, h->v[L], n));
  }
  TRY(alloc_buf(h, h->gA, h->u[2].cap)); TRY(alloc_buf(h, h->gB, h->u[2].cap));
  TRY(alloc_buf(h, h->p3, B * (size_t)(8 * g.yh) * (8 * g.yw) * 80));
  if (cfg->bits_back) {
    Buf* bb2[] = {&h->zml, &h->mzml, &h->vzml, &h->g_zml};
    for (Buf* b : bb2) TRY(alloc_buf(h, *b, 2 * nz));
    TRY(alloc_buf(h, h->jac_lv, nz));
    TRY(alloc_buf(h, h->lrtab2, kMaxIts));
  }
  TRY(alloc_buf(h, h->xq, B * g.H * g.W * 3));
  {
    int hh = g.H, ww = g.W;
    for (int k = 1; k < 5; ++k) {
      hh = (hh + 1) / 2; ww = (ww + 1) / 2;
      Buf a, b;
      TRY(alloc_buf(h, a, B * hh * ww * 3)); TRY(alloc_buf(h, b, B * hh * ww * 3));
      h->msA[k] = a.p; h->msB[k] = b.p;
    }
    void* p = nullptr;
    TRY(dev_alloc(h, &p, sizeof(double) * 5 * B * 6)); h->ms_stats = (double*)p;
    TRY(dev_alloc(h, &p, sizeof(int) * 8)); h->ms_counts = (int*)p;
    if (msssim_init() != 0) return fail(SGA_ERR_HIP);
  }
  TRY(alloc_buf(h, h->scratch, 8 + B * 8));
  {
    void* p = nullptr;
    TRY(dev_alloc(h, &p, 17 * 128));      // k_step_boundary: top counter + 16 group counters, one cache line each
    if (hipMemset(p, 0, 17 * 128) != hipSuccess) return fail(SGA_ERR_HIP);
    h->ticket = (unsigned*)p;
  }
  {
    void* p = nullptr;
    TRY(dev_alloc(h, &p, 256));
    if (hipMemset(p, 0, 256) != hipSuccess) return fail(SGA_ERR_HIP);
    h->zeros = (float*)p;
#ifdef SGA_CLOCK_PROBE
    if (const char* e = PROBE_ENV("SGA_CLOCK_PROBE")) {      // (a PROBE=1 build is not an EXPERIMENTS=1 build: not LAB_ENV)
      h->clk_mode = e[0] == '1' ? 1 : (e[0] == '2' ? 2 : 0);
      if (h->clk_mode) {
        void* q = nullptr;
        TRY(dev_alloc(h, &q, (size_t)40 * 16384 * 6 * sizeof(unsigned long long)));
        h->clk_probe = (unsigned long long*)q;
      }
    }
#endif
  }
  TRY(alloc_buf(h, h->part, (size_t)48 << 20));      // 192 MiB
  TRY(alloc_buf(h, h->partB, (size_t)8 << 20));      // 32 MiB
  h->cur_part = &h->part;
  TRY(alloc_buf(h, h->trace, (size_t)kMaxIts * 4));
  TRY(alloc_buf(h, h->Ttab, kMaxIts)); TRY(alloc_buf(h, h->lrtab, kMaxIts));
  {
    void* p = nullptr;
    TRY(dev_alloc(h, &p, sizeof(ImgSums) * B));
    h->sums = (ImgSums*)p;
    if (hipMemset(p, 0, sizeof(ImgSums) * B) != hipSuccess) return fail(SGA_ERR_HIP);
    TRY(dev_alloc(h, &p, sizeof(StepCtx)));
    h->ctx = (StepCtx*)p;
    if (hipMemset(p, 0, sizeof(StepCtx)) != hipSuccess) return fail(SGA_ERR_HIP);
    TRY(dev_alloc(h, &p, sizeof(int4) * B));
    h->img_ids = (int4*)p;
    h->img_keys.resize(B);
    for (size_t i = 0; i < B; ++i) h->img_keys[i] = int4{(int)i, 0, 0, 0};
    if (hipMemcpy(p, h->img_keys.data(), sizeof(int4) * B, hipMemcpyHostToDevice) != hipSuccess) return fail(SGA_ERR_HIP);
  }
  if (hipDeviceSynchronize() != hipSuccess) return fail(SGA_ERR_HIP);
  const char* env = getenv("SGA_NO_GRAPH");
  h->use_graph = !(env && env[0] == '1');
  env = getenv("SGA_NO_OVERLAP");
  h->overlap = !(env && env[0] == '1');
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // The side stream has normal priority (a high-priority one measured the same throughput).
    int prio = 0;
    if (const char* pr = LAB_ENV("SGA_SIDE_PRIORITY")) prio = atoi(pr);   // experiments only
    const unsigned evflags = hipEventDisableTiming;
    // Experiment: SGA_SIDE_CU_MASK=<hex words, lowest first, comma separated> confines the hyper branch's
    // stream to the CUs whose bits are set (hipExtStreamCreateWithCUMask).  A mask belongs to a stream, and kernel
    // nodes of a replayed hipGraph do not inherit it, so this only acts on eager launches (SGA_NO_GRAPH=1).
    bool masked = false;
    if (const char* cm = LAB_ENV("SGA_SIDE_CU_MASK")) {
      std::vector<uint32_t> words;
      for (const char* q = cm; *q;) {
        char* end = nullptr;
        words.push_back((uint32_t)strtoul(q, &end, 16));
        if (end == q) break;
        q = *end == ',' ? end + 1 : end;
      }
      masked = !words.empty() && hipExtStreamCreateWithCUMask(&h->sB, (uint32_t)words.size(), words.data()) == hipSuccess;
      if (!masked) (void)hipGetLastError();
    }
    h->side_masked = masked;
    // Experiment: SGA_HYBRID=1 runs the same hybrid replay with an unmasked side stream of priority SGA_SIDE_PRIORITY
    // (hardware-queue priorities act on eager streams; graph nodes ignore them)
    if (const char* hy = LAB_ENV("SGA_HYBRID")) h->side_hybrid = hy[0] == '1';
    {
      // LOW priority (numerically greatest): measured at cfg 2, the iteration takes 1773-1778 us on a low-priority launch stream,
      // 1773-1775 on the caller's normal one and 1806-1816 on a HIGH-priority one (the hyper branch's kernels, on the graph's
      // normal-priority internal stream, then lose every dispatch arbitration and become the critical path);
      // profiles/r06_launch_stream_priority.txt.  Laboratory: SGA_LAUNCH_STREAM=caller (rounds 1-5: graphs on the caller's stream) | high
      const char* ls = LAB_ENV("SGA_LAUNCH_STREAM");
      const int want = (ls && strcmp(ls, "high") == 0) ? hi : (lo != 0 ? lo : hi);
      const bool own = !(ls && strcmp(ls, "caller") == 0) && want != 0;      // (a device without stream priorities: the caller's stream)
      if (own && (hipStreamCreateWithPriority(&h->sG, hipStreamNonBlocking, want) != hipSuccess ||
                  hipEventCreateWithFlags(&h->ev_in, evflags) != hipSuccess || hipEventCreateWithFlags(&h->ev_out, evflags) != hipSuccess))
        return fail(SGA_ERR_HIP);
    }
    if ((!masked && hipStreamCreateWithPriority(&h->sB, hipStreamNonBlocking, prio) != hipSuccess) ||
        hipEventCreateWithFlags(&h->ev_fork, evflags) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, evflags) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork_cap, evflags) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join_cap, evflags) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork2_cap, evflags) != hipSuccess)
      return fail(SGA_ERR_HIP);
  }
  env = getenv("SGA_PRECISION");
  h->x2 = cfg->precision == SGA_PRECISION_BF16X2 ||
          (cfg->precision == SGA_PRECISION_DEFAULT && env && strcmp(env, "bf16x2") == 0);
  h->x3 = h->x2 || cfg->precision == SGA_PRECISION_BF16X3 ||
          (cfg->precision == SGA_PRECISION_DEFAULT && env && strcmp(env, "bf16x3") == 0);
  env = LAB_ENV("SGA_DEBUG_DUMP");
  if (env && env[0]) {
    h->dump_path = env;
    void* p = nullptr;
    if (dev_alloc(h, &p, (size_t)kMaxIts * 16 * 8) != SGA_OK) return fail(SGA_ERR_NOMEM);
    h->dump = (unsigned long long*)p;
    h->dump_probe = LAB_ENV("SGA_DEBUG_PROBE") != nullptr;
  }
  env = getenv("SGA_X3_FORK");
  h->x3_fork = !(env && env[0] == '0');
  env = LAB_ENV("SGA_SPLIT256");
  h->split256 = !(env && env[0] == '0');
  env = LAB_ENV("SGA_BM256");
  h->bm256 = !(env && env[0] == '0');
  env = LAB_ENV("SGA_BM256_SPLIT");
  h->bm256_split = h->bm256 && !(env && env[0] == '0');
  env = LAB_ENV("SGA_BM64_MAX");
  if (env) h->bm64_max = atoi(env);
  env = LAB_ENV("SGA_FORK_AT");
  if (env) { h->fork_at = atoi(env); h->fork_auto = false; }
  env = getenv("SGA_FORK_NAME");       // experiments: pin the named fork point(s) ("start" = at the first launch)
  if (env) { h->fork_auto = false; h->fork_name = strcmp(env, "start") == 0 ? nullptr : strdup(env); }
  env = LAB_ENV("SGA_FORK2_NAME");
  if (env) { h->fork_auto = false; h->fork2_name = strcmp(env, "none") == 0 ? nullptr : strdup(env); }
  env = LAB_ENV("SGA_SIDE_LAST");
  if (env) h->side_last = env[0] == '1';
  env = LAB_ENV("SGA_SIDE_TARGET");
  if (env) h->side_target = h->side_target_alone = atoi(env);
  env = LAB_ENV("SGA_FUSED_GDN");
  h->fused_gdn = !(env && env[0] == '0');
  // gdn_fused.hip has instances for C / 32 in {2, 4, 6, 8}; wider models (num_filters = 320, 384, ...) take the generic
  // gather-GEMM GDN with an ordinary split-K reduce (conv_mfma.hip tiles any channel count)
  if (C / 32 != 2 && C / 32 != 4 && C / 32 != 6 && C / 32 != 8) h->fused_gdn = false;
  env = LAB_ENV("SGA_KEEP_U");
  h->keep_u = env && env[0] == '1';
  env = LAB_ENV("SGA_FORK_DELAY_US");
  if (env) h->fork_delay_us = atoi(env);
  env = LAB_ENV("SGA_MAIN_WAVE_PRIO");
  if (env) { h->main_wave_prio = atoi(env); g_deconv3_prio = h->main_wave_prio; }
  env = LAB_ENV("SGA_SIDE_WAVE_PRIO");
  if (env) h->side_wave_prio = atoi(env);
  env = LAB_ENV("SGA_FUSED_POST64");
  h->fused_post64 = !(env && env[0] == '0');
  env = LAB_ENV("SGA_FUSED_POST");
  h->fused_post = !(env && env[0] == '0');
  env = LAB_ENV("SGA_PLAN_TILES");
  h->plan_tiles = !(env && env[0] == '0');
  env = LAB_ENV("SGA_FUSED_BOUNDARY");
  h->fused_boundary = !(env && env[0] == '0');
  env = LAB_ENV("SGA_FUSED_MSE");
  h->fused_mse = !(env && env[0] == '0');
  env = getenv("SGA_NO_SPLITK");
  h->no_splitk = env && env[0] == '1';
  env = getenv("SGA_PROFILE_BY_LAYER");
  h->profile_by_layer = env && env[0] == '1';
#undef TRY
  *out = h;
  return SGA_OK;
}

int sga_destroy(sga_handle* h) {
  if (!h) return SGA_ERR_BAD_ARG;
  (void)hipDeviceSynchronize();
  if (h->clk_mode == 2) {      // measurement: clocks of the LAST replay of the captured step
    std::vector<unsigned long long> t(6 * 16384);
    for (size_t s = 0; s < h->clk_slots.size(); ++s) {
      if (hipMemcpy(t.data(), h->clk_probe + s * (size_t)(6 * 16384), t.size() * sizeof(t[0]), hipMemcpyDeviceToHost) != hipSuccess) break;
      if (strncmp(h->clk_slots[s].name, "gdn:", 4) == 0) {      // raw [grid][8] u64 wall-clock stamps (gdn_fused.hip)
        if (const char* dir = PROBE_ENV("SGA_CLOCK_PROBE_DUMP")) {
          char fn[512];
          snprintf(fn, sizeof(fn), "%s/gdn_slot_%02zu.bin", dir, s);
          if (FILE* f = fopen(fn, "wb")) { fwrite(t.data(), sizeof(t[0]), 8 * (size_t)h->clk_slots[s].grid, f); fclose(f); }
        }
        fprintf(stderr, "clock_probe(graph) slot %zu %s grid %d\n", s, h->clk_slots[s].name, h->clk_slots[s].grid);
        continue;
      }
      double c = 0, w = 0, wmin = 1e30, wmax = 0;
      for (int i = 0; i < h->clk_slots[s].grid; ++i) {
        c += (double)t[6 * i]; w += (double)t[6 * i + 1];
        wmin = std::min(wmin, (double)t[6 * i + 1]); wmax = std::max(wmax, (double)t[6 * i + 1]);
      }
      const int g = h->clk_slots[s].grid;
      if (const char* dir = PROBE_ENV("SGA_CLOCK_PROBE_DUMP")) {      // raw [grid][6] u64: K-loop cycles, K-loop wall ticks, hw_id | xcc_id << 32, wall at K-loop start, at entry, at exit
        char fn[512];
        snprintf(fn, sizeof(fn), "%s/clk_slot_%02zu.bin", dir, s);
        if (FILE* f = fopen(fn, "wb")) { fwrite(t.data(), sizeof(t[0]), 6 * (size_t)g, f); fclose(f); }
      }
      fprintf(stderr, "clock_probe(graph) %s grid %d: %.0f MHz in the K loop; K loop per workgroup mean %.1f us, min %.1f, max %.1f\n",
              h->clk_slots[s].name, g, w > 0 ? 100.0 * c / w : 0.0, w / g / 100.0, wmin / 100.0, wmax / 100.0);
    }
  }
  free_all(h);
  delete h;
  return SGA_OK;
}

int sga_last_error(const sga_handle* h, char* msg, int msg_len) {
  if (!h) return SGA_ERR_BAD_ARG;
  if (msg && msg_len > 0) {
    strncpy(msg, h->last_msg, (size_t)msg_len - 1);
    msg[msg_len - 1] = 0;
  }
  return h->last_hip_error;
}

int sga_latent_shape(const sga_handle* h, int H, int W, int* yh, int* yw, int* zh, int* zw) {
  if (!h || H <= 0 || W <= 0) return SGA_ERR_BAD_ARG;
  const Geom g = make_geom(1, H, W);
  if (yh) *yh = g.yh;
  if (yw) *yw = g.yw;
  if (zh) *zh = g.zh;
  if (zw) *zw = g.zw;
  return SGA_OK;
}

int sga_set_image_ids(sga_handle* h, const int32_t* ids, int n) {
  if (!h || n < 0 || n > h->cfg.max_batch || (n > 0 && !ids)) return SGA_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0) return SGA_ERR_BAD_ARG;
  for (int i = 0; i < h->cfg.max_batch; ++i) h->img_keys[i].x = i < n ? ids[i] : i;
  // the captured step graph reads this array through a fixed device pointer: wait for work in
  // flight, then overwrite the contents (no re-capture needed)
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(h->img_ids, h->img_keys.data(), sizeof(int4) * h->img_keys.size(), hipMemcpyHostToDevice));
  return SGA_OK;
}

int sga_set_image_seeds(sga_handle* h, const uint64_t* seeds, int n) {
  if (!h || n < 0 || n > h->cfg.max_batch || (n > 0 && !seeds)) return SGA_ERR_BAD_ARG;
  for (int i = 0; i < h->cfg.max_batch; ++i) {
    int4& k = h->img_keys[i];
    if (i < n) { k.y = (int)(unsigned)(seeds[i] & 0xffffffffu); k.z = (int)(unsigned)(seeds[i] >> 32); k.w = 1; }
    else { k.y = 0; k.z = 0; k.w = 0; }
  }
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(h->img_ids, h->img_keys.data(), sizeof(int4) * h->img_keys.size(), hipMemcpyHostToDevice));
  return SGA_OK;
}

int sga_encode(sga_handle* h, const float* x, int B, int H, int W, float* y, float* z,
               void* stream) {
  if (!h || !x || !y || !z) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  SGACHK(ensure_borders(h, g, st));
  return encode_impl(h, g, x, y, z, st);
}

int sga_step_grads(sga_handle* h, const float* x, int B, int H, int W, const float* y,
                   const float* z, float T, float lambda, float loss_scale, uint64_t seed,
                   uint32_t it, const float* u_y, const float* u_z, float* gy, float* gz,
                   float* scalars, float* psnr, void* stream) {
  if (!h || !x || !y || !z || !(T > 0.f)) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  SGACHK(ensure_borders(h, g, st));
  HIPCHK(h, launch_set_ctx(h->ctx, (int)it, 0, T, 0.f, lambda, loss_scale, seed, st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (B), st));
  SGACHK(sga_step_core(h, g, x, y, z, u_y, u_z, st));
  const int64_t ny = (int64_t)B * g.yh * g.yw * h->C, nz = (int64_t)B * g.zh * g.zw * h->C;
  if (gy) HIPCHK(h, launch_combine_grad(h->g_yt_dist.p, h->g_yt_rate.p, h->dyt.p, gy, ny, st));
  if (gz) HIPCHK(h, launch_combine_grad(h->g_zt_hs.p, h->g_zt_eb.p, h->dzt.p, gz, nz, st));
  HIPCHK(h, launch_finalize_step(h->sums, h->ctx, B, H, W, scalars, psnr, nullptr, st));
  return SGA_OK;
}

int sga_adam(sga_handle* h, float* p, const float* g, float* m, float* v, int64_t n, int t,
             double lr, double beta1, double beta2, double eps, void* stream) {
  if (!h || !p || !g || !m || !v || n <= 0 || t <= 0) return SGA_ERR_BAD_ARG;
  // adam.py:40-42 in double, cast to float32 at the multiply
  const double lr_t = lr * (std::sqrt(1.0 - std::pow(beta2, t)) / (1.0 - std::pow(beta1, t)));
  HIPCHK(h, launch_adam(p, g, m, v, n, (float)lr_t, beta1, beta2, eps, (hipStream_t)stream));
  return SGA_OK;
}

int sga_eval(sga_handle* h, const float* x, int B, int H, int W, const float* y_hat,
             const float* z_hat, float* metrics, float* x_hat, void* stream) {
  if (!h || !x || !y_hat || !z_hat) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  SGACHK(ensure_borders(h, g, st));
  return eval_impl(h, g, x, y_hat, z_hat, metrics, x_hat, st);
}

// ---- the per-batch loop in three pieces (sga_run = begin + steps(its) + round/eval) -----------
int sga_run_begin(sga_handle* h, const float* x, int B, int H, int W, float lambda, float loss_scale,
                  int its, double lr, double annealing_rate, int t0, double T_ub, uint64_t seed,
                  const float* y0, const float* z0, void* stream) {
  if (!h || !x || its < 0 || its > kMaxIts || (y0 == nullptr) != (z0 == nullptr)) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  const int C = h->C;
  const int64_t ny = (int64_t)B * g.yh * g.yw * C, nz = (int64_t)B * g.zh * g.zw * C;
  h->run_its = -1;
  SGACHK(ensure_borders(h, g, st));
  // x is kept in handle-owned memory so the captured graph does not depend on caller pointers
  HIPCHK(h, launch_copy(h->xin.p, x, (int64_t)((size_t)B * H * W * 3), st));
  if (y0) {
    HIPCHK(h, launch_copy(h->y.p, y0, (int64_t)(ny), st));
    HIPCHK(h, launch_copy(h->z.p, z0, (int64_t)(nz), st));
  } else {
    SGACHK(encode_impl(h, g, h->xin.p, h->y.p, h->z.p, st));
  }
  // fresh optimiser per batch (sga.py:208)
  HIPCHK(h, launch_fill(h->my.p, 0.f, (int64_t)(ny), st));
  HIPCHK(h, launch_fill(h->vy.p, 0.f, (int64_t)(ny), st));
  HIPCHK(h, launch_fill(h->mz.p, 0.f, (int64_t)(nz), st));
  HIPCHK(h, launch_fill(h->vz.p, 0.f, (int64_t)(nz), st));
  if (its > 0) {
    // host tables: utils.py:166-180 ('exp0') and adam.py:40-42, evaluated in double
    h->hT.resize(its); h->hLr.resize(its);
    for (int it = 0; it < its; ++it) {
      double tau = h->sched == SGA_SCHED_EXP ? std::exp(-annealing_rate * (double)it)
                                             : T_ub * std::exp(-annealing_rate * (double)(it - t0));
      tau = std::fmin(std::fmax(tau, 1e-8), T_ub);
      h->hT[it] = (float)tau;
      const int t = it + 1;
      h->hLr[it] = (float)(lr * (std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.9, t))));
    }
    // previous run's tables may still be in use on this stream only; same-stream order suffices
    HIPCHK(h, hipMemcpyAsync(h->Ttab.p, h->hT.data(), its * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->lrtab.p, h->hLr.data(), its * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(h, hipStreamSynchronize(st));   // host vectors may now be reused
    if (h->dump) (void)hipMemsetAsync(h->dump, 0, (size_t)its * 16 * 8, st);
  }
  h->run_B = B; h->run_H = H; h->run_W = W; h->run_its = its; h->run_it = 0;
  h->run_lambda = lambda; h->run_loss_scale = loss_scale; h->run_seed = seed;
  return SGA_OK;
}

int sga_run_steps(sga_handle* h, int n, void* stream) {
  if (!h || n < 0 || h->run_its < 0) return SGA_ERR_BAD_ARG;
  LaunchStream ls(h, stream);
  hipStream_t st = ls.run;
  const int B = h->run_B, H = h->run_H, W = h->run_W, its = h->run_its;
  if (n > its - h->run_it) n = its - h->run_it;
  if (n == 0) return SGA_OK;
  const Geom g = make_geom(B, H, W);
  const int C = h->C;
  const int64_t ny = (int64_t)B * g.yh * g.yw * C, nz = (int64_t)B * g.zh * g.zw * C;
  // a one-shot call at ANOTHER size between two calls of an open run (sga_base_compress, sga_eval) leaves the zero borders of the
  // gradient image valid for its geometry, not this one: re-check (one compare when nothing changed)
  SGACHK(ensure_borders(h, g, st));
  // other entry points (sga_step_grads, sga_eval) may have used the step context and the sums
  // the step context of the first of these iterations; k_finalize_step advances it from then on
  HIPCHK(h, launch_set_ctx(h->ctx, h->run_it, its, h->hT[h->run_it], h->hLr[h->run_it], h->run_lambda,
                           h->run_loss_scale, h->run_seed, st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (B), st));

  // One iteration = forward/backward on the relaxed latents that are already in the workspace, then ONE launch
  // for Adam, the relaxation for the next iteration and the scalars / context advance (k_step_boundary).  The
  // relaxation of the first iteration of this call is launched here (counter-based noise: re-drawing it after
  // another entry point used the workspace gives the same values).
  const bool fb = h->fused_boundary && !(h->dump);
  if (fb)
    HIPCHK(h, launch_sample_yz(h->y.p, h->yt.p, h->dyt.p, ny, h->z.p, h->zt.p, h->dzt.p, nz, h->ctx, h->relax,
                               h->img_ids, B, st));
  auto enqueue_step = [&](hipStream_t s) -> int {
    if (fb) {
      SGACHK(rd_forward_backward(h, g, h->xin.p, true, s));
      HIPCHK(h, launch_step_boundary(h->y.p, h->g_yt_dist.p, h->g_yt_rate.p, h->dyt.p, h->my.p, h->vy.p, h->yt.p, ny,
                                     h->z.p, h->g_zt_hs.p, h->g_zt_eb.p, h->dzt.p, h->mz.p, h->vz.p, h->zt.p, nz,
                                     h->ctx, h->relax, h->img_ids, B, H, W, h->sums, h->trace.p, h->Ttab.p,
                                     h->lrtab.p, h->ticket, s));
      return SGA_OK;
    }
    SGACHK(sga_step_core(h, g, h->xin.p, h->y.p, h->z.p, nullptr, nullptr, s));
    HIPCHK(h, launch_adam_latent_yz(h->y.p, h->g_yt_dist.p, h->g_yt_rate.p, h->dyt.p, h->my.p, h->vy.p, ny,
                                    h->z.p, h->g_zt_hs.p, h->g_zt_eb.p, h->dzt.p, h->mz.p, h->vz.p, nz, h->ctx, s));
    HIPCHK(h, launch_finalize_step(h->sums, h->ctx, B, H, W, nullptr, nullptr, h->trace.p, s, h->Ttab.p, h->lrtab.p));
    if (h->dump) {      // (laboratory, SGA_DEBUG_DUMP: per-iteration checksums of the step's buffers; the row is chosen on the device,
                        //  so the launches replay with the graph)
      const size_t B_ = (size_t)B;
      const int64_t n_ms = (int64_t)B_ * g.hsh * g.hsw * 2 * C, n_hs1 = (int64_t)B_ * g.hsh * g.hsw * h->C15,
                    n_hs0 = (int64_t)B_ * (2 * g.zh) * (2 * g.zw) * C;
      const float* bufs[16] = {h->yt.p, h->zt.p, h->g_yt_dist.p, h->g_yt_rate.p, h->g_zt_hs.p, h->g_zt_eb.p, h->ms.p, h->g_ms.p,
                               h->hs1.p, h->g_hs1.p, h->hs0.p, h->g_hs0.p, h->y.p, h->z.p, h->v[2].p, h->gB.p};
      const int64_t ns[16] = {ny, nz, ny, ny, nz, nz, n_ms, n_ms, n_hs1, n_hs1, n_hs0, n_hs0, ny, nz, ny * 64, ny * 4};
      for (int k = 0; k < (h->dump_probe ? 13 : 16); ++k) HIPCHK(h, launch_checksum(bufs[k], ns[k], h->dump + k, s, h->ctx));
    }
    return SGA_OK;
  };

  // ---- hybrid replay (side stream confined to a CU subset) ------------------------------------------------------
  // A CU mask is a property of a stream's hardware queue; the kernel nodes of a replayed hipGraph run on the
  // runtime's own streams and do not inherit it.  So only the MAIN chain (the synthesis branch, 20 launches) is a
  // captured graph; the hyper branch is launched eagerly on the masked stream every iteration, and the two meet
  // at the boundary kernel through a pair of events:
  //   sB: wait(ev_iter: relaxed latents of this iteration exist) . hyper branch . record(ev_side)
  //   st: graph(main chain) . wait(ev_side) . boundary kernel (Adam + next relaxation) . record(ev_iter)
  // Same kernels, same arguments, same order per stream as the two-stream graph: results are bit-identical.
  if ((h->side_masked || h->side_hybrid) && h->use_graph && !h->profiling && fb && h->overlap && !h->x3) {
    if (!h->graph_main || h->gmain_B != B || h->gmain_H != H || h->gmain_W != W) {
      if (h->graph_main) { (void)hipGraphExecDestroy(h->graph_main); h->graph_main = nullptr; }
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        h->branch_only = 1;
        const int rc = rd_forward_backward(h, g, h->xin.p, true, st);
        h->branch_only = 0;
        const hipError_t ec = hipStreamEndCapture(st, &graph);
        if (rc == SGA_OK && ec == hipSuccess && graph &&
            hipGraphInstantiate(&h->graph_main, graph, nullptr, nullptr, 0) == hipSuccess) {
          h->gmain_B = B; h->gmain_H = H; h->gmain_W = W;
        } else {
          h->graph_main = nullptr;
        }
        if (graph) (void)hipGraphDestroy(graph);
      }
      (void)hipGetLastError();
    }
    if (h->graph_main) {
      HIPCHK(h, hipEventRecord(h->ev_fork, st));                  // ev_iter: the relaxed latents of iteration run_it exist
      for (int k = 0; k < n; ++k) {
        static const bool skip_side = LAB_ENV("SGA_SKIP_SIDE") != nullptr;      // timing experiment only: results are wrong
        HIPCHK(h, hipStreamWaitEvent(h->sB, h->ev_fork, 0));
        h->branch_only = 2;
        const int rs = skip_side ? SGA_OK : rd_forward_backward(h, g, h->xin.p, true, h->sB);
        h->branch_only = 0;
        SGACHK(rs);
        HIPCHK(h, hipEventRecord(h->ev_join, h->sB));
        HIPCHK(h, hipGraphLaunch(h->graph_main, st));
        HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
        HIPCHK(h, launch_step_boundary(h->y.p, h->g_yt_dist.p, h->g_yt_rate.p, h->dyt.p, h->my.p, h->vy.p, h->yt.p, ny,
                                       h->z.p, h->g_zt_hs.p, h->g_zt_eb.p, h->dzt.p, h->mz.p, h->vz.p, h->zt.p, nz,
                                       h->ctx, h->relax, h->img_ids, B, H, W, h->sums, h->trace.p, h->Ttab.p,
                                       h->lrtab.p, h->ticket, st));
        HIPCHK(h, hipEventRecord(h->ev_fork, st));
      }
      h->run_it += n;
      return SGA_OK;
    }
  }

  bool graphed = false;
  hipGraphExec_t step_graph = nullptr;
  int tuned_done = 0;              // iterations of this call already run by the fork-point candidates
  if (h->use_graph && !h->profiling) {
    auto capture = [&]() -> hipGraphExec_t {
      hipGraphExec_t ex = nullptr;
      hipGraph_t graph = nullptr;
      ++h->n_captures;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = enqueue_step(st);
        const hipError_t ec = hipStreamEndCapture(st, &graph);
        if (!(rc == SGA_OK && ec == hipSuccess && graph && hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess))
          ex = nullptr;
        if (graph) (void)hipGraphDestroy(graph);
      }
      (void)hipGetLastError();
      return ex;
    };
    // Where the hyper branch is forked does not change a single bit of the result (same kernels, same arguments, other
    // graph edges), but it decides which main-chain launches its kernels fall on: +-50 us per iteration at cfg 2, and the
    // best point differs between geometries (cfg 2, B = 8: before gs3.fwd 1780 us, at the start 1840 us; one Kodak image:
    // 1.59 against 1.52 ms).  So a call with enough iterations left times the three candidates once per geometry -- on
    // the run's own iterations: 8 replays of each candidate graph, which count -- and keeps the fastest graph
    // (DESIGN.md 3.7).  SGA_FORK_AT=<n> pins the point instead.
    const bool tune = h->fork_auto && fb && h->overlap && (!h->x3 || h->x3_fork) && !h->gprof && n >= 100;
    const sga_handle::GraphKey key{0, B, H, W, h->relax, h->scale_bound, h->gprof ? 1 : 0};
    sga_handle::GraphEntry* e = find_graph(h, key);
    if (e && tune && !e->tuned) {      // built untimed by a short call: this call has the iterations to time the candidates
      erase_graph(h, e, st);
      e = nullptr;
    }
    if (!e) {
      hipGraphExec_t ex = nullptr;
      bool tuned = false;
      if (!tune) {
        if (h->fork_auto) h->fork_name = tuned_fork_for(h, 0, B, H, W, &tuned);
        ex = capture();
      } else {
        SGACHK(timed_fork_choice(h, st, B, H, W, capture, &ex, &tuned_done));
        tuned = ex != nullptr;
        if (!ex) { h->fork_name = nullptr; ex = capture(); }
      }
      if (ex) e = insert_graph(h, key, ex, tuned, h->fork_name, h->gprof_in_graph, st);
    } else {
      if (h->fork_auto) h->fork_name = e->fork_name;
      if (h->gprof) h->gprof_in_graph = e->stamp_in_graph;
    }
    h->graph_tuned = e && e->tuned;
    step_graph = e ? e->exec : nullptr;
    graphed = step_graph != nullptr;
  }
  for (int k = tuned_done; k < n; ++k) {
    h->dbg_it = h->run_it + k;
    if (graphed) HIPCHK(h, hipGraphLaunch(step_graph, st));
    else SGACHK(enqueue_step(st));
    if (graphed && h->gprof && h->gprof_in_graph) {      // measurement: the stamped kernel's span in this replay
      unsigned long long t[2] = {0, 0};
      const unsigned long long reset[2] = {~0ull, 0ull};
      HIPCHK(h, hipStreamSynchronize(st));
      HIPCHK(h, hipMemcpy(t, h->gstamp, sizeof(t), hipMemcpyDeviceToHost));
      HIPCHK(h, hipMemcpy(h->gstamp, reset, sizeof(reset), hipMemcpyHostToDevice));
      if (t[1] > t[0]) { h->gprof_ms += (double)(t[1] - t[0]) * 1e-5; h->gprof_flops += h->gprof_flops_launch; h->gprof_n += 1; }
    }
  }
  h->dbg_it = -1;
  h->run_it += n;
  if (h->dump && h->run_it == its) {
    std::vector<unsigned long long> host((size_t)its * 16);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host.data(), h->dump, host.size() * 8, hipMemcpyDeviceToHost);
    char fn[512];
    snprintf(fn, sizeof(fn), "%s.%d", h->dump_path, h->dump_run);
    if (FILE* f = fopen(fn, "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    if (LAB_ENV("SGA_DEBUG_DUMP_BUFS")) {      // raw z_tilde, its rate gradient and the Jacobian as the last iteration left them
      const float* bufs[3] = {h->zt.p, h->g_zt_eb.p, h->g_zt_hs.p};
      const char* tags[3] = {"zt", "gzeb", "gzhs"};
      std::vector<float> hb((size_t)nz);
      for (int k = 0; k < 3; ++k) {
        (void)hipMemcpy(hb.data(), bufs[k], hb.size() * 4, hipMemcpyDeviceToHost);
        snprintf(fn, sizeof(fn), "%s.%s.%d", h->dump_path, tags[k], h->dump_run);
        if (FILE* f = fopen(fn, "wb")) { fwrite(hb.data(), 4, hb.size(), f); fclose(f); }
      }
    }
    h->dump_run++;
  }
  return SGA_OK;
}

int sga_run_state(sga_handle* h, int set, float* y, float* z, float* trace, void* stream) {
  if (!h || h->run_its < 0) return SGA_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(h->run_B, h->run_H, h->run_W);
  const int64_t ny = (int64_t)h->run_B * g.yh * g.yw * h->C, nz = (int64_t)h->run_B * g.zh * g.zw * h->C;
  if (set) {
    if (!y || !z) return SGA_ERR_BAD_ARG;
    HIPCHK(h, launch_copy(h->y.p, y, ny, st));
    HIPCHK(h, launch_copy(h->z.p, z, nz, st));
    return SGA_OK;
  }
  if (y) HIPCHK(h, launch_copy(y, h->y.p, ny, st));
  if (z) HIPCHK(h, launch_copy(z, h->z.p, nz, st));
  if (trace && h->run_it > 0) HIPCHK(h, launch_copy(trace, h->trace.p, (int64_t)h->run_it * 4, st));
  return SGA_OK;
}

int sga_run(sga_handle* h, const float* x, int B, int H, int W, float lambda, float loss_scale,
            int its, double lr, double annealing_rate, int t0, double T_ub, uint64_t seed,
            const float* y0, const float* z0, float* y_hat, float* z_hat, float* metrics,
            float* trace, void* stream) {
  SGACHK(sga_run_begin(h, x, B, H, W, lambda, loss_scale, its, lr, annealing_rate, t0, T_ub, seed, y0, z0, stream));
  SGACHK(sga_run_steps(h, its, stream));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  const int64_t ny = (int64_t)B * g.yh * g.yw * h->C, nz = (int64_t)B * g.zh * g.zw * h->C;
  if (trace && its > 0) HIPCHK(h, launch_copy(trace, h->trace.p, (int64_t)((size_t)its * 4), st));
  // sga.py:240-247: round (half-to-even) and evaluate with the latents fed directly
  float* yh_dst = y_hat ? y_hat : h->g_yt_dist.p;
  float* zh_dst = z_hat ? z_hat : h->g_zt_hs.p;
  HIPCHK(h, launch_round(h->y.p, yh_dst, ny, st));
  HIPCHK(h, launch_round(h->z.p, zh_dst, nz, st));
  if (metrics) SGACHK(eval_impl(h, g, h->xin.p, yh_dst, zh_dst, metrics, nullptr, st));
  return SGA_OK;
}

// tfc `_quantize(.., 'dequantize')` with centring, as map.py:83,101 uses it: z_hat = round(z - median)
// + median; (mu, .) = h_s(z) with z AS GIVEN (map.py computes mu from the placeholder); y_hat =
// round(y - mu) + mu.
int sga_quantize_centered(sga_handle* h, const float* y, const float* z, int B, int H, int W,
                          const float* medians, float* y_hat, float* z_hat, void* stream) {
  if (!h || !y || !z || !y_hat || !z_hat) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  const int C = h->C;
  const int64_t nz = (int64_t)B * g.zh * g.zw * C;
  h->cur_part = &h->part;
  SGACHK(deconv_fwd(h, h->hs_f[0], h->hs_bias[0], z, B, g.zh, g.zw, h->hs0.p, EPI_BIAS_RELU, st));
  SGACHK(deconv_fwd(h, h->hs_f[1], h->hs_bias[1], h->hs0.p, B, 2 * g.zh, 2 * g.zw, h->hs1.p, EPI_BIAS_RELU, st));
  SGACHK(conv3(h, h->hs_f[2], h->hs_bias[2], h->hs1.p, h->C15, B, g.hsh, g.hsw, h->ms.p, true, EPI_BIAS, nullptr, st));
  HIPCHK(h, launch_round_centered(y, h->ms.p, B, g.yh, g.yw, g.hsh, g.hsw, C, y_hat, st));
  HIPCHK(h, launch_round_median(z, medians, nz, C, z_hat, st));
  return SGA_OK;
}

int sga_base_compress_bound(sga_handle* h, const float* x, int B, int H, int W, const float* medians, float scale_bound,
                            float* y_hat, float* z_hat, float* metrics, void* stream) {
  if (!h || !(scale_bound >= 0.f) || !(scale_bound < 1e30f)) return SGA_ERR_BAD_ARG;
  // the call launches eagerly (no captured graph is involved) and the bound crosses to the device BY VALUE as an argument of
  // this call's k_gaussian launch: the host field is read while the launches are enqueued and restored before returning, so
  // nothing is synchronised, no cached step graph is dropped and no launch in flight sees the temporary value
  const float keep = h->scale_bound;
  h->scale_bound = scale_bound;
  const int rc = sga_base_compress(h, x, B, H, W, medians, y_hat, z_hat, metrics, stream);
  h->scale_bound = keep;
  return rc;
}

int sga_base_compress(sga_handle* h, const float* x, int B, int H, int W, const float* medians,
                      float* y_hat, float* z_hat, float* metrics, void* stream) {
  if (!h || !x || !y_hat || !z_hat) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  const int C = h->C;
  const int64_t nz = (int64_t)B * g.zh * g.zw * C;
  SGACHK(ensure_borders(h, g, st));
  // y, z go to scratch -- two gradient buffers of the same sizes that every SGA iteration rewrites before it reads them --
  // NOT to h->y / h->z: those are the live latents of a run opened by sga_run_begin, which a one-shot encode between two
  // sga_run_steps calls must leave alone (tests/test_gpu_configs.py::test_base_compress_inside_an_open_run)
  float* const ybuf = h->g_yt_dist.p; float* const zbuf = h->g_zt_eb.p;
  SGACHK(encode_impl(h, g, x, ybuf, zbuf, st));
  HIPCHK(h, launch_round_median(zbuf, medians, nz, C, z_hat, st));             // mbt2018.py:69
  // mu = h_s(z_hat)[..., :C] (mbt2018.py:70-76), y_hat = round(y - mu) + mu (mbt2018.py:80)
  SGACHK(deconv_fwd(h, h->hs_f[0], h->hs_bias[0], z_hat, B, g.zh, g.zw, h->hs0.p, EPI_BIAS_RELU, st));
  SGACHK(deconv_fwd(h, h->hs_f[1], h->hs_bias[1], h->hs0.p, B, 2 * g.zh, 2 * g.zw, h->hs1.p, EPI_BIAS_RELU, st));
  SGACHK(conv3(h, h->hs_f[2], h->hs_bias[2], h->hs1.p, h->C15, B, g.hsh, g.hsw, h->ms.p, true, EPI_BIAS, nullptr, st));
  HIPCHK(h, launch_round_centered(ybuf, h->ms.p, B, g.yh, g.yw, g.hsh, g.hsw, C, y_hat, st));
  if (metrics) SGACHK(eval_impl(h, g, x, y_hat, z_hat, metrics, nullptr, st));
  return SGA_OK;
}

// ---- per-layer operator surface ---------------------------------------------------------------
int sga_op_layer_fwd(sga_handle* h, int layer, const float* in, int B, int Hin, int Win,
                     float* out, void* stream) {
  if (!h || !in || !out || B <= 0 || Hin <= 0 || Win <= 0) return SGA_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int C = h->C;
  const int Ho2 = cdiv(Hin, 2), Wo2 = cdiv(Win, 2);
  const size_t npx_in = (size_t)B * Hin * Win;
  switch (layer) {
    case SGA_GA0: {
      const int Hp = 2 * Ho2 + 4, Wp = 2 * Wo2 + 4;
      if ((size_t)B * Hp * Wp * 3 > h->xpad.cap || (size_t)B * Ho2 * Wo2 * C > h->u[2].cap) return SGA_ERR_BAD_SHAPE;
      HIPCHK(h, launch_fill(h->xpad.p, 0.f, (int64_t)(h->xpad.cap), st));
      h->borders_valid = false;
      HIPCHK(h, launch_pad_image(in, B, Hin, Win, Hp, Wp, h->xpad.p, st));
      SGACHK(conv_smallc(h, h->ga_f[0], h->ga_bias[0], h->xpad.p, B, Hp, Wp, Ho2, Wo2, h->u[2].p, st));
      return gdn_fwd(h, h->ga_gdn[0], h->ga_beta[0], h->u[2].p, B, Ho2, Wo2, nullptr, out, false, st);
    }
    case SGA_GA1: case SGA_GA2: {
      const int i = layer - SGA_GA0;
      if ((size_t)B * Ho2 * Wo2 * C > h->u[2].cap) return SGA_ERR_BAD_SHAPE;
      SGACHK(conv5s2(h, h->ga_f[i], h->ga_bias[i], in, B, Hin, Win, Ho2, Wo2, h->u[2].p, EPI_BIAS, nullptr, st));
      return gdn_fwd(h, h->ga_gdn[i], h->ga_beta[i], h->u[2].p, B, Ho2, Wo2, nullptr, out, false, st);
    }
    case SGA_GA3:
      return conv5s2(h, h->ga_f[3], h->ga_bias[3], in, B, Hin, Win, Ho2, Wo2, out, EPI_BIAS, nullptr, st);
    case SGA_GS0: case SGA_GS1: case SGA_GS2: {
      const int i = layer - SGA_GS0;
      if (npx_in * 4 * C > h->u[2].cap) return SGA_ERR_BAD_SHAPE;
      SGACHK(deconv_fwd(h, h->gs_f[i], h->gs_bias[i], in, B, Hin, Win, h->u[2].p, EPI_BIAS, st));
      return gdn_fwd(h, h->gs_gdn_f[i], h->gs_beta[i], h->u[2].p, B, 2 * Hin, 2 * Win, h->s[2].p, out, true, st);
    }
    case SGA_GS3:
      return deconv_to3(h, h->gs_f[3], h->gs_bias[3], in, B, Hin, Win, 2 * Hin, 2 * Win, out, st);
    case SGA_HA0:
      return conv3(h, h->ha_f[0], h->ha_bias[0], in, C, B, Hin, Win, out, false, EPI_BIAS_RELU, nullptr, st);
    case SGA_HA1:
      return conv5s2(h, h->ha_f[1], h->ha_bias[1], in, B, Hin, Win, Ho2, Wo2, out, EPI_BIAS_RELU, nullptr, st);
    case SGA_HA2:
      return conv5s2(h, h->ha_f[2], nullptr, in, B, Hin, Win, Ho2, Wo2, out, EPI_BIAS, nullptr, st);
    case SGA_HS0:
      return deconv_fwd(h, h->hs_f[0], h->hs_bias[0], in, B, Hin, Win, out, EPI_BIAS_RELU, st);
    case SGA_HS1:
      return deconv_fwd(h, h->hs_f[1], h->hs_bias[1], in, B, Hin, Win, out, EPI_BIAS_RELU, st);
    case SGA_HS2:
      return conv3(h, h->hs_f[2], h->hs_bias[2], in, h->C15, B, Hin, Win, out, true, EPI_BIAS, nullptr, st);
  }
  return SGA_ERR_UNSUPPORTED;
}

int sga_op_layer_bwd(sga_handle* h, int layer, const float* in, const float* g_out, int B,
                     int Hin, int Win, float* g_in, void* stream) {
  if (!h || !in || !g_out || !g_in || B <= 0 || Hin <= 0 || Win <= 0) return SGA_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int C = h->C;
  const size_t npx_in = (size_t)B * Hin * Win;
  switch (layer) {
    case SGA_GS0: case SGA_GS1: case SGA_GS2: {
      const int i = layer - SGA_GS0;
      if (npx_in * 4 * C > h->u[2].cap) return SGA_ERR_BAD_SHAPE;
      // recompute u and s = sqrt(n), then IGDN backward, then the transposed conv's data-gradient
      SGACHK(deconv_fwd(h, h->gs_f[i], h->gs_bias[i], in, B, Hin, Win, h->u[2].p, EPI_BIAS, st));
      SGACHK(gdn_fwd(h, h->gs_gdn_f[i], h->gs_beta[i], h->u[2].p, B, 2 * Hin, 2 * Win, h->s[2].p, h->v[2].p, true, st));
      SGACHK(igdn_bwd(h, h->gs_gdn_b[i], g_out, h->u[2].p, h->s[2].p, B, 2 * Hin, 2 * Win, h->gB.p, st));
      return conv5s2(h, h->gs_b[i], nullptr, h->gB.p, B, 2 * Hin, 2 * Win, Hin, Win, g_in, EPI_BIAS, nullptr, st);
    }
    case SGA_GS3: {
      const int Hp = 2 * Hin + 4, Wp = 2 * Win + 4;
      if ((size_t)B * Hp * Wp * 3 > h->gpad.cap) return SGA_ERR_BAD_SHAPE;
      HIPCHK(h, launch_fill(h->gpad.p, 0.f, (int64_t)(h->gpad.cap), st));
      h->borders_valid = false;
      HIPCHK(h, launch_pad_image(g_out, B, 2 * Hin, 2 * Win, Hp, Wp, h->gpad.p, st));
      return conv_smallc(h, h->gs_b[3], nullptr, h->gpad.p, B, Hp, Wp, Hin, Win, g_in, st);
    }
    case SGA_HS0: case SGA_HS1: {
      // g_out is the gradient w.r.t. the ReLU output: mask with the recomputed activation
      const int i = layer - SGA_HS0;
      const int Co = (i == 0) ? C : h->C15;
      if (npx_in * 4 * Co > h->g_hs1.cap || npx_in * 4 * Co > h->hs1.cap) return SGA_ERR_BAD_SHAPE;
      SGACHK(deconv_fwd(h, h->hs_f[i], h->hs_bias[i], in, B, Hin, Win, h->hs1.p, EPI_BIAS_RELU, st));
      HIPCHK(h, launch_relu_mask(g_out, h->hs1.p, h->g_hs1.p, (int64_t)npx_in * 4 * Co, st));
      return conv5s2(h, h->hs_b[i], nullptr, h->g_hs1.p, B, 2 * Hin, 2 * Win, Hin, Win, g_in, EPI_BIAS, nullptr, st);
    }
    case SGA_HS2:
      return conv3(h, h->hs_b[2], nullptr, g_out, 2 * C, B, Hin, Win, g_in, false, EPI_BIAS, nullptr, st);
  }
  return SGA_ERR_UNSUPPORTED;
}

int sga_op_sample(sga_handle* h, const float* v, const float* u, int64_t n, float T, float* v_tilde,
                  float* dvt_dv, void* stream) {
  if (!h || !v || !u || !v_tilde || n <= 0 || !(T > 0.f)) return SGA_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, launch_set_ctx(h->ctx, 0, 0, T, 0.f, 0.f, 1.f, 0, st));
  HIPCHK(h, launch_sample(v, u, h->ctx, 0, v_tilde, dvt_dv, n, st));
  return SGA_OK;
}

int sga_op_factorized_likelihood(sga_handle* h, const float* v, int64_t n_pix, float* p,
                                 float* dp_dv, void* stream) {
  if (!h || !v || n_pix <= 0 || n_pix * h->C > 0x7fffffffLL) return SGA_ERR_BAD_ARG;
  HIPCHK(h, launch_factorized(v, h->eb_packed, nullptr, 1, (int)n_pix, h->C, 1.f, nullptr, nullptr, p,
                              dp_dv, (hipStream_t)stream));
  return SGA_OK;
}

int sga_op_gaussian_likelihood_bound(sga_handle* h, const float* y, const float* mu, const float* sigma_raw, int64_t n,
                                     float scale_bound, float* p, float* dp_dy, float* dp_dmu, float* dp_dsraw,
                                     void* stream) {
  if (!h || !y || !mu || !sigma_raw || n <= 0 || !(scale_bound >= 0.f) || !(scale_bound < 1e30f)) return SGA_ERR_BAD_ARG;
  HIPCHK(h, launch_gaussian_op(y, mu, sigma_raw, n, scale_bound, p, dp_dy, dp_dmu, dp_dsraw, (hipStream_t)stream));
  return SGA_OK;
}

int sga_op_gaussian_likelihood(sga_handle* h, const float* y, const float* mu,
                               const float* sigma_raw, int64_t n, float* p, float* dp_dy,
                               float* dp_dmu, float* dp_dsraw, void* stream) {
  if (!h) return SGA_ERR_BAD_ARG;
  return sga_op_gaussian_likelihood_bound(h, y, mu, sigma_raw, n, h->scale_bound, p, dp_dy, dp_dmu, dp_dsraw, stream);
}

// The rate half of the graph with fed intermediates (the reference can feed y_tilde / z_tilde and
// fetch any tensor, sga.py:219-225): exactly the kernels the SGA step launches for sga.py:100-104
// and :126-146 -- factorized mass of z_tilde and box-Gaussian mass of y_tilde under (mu |
// sigma_raw) = ms, both with lower_bound and its gradient rule (math_ops.py:63-76) -- and their
// gradients of train_bpp = loss_scale * sum_b (y_bpp + z_bpp).
int sga_op_rate_terms(sga_handle* h, const float* y_tilde, const float* z_tilde, const float* ms, int B,
                      int H, int W, float loss_scale, float* g_yt, float* g_ms, float* g_zt,
                      float* metrics, void* stream) {
  if (!h || !y_tilde || !z_tilde || !ms) return SGA_ERR_BAD_ARG;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  const float il = inv_ln2_hw(g);
  HIPCHK(h, launch_set_ctx(h->ctx, 0, 0, 1.f, 0.f, 0.f, loss_scale, 0, st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * B, st));
  HIPCHK(h, launch_factorized(z_tilde, h->eb_packed, h->ctx, B, g.zh * g.zw, h->C, il, h->sums, g_zt, nullptr,
                              nullptr, st));
  HIPCHK(h, launch_gaussian(y_tilde, ms, h->ctx, B, g.yh, g.yw, g.hsh, g.hsw, h->C, il, h->scale_bound, h->sums, g_yt,
                            g_ms, st));
  if (metrics) {   // [B][7] in the order of sga.py:183; only est_bpp / est_y_bpp / est_z_bpp are meaningful
    HIPCHK(h, launch_finalize_eval(h->sums, B, H, W, metrics, st));
  } else {
    HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * B, st));
  }
  return SGA_OK;
}

// The bound is a launch argument of k_gaussian, i.e. baked into a captured step graph: it is part of the graph cache's KEY.
int sga_set_scale_bound(sga_handle* h, float scale_bound) {
  if (!h || !(scale_bound >= 0.f) || !(scale_bound < 1e30f)) return SGA_ERR_BAD_ARG;
  // the bound is part of the graph cache's key: the graphs captured under the other value stay cached, nothing is synchronised
  h->scale_bound = scale_bound;
  return SGA_OK;
}

int sga_set_relaxation(sga_handle* h, int relaxation, int schedule) {
  if (!h || relaxation < 0 || relaxation > SGA_RELAX_NONE || schedule < 0 || schedule > SGA_SCHED_EXP)
    return SGA_ERR_BAD_ARG;
  h->relax = relaxation;
  h->sched = schedule;
  return SGA_OK;
}

int sga_profile_begin(sga_handle* h) {
  if (!h) return SGA_ERR_BAD_ARG;
  for (auto& r : h->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  h->prof.clear();
  h->profiling = true;
  return SGA_OK;
}

// One kernel symbol timed inside the hipGraph replay that sga_run / sga_run_steps time (the roofline of what is timed).
int sga_profile_graph_begin(sga_handle* h, const char* kernel_name) {
  if (!h || !kernel_name || !kernel_name[0]) return SGA_ERR_BAD_ARG;
  if (!h->gstamp) {
    void* p = nullptr;
    SGACHK(dev_alloc(h, &p, 256));
    h->gstamp = (unsigned long long*)p;
  }
  HIPCHK(h, hipDeviceSynchronize());
  {
    const unsigned long long reset[2] = {~0ull, 0ull};
    HIPCHK(h, hipMemcpy(h->gstamp, reset, sizeof(reset), hipMemcpyHostToDevice));
  }
  for (size_t i = h->graphs.size(); i-- > 0;)      // an older stamped graph may carry another kernel's stamp: capture afresh
    if (h->graphs[i].key.stamped) { drop_graph(h, h->graphs[i].exec); h->graphs.erase(h->graphs.begin() + (long)i); }
  strncpy(h->gprof_name, kernel_name, sizeof(h->gprof_name) - 1);
  h->gprof_name[sizeof(h->gprof_name) - 1] = 0;
  h->gprof = true; h->gprof_in_graph = false;
  h->gprof_ms = 0.0; h->gprof_flops = 0.0; h->gprof_n = 0;
  return SGA_OK;
}

int sga_get_fork_point(const sga_handle* h, char* name, int name_len) {
  if (!h || !name || name_len <= 0) return SGA_ERR_BAD_ARG;
  const char* s = (h->graph_tuned || h->bb_graph_tuned) ? (h->fork_name ? h->fork_name : "start") : (h->fork_auto ? "untimed" : "pinned");
  strncpy(name, s, (size_t)name_len - 1);
  name[name_len - 1] = 0;
  return SGA_OK;
}

int sga_debug_counter(const sga_handle* h, int which, long long* value) {
  if (!h || !value) return SGA_ERR_BAD_ARG;
  switch (which) {
    case SGA_COUNTER_GRAPH_CAPTURES: *value = h->n_captures; return SGA_OK;
    case SGA_COUNTER_GRAPHS_CACHED: *value = (long long)h->graphs.size(); return SGA_OK;
    case SGA_COUNTER_GRAPH_EVICTIONS: *value = h->n_graph_evictions; return SGA_OK;
    case SGA_COUNTER_GRAPHS_DROPPED: *value = h->n_dropped; return SGA_OK;
  }
  return SGA_ERR_BAD_ARG;
}

int sga_profile_graph_end(sga_handle* h, sga_kernel_stat* out) {
  if (!h || !out) return SGA_ERR_BAD_ARG;
  HIPCHK(h, hipDeviceSynchronize());
  memset(out, 0, sizeof(*out));
  strncpy(out->name, h->gprof_name, sizeof(out->name) - 1);
  out->launches = h->gprof_n; out->ms_total = h->gprof_ms; out->flops_total = h->gprof_flops;
  h->gprof = false; h->gprof_in_graph = false;
  for (size_t i = h->graphs.size(); i-- > 0;)      // the production graphs are still cached under their own keys
    if (h->graphs[i].key.stamped) { drop_graph(h, h->graphs[i].exec); h->graphs.erase(h->graphs.begin() + (long)i); }
  return SGA_OK;
}

int sga_profile_end(sga_handle* h, sga_kernel_stat* out, int max_out, int* n_out) {
  if (!h || !n_out) return SGA_ERR_BAD_ARG;
  h->profiling = false;
  HIPCHK(h, hipDeviceSynchronize());
  std::vector<sga_kernel_stat> agg;
  for (auto& r : h->prof) {
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, r.a, r.b));
    size_t k = 0;
    for (; k < agg.size(); ++k)
      if (strncmp(agg[k].name, r.name, sizeof(agg[k].name)) == 0) break;
    if (k == agg.size()) {
      sga_kernel_stat s0;
      memset(&s0, 0, sizeof(s0));
      strncpy(s0.name, r.name, sizeof(s0.name) - 1);
      agg.push_back(s0);
    }
    agg[k].launches += 1;
    agg[k].ms_total += ms;
    agg[k].flops_total += r.flops;
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  h->prof.clear();
  *n_out = (int)agg.size();
  for (int k = 0; k < (int)agg.size() && k < max_out && out; ++k) out[k] = agg[k];
  return SGA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// bits-back variant (bb_sga.py)
// ---------------------------------------------------------------------------------------------
namespace {

// (z_mean | z_logvar) = h_a(y_tilde), 2C output channels (bb_sga.py:69,93-94)
int bb_init_z_impl(sga_handle* h, const Geom& g, const float* y_tilde, float* zml, hipStream_t st) {
  const int B = g.B, C = h->C;
  float* t0 = h->hs1.p; float* t1 = h->hs0.p;
  h->cur_part = &h->part;
  SGACHK(conv3(h, h->ha_f[0], h->ha_bias[0], y_tilde, C, B, g.yh, g.yw, t0, false, EPI_BIAS_RELU, nullptr, st));
  SGACHK(conv5s2(h, h->ha_f[1], h->ha_bias[1], t0, B, g.yh, g.yw, g.zh1, g.zw1, t1, EPI_BIAS_RELU, nullptr, st));
  SGACHK(conv5s2(h, h->ha_f[2], nullptr, t1, B, g.zh1, g.zw1, g.zh, g.zw, zml, EPI_BIAS, nullptr, st));
  return SGA_OK;
}

// one evaluation of the bits-back objective and its gradients; results in the workspace:
// g_yt_dist/g_yt_rate/dyt (stage 1) and g_zml
int bb_step_core(sga_handle* h, const Geom& g, const float* x, const float* y, const float* zml,
                 const float* u_y, const float* eps, bool rate_only, int eps_stream,
                 hipStream_t st) {
  const int C = h->C;
  const int64_t ny = (int64_t)g.B * g.yh * g.yw * C;
  if (rate_only) {
    if (y != h->yt.p)
      HIPCHK(h, launch_copy(h->yt.p, y, (int64_t)(ny), st));
  } else {
    HIPCHK(h, launch_sample(y, u_y, h->ctx, 0, h->yt.p, h->dyt.p, ny, st, 0, h->img_ids, ny / g.B));
  }
  HIPCHK(h, launch_bb_sample_z(zml, eps, h->ctx, eps_stream, g.B, g.zh * g.zw, C, h->zt.p,
                               h->jac_lv.p, h->sums, st, h->img_ids));
  SGACHK(rd_forward_backward(h, g, x, true, st, /*density=*/true, /*do_synth=*/!rate_only));
  HIPCHK(h, launch_bb_zgrad(h->g_zt_hs.p, h->g_zt_eb.p, h->jac_lv.p, h->ctx, inv_ln2_hw(g),
                            (int64_t)g.B * g.zh * g.zw, C, h->g_zml.p, st));
  return SGA_OK;
}

int bb_eval_impl(sga_handle* h, const Geom& g, const float* x, const float* y_hat, const float* zml,
                 const float* eps, float* metrics, hipStream_t st) {
  const int C = h->C;
  const int64_t ny = (int64_t)g.B * g.yh * g.yw * C;
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (g.B), st));
  if (y_hat != h->yt.p)
    HIPCHK(h, launch_copy(h->yt.p, y_hat, (int64_t)(ny), st));
  HIPCHK(h, launch_bb_sample_z(zml, eps, h->ctx, 3, g.B, g.zh * g.zw, C, h->zt.p, nullptr, h->sums, st,
                               h->img_ids));
  SGACHK(rd_forward_backward(h, g, x, false, st, true, true));
  if (metrics) {
    HIPCHK(h, launch_finalize_eval_bb(h->sums, g.B, g.H, g.W, metrics, st));
    if (g.H >= 176 && g.W >= 176)
      HIPCHK(h, launch_msssim(h->xq.p, x, g.B, g.H, g.W, h->msA, h->msB, h->ms_stats, h->ms_counts,
                              metrics, 8, st));
  }
  return SGA_OK;
}

// `n` replays of one bits-back iteration (stage 0 / 1) as a hipGraph captured once per geometry: the step context
// lives on the device and every launch argument is a workspace pointer, so an iteration is the same launch
// sequence whatever its number (as in sga_run_steps).  Eager when graphs are off or a profile is being taken.
template <typename F>
int bb_iterations(sga_handle* h, int stage, const Geom& g, int n, hipStream_t st, F&& enqueue) {
  bool graphed = false;
  hipGraphExec_t bb_graph = nullptr;
  int done = 0;                    // iterations already run by the fork-point candidates
  if (h->use_graph && !h->profiling && n > 0) {
    auto capture = [&]() -> hipGraphExec_t {
      hipGraphExec_t ex = nullptr;
      hipGraph_t graph = nullptr;
      ++h->n_captures;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = enqueue();
        const hipError_t ec = hipStreamEndCapture(st, &graph);
        if (!(rc == SGA_OK && ec == hipSuccess && graph && hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess))
          ex = nullptr;
        if (graph) (void)hipGraphDestroy(graph);
      }
      (void)hipGetLastError();
      return ex;
    };
    // stage 1 (the full step, bb_sga.py:203-236) forks the hyper branch like the plain SGA step: its fork point is timed the
    // same way (stage 2 is the hyper branch alone: nothing to fork)
    const sga_handle::GraphKey key{1 + stage, g.B, g.H, g.W, 0, h->scale_bound, 0};
    sga_handle::GraphEntry* e = find_graph(h, key);
    const bool tune = stage == 0 && h->fork_auto && h->overlap && (!h->x3 || h->x3_fork) && n >= 100 && !(e && e->tuned);
    if (e && tune) { erase_graph(h, e, st); e = nullptr; }
    if (!e) {
      hipGraphExec_t ex = nullptr;
      bool tuned = false;
      if (tune) {
        SGACHK(timed_fork_choice(h, st, g.B, g.H, g.W, capture, &ex, &done));
        tuned = ex != nullptr;
      } else if (stage == 0 && h->fork_auto) {
        h->fork_name = tuned_fork_for(h, 1, g.B, g.H, g.W, &tuned);
      }
      if (!ex) ex = capture();
      if (ex) e = insert_graph(h, key, ex, tuned, stage == 0 ? h->fork_name : nullptr, false, st);
    } else if (stage == 0 && h->fork_auto) {
      h->fork_name = e->fork_name;
    }
    if (stage == 0) h->bb_graph_tuned = e && e->tuned;
    bb_graph = e ? e->exec : nullptr;
    graphed = bb_graph != nullptr;
  }
  for (int it = done; it < n; ++it) {
    if (graphed) HIPCHK(h, hipGraphLaunch(bb_graph, st));
    else SGACHK(enqueue());
  }
  return SGA_OK;
}

// bb_sga.py:238-261 given y_hat in h->yt: (z_mean | z_logvar) = h_a(y_hat) (:247), fresh Adam, r_its rate-only
// iterations with the step sizes of h->lrtab2 (uploaded by the caller).  A pure function of (y_hat, seed, r_its, r_lr,
// loss_scale): the sender (sga_bb_run) and a receiver that has decoded y_hat (sga_bb_refine) get the same posterior
// parameters bit for bit -- what bits-back coding needs.
int bb_stage2(sga_handle* h, const Geom& g, int r_its, float loss_scale, uint64_t seed, hipStream_t st) {
  const int B = g.B, H = g.H, W = g.W;
  const int64_t nz2 = (int64_t)B * g.zh * g.zw * h->C * 2;
  SGACHK(bb_init_z_impl(h, g, h->yt.p, h->zml.p, st));       // bb_sga.py:247
  HIPCHK(h, launch_fill(h->mzml.p, 0.f, (int64_t)(nz2), st));
  HIPCHK(h, launch_fill(h->vzml.p, 0.f, (int64_t)(nz2), st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (B), st));
  HIPCHK(h, launch_set_ctx(h->ctx, -1, r_its, 1.f, 0.f, 0.f, loss_scale, seed, st));
  SGACHK(bb_iterations(h, 1, g, r_its, st, [&]() -> int {
    HIPCHK(h, launch_advance_ctx(h->ctx, h->Ttab.p, h->lrtab2.p, st));
    SGACHK(bb_step_core(h, g, nullptr, h->yt.p, h->zml.p, nullptr, nullptr, true, 2, st));
    HIPCHK(h, launch_adam_ctx(h->zml.p, h->g_zml.p, h->mzml.p, h->vzml.p, nz2, h->ctx, st));
    HIPCHK(h, launch_finalize_step(h->sums, h->ctx, B, H, W, nullptr, nullptr, h->trace.p, st));
    return SGA_OK;
  }));
  return SGA_OK;
}

}  // namespace

extern "C" {

int sga_bb_init_z(sga_handle* h, const float* y_tilde, int B, int H, int W, float* zml, void* stream) {
  if (!h || !y_tilde || !zml) return SGA_ERR_BAD_ARG;
  if (!h->cfg.bits_back) return SGA_ERR_UNSUPPORTED;
  SGACHK(check_shape(h, B, H, W));
  return bb_init_z_impl(h, make_geom(B, H, W), y_tilde, zml, (hipStream_t)stream);
}

int sga_bb_step_grads(sga_handle* h, const float* x, int B, int H, int W, const float* y,
                      const float* zml, float T, float lambda, float loss_scale, uint64_t seed,
                      uint32_t it, const float* u_y, const float* eps, int rate_only, float* gy,
                      float* gzml, float* scalars, float* psnr, void* stream) {
  if (!h || !y || !zml || !(T > 0.f) || (!rate_only && !x)) return SGA_ERR_BAD_ARG;
  if (!h->cfg.bits_back) return SGA_ERR_UNSUPPORTED;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  SGACHK(ensure_borders(h, g, st));
  HIPCHK(h, launch_set_ctx(h->ctx, (int)it, 0, T, 0.f, rate_only ? 0.f : lambda, loss_scale, seed, st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (B), st));
  SGACHK(bb_step_core(h, g, x, y, zml, u_y, eps, rate_only != 0, rate_only ? 2 : 1, st));
  const int64_t ny = (int64_t)B * g.yh * g.yw * h->C, nz2 = (int64_t)B * g.zh * g.zw * h->C * 2;
  if (gy && !rate_only)
    HIPCHK(h, launch_combine_grad(h->g_yt_dist.p, h->g_yt_rate.p, h->dyt.p, gy, ny, st));
  if (gzml)
    HIPCHK(h, launch_copy(gzml, h->g_zml.p, (int64_t)(nz2), st));
  HIPCHK(h, launch_finalize_step(h->sums, h->ctx, B, H, W, scalars, psnr, nullptr, st));
  return SGA_OK;
}

int sga_bb_eval(sga_handle* h, const float* x, int B, int H, int W, const float* y_hat,
                const float* zml, const float* eps, uint64_t seed, float* metrics, void* stream) {
  if (!h || !x || !y_hat || !zml) return SGA_ERR_BAD_ARG;
  if (!h->cfg.bits_back) return SGA_ERR_UNSUPPORTED;
  SGACHK(check_shape(h, B, H, W));
  hipStream_t st = (hipStream_t)stream;
  const Geom g = make_geom(B, H, W);
  SGACHK(ensure_borders(h, g, st));
  HIPCHK(h, launch_set_ctx(h->ctx, 0, 0, 1.f, 0.f, 0.f, 1.f, seed, st));
  return bb_eval_impl(h, g, x, y_hat, zml, eps, metrics, st);
}

int sga_bb_run(sga_handle* h, const float* x, int B, int H, int W, float lambda, float loss_scale,
               int its, int r_its, double lr, double r_lr, double annealing_rate, int t0,
               double T_ub, uint64_t seed, float* y_hat, float* zml_out, float* metrics,
               float* trace1, float* trace2, void* stream) {
  if (!h || !x || its < 0 || r_its < 0 || its > kMaxIts || r_its > kMaxIts) return SGA_ERR_BAD_ARG;
  if (!h->cfg.bits_back) return SGA_ERR_UNSUPPORTED;
  SGACHK(check_shape(h, B, H, W));
  LaunchStream ls(h, stream);
  hipStream_t st = ls.run;
  const Geom g = make_geom(B, H, W);
  const int C = h->C;
  const int64_t ny = (int64_t)B * g.yh * g.yw * C, nz2 = (int64_t)B * g.zh * g.zw * C * 2;
  SGACHK(ensure_borders(h, g, st));
  HIPCHK(h, launch_copy(h->xin.p, x, (int64_t)((size_t)B * H * W * 3), st));
  // bb_sga.py:202-204: y = g_a(x); (z_mean | z_logvar) = h_a(y) with the UNROUNDED y fed as y_tilde
  SGACHK(encode_impl(h, g, h->xin.p, h->y.p, h->zml.p, st));
  // host tables (double, cast per step)
  const int nmax = its > r_its ? its : r_its;
  h->hT.assign(nmax > 0 ? nmax : 1, 1.f); h->hLr.assign(nmax > 0 ? nmax : 1, 0.f);
  std::vector<float> lr2(nmax > 0 ? nmax : 1, 0.f);
  for (int it = 0; it < nmax; ++it) {
    double tau = T_ub * std::exp(-annealing_rate * (double)(it - t0));
    h->hT[it] = (float)std::fmin(std::fmax(tau, 1e-8), T_ub);
    const int t = it + 1;
    const double corr = std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.9, t));
    h->hLr[it] = (float)(lr * corr);
    lr2[it] = (float)(r_lr * corr);
  }
  HIPCHK(h, hipMemcpyAsync(h->Ttab.p, h->hT.data(), h->hT.size() * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->lrtab.p, h->hLr.data(), h->hLr.size() * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->lrtab2.p, lr2.data(), lr2.size() * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipStreamSynchronize(st));
  // ---- stage 1: R-D optimisation of [y, z_mean, z_logvar] (bb_sga.py:205-236) ----------------
  HIPCHK(h, launch_fill(h->my.p, 0.f, (int64_t)(ny), st));
  HIPCHK(h, launch_fill(h->vy.p, 0.f, (int64_t)(ny), st));
  HIPCHK(h, launch_fill(h->mzml.p, 0.f, (int64_t)(nz2), st));
  HIPCHK(h, launch_fill(h->vzml.p, 0.f, (int64_t)(nz2), st));
  HIPCHK(h, launch_fill((float*)h->sums, 0.f, (int64_t)(sizeof(ImgSums) / sizeof(float)) * (B), st));
  HIPCHK(h, launch_set_ctx(h->ctx, -1, its, 0.f, 0.f, lambda, loss_scale, seed, st));
  SGACHK(bb_iterations(h, 0, g, its, st, [&]() -> int {
    HIPCHK(h, launch_advance_ctx(h->ctx, h->Ttab.p, h->lrtab.p, st));
    SGACHK(bb_step_core(h, g, h->xin.p, h->y.p, h->zml.p, nullptr, nullptr, false, 1, st));
    HIPCHK(h, launch_adam_latent(h->y.p, h->g_yt_dist.p, h->g_yt_rate.p, h->dyt.p, h->my.p, h->vy.p, ny, h->ctx, st));
    HIPCHK(h, launch_adam_ctx(h->zml.p, h->g_zml.p, h->mzml.p, h->vzml.p, nz2, h->ctx, st));
    HIPCHK(h, launch_finalize_step(h->sums, h->ctx, B, H, W, nullptr, nullptr, h->trace.p, st));
    return SGA_OK;
  }));
  if (trace1 && its > 0)
    HIPCHK(h, launch_copy(trace1, h->trace.p, (int64_t)((size_t)its * 4), st));
  // ---- stage 2: fix y_tilde = round(y), rate optimisation of zml (bb_sga.py:238-261) ----------
  HIPCHK(h, launch_round(h->y.p, h->yt.p, ny, st));          // h->yt holds y_hat from here on
  if (y_hat) HIPCHK(h, launch_copy(y_hat, h->yt.p, (int64_t)(ny), st));
  SGACHK(bb_stage2(h, g, r_its, loss_scale, seed, st));
  if (trace2 && r_its > 0)
    HIPCHK(h, launch_copy(trace2, h->trace.p, (int64_t)((size_t)r_its * 4), st));
  if (zml_out) HIPCHK(h, launch_copy(zml_out, h->zml.p, (int64_t)(nz2), st));
  // ---- eval with a fresh eps draw (bb_sga.py:273-275) -------------------------------------------
  if (metrics) {
    HIPCHK(h, launch_set_ctx(h->ctx, 0, 0, 1.f, 0.f, 0.f, 1.f, seed, st));
    SGACHK(bb_eval_impl(h, g, h->xin.p, h->yt.p, h->zml.p, nullptr, metrics, st));
  }
  return SGA_OK;
}

int sga_bb_refine(sga_handle* h, const float* y_hat, int B, int H, int W, float loss_scale, int r_its, double r_lr,
                  uint64_t seed, float* zml_out, void* stream) {
  if (!h || !y_hat || !zml_out || r_its < 0 || r_its > kMaxIts) return SGA_ERR_BAD_ARG;
  if (!h->cfg.bits_back) return SGA_ERR_UNSUPPORTED;
  SGACHK(check_shape(h, B, H, W));
  LaunchStream ls(h, stream);
  hipStream_t st = ls.run;
  const Geom g = make_geom(B, H, W);
  const int64_t ny = (int64_t)B * g.yh * g.yw * h->C, nz2 = (int64_t)B * g.zh * g.zw * h->C * 2;
  h->hT.assign(r_its > 0 ? r_its : 1, 1.f);
  std::vector<float> lr2(r_its > 0 ? r_its : 1, 0.f);
  for (int it = 0; it < r_its; ++it) {
    const int t = it + 1;
    lr2[it] = (float)(r_lr * (std::sqrt(1.0 - std::pow(0.999, t)) / (1.0 - std::pow(0.9, t))));
  }
  HIPCHK(h, hipMemcpyAsync(h->Ttab.p, h->hT.data(), h->hT.size() * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->lrtab2.p, lr2.data(), lr2.size() * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipStreamSynchronize(st));
  HIPCHK(h, launch_copy(h->yt.p, y_hat, ny, st));
  SGACHK(bb_stage2(h, g, r_its, loss_scale, seed, st));
  HIPCHK(h, launch_copy(zml_out, h->zml.p, nz2, st));
  return SGA_OK;
}

int sga_op_factorized_density(sga_handle* h, const float* v, int64_t n_pix, float* p, float* dp_dv,
                              void* stream) {
  if (!h || !v || n_pix <= 0 || n_pix * h->C > 0x7fffffffLL) return SGA_ERR_BAD_ARG;
  HIPCHK(h, launch_factorized_pdf(v, h->eb_packed, nullptr, 1, (int)n_pix, h->C, 1.f, nullptr, nullptr,
                                  p, dp_dv, (hipStream_t)stream));
  return SGA_OK;
}

}  // extern "C"
