// GPU box, round 5: stand-alone harness for the persistent wave-specialised igdn2.bwd kernel (csrc/igdn_bwd_ws.hip) against
// the tile kernel (csrc/gdn_fused.hip) on random operands of a given shape: bit-equality (with the location of the first
// mismatches), time per launch of both, and -- built with -DSGA_CLOCK_PROBE=1 -- the per-phase stamps of the new kernel.
//   build:  scripts/r05/build_igdn_ws_bench.sh [extra -D flags]      run:  igdn_ws_bench C B Hg Wg [reps] [sched: 0 static, 1 dynamic]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../../improving-inference-for-neural-image-compression_amd/csrc/sga_common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f; }

template <typename T>
T* dev(const std::vector<T>& h) {
  T* p; CK(hipMalloc(&p, h.size() * sizeof(T) + 256)); CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}

static std::vector<float> frag(const std::vector<float>& w, int nslab, int N, int Kc) {
  const int NB = N / 32, qs = Kc / 8;
  std::vector<float> f((size_t)nslab * qs * NB * 256);
  for (int sl = 0; sl < nslab; ++sl) for (int q = 0; q < qs; ++q) for (int nb = 0; nb < NB; ++nb) for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r)
    f[((((size_t)sl * qs + q) * NB + nb) * 64 + lane) * 4 + r] = w[((size_t)sl * N + nb * 32 + (lane & 31)) * Kc + q * 8 + (lane >> 5) * 4 + r];
  return f;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 192, B = argc > 2 ? atoi(argv[2]) : 8, Hg = argc > 3 ? atoi(argv[3]) : 128, Wg = argc > 4 ? atoi(argv[4]) : 128;
  const int reps = argc > 5 ? atoi(argv[5]) : 30, dyn = argc > 6 ? atoi(argv[6]) : 1, nt = argc > 7 ? atoi(argv[7]) : 0;
  const int Hp = 2 * Hg + 4, Wp = 2 * Wg + 4;
  const long long M = (long long)B * Hg * Wg;
  unsigned seed = 12345;
  std::vector<float> gamma((size_t)C * C), wc((size_t)3 * C * 32, 0.f), pad((size_t)B * Hp * Wp * 3), v((size_t)M * C), s((size_t)M * C);
  for (auto& x : gamma) x = 0.02f * frand(seed);
  for (int st = 0; st < 3; ++st) for (int n = 0; n < C; ++n) for (int half = 0; half < 2; ++half) for (int kk = 0; kk < 15; ++kk)
    if (2 * st + half <= 4) wc[((size_t)st * C + n) * 32 + half * 16 + kk] = frand(seed) - 0.5f;
  for (auto& x : pad) x = frand(seed) - 0.5f;
  for (auto& x : v) x = frand(seed) - 0.5f;
  for (auto& x : s) x = 1.f + frand(seed);
  GdnArgs g;
  memset(&g, 0, sizeof(g));
  g.C = C; g.mode = GDN_IGDN_BWD; g.pro = GDN_PRO_CONV3; g.M = M;
  for (int p = 0; p < 4; ++p) g.nsplit[p] = 1;
  g.s_out = 1;
  g.pad = dev(pad); g.wc = dev(wc); g.Hg = Hg; g.Wg = Wg; g.Hp = Hp; g.Wp = Wp;
  g.w = dev(gamma); g.s = dev(s); g.v = dev(v);
  g.wf = dev(frag(gamma, 1, C, C)); g.wcf = dev(frag(wc, 3, C, 32));
  unsigned* sched; CK(hipMalloc(&sched, 256)); CK(hipMemset(sched, 0, 256));
  g.sched = dyn ? sched : nullptr; (void)nt;
  float *o0, *o1;
  CK(hipMalloc(&o0, (size_t)M * C * 4 + 256)); CK(hipMalloc(&o1, (size_t)M * C * 4 + 256));
  CK(hipMemset(o0, 0xff, (size_t)M * C * 4)); CK(hipMemset(o1, 0xee, (size_t)M * C * 4));
#ifdef SGA_CLOCK_PROBE
  unsigned long long* clk; const size_t nclk = (size_t)256 * 40 * 16 * 2;
  CK(hipMalloc(&clk, nclk * 8)); CK(hipMemset(clk, 0, nclk * 8));
#endif
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int k = 0; k < 2; ++k) {
    g.ws = k ? 2 : 0; g.out = k ? o1 : o0;
#ifdef SGA_CLOCK_PROBE
    g.clk = nullptr;
#endif
    for (int r = 0; r < 3; ++r) { int rc = launch_gdn_tile(g, st); if (rc) { printf("launch rc %d\n", rc); return 1; } }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) launch_gdn_tile(g, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[k], e0, e1));
  }
  char n0[96], n1[96];
  g.ws = 0; gdn_kernel_name(g, n0, sizeof(n0)); g.ws = 2; gdn_kernel_name(g, n1, sizeof(n1));
  const double gf = (2.0 * M * C * (double)C + 2.0 * M * 75.0 * C) * 1e-9;
  printf("C %d B %d grid %dx%d M %lld tiles %lld sched %s\n  %-44s %8.1f us  %6.1f TF/s\n  %-44s %8.1f us  %6.1f TF/s\n", C, B, Hg, Wg, M, (M + 63) / 64,
         dyn ? "dynamic" : "static", n0, 1e3 * ms[0] / reps, gf / (ms[0] / reps), n1, 1e3 * ms[1] / reps, gf / (ms[1] / reps));
  std::vector<float> h0((size_t)M * C), h1((size_t)M * C);
  CK(hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
  long long bad = 0; int shown = 0;
  for (size_t i = 0; i < h0.size(); ++i)
    if (memcmp(&h0[i], &h1[i], 4) != 0 && !(h0[i] == h1[i])) {
      ++bad;
      if (shown < 12) { printf("  mismatch row %lld (tile %lld, row-in-tile %lld) col %lld: tile %.9g ws %.9g\n", (long long)(i / C), (long long)(i / C / 64), (long long)(i / C % 64), (long long)(i % C), h0[i], h1[i]); ++shown; }
    }
  printf("  mismatching elements: %lld of %zu\n", bad, h0.size());
#ifdef SGA_CLOCK_PROBE
  // one stamped launch of the persistent kernel: [wg][phase][16] wall-clock ticks (100 MHz): slots 0..7 matrix wave 0, 8..15 memory wave 0
  g.ws = 2; g.out = o1; g.clk = clk;
  launch_gdn_tile(g, st); CK(hipStreamSynchronize(st));
  std::vector<unsigned long long> hc(nclk);
  CK(hipMemcpy(hc.data(), clk, nclk * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < nclk / 2; ++i) if (hc[i]) { t0 = std::min(t0, hc[i]); t1 = std::max(t1, hc[i]); }
  printf("  stamped launch span %.1f us\n", (t1 - t0) * 0.01);
  static const char* names[16] = {"m.start", "m.conv3", "m.Ba", "m.g->lds", "m.contract", "m.Bb", "m.n->lds", "m.Bc",
                                  "l.start", "l.epilogue", "l.request", "l.Ba", "l.fill", "l.Bb", "l.Bc", "-"};
  for (int wg : {0, 100, 255}) {
    printf("  wg %d (us since launch start)\n", wg);
    for (int ph = 0; ph < 12; ++ph) {
      const unsigned long long* r = &hc[((size_t)wg * 40 + ph) * 16];
      if (!r[0] && !r[8]) continue;
      printf("   ph %2d:", ph);
      for (int k = 0; k < 15; ++k) if (r[k]) printf(" %s %.2f", names[k], (r[k] - t0) * 0.01);
      printf("\n");
    }
  }
  // mean durations over all workgroups and steady-state phases (2..6)
  double acc[16] = {0}; long long cnt = 0;
  for (int wg = 0; wg < 256; ++wg) for (int ph = 2; ph <= 6; ++ph) {
    const unsigned long long* r = &hc[((size_t)wg * 40 + ph) * 16];
    if (!r[0] || !r[7] || !r[14]) continue;
    ++cnt;
    acc[1] += (r[1] - r[0]); acc[2] += (r[2] - r[1]); acc[3] += (r[3] - r[2]); acc[4] += (r[4] - r[3]); acc[5] += (r[5] - r[4]); acc[6] += (r[6] - r[5]); acc[7] += (r[7] - r[6]);
    acc[9] += (r[9] - r[8]); acc[10] += (r[10] - r[9]); acc[11] += (r[11] - r[10]); acc[12] += (r[12] - r[11]); acc[13] += (r[13] - r[12]); acc[14] += (r[14] - r[13]);
  }
  {
    double cy3 = 0, cyc = 0, w3 = 0, wc_ = 0; long long n = 0;
    const size_t H = nclk / 2;
    for (int wg = 0; wg < 256; ++wg) for (int ph = 2; ph <= 5; ++ph) {
      const size_t b = ((size_t)wg * 40 + ph) * 16;
      if (!hc[b] || !hc[b + 4]) continue;
      ++n; cy3 += (double)(hc[H + b + 1] - hc[H + b]); cyc += (double)(hc[H + b + 4] - hc[H + b + 3]);
      w3 += (double)(hc[b + 1] - hc[b]); wc_ += (double)(hc[b + 4] - hc[b + 3]);
    }
    if (n) printf("  matrix wave 0: conv3 %.0f cycles (MFMA-bound %d), contract %.0f cycles (MFMA-bound %d); shader clock %.0f / %.0f MHz\n",
                  cy3 / n, 120 * 64, cyc / n, C / 8 * (C / 64) * 4 * 64, cy3 / w3 * 100, cyc / wc_ * 100);
  }
  if (cnt) {
    printf("  steady-state means (us): matrix: conv3 %.2f | wait Ba %.2f | g->lds %.2f | contract %.2f | wait Bb %.2f | n->lds %.2f | wait Bc %.2f\n",
           acc[1] / cnt * 0.01, acc[2] / cnt * 0.01, acc[3] / cnt * 0.01, acc[4] / cnt * 0.01, acc[5] / cnt * 0.01, acc[6] / cnt * 0.01, acc[7] / cnt * 0.01);
    printf("                           memory: epilogue %.2f | request %.2f | wait Ba %.2f | fill %.2f | wait Bb %.2f | wait Bc %.2f\n",
           acc[9] / cnt * 0.01, acc[10] / cnt * 0.01, acc[11] / cnt * 0.01, acc[12] / cnt * 0.01, acc[13] / cnt * 0.01, acc[14] / cnt * 0.01);
  }
#endif
  return bad ? 2 : 0;
}
