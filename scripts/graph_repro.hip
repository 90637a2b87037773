// Stand-alone HIP repro attempts for the two orchestration defects of DESIGN_EXPERIMENTS.md A.7 / A.3 -- no libsga_hip, no
// PyTorch: only the HIP runtime.  Built and run by scripts/graph_repro.sh on the GPU box.
//
//   part 1 ("lifetime"): the graph life cycle of sga_run_steps -- three two-stream candidate graphs captured on the same pair
//     of streams (fork / join through events, ThreadLocal capture, ~30 kernel nodes with 700-byte argument blocks), each
//     replayed a few times and timed with an event pair, the two losers DESTROYED (hipGraphExecDestroy) while the winner
//     lives on, the winner replayed 2000 x, then a "next run" (pageable H2D copies, fills, more replays), then a geometry
//     change (synchronise, destroy, capture again).  Run under MALLOC_PERTURB_ (freed host memory is overwritten) a
//     use-after-free inside the runtime shows up as a crash or as a wrong result here.
//   part 2 ("visibility"): a producer kernel on stream A writes a buffer; the graph forks; the FIRST kernel of stream B reads
//     that buffer while a long kernel occupies stream A; every replay the producer writes a new value.  Any reader that sees
//     the previous replay's value is counted.  The reader's workgroups are arranged to read what a workgroup on ANOTHER XCD
//     wrote (blocks b, b + 8, ... share an XCD), the situation of k_factorized reading z_tilde after the fork.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(expr)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_));        \
      exit(2);                                                                                      \
    }                                                                                               \
  } while (0)

struct BigArgs {      // the size of ConvArgs: kernel arguments of this size live in the graph's kernarg pool
  float* dst; const float* src; int n; int add; int pad[170];
};

__global__ void k_work(const BigArgs a) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) a.dst[i] = a.src[i] + (float)a.add;
}

__global__ void k_spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 0);
}

// ---- part 1 -----------------------------------------------------------------------------------------------------------
struct Ctx {
  hipStream_t st, sb;
  hipEvent_t ev_fork, ev_join;
  float *a, *b, *c, *d;
  int n;
};

static hipGraphExec_t capture(Ctx& c, int fork_at, int nodes) {
  hipGraph_t g = nullptr;
  hipGraphExec_t ex = nullptr;
  CHK(hipStreamBeginCapture(c.st, hipStreamCaptureModeThreadLocal));
  BigArgs m;
  memset(&m, 0, sizeof(m));
  m.n = c.n;
  for (int k = 0; k < nodes; ++k) {
    if (k == fork_at) {
      CHK(hipEventRecord(c.ev_fork, c.st));
      CHK(hipStreamWaitEvent(c.sb, c.ev_fork, 0));
    }
    m.dst = (k & 1) ? c.a : c.b; m.src = (k & 1) ? c.b : c.a; m.add = 1;
    hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, c.st, m);
  }
  // the side branch's nodes are created after the main chain's (as the library does)
  for (int k = 0; k < 11; ++k) {
    m.dst = (k & 1) ? c.c : c.d; m.src = (k & 1) ? c.d : c.c; m.add = 1;
    hipLaunchKernelGGL(k_work, dim3(32), dim3(256), 0, c.sb, m);
  }
  CHK(hipEventRecord(c.ev_join, c.sb));
  CHK(hipStreamWaitEvent(c.st, c.ev_join, 0));
  m.dst = c.a; m.src = c.a; m.add = 0;
  hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, c.st, m);
  CHK(hipStreamEndCapture(c.st, &g));
  CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CHK(hipGraphDestroy(g));
  return ex;
}

static int part1(int rounds, bool destroy) {
  Ctx c;
  c.n = 1 << 16;
  CHK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
  CHK(hipStreamCreateWithPriority(&c.sb, hipStreamNonBlocking, 0));
  CHK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
  CHK(hipMalloc(&c.a, c.n * 4)); CHK(hipMalloc(&c.b, c.n * 4)); CHK(hipMalloc(&c.c, c.n * 4)); CHK(hipMalloc(&c.d, c.n * 4));
  std::vector<hipGraphExec_t> retired;
  std::vector<float> host(c.n, 0.f), back(c.n);
  long long bad = 0;
  for (int r = 0; r < rounds; ++r) {
    CHK(hipMemcpyAsync(c.a, host.data(), c.n * 4, hipMemcpyHostToDevice, c.st));      // pageable source, as sga_run_begin
    CHK(hipMemcpyAsync(c.c, host.data(), c.n * 4, hipMemcpyHostToDevice, c.st));
    CHK(hipStreamSynchronize(c.st));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipGraphExec_t best = nullptr;
    float best_ms = 0.f;
    int launches = 0;
    const int nodes = 19 + (r % 3);      // a "geometry": other node counts from round to round
    for (int cand = 0; cand < 3; ++cand) {
      hipGraphExec_t ex = capture(c, cand == 0 ? 0 : (cand == 1 ? 3 : 4), nodes);
      for (int k = 0; k < (cand == 0 ? 3 : 1); ++k) { CHK(hipGraphLaunch(ex, c.st)); ++launches; }
      CHK(hipEventRecord(e0, c.st));
      for (int k = 0; k < 8; ++k) { CHK(hipGraphLaunch(ex, c.st)); ++launches; }
      CHK(hipEventRecord(e1, c.st));
      CHK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      hipGraphExec_t loser = nullptr;
      if (!best || ms < best_ms) { loser = best; best = ex; best_ms = ms; } else loser = ex;
      if (loser) { if (destroy) CHK(hipGraphExecDestroy(loser)); else retired.push_back(loser); }
    }
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    for (int k = 0; k < 300; ++k) { CHK(hipGraphLaunch(best, c.st)); ++launches; }
    // "the next run on the handle": uploads, fills, replays of the surviving graph
    std::vector<float> tab(2000, 1.f);
    CHK(hipMemcpyAsync(c.d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c.st));
    CHK(hipStreamSynchronize(c.st));
    for (int k = 0; k < 300; ++k) { CHK(hipGraphLaunch(best, c.st)); ++launches; }
    CHK(hipMemcpyAsync(back.data(), c.a, c.n * 4, hipMemcpyDeviceToHost, c.st));
    CHK(hipStreamSynchronize(c.st));
    // every replay adds `nodes` to a (ping-pong a <-> b: an even count lands in a, an odd count in b and the last kernel copies a)
    // -- only consistency across the buffer is checked (every element went through the same chain)
    for (int i = 1; i < c.n; ++i) bad += back[i] != back[0];
    // geometry change: synchronise, drop the winner, next round captures again
    CHK(hipStreamSynchronize(c.st));
    if (destroy) CHK(hipGraphExecDestroy(best)); else retired.push_back(best);
    if ((r & 15) == 15) fprintf(stderr, "lifetime round %d ok (%d launches, a[0] = %.0f)\n", r + 1, launches, back[0]);
  }
  for (hipGraphExec_t ex : retired) CHK(hipGraphExecDestroy(ex));
  printf("part1 lifetime: rounds %d policy %s inconsistent elements %lld\n", rounds, destroy ? "destroy" : "retire", bad);
  return bad != 0;
}

// ---- part 2 -----------------------------------------------------------------------------------------------------------
__global__ void k_produce(int* buf, int n_per, const int* iter) {
  // block b writes the chunk that reader block (b + 1) % gridDim.x reads: another XCD when gridDim.x % 8 == 0
  const int chunk = (blockIdx.x + 1) % gridDim.x;
  const int v = *iter;
  for (int i = threadIdx.x; i < n_per; i += blockDim.x) buf[chunk * n_per + i] = v;
}
// The expected value is the reader's OWN count of reader launches (cache-bypassing loads; bumped by the last block), so that a
// reader that runs too early -- or sees a stale `iter` as well as a stale buffer -- is still caught.
__device__ unsigned g_reader_count = 0, g_reader_ticket = 0;
__global__ void k_consume(const int* buf, int n_per, const int* iter, unsigned long long* stale) {
  __shared__ int expect;
  if (threadIdx.x == 0) expect = (int)__hip_atomic_load(&g_reader_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  const int v = expect;
  int bad = 0;
  for (int i = threadIdx.x; i < n_per; i += blockDim.x) bad += buf[blockIdx.x * n_per + i] != v;
  if (threadIdx.x == 0 && *iter != v) ++bad;
  if (bad) atomicAdd(stale, (unsigned long long)bad);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(&g_reader_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(&g_reader_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_add(&g_reader_count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__global__ void k_reset_reader() { g_reader_count = 0; g_reader_ticket = 0; }
__global__ void k_advance(int* iter) { *iter += 1; }

static int part2(int replays, int grid, long long spin_ticks) {
  hipStream_t st, sb;
  hipEvent_t evf, evj;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CHK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, 0));
  CHK(hipEventCreateWithFlags(&evf, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&evj, hipEventDisableTiming));
  const int n_per = 4096;
  int *buf, *iter, *sink;
  unsigned long long* stale;
  CHK(hipMalloc(&buf, (size_t)grid * n_per * 4)); CHK(hipMalloc(&iter, 4)); CHK(hipMalloc(&sink, 4));
  CHK(hipMalloc(&stale, 8));
  CHK(hipMemset(buf, 0xff, (size_t)grid * n_per * 4)); CHK(hipMemset(iter, 0, 4)); CHK(hipMemset(stale, 0, 8)); CHK(hipMemset(sink, 0, 4));
  hipLaunchKernelGGL(k_reset_reader, dim3(1), dim3(1), 0, st);
  CHK(hipStreamSynchronize(st));
  hipGraph_t g = nullptr;
  hipGraphExec_t ex = nullptr;
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_produce, dim3(grid), dim3(256), 0, st, buf, n_per, iter);
  CHK(hipEventRecord(evf, st));
  CHK(hipStreamWaitEvent(sb, evf, 0));
  hipLaunchKernelGGL(k_spin, dim3(512), dim3(256), 0, st, spin_ticks, sink);          // the main chain goes on
  hipLaunchKernelGGL(k_consume, dim3(grid), dim3(256), 0, sb, buf, n_per, iter, stale);   // first side-stream kernel
  CHK(hipEventRecord(evj, sb));
  CHK(hipStreamWaitEvent(st, evj, 0));
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, iter);
  CHK(hipStreamEndCapture(st, &g));
  CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CHK(hipGraphDestroy(g));
  for (int k = 0; k < replays; ++k) CHK(hipGraphLaunch(ex, st));
  CHK(hipStreamSynchronize(st));
  unsigned long long h = 0;
  CHK(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
  printf("part2 visibility: replays %d grid %d stale words %llu\n", replays, grid, h);
  CHK(hipGraphExecDestroy(ex));
  return h != 0;
}

// ---- part 3: hipGraphExecDestroy while replays of that graph are still in flight ----------------------------------------
// (CUDA frees an in-flight executable graph asynchronously on completion; what does this runtime do?)  Rounds of: capture,
// queue `inflight` replays, destroy WITHOUT synchronising, capture the next graph at once and replay it.
static int part3(int rounds, int inflight) {
  Ctx c;
  c.n = 1 << 20;
  CHK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
  CHK(hipStreamCreateWithPriority(&c.sb, hipStreamNonBlocking, 0));
  CHK(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
  CHK(hipMalloc(&c.a, c.n * 4)); CHK(hipMalloc(&c.b, c.n * 4)); CHK(hipMalloc(&c.c, c.n * 4)); CHK(hipMalloc(&c.d, c.n * 4));
  CHK(hipMemset(c.a, 0, c.n * 4)); CHK(hipMemset(c.b, 0, c.n * 4)); CHK(hipMemset(c.c, 0, c.n * 4)); CHK(hipMemset(c.d, 0, c.n * 4));
  std::vector<float> back(c.n);
  long long bad = 0;
  double expect = 0.0;
  for (int r = 0; r < rounds; ++r) {
    const int nodes = 20 + (r % 3) * 2;          // even: every replay adds `nodes` to a
    hipGraphExec_t ex = capture(c, 3, nodes);
    for (int k = 0; k < inflight; ++k) CHK(hipGraphLaunch(ex, c.st));
    expect += (double)nodes * inflight;
    CHK(hipGraphExecDestroy(ex));                // replays still queued / running
  }
  CHK(hipMemcpyAsync(back.data(), c.a, c.n * 4, hipMemcpyDeviceToHost, c.st));
  CHK(hipStreamSynchronize(c.st));
  for (int i = 0; i < c.n; ++i) bad += (double)back[i] != expect;
  printf("part3 in-flight destroy: rounds %d x %d replays, a[0] = %.0f expected %.0f, wrong elements %lld\n", rounds, inflight,
         back[0], expect, bad);
  return bad != 0;
}

// ---- part 4: the reader is a ROOT node of the graph on the second stream ------------------------------------------------
// The library's fork "at the start" records the fork event before the first captured launch: the hyper branch's first kernel
// (k_factorized, reads z_tilde) then has NO predecessor inside the graph, and what it reads was written by the LAST kernel of
// the PREVIOUS hipGraphLaunch on the launching stream.  Is a side-stream root node ordered (and its caches acquired) after
// the previous launch's tail?
static int part4(int replays, int grid, long long spin_ticks, int head_node) {
  hipStream_t st, sb;
  hipEvent_t evf, evj;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CHK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, 0));
  CHK(hipEventCreateWithFlags(&evf, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&evj, hipEventDisableTiming));
  const int n_per = 4096;
  int *buf, *iter, *sink;
  unsigned long long* stale;
  CHK(hipMalloc(&buf, (size_t)grid * n_per * 4)); CHK(hipMalloc(&iter, 4)); CHK(hipMalloc(&sink, 4));
  CHK(hipMalloc(&stale, 8));
  CHK(hipMemset(buf, 0, (size_t)grid * n_per * 4)); CHK(hipMemset(iter, 0, 4)); CHK(hipMemset(stale, 0, 8)); CHK(hipMemset(sink, 0, 4));
  hipLaunchKernelGGL(k_reset_reader, dim3(1), dim3(1), 0, st);
  CHK(hipStreamSynchronize(st));
  hipGraph_t g = nullptr;
  hipGraphExec_t ex = nullptr;
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  if (head_node) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 0ll, sink);     // the work-around: a head node before the fork
  CHK(hipEventRecord(evf, st));                                                         // fork before the first launch
  CHK(hipStreamWaitEvent(sb, evf, 0));
  hipLaunchKernelGGL(k_spin, dim3(512), dim3(256), 0, st, spin_ticks, sink);            // main chain
  hipLaunchKernelGGL(k_consume, dim3(grid), dim3(256), 0, sb, buf, n_per, iter, stale); // side ROOT node: reads the previous replay's tail
  CHK(hipEventRecord(evj, sb));
  CHK(hipStreamWaitEvent(st, evj, 0));
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, iter);
  hipLaunchKernelGGL(k_produce, dim3(grid), dim3(256), 0, st, buf, n_per, iter);        // tail: writes for the NEXT replay
  CHK(hipStreamEndCapture(st, &g));
  CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CHK(hipGraphDestroy(g));
  for (int k = 0; k < replays; ++k) CHK(hipGraphLaunch(ex, st));
  CHK(hipStreamSynchronize(st));
  unsigned long long h = 0;
  CHK(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
  printf("part4 side-stream root node: replays %d grid %d head_node %d stale words %llu\n", replays, grid, head_node, h);
  CHK(hipGraphExecDestroy(ex));
  return h != 0;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  int rc = 0;
  if (!strcmp(what, "lifetime") || !strcmp(what, "all")) rc |= part1(argc > 2 ? atoi(argv[2]) : 64, !(argc > 3 && !strcmp(argv[3], "retire")));
  if (!strcmp(what, "visibility") || !strcmp(what, "all")) {
    rc |= part2(20000, 96, 2000);       // 20 us of main-chain work beside the reader
    rc |= part2(20000, 8, 200);
    rc |= part2(5000, 1024, 20000);
  }
  if (!strcmp(what, "inflight")) rc |= part3(argc > 2 ? atoi(argv[2]) : 50, argc > 3 ? atoi(argv[3]) : 40);
  if (!strcmp(what, "rootfork") || !strcmp(what, "all")) {
    const int head = argc > 2 ? atoi(argv[2]) : 0;
    rc |= part4(20000, 96, 2000, head);
    rc |= part4(20000, 8, 20000, head);
    rc |= part4(20000, 1024, 200, head);
  }
  printf("graph_repro %s\n", rc ? "FAILED" : "clean");
  return rc;
}
