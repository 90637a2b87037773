// Replays a C-ABI call log of libsga_hip (SGA_CALL_LOG=<file>, sga_amd/_lib.py: one text line per call) from a process WITHOUT
// Python or PyTorch: the HIP runtime this program links (ROCm's own, not the one PyTorch bundles), plain hipMalloc'd buffers,
// one non-blocking stream per handle.  Round 6, defect (a): the host SIGSEGV after mid-life hipGraphExecDestroy needs the HISTORY of
// a pytest process (tests/test_gpu_configs.py as of round 5: ~60 handles, ~100 destroyed graphs); replayed here the same sequence
// of sga_create / sga_encode / sga_step_grads / sga_run / sga_bb_run / sga_destroy can be put under rocgdb, valgrind or the
// `make ASAN=1` host build, which a PyTorch process cannot (DESIGN_EXPERIMENTS.md A.8).
//
//   hipcc -O1 -g -I include tests/c_client/sga_replay.cpp -o sga_replay -ldl        (the library is dlopen'ed: SGA_LIB or argv[2])
//   SGA_GRAPH_DROP=destroy ./sga_replay calls.txt [libsga_hip.so] [-v]
//
// Data: weights are generated here with the scales of make_synthetic_weights (sga_amd/weights.py) -- the VALUES do not matter for
// a crash hunt, the launch sequence does; every pointer argument of a call is served from its own pre-filled 48 MiB device slot
// (values in [0, 1): valid images, latents, uniforms).  A handle argument is `h<n>` = the n-th sga_create of the log.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "sga_hip.h"

#define HIPOK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); exit(2); } } while (0)

namespace {

void* g_lib = nullptr;
template <typename F> F sym(const char* name) {
  void* p = dlsym(g_lib, name);
  if (!p) { fprintf(stderr, "missing symbol %s\n", name); exit(2); }
  return reinterpret_cast<F>(p);
}

struct Weights {
  std::vector<std::vector<float>> store;
  sga_weights w;
};

const float* tensor(Weights& W, std::mt19937& rng, size_t n, float scale, float mean = 0.f, bool uniform = false) {
  std::vector<float> v(n);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  for (size_t i = 0; i < n; ++i) v[i] = mean + scale * (uniform ? ud(rng) : nd(rng));
  W.store.push_back(std::move(v));
  return W.store.back().data();
}

// effective tensors with the shapes of layer_shapes() and the scales of make_synthetic_weights (sga_amd/weights.py)
Weights* make_weights(int C, bool bb) {
  Weights* W = new Weights();
  W->store.reserve(64);
  memset(&W->w, 0, sizeof(W->w));
  std::mt19937 rng(1234u + (unsigned)C + (bb ? 7u : 0u));
  const int C15 = (int)(C * 1.5), hao = bb ? 2 * C : C;
  auto conv = [&](int kh, int kw, int ci, int co, float gain, bool deconv) {
    const float fan = (float)(kh * kw * ci) / (deconv ? 4.f : 1.f);
    return tensor(*W, rng, (size_t)kh * kw * ci * co, gain / std::sqrt(fan));
  };
  auto gdn = [&](const float** beta, const float** gamma) {
    *beta = tensor(*W, rng, (size_t)C, 0.1f, 1.0f, true);
    std::vector<float> g((size_t)C * C);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    for (int i = 0; i < C; ++i)
      for (int j = 0; j < C; ++j) g[(size_t)i * C + j] = (i == j ? 0.1f : 0.f) + (0.2f / C) * ud(rng);
    W->store.push_back(std::move(g));
    *gamma = W->store.back().data();
  };
  const float gag[4] = {3.f, 2.f, 2.f, 2.f}, gsg[4] = {0.27f, 1.f, 1.f, 0.15f};
  for (int k = 0; k < 4; ++k) {
    W->w.ga_kernel[k] = conv(5, 5, k == 0 ? 3 : C, C, gag[k], false);
    W->w.ga_bias[k] = tensor(*W, rng, (size_t)C, k == 3 ? 0.3f : 0.05f);
    W->w.gs_kernel[k] = conv(5, 5, C, k == 3 ? 3 : C, gsg[k], true);
    W->w.gs_bias[k] = tensor(*W, rng, (size_t)(k == 3 ? 3 : C), k == 3 ? 0.02f : 0.05f, k == 3 ? 0.5f : 0.f);
  }
  for (int k = 0; k < 3; ++k) { gdn(&W->w.ga_beta[k], &W->w.ga_gamma[k]); gdn(&W->w.gs_beta[k], &W->w.gs_gamma[k]); }
  W->w.ha_kernel[0] = conv(3, 3, C, C, 1.f, false); W->w.ha_bias[0] = tensor(*W, rng, (size_t)C, 0.05f);
  W->w.ha_kernel[1] = conv(5, 5, C, C, 1.4f, false); W->w.ha_bias[1] = tensor(*W, rng, (size_t)C, 0.05f);
  W->w.ha_kernel[2] = conv(5, 5, C, hao, bb ? 0.5f : 3.f, false);      // bb: small, so that exp(0.5 * logvar) stays finite
  W->w.hs_kernel[0] = conv(5, 5, C, C, 1.f, true); W->w.hs_bias[0] = tensor(*W, rng, (size_t)C, 0.1f, 0.1f);
  W->w.hs_kernel[1] = conv(5, 5, C, C15, 1.4f, true); W->w.hs_bias[1] = tensor(*W, rng, (size_t)C15, 0.1f, 0.1f);
  W->w.hs_kernel[2] = conv(3, 3, C15, 2 * C, 0.5f, false); W->w.hs_bias[2] = tensor(*W, rng, (size_t)2 * C, 0.05f, 0.15f);
  const int dims[5] = {1, 3, 3, 3, 1};
  for (int k = 0; k < 4; ++k) {
    const float init = std::log(std::expm1(1.f / std::pow(10.f, 0.25f) / (float)dims[k + 1]));
    const float sp = std::log1p(std::exp(init));      // softplus(init)
    W->w.eb_matrix[k] = tensor(*W, rng, (size_t)C * dims[k + 1] * dims[k], 0.02f, sp);
    W->w.eb_bias[k] = tensor(*W, rng, (size_t)C * dims[k + 1], 1.f, -0.5f, true);
    if (k < 3) W->w.eb_factor[k] = tensor(*W, rng, (size_t)C * dims[k + 1], 0.25f);
  }
  return W;
}

struct Handle { sga_handle* h = nullptr; hipStream_t st = nullptr; int C = 0; };

constexpr size_t kSlotBytes = 48u << 20;
constexpr int kSlots = 24;
char* g_pool = nullptr;
void* slot(int k) { return g_pool + (size_t)k * kSlotBytes; }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s calls.txt [libsga_hip.so] [-v]\n", argv[0]); return 1; }
  bool verbose = false;
  const char* libpath = getenv("SGA_LIB");
  for (int i = 2; i < argc; ++i) { if (!strcmp(argv[i], "-v")) verbose = true; else libpath = argv[i]; }
  if (!libpath) libpath = "improving-inference-for-neural-image-compression_amd/libsga_hip.so";
  g_lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
  if (!g_lib) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 2; }
  auto p_create = sym<int (*)(sga_handle**, const sga_config*, const sga_weights*)>("sga_create");
  auto p_destroy = sym<int (*)(sga_handle*)>("sga_destroy");
  auto p_last_error = sym<int (*)(const sga_handle*, char*, int)>("sga_last_error");
  auto p_latent = sym<int (*)(const sga_handle*, int, int, int*, int*, int*, int*)>("sga_latent_shape");
  auto p_set_bound = sym<int (*)(sga_handle*, float)>("sga_set_scale_bound");
  auto p_set_relax = sym<int (*)(sga_handle*, int, int)>("sga_set_relaxation");
  auto p_encode = sym<int (*)(sga_handle*, const float*, int, int, int, float*, float*, void*)>("sga_encode");
  auto p_step = sym<int (*)(sga_handle*, const float*, int, int, int, const float*, const float*, float, float, float, uint64_t, uint32_t,
                            const float*, const float*, float*, float*, float*, float*, void*)>("sga_step_grads");
  auto p_run = sym<int (*)(sga_handle*, const float*, int, int, int, float, float, int, double, double, int, double, uint64_t, const float*,
                           const float*, float*, float*, float*, float*, void*)>("sga_run");
  auto p_run_begin = sym<int (*)(sga_handle*, const float*, int, int, int, float, float, int, double, double, int, double, uint64_t,
                                 const float*, const float*, void*)>("sga_run_begin");
  auto p_run_steps = sym<int (*)(sga_handle*, int, void*)>("sga_run_steps");
  auto p_run_state = sym<int (*)(sga_handle*, int, float*, float*, float*, void*)>("sga_run_state");
  auto p_eval = sym<int (*)(sga_handle*, const float*, int, int, int, const float*, const float*, float*, float*, void*)>("sga_eval");
  auto p_base = sym<int (*)(sga_handle*, const float*, int, int, int, const float*, float*, float*, float*, void*)>("sga_base_compress");
  auto p_base_b = sym<int (*)(sga_handle*, const float*, int, int, int, const float*, float, float*, float*, float*, void*)>("sga_base_compress_bound");
  auto p_bb_init = sym<int (*)(sga_handle*, const float*, int, int, int, float*, void*)>("sga_bb_init_z");
  auto p_bb_step = sym<int (*)(sga_handle*, const float*, int, int, int, const float*, const float*, float, float, float, uint64_t, uint32_t,
                               const float*, const float*, int, float*, float*, float*, float*, void*)>("sga_bb_step_grads");
  auto p_bb_run = sym<int (*)(sga_handle*, const float*, int, int, int, float, float, int, int, double, double, double, int, double, uint64_t,
                              float*, float*, float*, float*, float*, void*)>("sga_bb_run");
  auto p_bb_refine = sym<int (*)(sga_handle*, const float*, int, int, int, float, int, double, uint64_t, float*, void*)>("sga_bb_refine");
  auto p_bb_eval = sym<int (*)(sga_handle*, const float*, int, int, int, const float*, const float*, const float*, uint64_t, float*, void*)>("sga_bb_eval");
  auto p_counter = sym<int (*)(const sga_handle*, int, long long*)>("sga_debug_counter");
  auto p_fork = sym<int (*)(const sga_handle*, char*, int)>("sga_get_fork_point");
  auto p_prof_begin = sym<int (*)(sga_handle*)>("sga_profile_begin");
  auto p_prof_end = sym<int (*)(sga_handle*, sga_kernel_stat*, int, int*)>("sga_profile_end");

  HIPOK(hipSetDevice(0));
  HIPOK(hipMalloc((void**)&g_pool, kSlotBytes * kSlots));
  {      // every slot: values in [0, 1)
    std::vector<float> host(kSlotBytes / 4);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> ud(1e-3f, 0.999f);
    for (auto& v : host) v = ud(rng);
    for (int k = 0; k < kSlots; ++k) HIPOK(hipMemcpy(slot(k), host.data(), kSlotBytes, hipMemcpyHostToDevice));
  }
  std::map<std::pair<int, int>, Weights*> wcache;
  std::vector<Handle> handles;
  FILE* f = fopen(argv[1], "r");
  if (!f) { perror(argv[1]); return 1; }
  char line[1024];
  long nline = 0, nskipped = 0;
  while (fgets(line, sizeof(line), f)) {
    ++nline;
    std::istringstream is(line);
    std::string name;
    is >> name;
    std::vector<std::string> a;
    for (std::string t; is >> t;) a.push_back(t);
    if (verbose) { fprintf(stderr, "[%ld] %s", nline, line); }
    auto I = [&](size_t k) { return k < a.size() ? atoi(a[k].c_str()) : 0; };
    auto D = [&](size_t k) { return k < a.size() ? atof(a[k].c_str()) : 0.0; };
    auto U = [&](size_t k) { return k < a.size() ? strtoull(a[k].c_str(), nullptr, 10) : 0ull; };
    // pointer argument k: its own slot when the log says non-null
    auto P = [&](size_t k) -> float* { return (k < a.size() && a[k] == "P") ? (float*)slot((int)(k % kSlots)) : nullptr; };
    auto H = [&]() -> Handle* {
      if (a.empty() || a[0].size() < 2 || a[0][0] != 'h') return nullptr;
      const size_t id = (size_t)atoi(a[0].c_str() + 1);
      return id < handles.size() && handles[id].h ? &handles[id] : nullptr;
    };
    int rc = 0;
    if (name == "sga_create") {
      sga_config cfg;
      memset(&cfg, 0, sizeof(cfg));
      cfg.num_filters = I(0); cfg.max_batch = I(1); cfg.max_height = I(2); cfg.max_width = I(3);
      cfg.bits_back = I(4); cfg.precision = I(5); cfg.scale_bound = (float)D(6);
      auto key = std::make_pair(cfg.num_filters, cfg.bits_back);
      if (!wcache.count(key)) wcache[key] = make_weights(cfg.num_filters, cfg.bits_back != 0);
      Handle h;
      h.C = cfg.num_filters;
      HIPOK(hipStreamCreateWithFlags(&h.st, hipStreamNonBlocking));
      rc = p_create(&h.h, &cfg, &wcache[key]->w);
      if (rc != 0) { fprintf(stderr, "line %ld: sga_create -> %d\n", nline, rc); return 3; }
      handles.push_back(h);
      continue;
    }
    Handle* h = H();
    if (!h) { ++nskipped; continue; }      // a call on a handle this replay does not have (or a handle-less entry point)
    void* st = (void*)h->st;
    if (name == "sga_destroy") { HIPOK(hipStreamSynchronize(h->st)); rc = p_destroy(h->h); h->h = nullptr; HIPOK(hipStreamDestroy(h->st)); }
    else if (name == "sga_latent_shape") { int q[4]; rc = p_latent(h->h, I(1), I(2), q, q + 1, q + 2, q + 3); }
    else if (name == "sga_set_scale_bound") rc = p_set_bound(h->h, (float)D(1));
    else if (name == "sga_set_relaxation") rc = p_set_relax(h->h, I(1), I(2));
    else if (name == "sga_encode") rc = p_encode(h->h, P(1), I(2), I(3), I(4), P(5), P(6), st);
    else if (name == "sga_step_grads")
      rc = p_step(h->h, P(1), I(2), I(3), I(4), P(5), P(6), (float)D(7), (float)D(8), (float)D(9), U(10), (uint32_t)U(11), P(12), P(13), P(14),
                  P(15), P(16), P(17), st);
    else if (name == "sga_run")
      rc = p_run(h->h, P(1), I(2), I(3), I(4), (float)D(5), (float)D(6), I(7), D(8), D(9), I(10), D(11), U(12), P(13), P(14), P(15), P(16),
                 P(17), P(18), st);
    else if (name == "sga_run_begin")
      rc = p_run_begin(h->h, P(1), I(2), I(3), I(4), (float)D(5), (float)D(6), I(7), D(8), D(9), I(10), D(11), U(12), P(13), P(14), st);
    else if (name == "sga_run_steps") rc = p_run_steps(h->h, I(1), st);
    else if (name == "sga_run_state") rc = p_run_state(h->h, I(1), P(2), P(3), P(4), st);
    else if (name == "sga_eval") rc = p_eval(h->h, P(1), I(2), I(3), I(4), P(5), P(6), P(7), P(8), st);
    else if (name == "sga_base_compress") rc = p_base(h->h, P(1), I(2), I(3), I(4), P(5), P(6), P(7), P(8), st);
    else if (name == "sga_base_compress_bound") rc = p_base_b(h->h, P(1), I(2), I(3), I(4), P(5), (float)D(6), P(7), P(8), P(9), st);
    else if (name == "sga_bb_init_z") rc = p_bb_init(h->h, P(1), I(2), I(3), I(4), P(5), st);
    else if (name == "sga_bb_step_grads")
      rc = p_bb_step(h->h, P(1), I(2), I(3), I(4), P(5), P(6), (float)D(7), (float)D(8), (float)D(9), U(10), (uint32_t)U(11), P(12), P(13), I(14),
                     P(15), P(16), P(17), P(18), st);
    else if (name == "sga_bb_run")
      rc = p_bb_run(h->h, P(1), I(2), I(3), I(4), (float)D(5), (float)D(6), I(7), I(8), D(9), D(10), D(11), I(12), D(13), U(14), P(15), P(16),
                    P(17), P(18), P(19), st);
    else if (name == "sga_bb_refine") rc = p_bb_refine(h->h, P(1), I(2), I(3), I(4), (float)D(5), I(6), D(7), U(8), P(9), st);
    else if (name == "sga_bb_eval") rc = p_bb_eval(h->h, P(1), I(2), I(3), I(4), P(5), P(6), P(7), U(8), P(9), st);
    else if (name == "sga_debug_counter") { long long v = 0; rc = p_counter(h->h, I(1), &v); }
    else if (name == "sga_get_fork_point") { char b[32]; rc = p_fork(h->h, b, 32); }
    else if (name == "sga_profile_begin") rc = p_prof_begin(h->h);
    else if (name == "sga_profile_end") { static sga_kernel_stat ks[256]; int n = 0; rc = p_prof_end(h->h, ks, 256, &n); }
    else if (name == "sga_last_error") { char b[256]; (void)p_last_error(h->h, b, 256); }
    else { ++nskipped; continue; }
    if (rc != 0) {
      char msg[256] = {0};
      if (h->h) (void)p_last_error(h->h, msg, sizeof(msg));
      fprintf(stderr, "line %ld: %s -> %d %s\n", nline, name.c_str(), rc, msg);
    }
    // the Python host synchronises when it reads results back (metrics_to_dict, .cpu()): after every call that returns data
    if (name == "sga_run" || name == "sga_bb_run" || name == "sga_step_grads" || name == "sga_bb_step_grads" || name == "sga_eval" ||
        name == "sga_encode" || name == "sga_base_compress" || name == "sga_base_compress_bound" || name == "sga_run_state")
      if (h->h) HIPOK(hipStreamSynchronize(h->st));
  }
  fclose(f);
  for (auto& h : handles) if (h.h) { (void)p_destroy(h.h); (void)hipStreamDestroy(h.st); }
  HIPOK(hipDeviceSynchronize());
  printf("replayed %ld lines (%ld skipped), %zu handles: no crash\n", nline, nskipped, handles.size());
  return 0;
}
