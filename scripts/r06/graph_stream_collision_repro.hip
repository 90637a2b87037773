// Stand-alone reproducer (no libsga_hip, no PyTorch) of the HIP-runtime defect behind "defect (a)" -- the host SIGSEGV inside
// hipGraphLaunch that rounds 3-5 saw after mid-life hipGraphExecDestroy calls (DESIGN_EXPERIMENTS.md A.8a / A.13).
//
// What the native backtrace of round 6 showed (profiles/r06_defect_a_native_backtrace.txt; libamdhip64.so of ROCm 7.0.2 as bundled
// with PyTorch 2.10.0+rocm7.0, offsets in that file):
//   hipGraphLaunch -> GraphExec::Run (+0xaf880) -> Graph::UpdateStreams (+0xaed90), which does, in effect,
//       streams_.resize(max_streams_);  streams_[0] = launch_stream;
//       for (i = 1, j = 0; i < streams_.size(); ++j)
//         if (queue_of(launch_stream) != queue_of(parallel_streams[j])) streams_[i++] = parallel_streams[j];
//   -- the loop bounds i, NOT j.  GraphExec::Init (+0xafe20) creates max_streams_ internal streams for a graph that needs
//   max_streams_ - 1 of them ("one spare"), so ONE internal stream that shares the launch stream's hardware queue is
//   tolerated; when ALL of them do, j runs past the end of parallel_streams_ and the next 8 bytes of heap are used as a
//   hip::Stream* (fault address 0x1a8 = NULL + offsetof(vdev) in the crash that was caught).
// When do all internal streams of a two-stream graph land on the launch stream's hardware queue?  A new stream takes the
// least-referenced of the GPU_MAX_HW_QUEUES hardware queues of its priority class.  A process that only CREATES streams keeps
// the reference counts within one of each other and two consecutive creations alternate; DESTROYING streams (what
// hipGraphExecDestroy does to a graph's internal streams, and hipStreamDestroy to any other) unbalances the counts, and
// the next graph's streams can both go to the emptier queue -- the launch stream's, if that is where it lives.
//
// This program makes that happen on purpose: a launch stream, a capture partner, NBAL ballast streams; the ballast streams
// selected by <mask> are destroyed; a two-stream graph (fork / join through events) is captured, instantiated and launched.
//   usage:  graph_stream_collision_repro <mask> [launch-stream priority: 0 normal (default) | -1 high]
//   run:    for m in $(seq 0 63); do MALLOC_PERTURB_=165 GPU_MAX_HW_QUEUES=2 ./repro $m 0; done     (scripts/r06/s04_*.sh)
// MALLOC_PERTURB_ makes the out-of-bounds slot non-zero garbage, so an overrun faults instead of depending on heap history.
// Expected: with a NORMAL-priority launch stream some masks die with SIGSEGV inside hipGraphLaunch; with a HIGH-priority launch
// stream (its own hardware-queue class: the internal streams are always created with normal priority) none does -- which is
// the work-around libsga_hip ships (sga_handle::sG).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_add(float* p, int n, float v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] += v;
}

int main(int argc, char** argv) {
  const unsigned mask = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 0) : 0u;
  const int prio = argc > 2 ? atoi(argv[2]) : 0;
  constexpr int NBAL = 6, N = 1 << 16;
  setenv("GPU_MAX_HW_QUEUES", "2", 0);      // what sga_amd/__init__.py sets (two hardware queues: main chain + hyper branch)
  CHK(hipSetDevice(0));
  float *a = nullptr, *b = nullptr;
  CHK(hipMalloc((void**)&a, N * sizeof(float)));
  CHK(hipMalloc((void**)&b, N * sizeof(float)));
  CHK(hipMemset(a, 0, N * sizeof(float)));
  CHK(hipMemset(b, 0, N * sizeof(float)));
  hipStream_t L = nullptr, S = nullptr, bal[NBAL];
  CHK(hipStreamCreateWithPriority(&L, hipStreamNonBlocking, prio));      // the launch stream (the caller's stream of sga_run)
  CHK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));               // capture partner (the handle's second stream)
  for (int k = 0; k < NBAL; ++k) CHK(hipStreamCreateWithFlags(&bal[k], hipStreamNonBlocking));
  // every stream gets its hardware queue when it is first used
  hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, L, a, N, 0.f);
  hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, S, b, N, 0.f);
  for (int k = 0; k < NBAL; ++k) hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, bal[k], b, N, 0.f);
  CHK(hipDeviceSynchronize());
  for (int k = 0; k < NBAL; ++k)
    if (mask & (1u << k)) { CHK(hipStreamDestroy(bal[k])); bal[k] = nullptr; }      // unbalance the hardware queues' reference counts
  hipEvent_t ef = nullptr, ej = nullptr;
  CHK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CHK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  hipGraph_t g = nullptr;
  hipGraphExec_t ex = nullptr;
  CHK(hipStreamBeginCapture(L, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, L, a, N, 1.f);
  CHK(hipEventRecord(ef, L));
  CHK(hipStreamWaitEvent(S, ef, 0));
  for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, L, a, N, 1.f);
  for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, S, b, N, 1.f);
  CHK(hipEventRecord(ej, S));
  CHK(hipStreamWaitEvent(L, ej, 0));
  hipLaunchKernelGGL(k_add, dim3(16), dim3(256), 0, L, a, N, 1.f);
  CHK(hipStreamEndCapture(L, &g));
  CHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));      // creates the graph's internal streams
  CHK(hipGraphDestroy(g));
  fprintf(stderr, "mask %2u prio %d: launching\n", mask, prio);
  for (int r = 0; r < 3; ++r) CHK(hipGraphLaunch(ex, L));     // Graph::UpdateStreams runs in here
  CHK(hipStreamSynchronize(L));
  float ha = 0.f, hb = 0.f;
  CHK(hipMemcpy(&ha, a, sizeof(float), hipMemcpyDeviceToHost));
  CHK(hipMemcpy(&hb, b, sizeof(float), hipMemcpyDeviceToHost));
  CHK(hipGraphExecDestroy(ex));
  printf("mask %2u prio %d: survived (a = %.0f, b = %.0f; expected 18, 12)\n", mask, prio, ha, hb);
  return (ha == 18.f && hb == 12.f) ? 0 : 3;
}
