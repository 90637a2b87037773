"""Regression harness for "defect (a)" (DESIGN_EXPERIMENTS.md A.13), run as a SUBPROCESS by tests/test_gpu_graph_streams.py.

The HIP runtime bundled with PyTorch-ROCm 2.10 (ROCm 7.0.2) overruns a graph's internal stream list inside hipGraphLaunch when ALL
of the graph's internal streams share the LAUNCH stream's hardware queue; whether they do depends on which streams the process has
created and destroyed before (a new stream takes the least-referenced hardware queue of its priority class).  This harness builds
such histories on purpose -- the pattern of scripts/r06/graph_stream_collision_repro.hip, which kills 7 of 64 masks with a
normal-priority launch stream -- in a process that has PyTorch AND libsga_hip in it: for every mask, six normal-priority ballast
streams are created and used, the masked ones destroyed, and a fresh handle runs 110 iterations (three fork-point candidates are
captured, instantiated, launched, two destroyed; the winner replayed).  libsga_hip launches its graphs on a stream of another
priority class (sga_handle::sG), so every mask must survive and give the first mask's result bit for bit.

    MALLOC_PERTURB_=165 python tests/tools/graph_stream_stress.py [--control]

--control: the laboratory build with SGA_LAUNCH_STREAM=caller = the behaviour of rounds 1-5 (graphs on the caller's stream): on the
7.0.2 runtime this is EXPECTED TO DIE with SIGSEGV (scripts/r06/s05_*.sh runs it as the control; the test suite does not)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
control = "--control" in sys.argv
if control:
    os.environ["SGA_LAUNCH_STREAM"] = "caller"
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sga_amd  # noqa: E402
from sga_amd.codec import SGACodec  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so.7")      # resolves to the runtime already in the process (PyTorch's)
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipDeviceSynchronize.argtypes = []


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> hipError %d" % (what, rc))


C, B, H, W, NBAL = 64, 1, 64, 64, 6
w = sga_amd.make_synthetic_weights(C, seed=0)
x = np.random.RandomState(3).rand(B, H, W, 3).astype(np.float32)
scratch = torch.zeros(1024, device="cuda")
ref = None
masks = range(64) if "--quick" not in sys.argv else (0, 10, 34, 40, 42, 43, 46, 58, 63)
for mask in masks:
    bal = []
    for _ in range(NBAL):
        s = ctypes.c_void_p()
        chk(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1), "hipStreamCreateWithFlags")      # hipStreamNonBlocking
        chk(hip.hipMemsetAsync(ctypes.c_void_p(scratch.data_ptr()), 0, 256, s), "hipMemsetAsync")      # first use: the stream gets its hardware queue
        bal.append(s)
    chk(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
    for k in range(NBAL):
        if mask & (1 << k):
            chk(hip.hipStreamDestroy(bal[k]), "hipStreamDestroy")
            bal[k] = None
    codec = SGACodec(w, C, B, H, W, lab=control)
    y_hat, z_hat, met, _ = codec.run(x, 0.01, its=110, seed=1)          # >= 100 iterations: the three fork-point candidates are timed
    again = codec.run(x, 0.01, its=30, seed=1)                          # the cached winner, launched again
    out = (y_hat.cpu(), z_hat.cpu(), again[0].cpu())
    if ref is None:
        ref = out
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2]), mask
    assert codec.counter("dropped") == 2, codec.counter("dropped")
    codec.close()
    for s in bal:
        if s is not None:
            chk(hip.hipStreamDestroy(s), "hipStreamDestroy")
    print("mask %2d ok" % mask, flush=True)
print("survived %d stream histories" % len(list(masks)), flush=True)
