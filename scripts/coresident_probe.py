"""GPU box: does a co-resident low-footprint wave find matrix-pipe time in the bubbles of the big convolution's K loop?
(DESIGN_EXPERIMENTS.md A.7.)  The 256-row convolution instance holds one 8-wave workgroup per CU (117 KB LDS, 176 VGPRs);
the filler (scripts/coresident_probe.hip: 4 waves, 32 KB LDS, register-only MFMAs) fits beside it.  Times, in us:
conv alone, filler alone, both started together on two streams.  "absorbed" = alone + alone - together."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "scripts", "coresident_probe.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "scripts", "coresident_probe.hip"), "-o", so], check=True)
import torch
import sga_amd
from sga_amd.codec import SGACodec

lib = C.CDLL(so)
lib.filler_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
Cn, B = 192, 8
codec = SGACodec(sga_amd.make_synthetic_weights(Cn, 0), Cn, B, 512, 512)
# GA1 forward on [B,128,128,C]: the stride-2 5x5 convolution of gs2.bwd's geometry (128 256-row tiles split in two) + a GDN launch
inp = torch.rand(B, 128, 128, Cn, device="cuda")
out = torch.empty(256 * 256 * 4, device="cuda")
s2 = torch.cuda.Stream()


def conv():
    codec.layer_fwd("GA1", inp)


def filler(n, grid, prio):
    lib.filler_launch(C.c_void_p(out.data_ptr()), grid, n, prio, C.c_void_p(s2.cuda_stream))


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        s2.synchronize(); torch.cuda.synchronize()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return best


for _ in range(3):
    conv()
t0 = timed(lambda: None)        # host-side synchronisation overhead inside the bracket
t_conv = timed(conv) - t0
print("conv + GDN alone                      %8.1f us" % t_conv)
for grid in (256, 512):
    for n in (400, 800, 1600):
        for prio in (0,):
            t_f = timed(lambda: filler(n, grid, prio)) - t0

            def both():
                s2.wait_stream(torch.cuda.current_stream())
                conv()              # launches on the codec's stream, returns after enqueue
                filler(n, grid, prio)
            t_b = timed(both) - t0
            print("filler grid %4d n %5d prio %d: alone %8.1f us, together %8.1f us, absorbed %6.1f us (%.0f %% of the filler)"
                  % (grid, n, prio, t_f, t_b, t_conv + t_f - t_b, 100 * (t_conv + t_f - t_b) / t_f))
