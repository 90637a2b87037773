"""GPU box: does confining the hyper branch's stream to a subset of CUs (hipExtStreamCreateWithCUMask) help the main chain?
A CU mask is a stream property and is not inherited by the kernel nodes of a replayed hipGraph, so the comparison is
between EAGER two-stream runs (SGA_NO_GRAPH=1): unmasked side stream vs masks of 16 / 32 / 64 / 128 CUs, in two layouts
(the low bits, and every k-th bit).  Prints us per SGA iteration at the bench shape; the graph replay is the reference line."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = 192, 8, 256, 256
codec = SGACodec(sga_amd.make_synthetic_weights(C, 0), C, B, H, W)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).cuda()
codec.run(x, 0.01, its=100, metrics=False); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.time(); codec.run(x, 0.01, its=400, metrics=False); torch.cuda.synchronize()
    best = min(best, (time.time() - t) / 400)
print("%%.1f" %% (best * 1e6))
''' % ROOT


def run(env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-c", CODE], env=e, capture_output=True, text=True).stdout.strip()


def words(bits):
    w = [0] * 8
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return ",".join("%x" % v for v in w)


print("graph replay (production)        ", run({}))
for n, stride in ((16, 16), (24, 0), (32, 8), (48, 0), (64, 4)):
    bits = list(range(0, 256, stride)) if stride else [32 * x + (32 // (n // 8)) * j for x in range(8) for j in range(n // 8)]
    print("hybrid (graph main + eager masked side), %d CUs spread" % len(bits), run({"SGA_SIDE_CU_MASK": words(bits)}))
print("eager, two streams, no mask      ", run({"SGA_NO_GRAPH": "1"}))
print("eager, single stream             ", run({"SGA_NO_GRAPH": "1", "SGA_NO_OVERLAP": "1"}))
for n in (16, 32, 64, 128):
    print("eager, side stream on CUs 0..%-3d  " % (n - 1), run({"SGA_NO_GRAPH": "1", "SGA_SIDE_CU_MASK": words(range(n))}))
    print("eager, side stream on every %d-th CU" % (256 // n), run({"SGA_NO_GRAPH": "1", "SGA_SIDE_CU_MASK": words(range(0, 256, 256 // n))}))
