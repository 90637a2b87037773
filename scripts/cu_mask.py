"""GPU box: does confining the hyper branch's stream to a subset of CUs (hipExtStreamCreateWithCUMask, SGA_SIDE_CU_MASK)
help the main chain?  (DESIGN_EXPERIMENTS.md A.6; output: profiles/r03_cu_mask.txt)

A CU mask is a property of a stream's hardware queue and is not inherited by the kernel nodes of a replayed hipGraph, so
(1) EAGER two-stream runs (SGA_NO_GRAPH=1) compare an unmasked side stream with masks of 16..128 CUs, and
(2) the HYBRID replay (sga_api.hip: graph for the main chain, eager hyper branch on the masked stream; active whenever
    SGA_SIDE_CU_MASK is set and graphs are on) is swept over whole-XCD and partial masks, and run with the branch skipped
    (SGA_SKIP_SIDE=1: wrong results, timing only) to get the main chain's own time.
Mask bit i = XCD (i % 8), CU (i / 8) of that XCD on this chip.  Prints us per SGA iteration at the bench shape."""
import os
import subprocess
import sys

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
    return subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip()


def words(bits):
    w = [0] * 8
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return ",".join("%x" % v for v in w)


def xcd(x, ncu=32, first=0):
    return [8 * c + x for c in range(first, first + ncu)]


if __name__ == "__main__":
    print("graph replay (production)                 ", run({}))
    print("graph, single stream (SGA_NO_OVERLAP=1)   ", run({"SGA_NO_OVERLAP": "1"}))
    print("eager, two streams, no mask               ", run({"SGA_NO_GRAPH": "1"}))
    print("eager, single stream                      ", run({"SGA_NO_GRAPH": "1", "SGA_NO_OVERLAP": "1"}))
    for n in (16, 32, 64, 128):
        print("eager, side stream on mask bits 0..%-3d     " % (n - 1), run({"SGA_NO_GRAPH": "1", "SGA_SIDE_CU_MASK": words(range(n))}))
        print("eager, side stream on every %3d-th mask bit" % (256 // n), run({"SGA_NO_GRAPH": "1", "SGA_SIDE_CU_MASK": words(range(0, 256, 256 // n))}))
    allm = words(range(256))
    print("hybrid, all 256 CUs in the mask            ", run({"SGA_SIDE_CU_MASK": allm}))
    print("hybrid, main chain ONLY (branch skipped)   ", run({"SGA_SIDE_CU_MASK": allm, "SGA_SKIP_SIDE": "1"}))
    for x in range(8):
        print("hybrid, whole XCD %d                        " % x, run({"SGA_SIDE_CU_MASK": words(xcd(x))}))
    print("hybrid, 24 / 16 CUs of XCD 0               ", run({"SGA_SIDE_CU_MASK": words(xcd(0, 24))}), run({"SGA_SIDE_CU_MASK": words(xcd(0, 16))}))
    print("hybrid, XCD 0 + half of XCD 1 / XCDs 0 and 1", run({"SGA_SIDE_CU_MASK": words(xcd(0) + xcd(1, 16))}), run({"SGA_SIDE_CU_MASK": words(xcd(0) + xcd(1))}))
    for tgt in (96, 192, 256):
        print("hybrid, whole XCD 0, SGA_SIDE_TARGET=%-3d    " % tgt, run({"SGA_SIDE_CU_MASK": words(xcd(0)), "SGA_SIDE_TARGET": str(tgt)}))
