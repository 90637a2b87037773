"""GPU box, round 5 experiment: does the cfg-2 batch run faster as TWO half-batches on two streams?

The iteration's ~300 us over the MFMA ideal are latency / occupancy of the 16^2 ... 64^2 stages.  Two handles at B = 4, each
replaying its own two-stream graph on its own stream, give the chip a second, independent chain to fill those gaps with (images
are independent: nothing in the optimisation couples them).  Measured here, per SGA iteration of 8 images:
  one handle B = 8 | one handle B = 4 (x 2 = run back to back) | two handles B = 4 interleaved on two streams (lock-step / free-running)
Results go to gpurun_out/r05_half_batch_pipeline.txt.
"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
import sga_amd
from sga_amd.codec import SGACodec

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
Cn, H, W = 192, 256, 256
weights = sga_amd.make_synthetic_weights(Cn, seed=0)
x = torch.rand(8, H, W, 3, generator=torch.Generator().manual_seed(1000)).to(dev)
out = open("gpurun_out/r05_half_batch_pipeline.txt", "w")


def say(*a):
    s = " ".join(str(v) for v in a)
    print(s)
    out.write(s + "\n")
    out.flush()


def steps(c, n):
    st = c.lib.sga_run_steps(c.handle, int(n), C.c_void_p(c.stream.cuda_stream))
    assert st == 0, st


def begin(c, xs, seed):
    c.run_begin(xs, 0.01, its=10000, seed=seed)
    c.run_steps(80)                    # captures the graph and times the fork point
    torch.cuda.synchronize()


def timed(fn, its):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / its * 1e6


ITS = 400
for precision in sys.argv[1:] or ["f32"]:
    c8 = SGACodec(weights, Cn, 8, H, W, device=dev, precision=precision)
    begin(c8, x, 1)
    t8 = min(timed(lambda: steps(c8, ITS), ITS) for _ in range(2))
    say(precision, "one handle  B=8 : %.1f us per iteration of 8 images  (fork %s)" % (t8, c8.fork_point()))
    c8.close()

    ca = SGACodec(weights, Cn, 4, H, W, device=dev, precision=precision)
    cb = SGACodec(weights, Cn, 4, H, W, device=dev, precision=precision)
    begin(ca, x[:4], 1)
    begin(cb, x[4:], 2)
    t4 = min(timed(lambda: steps(ca, ITS), ITS) for _ in range(2))
    say(precision, "one handle  B=4 : %.1f us per iteration of 4 images -> %.1f per 8  (fork %s)" % (t4, 2 * t4, ca.fork_point()))

    def lockstep():
        for _ in range(ITS):
            steps(ca, 1)
            steps(cb, 1)
    tl = min(timed(lockstep, ITS) for _ in range(2))
    say(precision, "two handles B=4, one graph launch each in turn : %.1f us per iteration of 8 images" % tl)

    def chunks():
        for _ in range(ITS // 20):
            steps(ca, 20)
            steps(cb, 20)
    tc = min(timed(chunks, ITS) for _ in range(2))
    say(precision, "two handles B=4, 20 launches each in turn       : %.1f us per iteration of 8 images" % tc)

    # half an iteration of offset: A runs ahead by one launch and B is held back by an event half-way?  Simplest skew: start B
    # after A has been given a head start of one iteration, then lock-step.
    def skewed():
        steps(ca, 1)
        for _ in range(ITS - 1):
            steps(cb, 1)
            steps(ca, 1)
        steps(cb, 1)
    ts = min(timed(skewed, ITS) for _ in range(2))
    say(precision, "two handles B=4, A one launch ahead             : %.1f us per iteration of 8 images" % ts)
    say(precision, "ratio two-stream / one handle B=8: lock-step %.3f, chunks %.3f, skewed %.3f" % (tl / t8, tc / t8, ts / t8))
    ca.close()
    cb.close()
