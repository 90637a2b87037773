"""GPU box, round 5: the graph cache under EVICTION pressure -- 24 keys (B = 1..8 at three sizes) on a 16-entry cache, short complete
runs (300 iterations: enough for the timed fork choice), several passes, under the drop policy given by SGA_GRAPH_DROP (default
retire; `destroy` = hipGraphExecDestroy behind a synchronisation, the policy that crashed the round-3 / round-4 suites 3 times in 8).
Every key's result must repeat BIT-equal.  Appends to gpurun_out/r05_soak_evictions.txt."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import sga_amd
from sga_amd.codec import SGACodec
C = 192
MINUTES = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
codec = SGACodec(sga_amd.make_synthetic_weights(C, seed=0), C, 8, 320, 320)
rng = np.random.RandomState(0)
keys = [(b, s, s) for s in (192, 256, 320) for b in range(1, 9)]
imgs = {k: rng.rand(*k, 3).astype(np.float32) for k in keys}
out = open("gpurun_out/r05_soak_evictions.txt", "a")
def say(s):
    print(s); out.write(s + "\n"); out.flush()
first, runs, mism, t0, passes = {}, 0, 0, time.time(), 0
while time.time() - t0 < 60 * MINUTES:
    for k in keys:
        y, z, met, _ = codec.run(imgs[k], 0.01, its=300, seed=3)
        res = (y.cpu().numpy(), z.cpu().numpy(), met.cpu().numpy())
        if k not in first:
            first[k] = res
        elif not all(np.array_equal(a, b) for a, b in zip(first[k], res)):
            mism += 1
        runs += 1
    passes += 1
say("policy %s: %d passes over %d keys (%d runs of 300 iterations) in %.0f s: captures %d, cached %d, evictions %d, dropped %d, "
    "repeats bit-equal: %s" % ("destroy (the only policy since round 6)", passes, len(keys), runs, time.time() - t0, codec.counter("captures"),
                               codec.counter("cached"), codec.counter("evictions"), codec.counter("dropped"), mism == 0))
codec.close()
sys.exit(1 if mism else 0)
