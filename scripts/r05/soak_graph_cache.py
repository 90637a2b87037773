"""GPU box, round 5: soak of ONE handle under a service's pattern -- complete 2000-iteration runs alternating between geometries
(cfg 2 B = 8, B = 3 ragged, B = 1; one 512 x 768 image; relaxation and sigma-bound toggles), ~10 minutes.  Every geometry's first
result is the reference for its later repeats (same seed -> BIT-equal latents and metrics); the graph cache must stop capturing after
the first pass; nothing may go non-finite.  Writes gpurun_out/r05_soak_graph_cache.txt."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import sga_amd
from sga_amd.codec import SGACodec

C = 192
MINUTES = float(sys.argv[1]) if len(sys.argv) > 1 else 9.0
PRECISION = sys.argv[2] if len(sys.argv) > 2 else "f32"
codec = SGACodec(sga_amd.make_synthetic_weights(C, seed=0), C, 8, 512, 768, precision=PRECISION)
rng = np.random.RandomState(0)
imgs = {(8, 256, 256): rng.rand(8, 256, 256, 3).astype(np.float32), (3, 256, 256): rng.rand(3, 256, 256, 3).astype(np.float32),
        (1, 256, 256): rng.rand(1, 256, 256, 3).astype(np.float32), (1, 512, 768): rng.rand(1, 512, 768, 3).astype(np.float32)}
plan = [((8, 256, 256), "sga", 0.0), ((1, 512, 768), "sga", 0.0), ((3, 256, 256), "sga", 0.0), ((8, 256, 256), "sga", 0.11),
        ((1, 256, 256), "sga", 0.0), ((8, 256, 256), "unoise", 0.0), ((1, 512, 768), "sga", 0.11)]
out = open("gpurun_out/r05_soak_graph_cache%s.txt" % ("" if PRECISION == "f32" else "_" + PRECISION), "w")
def say(s):
    print(s); out.write(s + "\n"); out.flush()
first, runs, mism, t0 = {}, 0, 0, time.time()
captures_after_first_pass = None
while time.time() - t0 < 60 * MINUTES:
    for key in plan:
        shape, relax, bound = key
        codec.set_relaxation(relax, "exp0")
        codec.set_scale_bound(bound)
        y, z, met, _ = codec.run(imgs[shape], 0.01, its=2000, seed=11)
        res = (y.cpu().numpy(), z.cpu().numpy(), met.cpu().numpy())
        assert np.isfinite(res[2][:, [0, 1, 4]]).all(), (key, runs)
        if key not in first:
            first[key] = res
        elif not all(np.array_equal(a, b) for a, b in zip(first[key], res)):
            mism += 1
            say("MISMATCH at run %d %s" % (runs, key))
        runs += 1
    if captures_after_first_pass is None:
        captures_after_first_pass = codec.counter("captures")
    say("%.0f s: %d complete runs, captures %d (after the first pass: %d), cached %d, evictions %d, dropped %d, mismatches %d" %
        (time.time() - t0, runs, codec.counter("captures"), captures_after_first_pass, codec.counter("cached"),
         codec.counter("evictions"), codec.counter("dropped"), mism))
codec.set_relaxation("sga", "exp0")
say("soak (%s): %d complete 2000-iteration runs on one handle over %d (geometry, relaxation, bound) keys in %.0f s; repeats bit-equal: %s; "
    "captures after the first pass: %d" % (PRECISION, runs, len(plan), time.time() - t0, mism == 0, codec.counter("captures") - captures_after_first_pass))
codec.close()
sys.exit(1 if mism or codec is None else 0)
