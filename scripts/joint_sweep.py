"""GPU box: JOINT sweep of the schedule / variant switches on the iteration time at cfg 2.  The earlier sweeps moved one knob at
a time around the production point; where the hyper branch's launches fall on the main chain decides +-50 us, so a kernel
variant that loses at the production fork point may win at another.  One process per configuration (several switches are
read once per process); prints us per iteration (best of 2 x 300 graph replays), sorted at the end."""
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, sga_amd
from sga_amd.codec import SGACodec
codec = SGACodec(sga_amd.make_synthetic_weights(192, 0), 192, 8, 256, 256)
x = torch.rand(8, 256, 256, 3, generator=torch.Generator().manual_seed(1000)).cuda()
codec.run(x, 0.01, its=60, metrics=False); torch.cuda.synchronize()
best = 1e9
for _ in range(2):
    t = time.time(); codec.run(x, 0.01, its=300, metrics=False); torch.cuda.synchronize()
    best = min(best, (time.time() - t) / 300)
print("%%.1f" %% (best * 1e6))
''' % ROOT

KNOBS = {
    "SGA_FUSED_POST64": ["0", "1"],
    "SGA_BN96_AS_192": ["0", "1"],
    "SGA_GS3_GEMM": ["0", "1"],
    "SGA_FORK_AT": ["0", "2", "3", "4"],
    "SGA_SIDE_TARGET": ["256", "512"],
    "SGA_REDUCE_BATCH": ["1", "2"],
}

if __name__ == "__main__":
    if len(sys.argv) > 1:      # python scripts/joint_sweep.py '{"SGA_FORK_AT": ["4", "5"], ...}'
        import json
        KNOBS = json.loads(sys.argv[1])
    names = list(KNOBS)
    res = []
    for vals in itertools.product(*[KNOBS[n] for n in names]):
        env = dict(os.environ, **dict(zip(names, vals)))
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        out = r.stdout.strip().splitlines()
        us = float(out[-1]) if out else float("nan")
        tag = " ".join("%s=%s" % (n.replace("SGA_", ""), v) for n, v in zip(names, vals))
        print("%8.1f  %s" % (us, tag), flush=True)
        res.append((us, tag))
    print("---- best 15")
    for us, tag in sorted(res)[:15]:
        print("%8.1f  %s" % (us, tag))
