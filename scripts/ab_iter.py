"""GPU box: A/B of environment switches on the SGA iteration time at the bench shape (cfg 2).
usage: python scripts/ab_iter.py [--rounds N] "" "SGA_X=1" "SGA_X=1 SGA_Y=0" ...   (each argument = one variant's environment)
LAB=1 in a variant's environment: the laboratory build (libsga_hip_lab.so) with its extra switches; B, H, W, C: the shape.
The variants run in separate processes, alternating, N rounds; prints us per iteration (best of 3 x 400 graph replays)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = int(os.environ.get("C", 192)), int(os.environ.get("B", 8)), int(os.environ.get("H", 256)), int(os.environ.get("W", 256))
codec = SGACodec(sga_amd.make_synthetic_weights(C, 0), C, B, H, W, lab=bool(os.environ.get("LAB")))
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).cuda()
codec.run(x, 0.01, its=100, metrics=False); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.time(); codec.run(x, 0.01, its=400, metrics=False); torch.cuda.synchronize()
    best = min(best, (time.time() - t) / 400)
print("%%.1f" %% (best * 1e6))
''' % ROOT


def run(envs):
    env = dict(os.environ)
    for kv in envs.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    return r.stdout.strip() or ("FAILED: " + r.stderr.strip()[-300:])


if __name__ == "__main__":
    args = sys.argv[1:]
    rounds = 2
    if args and args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    res = {a: [] for a in args}
    for _ in range(rounds):
        for a in args:
            res[a].append(run(a))
    for a in args:
        print("%-50s %s" % (a or "(default)", "  ".join(res[a])), flush=True)
