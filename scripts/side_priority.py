"""GPU box: hardware-queue priorities for the two chains of an SGA iteration (DESIGN_EXPERIMENTS.md A.7).

Graph kernel nodes ignore stream priorities, so this uses the HYBRID replay (sga_api.hip: the main chain is a captured
graph launched on the caller's stream, the hyper branch is launched eagerly on the handle's side stream every iteration)
with SGA_HYBRID=1 (no CU mask), the side stream created with priority SGA_SIDE_PRIORITY (HIP: -1 high, 0 normal, 1 low)
and the caller's stream with SGA_MAIN_PRIORITY.  Prints us per SGA iteration at the bench shape."""
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
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True)
    return r.stdout.strip() or ("FAILED: " + r.stderr.strip()[-200:])


if __name__ == "__main__":
    print("graph replay (production)            ", run({}), flush=True)
    print("hybrid, priorities main 0 / side 0   ", run({"SGA_HYBRID": "1"}), flush=True)
    for mp in ("0", "-1"):
        for sp in ("-1", "0", "1"):
            print("hybrid, main %2s / side %2s            " % (mp, sp),
                  run({"SGA_HYBRID": "1", "SGA_MAIN_PRIORITY": mp, "SGA_SIDE_PRIORITY": sp}), flush=True)
    print("hybrid, main chain only (SGA_SKIP_SIDE)", run({"SGA_HYBRID": "1", "SGA_SKIP_SIDE": "1"}), flush=True)
