"""GPU box: span of one kernel symbol INSIDE the graph replay (earliest workgroup entry to latest exit, sga_profile_graph_begin)
for a list of environment variants, alternating.  usage: python scripts/in_graph_stamp.py "" "SGA_XCD_REMAP=0" ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys
sys.path.insert(0, %r)
import torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = 192, 8, 256, 256
codec = SGACodec(sga_amd.make_synthetic_weights(C, 0), C, B, H, W)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).cuda()
codec.run(x, 0.01, its=50, metrics=False)
codec.profile_graph_begin(os.environ.get("SYM", "conv_mfma_kernel<2,3,4,2,0,false,0,1>"))
codec.run(x, 0.01, its=300, metrics=False)
g = codec.profile_graph_end()
print("%%.1f" %% (1e3 * g["ms_total"] / max(g["launches"], 1)))
''' % ROOT


def run(envs):
    env = dict(os.environ)
    for kv in envs.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    return r.stdout.strip() or ("FAILED: " + r.stderr.strip()[-300:])


if __name__ == "__main__":
    args = sys.argv[1:] or [""]
    res = {a: [] for a in args}
    for _ in range(3):
        for a in args:
            res[a].append(run(a))
    for a in args:
        print("%-40s %s" % (a or "(default)", "  ".join(res[a])), flush=True)
