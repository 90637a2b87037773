"""GPU box: us per SGA iteration (bench shape, or C/B/H/W from the environment; graph replay) under different environment knobs.
usage: python scripts/env_sweep.py "" "SGA_GRAPH_UNROLL=4" "A=1;B=2" ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = (int(os.environ.get(k, d)) for k, d in (('C', 192), ('B', 8), ('H', 256), ('W', 256)))
c = SGACodec(sga_amd.make_synthetic_weights(C, 0), C, B, H, W)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).cuda()
y, z = c.encode(x)
c.run(x, 0.01, its=40, metrics=False); torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    t = time.time(); r = c.run(x, 0.01, its=400, seed=7, metrics=False); torch.cuda.synchronize()
    best = min(best, (time.time() - t) / 400)
print("%%.1f %%s" %% (best * 1e6, float(r[0].double().sum())))
''' % ROOT
for cfg in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in filter(None, cfg.split(";")):
        k, v = kv.split("=")
        env[k] = v
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-48s us/it, checksum:" % (cfg or "(default)"), (out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]), out.stderr.strip()[-200:] if os.environ.get("SWEEP_STDERR") else "", flush=True)
