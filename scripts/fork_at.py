"""GPU-box: ms per SGA iteration (bench shape, graph replay) as a function of where the hyper branch is forked."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, sga_amd
from sga_amd.codec import SGACodec
c = SGACodec(sga_amd.make_synthetic_weights(192, 0), 192, 8, 256, 256)
x = torch.rand(8, 256, 256, 3, generator=torch.Generator().manual_seed(1000)).cuda()
c.run(x, 0.01, its=30, metrics=False); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.time(); c.run(x, 0.01, its=200, metrics=False); torch.cuda.synchronize()
    best = min(best, (time.time() - t) / 200)
print("%%.1f" %% (best * 1e6))
''' % ROOT
for fa in sys.argv[1:] or [str(k) for k in range(0, 15)]:
    env = dict(os.environ, SGA_FORK_AT=fa)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("SGA_FORK_AT =", fa, "us/it:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
