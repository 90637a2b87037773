"""Determinism probe: identical runs must give identical latents and metrics.
Prints how many distinct outcomes N identical runs produced (1 = reproducible)."""
import os, sys, hashlib, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = (int(v) for v in os.environ.get("SHAPE", "192,8,256,256").split(","))
w = sga_amd.make_synthetic_weights(C, 0)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(0)).numpy()
its = int(os.environ.get("ITS", 300)); reruns = int(os.environ.get("RERUNS", 16))
for prec in sys.argv[1:] or ("f32", "bf16x3"):
    c = SGACodec(w, C, B, H, W, precision=prec)
    lat, met = [], []
    for i in range(reruns + 1):
        y, z, m, _ = c.run(x, 0.01, its=its, seed=0)
        lat.append(hashlib.sha1(y.cpu().numpy().tobytes() + z.cpu().numpy().tobytes()).hexdigest()[:8])
        met.append(hashlib.sha1(m.cpu().numpy().tobytes()).hexdigest()[:8])
    cl, cm = collections.Counter(lat), collections.Counter(met)
    print(prec, f"its={its} runs={reruns + 1}: distinct latents {len(cl)} {sorted(cl.values(), reverse=True)}, "
          f"distinct metrics {len(cm)} {sorted(cm.values(), reverse=True)}; run0 in majority: {cl.most_common(1)[0][0] == lat[0]}", flush=True)
    c.close()
