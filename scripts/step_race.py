"""Determinism probe for ONE SGA evaluation: identical calls must give identical gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = 192, 8, 256, 256
w = sga_amd.make_synthetic_weights(C, 0)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(0)).numpy()
for prec in sys.argv[1:] or ("f32", "bf16x3"):
    c = SGACodec(w, C, B, H, W, precision=prec)
    y, z = c.encode(x)
    ref = c.step_grads(x, y, z, 0.3, 0.01, seed=5, it=7)
    bad = {"gy": 0, "gz": 0}
    for i in range(40):
        got = c.step_grads(x, y, z, 0.3, 0.01, seed=5, it=7)
        for k in bad:
            d = (got[k] != ref[k])
            if d.any():
                bad[k] += 1
                if bad[k] <= 3:
                    idx = d.nonzero()
                    print(prec, k, "call", i, "ndiff", int(d.sum()), "first idx", idx[0].tolist(), "last idx", idx[-1].tolist(),
                          "max|d|", float((got[k] - ref[k]).abs().max()), "scale", float(ref[k].abs().max()))
    print(prec, "calls with different gradients out of 40:", bad, flush=True)
    c.close()
