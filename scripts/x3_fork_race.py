"""bf16x3 with the hyper branch on the second stream (SGA_X3_FORK=1): are identical runs identical?
Round 1 saw 7 different outcomes in 25 two-stream runs of this mode (DESIGN_EXPERIMENTS.md A.3) and made it single-stream.
Usage: SGA_X3_FORK=1 python scripts/x3_fork_race.py [runs] [its] [graph|eager]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
its = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if len(sys.argv) > 3 and sys.argv[3] == "eager":
    os.environ["SGA_NO_GRAPH"] = "1"
import numpy as np
import torch
import sga_amd
from sga_amd.codec import SGACodec

C, B, H, W = 192, 8, 256, 256
w = sga_amd.make_synthetic_weights(C, 0)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(0)).numpy()
for prec in ("bf16x3", "f32"):
    c = SGACodec(w, C, B, H, W, precision=prec)
    outs = {}
    first_bad = None
    for r in range(runs):
        y_hat, z_hat, met, tr = c.run(x, 0.01, its=its, seed=3, trace=True)
        torch.cuda.synchronize()
        tr = tr.cpu().numpy() if hasattr(tr, "cpu") else np.asarray(tr)
        key = hashlib.sha1(y_hat.cpu().numpy().tobytes() + z_hat.cpu().numpy().tobytes()).hexdigest()[:12]
        if key not in outs:
            outs[key] = (r, tr.copy())
        elif False:
            pass
    keys = list(outs)
    print(prec, "runs", runs, "its", its, "distinct outcomes", len(keys), flush=True)
    if len(keys) > 1:
        ref = outs[keys[0]][1]
        for k in keys[1:]:
            d = np.abs(outs[k][1] - ref).max(axis=1)
            it0 = int(np.argmax(d > 0)) if (d > 0).any() else -1
            print("  outcome", k, "first seen in run", outs[k][0], "trace first differs at iteration", it0)
    c.close()
