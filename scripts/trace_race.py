"""Locate the first SGA iteration at which two identical runs diverge, and in which quantity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = 192, 8, 256, 256
w = sga_amd.make_synthetic_weights(C, 0)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(0)).numpy()
its = int(os.environ.get("ITS", 2000)); reruns = int(os.environ.get("RERUNS", 16))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
c = SGACodec(w, C, B, H, W, precision=prec)
_, _, _, tr0 = c.run(x, 0.01, its=its, seed=0, trace=True)
tr0 = tr0.cpu().numpy()
for i in range(reruns):
    _, _, _, tr = c.run(x, 0.01, its=its, seed=0, trace=True)
    tr = tr.cpu().numpy()
    d = (tr != tr0)
    if d.any():
        it = int(np.argmax(d.any(1)))
        print("rerun", i, "first differing iteration", it, "fields [loss mse bpp psnr] differ:", d[it].tolist(),
              "rel dev:", np.array2string(np.abs(tr[it] - tr0[it]) / np.abs(tr0[it]), precision=2),
              "next it:", d[min(it + 1, its - 1)].tolist(), flush=True)
print("done")
