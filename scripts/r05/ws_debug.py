"""GPU box: where do the persistent igdn2.bwd kernel and the tile kernel differ at (C, B, H, W)?  (laboratory build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = [int(a) for a in sys.argv[1:5]]
w = sga_amd.make_synthetic_weights(C, seed=0)
os.environ["SGA_IGDN_WS_SCHED"] = sys.argv[5] if len(sys.argv) > 5 else "dynamic"
os.environ["SGA_IGDN_WS"] = "2"; ws = SGACodec(w, C, B, H, W, lab=True)
os.environ["SGA_IGDN_WS"] = "0"; tile = SGACodec(w, C, B, H, W, lab=True)
x = np.random.RandomState(C + H).rand(B, H, W, 3).astype(np.float32)
y, z = tile.encode(x)
rb = tile.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
for rep in range(4):
    ra = ws.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    d = (ra["gy"] != rb["gy"])
    n = int(d.sum())
    print("rep", rep, "differing gy elements", n, "of", d.numel(), "max abs diff", float((ra["gy"] - rb["gy"]).abs().max()), "gy max", float(rb["gy"].abs().max()),
          "gz equal", bool(torch.equal(ra["gz"], rb["gz"])), "loss equal", ra["rd_loss"] == rb["rd_loss"])
    if n:
        idx = d.nonzero()[:5].tolist()
        print("  first at", idx, [float(ra["gy"][tuple(i)]) for i in idx], [float(rb["gy"][tuple(i)]) for i in idx])
        print("  images affected", sorted(set(d.nonzero()[:, 0].tolist())), "rows", sorted(set(d.nonzero()[:, 1].tolist()))[:20])
