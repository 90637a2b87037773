"""GPU box: ms per iteration of the bits-back run (cfg 5: bb_sga.py, Kodak size) next to the plain SGA run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = 192, int(os.environ.get("B", 3)), 512, 768
w = sga_amd.make_synthetic_weights(C, 0, bb=True)
w["ha.k2"] = w["ha.k2"] * 0.05
x = torch.rand(B, H, W, 3).cuda()
c = SGACodec(w, C, B, H, W, bits_back=True)
c.bb_run(x, 0.01, its=20, r_its=20); torch.cuda.synchronize()
for (a, b) in ((150, 0), (0, 150)):
    t = time.time(); c.bb_run(x, 0.01, its=a, r_its=b); torch.cuda.synchronize()
    print("bb stage", 1 if a else 2, "ms/it", (time.time() - t) / 150 * 1e3, "(includes encode + eval once)")
c.close()
w = sga_amd.make_synthetic_weights(C, 0)
c = SGACodec(w, C, B, H, W)
c.run(x, 0.01, its=20, metrics=False); torch.cuda.synchronize()
t = time.time(); c.run(x, 0.01, its=150, metrics=False); torch.cuda.synchronize()
print("sga ms/it", (time.time() - t) / 150 * 1e3)
