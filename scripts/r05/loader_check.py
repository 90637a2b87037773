"""GPU box: the loader-wave form of the 64-row convolution instance (laboratory, SGA_DEEP64_KIND=2) against the plain instance: same
split, same stages, same MFMA order -- encode, gradients and complete runs must be BIT-identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
os.environ["SGA_DEEP64_KIND"] = "2"
bad = 0
for (C, B, H, W) in [(192, 8, 256, 256), (192, 1, 256, 256), (192, 2, 200, 264), (192, 1, 512, 768), (192, 3, 37, 41)]:
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(3).rand(B, H, W, 3).astype(np.float32)
    os.environ["SGA_DEEP64"] = "0"; ref = SGACodec(w, C, B, H, W, lab=True)
    y, z = ref.encode(x)
    rb = ref.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    r0 = ref.run(x, 0.01, its=40, seed=1)
    os.environ["SGA_DEEP64"] = "3"; c = SGACodec(w, C, B, H, W, lab=True)
    y2, z2 = c.encode(x)
    c.profile_begin(); c.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5); names = sorted(set(k["name"] for k in c.profile_end()))
    ra = c.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    r1 = c.run(x, 0.01, its=40, seed=1); r2 = c.run(x, 0.01, its=40, seed=1)
    ok = (torch.equal(y, y2) and torch.equal(z, z2) and torch.equal(ra["gy"], rb["gy"]) and torch.equal(ra["gz"], rb["gz"]) and
          torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1]) and torch.equal(r1[0], r2[0]))
    used = any(n.replace(" ", "").endswith(",0,5>") for n in names)
    print(C, B, H, W, "loader instance used:", used, " bit-identical:", ok, flush=True)
    bad += (not ok) + (not used)
    c.close(); ref.close()
print("FAILED" if bad else "ALL IDENTICAL")
