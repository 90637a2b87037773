"""GPU box: the four-stage (DEEP) 64-row convolution instance (laboratory, SGA_DEEP64) against the default plan: another split,
so another float32 summation order -- gradients must agree to float32 rounding, runs must be reproducible."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
for (C, B, H, W) in [(192, 8, 256, 256), (192, 1, 256, 256), (192, 2, 200, 264), (192, 1, 512, 768)]:
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(3).rand(B, H, W, 3).astype(np.float32)
    os.environ["SGA_DEEP64"] = "0"; ref = SGACodec(w, C, B, H, W, lab=True)
    y, z = ref.encode(x)
    rb = ref.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    for mode in ("1", "2", "3"):
        os.environ["SGA_DEEP64"] = mode
        c = SGACodec(w, C, B, H, W, lab=True)
        y2, z2 = c.encode(x)
        ra = c.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
        ey = float((ra["gy"] - rb["gy"]).abs().max() / rb["gy"].abs().max()); ez = float((ra["gz"] - rb["gz"]).abs().max() / rb["gz"].abs().max())
        a1 = c.run(x, 0.01, its=30, seed=1); a2 = c.run(x, 0.01, its=30, seed=1)
        print(C, B, H, W, "DEEP64 =", mode, "enc", float((y2 - y).abs().max() / y.abs().max()), float((z2 - z).abs().max() / z.abs().max()),
              "gy", ey, "gz", ez, "loss", abs(ra["rd_loss"] / rb["rd_loss"] - 1), "reproducible", torch.equal(a1[0], a2[0]), flush=True)
        c.close()
    ref.close()
