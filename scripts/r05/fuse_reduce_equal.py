"""GPU box: the slab sum inside the convolution launch (SGA_FUSE_REDUCE, conv_mfma.hip) against the reduce launches: complete runs
must end in bit-identical latents -- the check for stale reads of another workgroup's slab (repeated, two-stream, several shapes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
bad = 0
for (C, B, H, W, its, prec) in [(192, 8, 256, 256, 300, "f32"), (192, 1, 256, 256, 300, "f32"), (64, 2, 64, 64, 400, "f32"), (128, 3, 37, 41, 300, "f32"),
                                (192, 1, 512, 768, 100, "f32"), (192, 8, 256, 256, 200, "bf16x3"), (256, 1, 200, 264, 100, "f32")]:
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(3).rand(B, H, W, 3).astype(np.float32)
    outs = {}
    for mode in ("0", "3"):
        os.environ["SGA_FUSE_REDUCE"] = mode
        c = SGACodec(w, C, B, H, W, lab=True, precision=prec)
        y0, z0 = c.encode(x)
        r = [c.run(x, 0.01, its=its, seed=s) for s in (1, 2, 1)]
        outs[mode] = (y0, z0, r)
        c.close()
    a, b = outs["0"], outs["3"]
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(p[0], q[0]) and torch.equal(p[1], q[1]) for p, q in zip(a[2], b[2]))
    rep = torch.equal(b[2][0][0], b[2][2][0])
    print(C, B, H, W, its, prec, "fused == separate:", ok, " fused reproducible:", rep, flush=True)
    bad += (not ok) + (not rep)
print("FAILED" if bad else "ALL EQUAL")
