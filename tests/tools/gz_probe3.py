"""GPU box: the hyper-synthesis data-gradients (hs2.bwd, hs1.bwd, hs0.bwd) as single operators at the B = 4, 256x256 latent
shapes, split-K target swept through the laboratory build's SGA_MAIN_TARGET, vs float64 autograd: which layer / split
produces the 4.8e-4 gz deviation of the step?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from oracle.sga_oracle import SGAOracle
from sga_amd.codec import SGACodec
C, B = 192, int(os.environ.get("B", 4))
w = sga_amd.make_synthetic_weights(C, seed=0)
o64 = SGAOracle(w, dtype=torch.float64)
shapes = {"HS2": (16, 16, 288), "HS1": (8, 8, 192), "HS0": (4, 4, 192)}
for layer, (Hi, Wi, ci) in shapes.items():
    rng = np.random.RandomState(3)
    x = rng.standard_normal((B, Hi, Wi, ci)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    out = o64.layer_fwd(layer, xt)
    g_out = rng.standard_normal(tuple(out.shape)).astype(np.float32)
    (want,) = torch.autograd.grad(out, xt, torch.tensor(g_out, dtype=torch.float64))
    want = want.numpy(); mx = np.abs(want).max()
    for tgt in [int(sys.argv[1])]:      # SGA_MAIN_TARGET is read once per process: one process per target
        for extra in ["", "SGA_BN96_AS_192=0"]:
            os.environ["SGA_MAIN_TARGET"] = str(tgt)
            if extra: os.environ["SGA_BN96_AS_192"] = "0"
            c = SGACodec(w, C, B, 256, 256, lab=True)
            c.profile_begin()
            got = c.layer_bwd(layer, x, g_out).cpu().numpy().astype(np.float64)
            st = c.profile_end()
            e = np.abs(got - want) / mx
            names = [k["name"].strip() for k in st if "<" in k["name"]]
            print("%s target %4d %-18s err per image %s  rows>5e-5: %s  %s" % (
                layer, tgt, extra, " ".join("%.1e" % e[b].max() for b in range(B)),
                np.nonzero((e.reshape(-1, ci) > 5e-5).any(1))[0][:12].tolist(), names[-1] if names else ""), flush=True)
            c.close()
            os.environ.pop("SGA_BN96_AS_192", None)
