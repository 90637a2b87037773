"""Is one layer's output bit-identical when another stream runs other layers at the same time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = 192, 8, 256, 256
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
w = sga_amd.make_synthetic_weights(C, 0)
c1 = SGACodec(w, C, B, H, W, precision=prec)
c2 = SGACodec(w, C, B, H, W, precision=prec)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
rng = np.random.RandomState(0)
shapes = {"GS0": (16, 16, C), "GS1": (32, 32, C), "GS2": (64, 64, C), "HS0": (4, 4, C), "HS1": (8, 8, C), "HS2": (16, 16, 288)}
inp = {k: torch.tensor(rng.standard_normal((B,) + v).astype(np.float32), device="cuda") for k, v in shapes.items()}
torch.cuda.synchronize()
for victim in ("GS2", "GS1", "GS0", "HS0", "HS1", "HS2"):
    with torch.cuda.stream(s1):
        ref = c1.layer_fwd(victim, inp[victim]).clone()
    torch.cuda.synchronize()
    for partner in (None, "HS0", "HS1", "GS2"):
        bad = 0
        for i in range(150):
            if partner:
                with torch.cuda.stream(s2):
                    for _ in range(6 if partner != "GS2" else 1):
                        c2.layer_fwd(partner, inp[partner])
            with torch.cuda.stream(s1):
                out = c1.layer_fwd(victim, inp[victim])
                bad += int(not torch.equal(out, ref))
        torch.cuda.synchronize()
        print(prec, "victim", victim, "partner", partner, "mismatching outputs:", bad, "/150", flush=True)
