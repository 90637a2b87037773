"""GPU-box: per-layer hipEvent timing of the conv launches of one SGA iteration (bench config)."""
import os, sys, json
os.environ["SGA_PROFILE_BY_LAYER"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, sga_amd
from sga_amd.codec import SGACodec
C, B, H, W = int(os.environ.get("C", 192)), int(os.environ.get("B", 8)), int(os.environ.get("H", 256)), int(os.environ.get("W", 256))
codec = SGACodec(sga_amd.make_synthetic_weights(C, 0), C, B, H, W, precision=os.environ.get("PREC", "f32"), lab=bool(os.environ.get("LAB")))
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).cuda()
codec.run(x, 0.01, its=5, metrics=False)
its = 40
codec.profile_begin(); codec.run(x, 0.01, its=its, metrics=False); st = codec.profile_end()
tot = sum(k["ms_total"] for k in st)
print(f"{'layer':44s} {'us/it':>9s} {'GF/launch':>10s} {'TF/s':>7s} {'%':>6s}")
for k in sorted(st, key=lambda k: -k["ms_total"]):
    n = k["launches"]
    print(f"{k['name']:44s} {1e3*k['ms_total']/its:9.1f} {k['flops_total']/n/1e9:10.3f} {k['flops_total']/k['ms_total']/1e9:7.1f} {100*k['ms_total']/tot:6.1f}")
print("total conv us/it", 1e3 * tot / its, "ideal us/it", sum(k["flops_total"] for k in st) / its / 157.3e12 * 1e6)
