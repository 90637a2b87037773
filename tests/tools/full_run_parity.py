"""SURVEY 8(c) known-answer tests 9 and 10: the COMPLETE 2000-step run, HIP path vs the CPU oracle
with identical Philox noise, and the seed-to-seed spread of the same run (calibrates what
'final BPP within 1e-3 / PSNR within 0.01 dB' can mean for a stochastic optimiser).

    python tests/tools/full_run_parity.py [C B H W] > gpurun_out/full_run_parity.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import sga_amd
from sga_amd.codec import SGACodec, metrics_to_dict
from oracle.sga_oracle import SGAOracle

C, B, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (64, 2, 64, 64)
its = int(os.environ.get("ITS", 2000))
w = sga_amd.make_synthetic_weights(C, seed=0)
x = np.random.RandomState(6).rand(B, H, W, 3).astype(np.float32)
out = dict(config=dict(C=C, B=B, H=H, W=W, its=its, lmbda=0.01))
orc = SGAOracle(w)
t = time.time()
yo, zo, mo, _ = orc.run(x, 0.01, its=its, seed=0)
out["oracle_seconds"] = time.time() - t
out["oracle"] = dict(est_bpp=mo["est_bpp"].tolist(), psnr=mo["psnr"].tolist())
for prec in ("f32", "bf16x3"):
    codec = SGACodec(w, C, B, H, W, precision=prec)
    res = {}
    for seed in range(6):
        y_hat, z_hat, met, _ = codec.run(x, 0.01, its=its, seed=seed)
        m = metrics_to_dict(met)
        res[seed] = dict(est_bpp=m["est_bpp"].tolist(), psnr=m["psnr"].tolist())
        if seed == 0:
            res["y_hat_differs_from_oracle_frac"] = float((y_hat.cpu().numpy() != yo).mean())
            res["z_hat_differs_from_oracle_frac"] = float((z_hat.cpu().numpy() != zo).mean())
    bpp = np.array([res[s]["est_bpp"] for s in range(6)])      # [seed, image]
    psnr = np.array([res[s]["psnr"] for s in range(6)])
    res["same_seed_vs_oracle"] = dict(d_bpp=(bpp[0] - mo["est_bpp"]).tolist(), d_psnr=(psnr[0] - mo["psnr"]).tolist(),
                                      d_bpp_mean=float(bpp[0].mean() - mo["est_bpp"].mean()),
                                      d_psnr_mean=float(psnr[0].mean() - mo["psnr"].mean()))
    res["seed_spread"] = dict(bpp_std_per_image=bpp.std(0, ddof=1).tolist(), psnr_std_per_image=psnr.std(0, ddof=1).tolist(),
                              bpp_range_of_batch_mean=float(np.ptp(bpp.mean(1))), psnr_range_of_batch_mean=float(np.ptp(psnr.mean(1))))
    out[prec] = res
    codec.close()
print(json.dumps(out, indent=1))
