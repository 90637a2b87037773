"""A TRAINED-LIKE operating point for the parity sets (test infrastructure; SURVEY.md 8(c) "trained-like").

`make_synthetic_weights` is an untrained model: 4.1 bpp / 11 dB at cfg 2, |y| of several bins, predicted scales anywhere.
A trained mean-scale hyperprior puts most of y_hat at 0 under small predicted scales -- the regime in which the sigma bound
(sga.py:130-133 vs mbt2018.py:77-80), the `lower_bound` branches and the 1/sigma gradients matter most.  This script FITS the
model (the oracle's layers under PyTorch autograd, the training objective of mbt2018.py:90-135: additive uniform noise on y
and z, factorized prior on z, conditional Gaussian with the 0.11 scale bound on y, lambda * 255^2 * mse + bpp) on low-pass
noise images until it compresses them properly, and stores the EFFECTIVE tensors (the dict the C ABI takes), rounded to
float16-representable values, as tests/golden/fitted_weights_c64.npz (+ its digest and end metrics in
fitted_weights_c64.json).  Deterministic inputs; the stored file, not a re-run of this script, defines the model.

    python tests/tools/fit_weights.py [steps=4000]        # ~10 min on 3 cores
    FIT_C=192 NTHREADS=4 python tests/tools/fit_weights.py 4000   # -> fitted_weights_c192.npz (round 5)
    FIT_C=256 FIT_LMBDA=0.08 NTHREADS=4 python tests/tools/fit_weights.py 4000   # -> fitted_weights_c256.npz (round 6: cfg 4's width and rate point)
    python tests/tools/fit_weights.py 4000 bb             # the mbt2018_bb model (cfg 5: h_a emits mean | logvar, bb_sga.py:69)
                                                          # -> fitted_weights_c64bb.npz, objective = the bits-back ELBO
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import sga_amd
from oracle.sga_oracle import SGAOracle, lower_bound, LIKELIHOOD_BOUND, SCALES_MIN

C = int(os.environ.get("FIT_C", "64"))        # FIT_C=192: the north star's width (round 5; ~40 min on 4 cores)
H, W, BATCH = 64, 64, 8
LMBDA = float(os.environ.get("FIT_LMBDA", "0.01"))      # FIT_LMBDA=0.08: cfg 4's rate point (README.md:60,105), used for C = 256 (round 6)
OUT = os.path.join(ROOT, "tests", "golden", "fitted_weights_c%d" % C)


def raw_from_effective(w):
    """Unconstrained training variables whose `effective()` image is w (tfc's parameterisations in spirit: non-negative
    beta / gamma as squares, softplus'd prior matrices, tanh'd prior factors)."""
    raw = {}
    for k, v in w.items():
        t = torch.tensor(v, dtype=torch.float32)
        if ".beta" in k or ".gamma" in k:
            t = torch.sqrt(t)
        elif k.startswith("eb.m"):
            t = torch.log(torch.expm1(t))
        elif k.startswith("eb.f"):
            t = torch.atanh(torch.clamp(t, -0.999, 0.999))
        raw[k] = t.clone().requires_grad_(True)
    return raw


def effective(raw):
    w = {}
    for k, t in raw.items():
        if ".beta" in k:
            w[k] = t * t + 1e-6
        elif ".gamma" in k:
            w[k] = t * t
        elif k.startswith("eb.m"):
            w[k] = torch.nn.functional.softplus(t)
        elif k.startswith("eb.f"):
            w[k] = torch.tanh(t)
        else:
            w[k] = t
    return w


def objective(orc, x, gen):
    """mbt2018.py:90-135 (training graph) on the oracle's layers."""
    y = orc.analysis(x)
    z = orc.hyper_analysis(y)
    z_t = z + (torch.rand(z.shape, generator=gen) - 0.5)
    y_t = y + (torch.rand(y.shape, generator=gen) - 0.5)
    p_z = lower_bound(orc.eb_likelihood(z_t), LIKELIHOOD_BOUND)
    ms = orc.hyper_synthesis(z_t)
    mu, sig = torch.split(ms, orc.C, dim=-1)
    mu, sigma = mu[:, :y.shape[1], :y.shape[2]], torch.exp(sig)[:, :y.shape[1], :y.shape[2]]
    p_y = lower_bound(orc.gauss_likelihood(y_t, mu, sigma, SCALES_MIN), LIKELIHOOD_BOUND)
    x_t = orc.synthesis(y_t)[:, :x.shape[1], :x.shape[2]]
    npx = x.shape[0] * x.shape[1] * x.shape[2]
    bpp = (-torch.log(p_y).sum() - torch.log(p_z).sum()) / (math.log(2) * npx)
    mse = ((x - x_t) ** 2).mean() * 255.0 ** 2
    return LMBDA * mse + bpp, bpp, mse


def objective_bb(orc, x, gen):
    """The bits-back model's training graph (bb_sga.py:93-158 without the SGA relaxation: y_tilde = y + U)."""
    y = orc.analysis(x)
    y_t = y + (torch.rand(y.shape, generator=gen) - 0.5)
    zml = orc.hyper_analysis(y_t)
    z_mean, z_logvar = torch.split(zml, orc.C, dim=-1)
    eps = torch.randn(z_mean.shape, generator=gen)
    out = orc.bb_objective(x, y_t, z_mean, z_logvar, eps, LMBDA, loss_scale=1.0 / x.shape[0])
    return out["rd_loss"], out["train_bpp"], out["train_mse"]


def main():
    global OUT
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    bb = len(sys.argv) > 2 and sys.argv[2] == "bb"
    torch.set_num_threads(int(os.environ.get("NTHREADS", "3")))
    torch.manual_seed(0)
    w0 = sga_amd.make_synthetic_weights(C, seed=0, bb=bb)
    if bb:
        OUT += "bb"
        w0["ha.k2"] = w0["ha.k2"].copy()
        w0["ha.k2"][..., C:] *= np.float32(0.05)        # start at logvar ~ 0 (sigma_q ~ 1) instead of |logvar| ~ 20
    raw = raw_from_effective(w0)
    orc = SGAOracle(w0, scale_bound=SCALES_MIN if bb else 0.0)
    opt = torch.optim.Adam(list(raw.values()), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [int(steps * 0.7), int(steps * 0.9)], 0.3)
    gen = torch.Generator().manual_seed(1)
    pool = torch.tensor(sga_amd.make_lowpass_images(512, H, W, seed=100))      # training images
    t0 = time.time()
    for it in range(steps):
        idx = torch.randint(0, pool.shape[0], (BATCH,), generator=gen)
        orc.w = effective(raw)
        loss, bpp, mse = (objective_bb if bb else objective)(orc, pool[idx], gen)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(raw.values()), 10.0)
        opt.step()
        sched.step()
        if it % 100 == 0 or it == steps - 1:
            print("it %5d loss %.4f bpp %.4f psnr %.2f dB  %.0f s" % (it, float(loss.detach()), float(bpp.detach()),
                  10 * math.log10(255.0 ** 2 / float(mse.detach())), time.time() - t0), flush=True)
    w = {k: v.detach().numpy().astype(np.float32) for k, v in effective(raw).items()}
    w = sga_amd.save_weights_npz(OUT + ".npz", w)
    sga_amd.check_weights(w, C, bb)
    if bb:
        orc = SGAOracle(w)
        x = sga_amd.make_lowpass_images(4, H, W, seed=7)
        with torch.no_grad():
            zml = orc.bb_init_z(orc.analysis(torch.tensor(x)))
        rep = dict(C=C, steps=steps, lmbda=LMBDA, bb=True, digest=sga_amd.weights_digest(w),
                   z_mean_absmax=float(zml[..., :C].abs().max()), z_logvar_min=float(zml[..., C:].min()),
                   z_logvar_max=float(zml[..., C:].max()), last_train_bpp=float(bpp.detach()), last_train_psnr=10 * math.log10(255.0 ** 2 / float(mse.detach())))
        with open(OUT + ".json", "w") as f:
            json.dump(rep, f, indent=1)
        print(json.dumps(rep, indent=1))
        return
    # end metrics of the STORED model on held-out images: one-shot compress (cfg 1) in both sigma-bound modes
    orc = SGAOracle(w)
    x = sga_amd.make_lowpass_images(4, H, W, seed=7)
    rep = dict(C=C, steps=steps, lmbda=LMBDA, digest=sga_amd.weights_digest(w))
    for name, sb in (("bound_0.11", SCALES_MIN), ("raw_sigma", 0.0)):
        y_hat, z_hat, m = orc.base_compress(x, scale_bound=sb)
        rep[name] = dict(est_bpp=float(m["est_bpp"].mean()), psnr=float(m["psnr"].mean()),
                         frac_zero_y_hat=float((torch.round(y_hat) == 0).float().mean()))
    with torch.no_grad():
        y, z = orc.encode(x)
        ms = orc.hyper_synthesis(torch.round(z))
        sigma = torch.exp(ms[..., C:])
        rep["sigma_quantiles_1_50_99"] = [float(q) for q in torch.quantile(sigma.flatten(), torch.tensor([0.01, 0.5, 0.99]))]
        rep["frac_sigma_below_0.11"] = float((sigma < SCALES_MIN).float().mean())
        rep["abs_y_max"] = float(y.abs().max())
    with open(OUT + ".json", "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
