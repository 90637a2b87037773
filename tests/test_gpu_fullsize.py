"""GPU parity at BASELINE.json's full size (cfg 2: B=8 images of 256x256, num_filters=192,
2000 SGA iterations), through the C ABI.

* every layer at the shape it has in cfg 2 and one complete SGA step vs the CPU oracle
  (a single full-size evaluation takes the oracle a second or two);
* the COMPLETE 2000-iteration run (4.7 s on the GPU; the oracle would need ~10 minutes) through
  properties that do not need the oracle: bit-reproducibility, the returned metrics are exactly the
  evaluation of the returned latents, the optimisation improves on its starting point for every
  image, rate fields add up, the reconstruction's PSNR recomputed in float64 on the host agrees,
  and the exact-adjoint identity <A x, g> = <x, A^T g> of the linear layers at full size.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402
from oracle import philox  # noqa: E402
from oracle.sga_oracle import SGAOracle  # noqa: E402

C, B, H, W = 192, 8, 256, 256
PRECISIONS = ["f32", "bf16x3"]
_STATE = {}


def setup(precision):
    from sga_amd.codec import SGACodec
    if "w" not in _STATE:
        _STATE["w"] = sga_amd.make_synthetic_weights(C, seed=0)
        _STATE["orc"] = SGAOracle(_STATE["w"])
        _STATE["orc64"] = SGAOracle(_STATE["w"], dtype=torch.float64)
        _STATE["x"] = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(0)).numpy()
    if precision not in _STATE:
        _STATE[precision] = SGACodec(_STATE["w"], C, B, H, W, precision=precision)
    return _STATE[precision], _STATE["orc"], _STATE["orc64"], _STATE["x"]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def report(gpu_out_dir, name, **kw):
    with open(os.path.join(gpu_out_dir, "parity_fullsize.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **kw)) + "\n")


# layer -> (Hin, Win, Cin) in cfg 2 (y is 16x16, z is 4x4)
LAYER_IN = {
    "GA0": (256, 256, 3), "GA1": (128, 128, C), "GA2": (64, 64, C), "GA3": (32, 32, C),
    "GS0": (16, 16, C), "GS1": (32, 32, C), "GS2": (64, 64, C), "GS3": (128, 128, C),
    "HA0": (16, 16, C), "HA1": (16, 16, C), "HA2": (8, 8, C),
    "HS0": (4, 4, C), "HS1": (8, 8, C), "HS2": (16, 16, 288),
}


def _input(layer, seed=0):
    Hi, Wi, ci = LAYER_IN[layer]
    rng = np.random.RandomState(seed + 31 * list(LAYER_IN).index(layer))
    return rng.standard_normal((B, Hi, Wi, ci)).astype(np.float32)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("layer", list(LAYER_IN))
def test_layer_forward_fullsize(layer, precision, gpu_out_dir):
    codec, orc, _, _ = setup(precision)
    x = _input(layer)
    want = orc.layer_fwd(layer, x).numpy()
    got = codec.layer_fwd(layer, x).cpu().numpy()
    e = rel_err(got, want)
    report(gpu_out_dir, "layer_fwd", layer=layer, precision=precision, rel_err=e)
    assert got.shape == want.shape and e < 2e-5, (layer, e)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("layer", ["GS0", "GS1", "GS2", "GS3", "HS0", "HS1", "HS2"])
def test_layer_backward_fullsize(layer, precision, gpu_out_dir):
    """data-gradients vs autograd of the float32 oracle (float64 at this size would need ~10 GB);
    tolerance covers two independent float32 summation orders."""
    codec, orc, _, _ = setup(precision)
    x = _input(layer, seed=3)
    xt = torch.tensor(x, requires_grad=True)
    out = orc.layer_fwd(layer, xt)
    g_out = np.random.RandomState(5).standard_normal(tuple(out.shape)).astype(np.float32)
    (want,) = torch.autograd.grad(out, xt, torch.tensor(g_out))
    got = codec.layer_bwd(layer, x, g_out).cpu().numpy()
    e = rel_err(got, want.numpy())
    report(gpu_out_dir, "layer_bwd", layer=layer, precision=precision, rel_err=e)
    assert e < 5e-5, (layer, e)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("layer", ["GS3", "HS2"])
def test_linear_layers_exact_adjoint_fullsize(layer, precision, gpu_out_dir):
    """No oracle: for the two activation-free layers, fwd(x) - fwd(0) = A x and bwd(g) = A^T g, so
    <A x, g> == <x, A^T g> up to float32 rounding (accumulated in float64 on the host)."""
    codec, _, _, _ = setup(precision)
    x = _input(layer, seed=7)
    Ax = (codec.layer_fwd(layer, x) - codec.layer_fwd(layer, np.zeros_like(x))).double()
    g = torch.tensor(np.random.RandomState(9).standard_normal(tuple(Ax.shape)).astype(np.float32), device="cuda")
    ATg = codec.layer_bwd(layer, x, g).double()
    lhs = float((Ax * g.double()).sum())
    rhs = float((torch.tensor(x, device="cuda").double() * ATg).sum())
    scale = float(Ax.norm() * g.double().norm())
    report(gpu_out_dir, "adjoint", layer=layer, precision=precision, lhs=lhs, rhs=rhs, rel=abs(lhs - rhs) / scale)
    assert abs(lhs - rhs) < 1e-6 * scale, (lhs, rhs, scale)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_step_fullsize_vs_oracle(precision, gpu_out_dir):
    """One complete SGA evaluation at cfg 2 size with Philox noise: loss terms and both latent
    gradients vs the oracle (float32 forward/backward on the CPU, identical uniforms)."""
    codec, orc, _, x = setup(precision)
    yo, zo = orc.encode(x)
    seed, it, T, lmbda = 0xC0FFEE, 901, 0.41, 0.01
    u_y = philox.sga_uniforms(yo.numel(), it, 0, seed)
    u_z = philox.sga_uniforms(zo.numel(), it, 1, seed)
    want = orc.step(x, yo, zo, T, u_y, u_z, lmbda)
    got = codec.step_grads(x, yo.numpy(), zo.numpy(), T, lmbda, seed=seed, it=it)
    errs = dict(gy=rel_err(got["gy"].cpu().numpy(), want["gy"].numpy()),
                gz=rel_err(got["gz"].cpu().numpy(), want["gz"].numpy()),
                rd_loss=abs(float(got["rd_loss"]) / float(want["rd_loss"]) - 1),
                mse=abs(float(got["train_mse"]) / float(want["train_mse"]) - 1),
                bpp=abs(float(got["train_bpp"]) / float(want["train_bpp"]) - 1))
    report(gpu_out_dir, "step_fullsize", precision=precision, **errs)
    assert errs["gy"] < 1e-4 and errs["gz"] < 1e-4, errs
    assert errs["rd_loss"] < 1e-5 and errs["mse"] < 1e-5 and errs["bpp"] < 1e-5, errs


@pytest.mark.parametrize("precision", PRECISIONS)
def test_complete_run_fullsize_properties(precision, gpu_out_dir):
    """The benchmarked workload itself: 2000 iterations, B=8, 256x256, C=192."""
    from sga_amd.codec import metrics_to_dict
    codec, _, _, x = setup(precision)
    lmbda = 0.01
    y_hat, z_hat, met, _ = codec.run(x, lmbda, its=2000, seed=0)
    y2, z2, met2, _ = codec.run(x, lmbda, its=2000, seed=0)
    # (1) bit-reproducible (no atomics on the gradient path, counter-based RNG)
    assert torch.equal(y_hat, y2) and torch.equal(z_hat, z2)
    assert torch.allclose(met, met2, rtol=1e-6, atol=0)      # sums use f64 atomics: order may differ
    # (2) integers, finite
    assert torch.equal(y_hat, torch.round(y_hat)) and torch.equal(z_hat, torch.round(z_hat))
    m = metrics_to_dict(met)
    assert all(np.isfinite(v).all() for v in m.values())
    # (3) the metrics ARE the evaluation of the returned latents (sga.py:240-247)
    m_eval = metrics_to_dict(codec.evaluate(x, y_hat, z_hat))
    for k in m:
        assert np.allclose(m[k], m_eval[k], rtol=1e-6, atol=0), k
    # (4) rate fields add up; PSNR follows from the MSE of the rounded reconstruction
    assert np.allclose(m["est_bpp"], m["est_y_bpp"] + m["est_z_bpp"], rtol=1e-6)
    x_hat = codec.reconstruct(y_hat, H, W).double().cpu().numpy()       # g_s(y_hat) through the layer ops
    q = np.rint(np.clip(x_hat, 0, 1) * 255.0)                            # sga.py:167-174
    mse = ((q - x.astype(np.float64) * 255.0) ** 2).mean(axis=(1, 2, 3))
    assert np.allclose(m["mse"], mse, rtol=1e-4), (m["mse"], mse)
    assert np.allclose(m["psnr"], 20 * np.log10(255.0) - 10 * np.log10(mse), atol=1e-3)
    # (5) SGA improves the rate-distortion objective of EVERY image over its starting point
    #     (the encoder's latents rounded: its=0), sga.py:143-163 with the eval fields
    _, _, met0, _ = codec.run(x, lmbda, its=0, seed=0)
    m0 = metrics_to_dict(met0)
    rd = lmbda * m["mse"] + m["est_bpp"]
    rd0 = lmbda * m0["mse"] + m0["est_bpp"]
    report(gpu_out_dir, "complete_run", precision=precision, rd_start=rd0.tolist(), rd_final=rd.tolist(),
           est_bpp_mean=float(m["est_bpp"].mean()), psnr_mean=float(m["psnr"].mean()))
    assert (rd < rd0).all(), (rd, rd0)
    # (6) a different seed takes a different path to a statistically equal result
    y3, _, met3, _ = codec.run(x, lmbda, its=2000, seed=1)
    m3 = metrics_to_dict(met3)
    assert not torch.equal(y3, y_hat)
    assert abs(m3["est_bpp"].mean() - m["est_bpp"].mean()) < 0.02
    assert abs(m3["psnr"].mean() - m["psnr"].mean()) < 0.05
