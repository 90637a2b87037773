"""GPU parity for cfg 5 (bb_sga.py: SGA + bits-back) through the C ABI vs the oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402
from oracle import philox  # noqa: E402
from oracle.sga_oracle import SGAOracle  # noqa: E402

_CACHE = {}


def setup(C=64, B=2, H=48, W=40):
    from sga_amd.codec import SGACodec
    key = (C, B, H, W)
    if key not in _CACHE:
        w = sga_amd.make_synthetic_weights(C, seed=0, bb=True)
        _CACHE[key] = (SGACodec(w, C, B, H, W, bits_back=True), SGAOracle(w),
                       SGAOracle(w, dtype=torch.float64))
    return _CACHE[key]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_prior_density_and_derivative():
    codec, orc, orc64 = setup()
    v = (np.random.RandomState(0).standard_normal((4, 5, 3, 64)) * 5).astype(np.float32)
    vt = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    p = orc64.eb_pdf(vt)
    (dp,) = torch.autograd.grad(p.sum(), vt)
    gp, gdp = codec.factorized_density(v)
    assert rel_err(gp.cpu().numpy(), p.detach().numpy()) < 2e-5
    assert rel_err(gdp.cpu().numpy(), dp.numpy()) < 2e-4


def test_init_z_matches_hyper_analysis():
    codec, orc, _ = setup()
    x = np.random.RandomState(1).rand(2, 48, 40, 3).astype(np.float32)
    y, zml_enc = codec.encode(x)                       # bits_back: z output is [.., 2C]
    yo = orc.analysis(torch.tensor(x))
    assert rel_err(y.cpu().numpy(), yo.numpy()) < 5e-5
    want = orc.bb_init_z(yo.numpy()).numpy()
    assert rel_err(zml_enc.cpu().numpy(), want) < 1e-4
    got = codec.bb_init_z(np.round(yo.numpy()), 48, 40)
    assert rel_err(got.cpu().numpy(), orc.bb_init_z(np.round(yo.numpy())).numpy()) < 1e-4


@pytest.mark.parametrize("shape", [(64, 2, 48, 40), (192, 1, 64, 64)])
def test_bb_step_grads(shape, gpu_out_dir):
    C, B, H, W = shape
    codec, orc, orc64 = setup(C, B, H, W)
    x = np.random.RandomState(2).rand(B, H, W, 3).astype(np.float32)
    y = orc.analysis(torch.tensor(x)).numpy()
    zml = orc.bb_init_z(y).numpy()
    rng = np.random.RandomState(3)
    u_y = rng.uniform(1e-4, 1 - 1e-4, (y.size, 2)).astype(np.float32)
    eps = rng.standard_normal(zml.size // 2).astype(np.float32)
    ref = orc64.bb_step(x, y, zml, 0.35, u_y, eps, 0.01)
    got = codec.bb_step_grads(x, y, zml, 0.35, 0.01, u_y=u_y, eps=eps)
    ey = rel_err(got["gy"].cpu().numpy(), ref["gy"].numpy())
    ez = rel_err(got["gzml"].cpu().numpy(), ref["gzml"].numpy())
    with open(os.path.join(gpu_out_dir, "parity_bb.jsonl"), "a") as f:
        f.write(json.dumps(dict(test="bb_step", C=C, rel_err_gy=ey, rel_err_gzml=ez)) + "\n")
    assert ey < 1e-4 and ez < 1e-4, (ey, ez)
    for k in ("rd_loss", "train_mse", "train_bpp"):
        assert abs(got[k] - ref[k]) <= 3e-5 * abs(ref[k]), (k, got[k], ref[k])
    # stage 2: rate only, y_tilde fed directly (bb_sga.py:252-254)
    y_hat = np.round(y)
    ref2 = orc64.bb_step(x, y_hat, zml, 1.0, None, eps, None, rate_only=True)
    got2 = codec.bb_step_grads(x, y_hat, zml, 1.0, 0.0, eps=eps, rate_only=True)
    assert rel_err(got2["gzml"].cpu().numpy(), ref2["gzml"].numpy()) < 1e-4
    assert abs(got2["train_bpp"] - ref2["train_bpp"]) <= 3e-5 * abs(ref2["train_bpp"])


def test_bb_eval():
    from sga_amd.codec import metrics_to_dict
    codec, orc, _ = setup()
    x = np.random.RandomState(4).rand(2, 48, 40, 3).astype(np.float32)
    y_hat = np.round(orc.analysis(torch.tensor(x)).numpy())
    zml = orc.bb_init_z(y_hat).numpy()
    eps = np.random.RandomState(5).standard_normal(zml.size // 2).astype(np.float32)
    want = orc.bb_evaluate(x, y_hat, zml, eps)
    got = metrics_to_dict(codec.bb_evaluate(x, y_hat, zml, eps=eps))
    for k in ("est_bpp", "est_y_bpp", "est_z_bpp", "est_bpp_back"):
        assert np.allclose(got[k], want[k], rtol=3e-5, atol=1e-7), (k, got[k], want[k])
    assert np.allclose(got["psnr"], want["psnr"], atol=5e-3)


def test_bb_run_short_vs_oracle(gpu_out_dir):
    """Both stages for a few iterations with device Philox (uniforms + Box-Muller normals)."""
    from sga_amd.codec import metrics_to_dict
    codec, orc, _ = setup()
    x = np.random.RandomState(6).rand(2, 48, 40, 3).astype(np.float32)
    y_hat, zml, met, tr1, tr2 = codec.bb_run(x, 0.01, its=25, r_its=25, t0=5, annealing_rate=0.03,
                                             seed=9, trace=True)
    yo, zo, mo, t1o, t2o = orc.bb_run(x, 0.01, its=25, r_its=25, t0=5, r=0.03, seed=9, trace=True)
    tr1, tr2 = tr1.cpu().numpy(), tr2.cpu().numpy()
    assert np.allclose(tr1[:, :3], t1o[:, :3], rtol=2e-3), np.abs(tr1[:, :3] / t1o[:, :3] - 1).max(0)
    assert np.allclose(tr2[:, 2], t2o, rtol=2e-3)
    assert (y_hat.cpu().numpy() != yo).mean() < 5e-3
    got = metrics_to_dict(met)
    assert np.allclose(got["est_bpp"], mo["est_bpp"], rtol=5e-3)
    assert np.allclose(got["est_bpp_back"], mo["est_bpp_back"], rtol=5e-3)
