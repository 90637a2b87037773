"""GPU parity at the shapes of BASELINE.json's other configurations (the benchmark line is cfg 2,
tests/test_gpu_fullsize.py): cfg 3 = Kodak 768x512 / 512x768 at num_filters=192, cfg 4 = Tecnick
1200x1200 at num_filters=256, cfg 5 = bits-back at Kodak size.  One image each: a single full-size
evaluation takes the CPU oracle a few seconds."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402
from oracle import philox  # noqa: E402
from oracle.sga_oracle import SGAOracle  # noqa: E402


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def report(gpu_out_dir, **kw):
    with open(os.path.join(gpu_out_dir, "parity_configs.jsonl"), "a") as f:
        f.write(json.dumps(kw) + "\n")


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("C,H,W,f64", [(192, 512, 768, True), (192, 768, 512, True), (256, 1200, 1200, False),
                                       (256, 96, 80, True)])
def test_step_at_config_shape(C, H, W, f64, precision, gpu_out_dir):
    """encode + one SGA evaluation with Philox noise vs the oracle (float64 where it fits in a few
    seconds; the float32 oracle's own gz is only good to ~1e-2 at Kodak size, the HIP path agrees
    with float64 to 3e-6 there)."""
    from sga_amd.codec import SGACodec, metrics_to_dict
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(1).rand(1, H, W, 3).astype(np.float32)
    orc = SGAOracle(w, dtype=torch.float64 if f64 else torch.float32)
    codec = SGACodec(w, C, 1, H, W, precision=precision)
    yo, zo = SGAOracle(w).encode(x)
    y, z = codec.encode(x)
    assert tuple(y.shape) == (1, -(-H // 16), -(-W // 16), C) and tuple(z.shape) == (1, -(-H // 64), -(-W // 64), C)
    e_enc = (rel_err(y.cpu().numpy(), yo.numpy()), rel_err(z.cpu().numpy(), zo.numpy()))
    seed, it, T, lmbda = 9, 3, 0.3, 0.05
    u_y = philox.sga_uniforms(yo.numel(), it, 0, seed)
    u_z = philox.sga_uniforms(zo.numel(), it, 1, seed)
    want = orc.step(x, yo, zo, T, u_y, u_z, lmbda)
    got = codec.step_grads(x, yo.numpy(), zo.numpy(), T, lmbda, seed=seed, it=it)
    errs = dict(enc_y=e_enc[0], enc_z=e_enc[1], gy=rel_err(got["gy"].cpu().numpy(), want["gy"].numpy()),
                gz=rel_err(got["gz"].cpu().numpy(), want["gz"].numpy()),
                rd_loss=abs(float(got["rd_loss"]) / float(want["rd_loss"]) - 1))
    report(gpu_out_dir, test="config_step", C=C, H=H, W=W, precision=precision, **errs)
    assert errs["enc_y"] < 2e-5 and errs["enc_z"] < 2e-5, errs
    assert errs["gy"] < 1e-4 and errs["gz"] < (1e-4 if f64 else 5e-4) and errs["rd_loss"] < 1e-5, errs
    # a short complete run at this shape: finite metrics, objective improves, reproducible
    a = codec.run(x, lmbda, its=40, t0=10, annealing_rate=0.02, seed=2)
    b = codec.run(x, lmbda, its=40, t0=10, annealing_rate=0.02, seed=2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    m, m0 = metrics_to_dict(a[2]), metrics_to_dict(codec.run(x, lmbda, its=0)[2])
    assert np.isfinite(m["est_bpp"]).all() and np.isfinite(m["psnr"]).all()
    assert (lmbda * m["mse"] + m["est_bpp"] < lmbda * m0["mse"] + m0["est_bpp"]).all()
    codec.close()


@pytest.mark.parametrize("H,W", [(512, 768)])
def test_bits_back_step_at_kodak_size(H, W, gpu_out_dir):
    """cfg 5: one bits-back evaluation (bb_sga.py:93-158) at Kodak size vs the float64 oracle."""
    from sga_amd.codec import SGACodec
    C = 192
    w = sga_amd.make_synthetic_weights(C, seed=0, bb=True)
    codec = SGACodec(w, C, 1, H, W, bits_back=True)
    orc, orc64 = SGAOracle(w), SGAOracle(w, dtype=torch.float64)
    x = np.random.RandomState(2).rand(1, H, W, 3).astype(np.float32)
    yo = orc.analysis(torch.tensor(x))
    zml = orc.bb_init_z(yo.numpy()).numpy()
    # the untrained synthetic h_a produces |mean|, |logvar| ~ 20 at this image size: exp(sigma_raw)
    # then overflows float32 (on the GPU as it would in TF) and float64 alike.  Keep the posterior
    # parameters in the range a trained model produces; everything downstream is unchanged.
    zml = np.concatenate([np.clip(zml[..., :C], -4, 4), np.clip(zml[..., C:], -4, 1)], -1).astype(np.float32)
    rng = np.random.RandomState(3)
    u_y = rng.uniform(1e-4, 1 - 1e-4, (yo.numel(), 2)).astype(np.float32)
    eps = rng.standard_normal(zml.size // 2).astype(np.float32)
    ref = orc64.bb_step(x, yo.numpy(), zml, 0.35, u_y, eps, 0.01)
    got = codec.bb_step_grads(x, yo.numpy(), zml, 0.35, 0.01, u_y=u_y, eps=eps)
    ey = rel_err(got["gy"].cpu().numpy(), ref["gy"].numpy())
    ez = rel_err(got["gzml"].cpu().numpy(), ref["gzml"].numpy())
    report(gpu_out_dir, test="config_bb_step", H=H, W=W, gy=ey, gzml=ez)
    assert ey < 1e-4 and ez < 2e-4, (ey, ez)
    codec.close()
