"""SURVEY.md 8(f)-1 on the GPU path: a TF-format tensor bundle of RAW variables (RDFT kernels, reparametrised GDN
beta / gamma, the prior's matrix / factor / quantiles; names as tfc 1.3 / Keras scope them) -> the importer
(`tf_checkpoint.load_effective_weights`, the counterpart of `Saver.restore`, sga.py:180-182) -> `SGACodec` -> the HIP
kernels, against the oracle built from effective tensors that THIS FILE computes independently of the importer (explicit
cos / sin real-Fourier basis in float64 instead of its FFT-built matrix, explicit reparametrisation formulas).  So a
layout slip in the importer (transposed RDFT basis, gamma un-reparametrised on the wrong axis, medians taken from the wrong
quantile) shows up as a parity failure of the device path, not only of a CPU round trip.  No real checkpoint exists
offline (README.md:100-103): what tfc writes into one stays unverified."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402
from oracle.sga_oracle import SGAOracle  # noqa: E402
from test_tf_checkpoint import write_bundle  # noqa: E402  (the test suite's own bundle writer)

PED = 2.0 ** -36


def real_fourier_basis(n):
    """Orthonormal real DFT basis of one axis, columns ordered as tfc's irdft_matrix orders them: the real parts of
    rfft bins 0..n//2, then the imaginary parts of the bins that have one (np.fft.rfft(e_x)[k] = exp(-2 pi i k x / n))."""
    x = np.arange(n, dtype=np.float64)
    cols = []
    for k in range(n // 2 + 1):
        self_conj = k == 0 or 2 * k == n
        cols.append(np.cos(2 * np.pi * k * x / n) * ((1.0 if self_conj else np.sqrt(2.0)) / np.sqrt(n)))
    for k in range(1, n // 2 + 1):
        if 2 * k != n:
            cols.append(-np.sin(2 * np.pi * k * x / n) * (np.sqrt(2.0) / np.sqrt(n)))
    return np.stack(cols, 1)                                     # [x, coefficient]


def make_raw_and_effective(C, seed, bb=False):
    """Raw checkpoint variables + the effective tensors they stand for, the latter computed here in float64."""
    eff = {k: v.astype(np.float64) for k, v in sga_amd.make_synthetic_weights(C, seed=seed, bb=bb).items()}
    rng = np.random.RandomState(seed + 100)
    scope = {"ga": "analysis_transform", "gs": "synthesis_transform", "ha": "hyper_analysis_transform",
             "hs": "mbt2018_hyper_synthesis_transform"}
    raw = {}
    for p, n in (("ga", 4), ("gs", 4), ("ha", 3), ("hs", 3)):
        for i in range(n):
            k = eff[f"{p}.k{i}"]
            kh, kw, ci, co = k.shape
            if p == "hs":                                        # kernel_parameterizer=None (nn_models.py:154,158,162)
                raw[f"{scope[p]}/layer_{i}/kernel"] = k
            else:                                                # RDFTParameterizer: the variable holds the coefficients
                B0, B1 = real_fourier_basis(kh), real_fourier_basis(kw)
                coef = np.einsum("ya,xb,yxio->abio", B0, B1, k)                  # analysis = transpose (orthonormal)
                coef32 = coef.astype(np.float32)
                raw[f"{scope[p]}/layer_{i}/kernel_rdft"] = coef32.reshape(kh * kw, ci * co)
                eff[f"{p}.k{i}"] = np.einsum("ya,xb,abio->yxio", B0, B1, coef32.astype(np.float64))   # what the file holds
            if f"{p}.b{i}" in eff:
                raw[f"{scope[p]}/layer_{i}/bias"] = eff[f"{p}.b{i}"]
    for p, g in (("ga", "gdn"), ("gs", "igdn")):
        for i in range(3):
            rb = np.sqrt(eff[f"{p}.beta{i}"] + PED).astype(np.float32)
            rg = np.sqrt(eff[f"{p}.gamma{i}"] + PED).astype(np.float32)
            # entries below the parameterizers' minima exercise the max(): gamma 0 -> effective 0, beta -> 1e-6
            rg[rng.rand(C, C) < 0.05] = 0.0
            rb[::17] = 0.0
            raw[f"{scope[p]}/layer_{i}/{g}_{i}/reparam_beta"] = rb
            raw[f"{scope[p]}/layer_{i}/{g}_{i}/reparam_gamma"] = rg
            eff[f"{p}.beta{i}"] = np.maximum(rb.astype(np.float64), np.sqrt(1e-6 + PED)) ** 2 - PED
            eff[f"{p}.gamma{i}"] = np.maximum(rg.astype(np.float64), 2.0 ** -18) ** 2 - PED
    prior = "entropy_bottleneck"
    for k in range(4):
        m = np.log(np.expm1(eff[f"eb.m{k}"])).astype(np.float32)
        raw[f"{prior}/matrix_{k}"] = m
        eff[f"eb.m{k}"] = np.logaddexp(0.0, m.astype(np.float64))                # softplus
        raw[f"{prior}/bias_{k}"] = eff[f"eb.b{k}"]
        if k < 3:
            f = np.arctanh(eff[f"eb.f{k}"]).astype(np.float32)
            raw[f"{prior}/factor_{k}"] = f
            eff[f"eb.f{k}"] = np.tanh(f.astype(np.float64))
    med = np.linspace(-0.4, 0.45, C)
    raw[f"{prior}/quantiles"] = np.stack([med - 9.0, med, med + 9.0], -1).reshape(C, 1, 3)
    eff["eb.medians"] = med
    raw["analysis_transform/layer_0/bias/Adam"] = np.zeros(C)                    # optimizer slot: must be ignored
    raw["global_step"] = np.zeros(())
    return ({k: np.asarray(v, np.float32) for k, v in raw.items()},
            {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in eff.items()})


@pytest.fixture(scope="module")
def bundle(tmp_path_factory):
    C = 64
    raw, eff = make_raw_and_effective(C, seed=5)
    root = tmp_path_factory.mktemp("ckpt")
    runname = "mbt2018-num_filters=64-lmbda=0.02"
    d = root / runname
    d.mkdir()
    prefix = str(d / "model.ckpt-2000000")
    write_bundle(prefix, raw, snappy=True)
    (d / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-2000000"\n')
    return C, str(root), runname, eff


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_bundle_to_hip_path_matches_oracle_on_independent_effective_weights(bundle, gpu_out_dir):
    from sga_amd.codec import SGACodec, metrics_to_dict
    from sga_amd.tf_checkpoint import load_effective_weights
    C, root, runname, eff = bundle
    w = load_effective_weights(os.path.join(root, runname), C)
    for k, v in eff.items():                                     # the importer's tensors == the independent ones
        assert w[k].shape == v.shape, k
        assert np.allclose(w[k], v, rtol=2e-6, atol=2e-7), (k, float(np.abs(w[k] - v).max()))
    assert (w["gs.gamma0"] == 0).sum() > 50 and np.isclose(w["ga.beta1"][0], 1e-6, rtol=1e-3)   # the max() branches are live
    B, H, W = 2, 64, 80
    codec = SGACodec(w, C, B, H, W)
    orc = SGAOracle({k: v for k, v in eff.items() if k != "eb.medians"}, dtype=torch.float64)
    x = np.random.RandomState(1).rand(B, H, W, 3).astype(np.float32)
    y, z = codec.encode(x)
    yo, zo = orc.encode(x)
    e_y, e_z = rel_err(y.cpu().numpy(), yo.numpy()), rel_err(z.cpu().numpy(), zo.numpy())
    assert e_y < 2e-5 and e_z < 5e-5, (e_y, e_z)
    rng = np.random.RandomState(2)
    y0, z0 = yo.numpy().astype(np.float32), zo.numpy().astype(np.float32)
    u_y = rng.uniform(1e-4, 1 - 1e-4, (y0.size, 2)).astype(np.float32)
    u_z = rng.uniform(1e-4, 1 - 1e-4, (z0.size, 2)).astype(np.float32)
    ref = orc.step(x, y0, z0, 0.3, u_y, u_z, 0.02)
    got = codec.step_grads(x, y0, z0, 0.3, 0.02, u_y=u_y, u_z=u_z)
    e_gy, e_gz = rel_err(got["gy"].cpu().numpy(), ref["gy"].numpy()), rel_err(got["gz"].cpu().numpy(), ref["gz"].numpy())
    assert e_gy < 1e-4 and e_gz < 1e-4, (e_gy, e_gz)
    assert abs(got["rd_loss"] - ref["rd_loss"]) <= 2e-5 * abs(ref["rd_loss"])
    # cfg 1 with the medians read from `quantiles` (mbt2018.py:69: z is rounded around them)
    assert codec.medians is not None and np.allclose(codec.medians, eff["eb.medians"])
    y_hat, z_hat, met = codec.base_compress(x, medians=codec.medians)
    yh, zh, want = SGAOracle({k: v for k, v in eff.items() if k != "eb.medians"}).base_compress(x, medians=eff["eb.medians"])
    assert float((np.abs(z_hat.cpu().numpy() - zh.numpy()) > 1e-3).mean()) < 5e-3
    assert float((np.abs(y_hat.cpu().numpy() - yh.numpy()) > 1e-3).mean()) < 5e-3
    m = metrics_to_dict(met)
    assert np.allclose(m["est_bpp"], want["est_bpp"], rtol=5e-3) and np.allclose(m["psnr"], want["psnr"], atol=0.05)
    frac = (z_hat.cpu().numpy() - np.round(z_hat.cpu().numpy()))
    assert np.abs(frac).max() > 0.3                                # the centring by non-zero medians is really applied
    with open(os.path.join(gpu_out_dir, "parity_checkpoint.jsonl"), "a") as f:
        import json
        f.write(json.dumps(dict(test="bundle_to_hip", enc_y=e_y, enc_z=e_z, gy=e_gy, gz=e_gz)) + "\n")
    codec.close()


@pytest.mark.parametrize("method", ["sga", "mbt2018", "map"])
def test_driver_compress_from_a_checkpoint_directory(bundle, method, tmp_path):
    """`python -m sga_amd.driver --checkpoint_dir <dir> compress <runname> <input>` WITHOUT --synthetic_weights: the
    weights come from the bundle (sga.py:180-182), lambda from the run name (sga.py:157-158), the medians from
    `quantiles`; per-image results equal a codec fed the same weights directly."""
    from sga_amd import driver
    from sga_amd.codec import SGACodec, metrics_to_dict
    from sga_amd.tf_checkpoint import load_effective_weights
    C, root, runname, eff = bundle
    X = (np.random.RandomState(3).rand(3, 48, 64, 3) * 255).astype(np.uint8)
    inp = tmp_path / "tiny.npy"
    np.save(inp, X)
    out = tmp_path / "res"
    driver.main(["--num_filters", str(C), "--checkpoint_dir", root, "compress", "--results_dir", str(out),
                 "--sga_its", "20", "--t0", "5", "--method", method, runname, str(inp)])
    files = os.listdir(out)
    assert files == [driver.result_filename("rd", method, 0.02, runname, str(inp))]
    res = dict(np.load(out / files[0]))
    w = load_effective_weights(os.path.join(root, runname), C)
    codec = SGACodec(w, C, 3, 48, 64)
    x = X.astype(np.float32) / 255.0
    if method == "sga":
        _, _, met, _ = codec.run(x, 0.02, its=20, t0=5, seed=0, loss_scale=1.0 / 3)
    elif method == "mbt2018":
        _, _, met = codec.base_compress(x, medians=codec.medians)
    else:
        codec.set_relaxation("none", "exp0")
        _, _, met, _ = driver.run_early_stop(codec, torch.tensor(x), 0.02, method="map", its=20, lr=0.005, seed=0,
                                             loss_scale=1.0 / 3, medians=codec.medians)
    m = metrics_to_dict(met)
    for k in ("est_bpp", "psnr", "est_y_bpp", "est_z_bpp"):
        assert np.allclose(res[k], m[k], rtol=1e-6), (k, res[k], m[k])
    codec.close()
