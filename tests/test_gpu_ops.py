"""GPU parity, operator level: every HIP kernel behind the C ABI against the CPU oracle on the
same seeded inputs (SURVEY.md 8(c) known-answer tests 1-7).  Tolerances are for float32
summation-order differences between the MFMA fmaf chain and oneDNN; written per test."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402
from oracle.sga_oracle import SGAOracle, AdamF32, lower_bound  # noqa: E402

_CODECS = {}


def get_codec(C, precision="f32"):
    from sga_amd.codec import SGACodec
    key = (C, precision)
    if key not in _CODECS:
        w = sga_amd.make_synthetic_weights(C, seed=0)
        _CODECS[key] = (SGACodec(w, C, max_batch=4, max_height=96, max_width=96, precision=precision),
                        SGAOracle(w), SGAOracle(w, dtype=torch.float64), w)
    return _CODECS[key]


PRECISIONS = ["f32", "bf16x3"]   # v_mfma_f32_32x32x2_f32 chain / exact 3 x bf16 operand split, same tolerances
# sga_config.scale_bound: 0 = raw sigma (sga.py:130-133, tfc layer never built), 0.11 = built layer (mbt2018.py:77-80)
SCALE_BOUNDS = [0.0, 0.11]


class scale_bound_of:
    """Run a block with the shared codec's sigma bound set to `sb`; restore the default (0) afterwards."""

    def __init__(self, codec, sb):
        self.codec, self.sb = codec, sb

    def __enter__(self):
        self.codec.set_scale_bound(self.sb)

    def __exit__(self, *exc):
        self.codec.set_scale_bound(0.0)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def report(gpu_out_dir, name, **kw):
    with open(os.path.join(gpu_out_dir, "parity_ops.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **kw)) + "\n")


def where_bad(a, b, tol):
    a = np.asarray(a); b = np.asarray(b)
    d = np.abs(a - b)
    idx = np.unravel_index(np.argmax(d), d.shape)
    return f"max|d|={d.max():.3e} at {idx}: got {a[idx]:.6g} want {b[idx]:.6g}; #bad={(d > tol * (np.abs(b).max() + 1e-30)).sum()}/{d.size}"


LAYER_IN = {  # layer -> (Hin, Win, channels as multiple: 'x'=3, 'c'=C, 'c15'=1.5C)
    "GA0": (21, 26, "x"), "GA1": (13, 10, "c"), "GA2": (8, 9, "c"), "GA3": (6, 5, "c"),
    "GS0": (3, 4, "c"), "GS1": (5, 7, "c"), "GS2": (6, 6, "c"), "GS3": (9, 11, "c"),
    "HA0": (6, 5, "c"), "HA1": (7, 6, "c"), "HA2": (5, 4, "c"),
    "HS0": (2, 3, "c"), "HS1": (4, 5, "c"), "HS2": (8, 7, "c15"),
}


def _layer_input(layer, C, B=2, seed=0):
    Hi, Wi, ch = LAYER_IN[layer]
    cin = {"x": 3, "c": C, "c15": int(1.5 * C)}[ch]
    rng = np.random.RandomState(seed + 17 * list(LAYER_IN).index(layer))
    return rng.standard_normal((B, Hi, Wi, cin)).astype(np.float32)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("C", [64, 128, 192, 256])
@pytest.mark.parametrize("layer", list(LAYER_IN))
def test_layer_forward(layer, C, precision, gpu_out_dir):
    """conv / transposed conv / 3x3 conv (+ GDN, IGDN, ReLU) vs torch.nn.functional on
    asymmetric random kernels and odd sizes (catches flip / transposition / padding errors)."""
    codec, orc, _, _ = get_codec(C, precision)
    x = _layer_input(layer, C)
    want = orc.layer_fwd(layer, x).numpy()
    got = codec.layer_fwd(layer, x).cpu().numpy()
    assert got.shape == want.shape
    e = rel_err(got, want)
    report(gpu_out_dir, "layer_fwd", layer=layer, C=C, precision=precision, rel_err=e)
    assert e < 2e-5, f"{layer} C={C}: {where_bad(got, want, 2e-5)}"


@pytest.mark.parametrize("C", [64, 192, 256])
@pytest.mark.parametrize("layer", list(LAYER_IN))
def test_layer_forward_and_backward_bf16x2(layer, C, gpu_out_dir):
    """The fast precision mode (SGA_PRECISION_BF16X2, include/sga_hip.h: convolution operands rounded to 16 mantissa bits --
    two bf16 planes, three plane products, f32 accumulation).  NOT f32-grade by construction (2^-17 per operand), so its own
    bound: 5e-5 of the output scale after a layer incl. its GDN / IGDN, forward and backward (measured: 1.1e-5 / 5.4e-6 at
    worst over the 42 cases; the f32-grade bounds above are 2e-5 / 5e-5).  Same asymmetric kernels, odd sizes and float64 references as the tests above; also checks that the mode is
    not silently the three-plane one (the results differ) nor the f32 one."""
    codec, orc, orc64, _ = get_codec(C, "bf16x2")
    x = _layer_input(layer, C)
    want = orc.layer_fwd(layer, x).numpy()
    got = codec.layer_fwd(layer, x).cpu().numpy()
    e = rel_err(got, want)
    eb = None
    if layer in ("GS0", "GS1", "GS2", "GS3", "HS0", "HS1", "HS2"):
        xb = _layer_input(layer, C, seed=3)
        xt = torch.tensor(xb, dtype=torch.float64, requires_grad=True)
        out = orc64.layer_fwd(layer, xt)
        g_out = np.random.RandomState(5).standard_normal(tuple(out.shape)).astype(np.float32)
        (wantb,) = torch.autograd.grad(out, xt, torch.tensor(g_out, dtype=torch.float64))
        eb = rel_err(codec.layer_bwd(layer, xb, g_out).cpu().numpy(), wantb.numpy())
    report(gpu_out_dir, "layer_bf16x2", layer=layer, C=C, rel_err_fwd=e, rel_err_bwd=eb)
    assert e < 5e-5, f"{layer} C={C}: {where_bad(got, want, 5e-5)}"
    assert eb is None or eb < 5e-5, (layer, C, eb)
    if layer in ("GS1", "GS2", "HS1", "HA1") and C >= 128:      # C-channel inputs on the bf16 pipe: the planes really are two
        x3 = get_codec(C, "bf16x3")[0].layer_fwd(layer, x).cpu().numpy()
        assert rel_err(got, x3) > 1e-7, "bf16x2 produced the bf16x3 result bit for bit: the two-plane loop did not run"


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("C", [64, 192, 256])
@pytest.mark.parametrize("layer", ["GS0", "GS1", "GS2", "GS3", "HS0", "HS1", "HS2"])
def test_layer_backward(layer, C, precision, gpu_out_dir):
    """data-gradient of each synthesis-side layer vs float64 autograd of the oracle."""
    codec, orc, orc64, _ = get_codec(C, precision)
    x = _layer_input(layer, C, seed=3)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    out = orc64.layer_fwd(layer, xt)
    g_out = np.random.RandomState(5).standard_normal(tuple(out.shape)).astype(np.float32)
    (want,) = torch.autograd.grad(out, xt, torch.tensor(g_out, dtype=torch.float64))
    got = codec.layer_bwd(layer, x, g_out).cpu().numpy()
    e = rel_err(got, want.numpy())
    report(gpu_out_dir, "layer_bwd", layer=layer, C=C, precision=precision, rel_err=e)
    assert e < 5e-5, f"{layer} C={C}: {where_bad(got, want.numpy(), 5e-5)}"


def test_impulse_orientation(gpu_out_dir):
    """Asymmetric impulse through the transposed conv: a single 1 at one input pixel/channel
    must reproduce the (un-flipped) kernel slice at out[2i+ky-2, 2j+kx-2]."""
    C = 64
    codec, orc, _, w = get_codec(C)
    x = np.zeros((1, 4, 5, C), np.float32)
    x[0, 1, 2, 7] = 1.0
    got = codec.layer_fwd("HS0", x).cpu().numpy()
    K = w["hs.k0"]
    want = np.zeros((1, 8, 10, C), np.float32)
    for ky in range(5):
        for kx in range(5):
            oy, ox = 2 * 1 + ky - 2, 2 * 2 + kx - 2
            if 0 <= oy < 8 and 0 <= ox < 10:
                want[0, oy, ox] += K[ky, kx, 7, :]
    want = np.maximum(want + w["hs.b0"], 0)
    assert rel_err(got, want) < 1e-6, where_bad(got, want, 1e-6)


@pytest.mark.parametrize("T", [0.5, 0.2, 0.05])
def test_sampler(T, gpu_out_dir):
    """SGA relaxation with injected uniforms: value and Jacobian vs oracle autograd
    (float64); includes exact integers (fl == ce) and values next to integers (clip active)."""
    codec, orc, orc64, _ = get_codec(64)
    rng = np.random.RandomState(0)
    v = (rng.standard_normal(4096) * 3).astype(np.float32)
    v[:8] = np.array([0.0, 1.0, -2.0, 3.0, 0.5, -0.5, 1e-6, 1 - 1e-6], np.float32)
    v[8:16] = np.float32(2.0) + np.array([1e-7, -1e-7, 5e-6, -5e-6, 2e-5, -2e-5, 1e-3, -1e-3], np.float32)
    u = rng.uniform(1e-6, 1 - 1e-6, (4096, 2)).astype(np.float32)
    vt = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    out = SGAOracle.sga_sample(vt, T, torch.tensor(u, dtype=torch.float64))
    (jac,) = torch.autograd.grad(out.sum(), vt)
    got_v, got_j = codec.sample(v, u, T)
    ev = rel_err(got_v.cpu().numpy(), out.detach().numpy())
    # jacobian: elementwise relative (magnitudes span orders)
    gj, wj = got_j.cpu().numpy().astype(np.float64), jac.numpy()
    ej = float(np.max(np.abs(gj - wj) / (np.abs(wj) + 1e-3 * np.abs(wj).max())))
    report(gpu_out_dir, "sampler", T=T, rel_err_v=ev, rel_err_jac=ej)
    assert ev < 1e-5
    assert ej < 2e-3, where_bad(gj, wj, 2e-3)


def test_sampler_limit_rounds(gpu_out_dir):
    """T -> small: v_tilde -> round-to-nearest w.p. -> 1 (SURVEY 8(c) test 4)."""
    codec, _, _, _ = get_codec(64)
    rng = np.random.RandomState(1)
    v = (rng.standard_normal(8192) * 3).astype(np.float32)
    v = v[np.abs(v - np.round(v)) < 0.3]
    u = rng.uniform(0.05, 0.95, (v.size, 2)).astype(np.float32)
    vt, _ = codec.sample(v, u, 0.02)
    assert np.abs(vt.cpu().numpy() - np.round(v)).max() < 1e-3


def test_factorized_likelihood(gpu_out_dir):
    codec, orc, orc64, _ = get_codec(64)
    rng = np.random.RandomState(2)
    v = (rng.standard_normal((3, 5, 4, 64)) * 6).astype(np.float32)
    vt = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    p = orc64.eb_likelihood(vt)
    (dp,) = torch.autograd.grad(p.sum(), vt)
    gp, gdp = codec.factorized_likelihood(v)
    e1, e2 = rel_err(gp.cpu().numpy(), p.detach().numpy()), rel_err(gdp.cpu().numpy(), dp.numpy())
    report(gpu_out_dir, "factorized", rel_err_p=e1, rel_err_dp=e2)
    assert e1 < 2e-5 and e2 < 1e-4
    # sum over integers of the box mass ~ 1 per channel (SURVEY 8(c) test 3)
    ks = np.arange(-400, 401, dtype=np.float32)   # init_scale=10 logistic tails reach far
    grid = np.repeat(ks[:, None], 64, axis=1)
    mass, _ = codec.factorized_likelihood(grid)
    assert np.allclose(mass.cpu().numpy().sum(0), 1.0, atol=1e-3)


@pytest.mark.parametrize("sb", SCALE_BOUNDS)
def test_gaussian_likelihood(sb, gpu_out_dir):
    codec, orc, orc64, _ = get_codec(64)
    rng = np.random.RandomState(3)
    n = 20000
    y = (rng.standard_normal(n) * 4).astype(np.float32)
    mu = rng.standard_normal(n).astype(np.float32)
    sr = (rng.standard_normal(n) * 1.2).astype(np.float32)
    sr[:100] = -4.0            # sigma = 0.018, below the 0.11 bound
    y[:50] = mu[:50]           # |y - mu| = 0: sign() = 0
    yt, mt, st = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (y, mu, sr))
    # straight max() on sigma for the unit op (no upstream sign): d/dsraw = 0 below the bound
    sigma = torch.clamp_min(torch.exp(st), sb) if sb > 0 else torch.exp(st)
    v = torch.abs(yt - mt)
    c = 2.0 ** -0.5
    p = 0.5 * torch.erfc(-c * ((0.5 - v) / sigma)) - 0.5 * torch.erfc(-c * ((-0.5 - v) / sigma))
    dy, dm, ds = torch.autograd.grad(p.sum(), [yt, mt, st])
    with scale_bound_of(codec, sb):
        gp, gdy, gdm, gds = (t.cpu().numpy() for t in codec.gaussian_likelihood(y, mu, sr))
    errs = dict(p=rel_err(gp, p.detach().numpy()), dy=rel_err(gdy, dy.numpy()),
                dmu=rel_err(gdm, dm.numpy()), dsr=rel_err(gds, ds.numpy()))
    report(gpu_out_dir, "gaussian", scale_bound=sb, **errs)
    assert errs["p"] < 1e-5 and errs["dy"] < 5e-5 and errs["dmu"] < 5e-5 and errs["dsr"] < 5e-5, errs
    # the two modes really differ on the 100 small-sigma elements and nowhere else
    bounded = torch.clamp_min(torch.exp(st), 0.11).detach().numpy() != torch.exp(st).detach().numpy()
    assert bounded[:100].all() and (gds[:100] == 0).all() == (sb > 0)
    # symmetry p(mu+d) = p(mu-d) and sum_k p = 1
    d = np.float32(1.37)
    a, *_ = codec.gaussian_likelihood(mu + d, mu, sr)
    b, *_ = codec.gaussian_likelihood(mu - d, mu, sr)
    assert np.allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-9)
    ks = np.arange(-400, 401, dtype=np.float32)
    tot, *_ = codec.gaussian_likelihood(ks, np.full_like(ks, 0.3), np.full_like(ks, 1.0))
    assert abs(float(tot.sum()) - 1.0) < 1e-4


@pytest.mark.parametrize("sb", SCALE_BOUNDS)
def test_gaussian_likelihood_reference_fixture(sb):
    """sga_op_gaussian_likelihood vs the outputs of the reference's own
    utils.box_convolved_gaussian_pdf (tests/golden/utils_reference.npz; float64 there, f32 here).
    The vendored formula takes sigma as given: with scale_bound = 0 every fixture row applies (incl.
    sigma = 1e-3), with 0.11 the rows at or above the bound."""
    codec, *_ = get_codec(64)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "utils_reference.npz"))
    ok = g["box_sigma"] >= sb
    assert (~ok).any() == (sb > 0)
    y, mu, sigma, want = (g[k][ok] for k in ("box_y", "box_mu", "box_sigma", "box_out"))
    with scale_bound_of(codec, sb):
        p, *_ = codec.gaussian_likelihood(y.astype(np.float32), mu.astype(np.float32),
                                          np.log(sigma).astype(np.float32))
    p = p.cpu().numpy().astype(np.float64)
    # inputs rounded to f32 move p by up to ~|dp/dy| * 2^-24 |y|: compare with a mixed tolerance
    assert np.allclose(p, want, rtol=2e-4, atol=1e-7), float(np.abs(p - want).max())


GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _prior_fixture_codec():
    """Codec + float64 oracle whose factorized prior is the one of tests/golden/prior_reference.npz."""
    from sga_amd.codec import SGACodec
    if "prior_fx" not in _CODECS:
        fx = np.load(os.path.join(GOLDEN, "prior_reference.npz"))
        C = int(fx["channels"])
        w = dict(sga_amd.make_synthetic_weights(C, seed=0))
        for k in fx.files:
            if k.startswith("eb."):
                w[k] = fx[k]
        _CODECS["prior_fx"] = (SGACodec(w, C, max_batch=4, max_height=96, max_width=96), SGAOracle(w, dtype=torch.float64), fx)
    return _CODECS["prior_fx"]


def test_factorized_prior_vs_reference_executed_fixture(gpu_out_dir):
    """sga_op_factorized_likelihood / sga_op_factorized_density (the kernels of sga.py:101 and
    bb_sga.py:109) against values produced by learned_prior.py's own code (`_logits_cdf`, `cdf_pdf`;
    tests/golden/prior_reference.npz).  Float32 kernels vs float64 reference: relative 2e-5 where the
    value is not in the far tail, absolute otherwise."""
    codec, _, fx = _prior_fixture_codec()
    v = fx["v"]
    p, dp = (t.cpu().numpy().astype(np.float64) for t in codec.factorized_likelihood(v))
    q, dq = (t.cpu().numpy().astype(np.float64) for t in codec.factorized_density(v))
    errs = dict(mass=float(np.abs(p - fx["mass_sign_trick"]).max() / fx["mass_sign_trick"].max()),
                dmass=float(np.abs(dp - fx["dmass_dv"]).max() / np.abs(fx["dmass_dv"]).max()),
                pdf=float(np.abs(q - fx["pdf"]).max() / fx["pdf"].max()),
                dpdf=float(np.abs(dq - fx["dpdf_dv_fd"]).max() / np.abs(fx["dpdf_dv_fd"]).max()))
    report(gpu_out_dir, "prior_reference_fixture", **errs)
    assert np.allclose(p, fx["mass_sign_trick"], rtol=2e-5, atol=2e-8), errs
    assert np.allclose(dp, fx["dmass_dv"], rtol=1e-4, atol=2e-7), errs
    assert np.allclose(q, fx["pdf"], rtol=2e-5, atol=2e-8), errs
    assert np.allclose(dq, fx["dpdf_dv_fd"], rtol=2e-4, atol=2e-7), errs
    # far tails keep digits thanks to the sign trick (a plain sigmoid difference is 0 in float32 there)
    tail = fx["mass_sign_trick"] < 1e-9
    tail &= fx["mass_sign_trick"] > 1e-30
    assert tail.sum() > 50 and (p[tail] > 0).all() and np.allclose(p[tail], fx["mass_sign_trick"][tail], rtol=1e-3, atol=0)


@pytest.mark.parametrize("sb", SCALE_BOUNDS)
def test_lower_bound_branches_in_the_step_kernels(sb, gpu_out_dir):
    """math_ops.py:63-76 inside the kernels of the SGA step (k_gaussian / k_factorized, driven through
    sga_op_rate_terms with fed intermediates): every combination of
      sigma below / above 0.11   x   sign of the gradient reaching sigma,
      p below / above likelihood_bound = 1e-9 (sga.py:134-136, 102-104),
    against float64 autograd of the oracle, whose bound gradient is pinned to the reference-executed
    fixture (tests/test_oracle.py).  sb = 0.11: a built tfc layer (mbt2018.py:77-80), the small sigmas are
    bounded and their gradient follows the rule; sb = 0: sga.py:130-133, raw sigma, every gradient passes."""
    codec, o64, _ = _prior_fixture_codec()
    C, B, H, W = codec.C, 1, 64, 64
    yh, yw, zh, zw = codec.latent_shape(H, W)
    hs, ws = 4 * zh, 4 * zw
    rng = np.random.RandomState(7)
    # element classes along the channel axis; (|y - mu|, sigma) chosen well away from every threshold
    cases = [("sig_lo_g_pos", 0.00, 0.05),    # sigma < 0.11, p falls when sigma grows  -> gradient BLOCKED
             ("sig_lo_g_neg", 0.70, 0.05),    # sigma < 0.11, p grows with sigma        -> passes, chain factor = sigma
             ("sig_hi_g_pos", 0.10, 0.50),    # sigma >= 0.11, either sign passes
             ("sig_hi_g_neg", 1.50, 0.50),
             ("p_below_bound", 3.60, 0.50),   # p ~ 2.8e-10 < 1e-9: -log2 uses 1e-9, gradient passes (it is < 0)
             ("p_zero", 6.00, 0.11)]          # p underflows to 0
    k = np.arange(C) % len(cases)
    dist = np.array([c[1] for c in cases], np.float32)[k]
    sigma = np.array([c[2] for c in cases], np.float32)[k]
    mu = rng.standard_normal((B, hs, ws, C)).astype(np.float32)
    sraw = np.broadcast_to(np.log(sigma), (B, hs, ws, C)).astype(np.float32).copy()
    ms = np.concatenate([mu, sraw], -1)
    sgn = np.where(rng.rand(B, yh, yw, C) < 0.5, -1.0, 1.0).astype(np.float32)
    yt = (mu[:, :yh, :yw] + sgn * dist).astype(np.float32)
    # z_tilde: bulk plus far-tail values where the factorized mass is below 1e-9
    zt = (rng.standard_normal((B, zh, zw, C)) * 3).astype(np.float32)
    zt[..., ::5] = np.where(rng.rand(B, zh, zw, len(range(0, C, 5))) < 0.5, -400.0, 400.0)
    with scale_bound_of(codec, sb):
        got = codec.rate_terms(yt, zt, ms, H, W, loss_scale=1.0)
    # float64 oracle of the same sub-graph (sga.py:100-104, 126-146)
    from oracle.sga_oracle import LIKELIHOOD_BOUND
    ytt = torch.tensor(yt, dtype=torch.float64, requires_grad=True)
    ztt = torch.tensor(zt, dtype=torch.float64, requires_grad=True)
    mst = torch.tensor(ms, dtype=torch.float64, requires_grad=True)
    mu_t, sr_t = mst[..., :C][:, :yh, :yw], mst[..., C:][:, :yh, :yw]
    p_y_raw = o64.gauss_likelihood(ytt, mu_t, torch.exp(sr_t), sb)
    p_z_raw = o64.eb_likelihood(ztt)
    p_y, p_z = lower_bound(p_y_raw, LIKELIHOOD_BOUND), lower_bound(p_z_raw, LIKELIHOOD_BOUND)
    den = np.log(2.0) * H * W
    y_bpp, z_bpp = -torch.log(p_y).sum() / den, -torch.log(p_z).sum() / den
    g_yt, g_ms, g_zt = torch.autograd.grad(y_bpp + z_bpp, [ytt, mst, ztt])
    # the crafted classes really are where they are meant to be
    pr = p_y_raw.detach().numpy()
    assert (pr[..., k == 4] < 3e-10).all() and (pr[..., k == 4] > 0).all() and (pr[..., k == 5] < 1e-30).all()
    assert (p_z_raw.detach().numpy()[..., ::5] < 1e-12).all()
    gs = g_ms[..., C:][:, :yh, :yw].numpy()
    if sb > 0:
        assert (gs[..., k == 0] == 0).all()                              # blocked
        assert (gs[..., k == 1] < 0).all()
    else:
        # raw sigma = 0.05: the mass at distance 0 is 1 - 2e-23 (its sigma-gradient passes but underflows);
        # at distance 0.7 the box [0.2, 1.2] sigma away holds 3e-5 of the mass and grows fast with sigma
        assert (gs[..., k == 1] < 0).all() and (pr[..., k == 1] < 1e-4).all()
    assert (gs[..., k == 2] > 0).all() and (gs[..., k == 3] < 0).all()
    assert (gs[..., k == 4] != 0).all()                                   # bounded p still has a gradient
    for name, a, b_, tol in (("g_yt", got["g_yt"], g_yt, 2e-4), ("g_ms", got["g_ms"], g_ms, 2e-4),
                             ("g_zt", got["g_zt"], g_zt, 1e-3)):     # deep-tail dp/dv in float32: 4e-4
        a, b_ = a.cpu().numpy().astype(np.float64), b_.numpy()
        for ci, c in enumerate(cases if name != "g_zt" else []):
            sel = (k == ci)
            parts = [a[..., :C][..., sel], b_[..., :C][..., sel]] if name == "g_ms" else [a[..., sel], b_[..., sel]]
            report(gpu_out_dir, "lower_bound_branch", grad=name, case=c[0], rel_err=rel_err(*parts))
        scale = np.abs(b_).max()
        assert np.allclose(a, b_, rtol=tol, atol=tol * 1e-3 * scale), (name, where_bad(a, b_, tol))
    # exact zeros where the rule blocks, also in float32
    a = got["g_ms"].cpu().numpy()[..., C:][:, :yh, :yw]
    if sb > 0:
        assert (a[..., k == 0] == 0).all()
    assert np.allclose(got["est_y_bpp"].cpu().numpy(), [float(y_bpp.detach())], rtol=2e-5)
    assert np.allclose(got["est_z_bpp"].cpu().numpy(), [float(z_bpp.detach())], rtol=2e-5)


def test_lower_bound_truth_table():
    """math_ops.py:63-76 (oracle side; the kernels' use of it is covered by step parity)."""
    x = torch.tensor([0.5, 0.5, 2.0, 2.0], requires_grad=True)
    g = torch.tensor([1.0, -1.0, 1.0, -1.0])
    (gx,) = torch.autograd.grad(lower_bound(x, 1.0), x, g)
    assert gx.tolist() == [0.0, -1.0, 1.0, -1.0]


def test_adam_bit_exact(gpu_out_dir):
    """sga_adam vs the f32-pinned restatement of adam.py: bit-for-bit over 50 updates,
    and vs the committed fixture generated from the reference's own adam.py."""
    codec, *_ = get_codec(64)
    rng = np.random.RandomState(4)
    n = 10007
    p0 = rng.standard_normal(n).astype(np.float32)
    opt = AdamF32(lr=0.005)
    p_ref = p0.copy()
    p = torch.tensor(p0, device="cuda")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, 51):
        g = (rng.standard_normal(n) * (1e-3 if t % 2 else 3.0)).astype(np.float32)
        (p_ref,) = opt.update([p_ref], [g])
        codec.adam(p, torch.tensor(g, device="cuda"), m, v, t, lr=0.005)
        assert np.array_equal(p.cpu().numpy(), p_ref), f"adam differs at t={t}"
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "adam_reference.npz"))
    p = torch.tensor(fx["p0"], device="cuda")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, int(fx["steps"]) + 1):
        codec.adam(p, torch.tensor(fx["grads"][t - 1], device="cuda"), m, v, t, lr=float(fx["lr"]))
        if t in fx["checkpoints"]:
            want = fx[f"p_after_{t}"]
            # the fixture is the reference's adam.py under numpy 2 (promotes to float64): the
            # float32 arithmetic of numpy 1.17 drifts from it by <= 7.3e-6 abs over 2000 updates
            # (measured with the f32-pinned restatement, tests/test_oracle.py)
            tol = 3e-7 if t <= 3 else 2e-5
            assert np.abs(p.cpu().numpy() - want).max() < tol, f"fixture t={t}"
