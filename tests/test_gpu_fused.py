"""The fused GDN tile kernel (csrc/gdn_fused.hip: split-K slab sum or 3-channel conv prologue +
resident-operand C x C contraction + epilogue, one launch) against the launches it replaces
(conv -> splitk_reduce -> stand-alone GDN kernel; SGA_FUSED_GDN=0), and the IGDN post-phase of the big
transposed convolution (conv_mfma.hip, POST = 1) against a separate IGDN launch: the same f32 fmaf chains in the
same order, so the two paths must agree BIT FOR BIT -- in the encoder (GDN forward,
nn_models.py:17-25), in one SGA step (IGDN forward / backward, nn_models.py:51-59) and over a short
run.  Parity of either path with the oracle is covered by the other GPU test files.

The ablation switches (SGA_FUSED_GDN, SGA_KEEP_U, SGA_FUSED_POST64, SGA_GS3_GEMM, SGA_FUSED_MSE, SGA_FUSED_BOUNDARY,
SGA_FORK_AT, SGA_FORK2_NAME) exist only in the LABORATORY build of the library (`make EXPERIMENTS=1` -> libsga_hip_lab.so,
`SGACodec(..., lab=True)`); the product library has none of them (tests/test_host.py).  With no switch set the two builds
run the same launches: `test_lab_build_equals_the_product_build` checks that bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402


def _pair(C, B, H, W):
    """fused: the tile kernels / POST phase with SGA_KEEP_U=1 (the IGDN also stores its input u and the data-gradient
    reads it: the arithmetic of the stand-alone launches); legacy: SGA_FUSED_GDN=0; default: what ships (s and v
    stored, u = v / s by v_rcp_f32 in igdn*.bwd -- not bit-equal by construction, compared to 1e-5 below and with
    the oracle everywhere else)."""
    from sga_amd.codec import SGACodec
    w = sga_amd.make_synthetic_weights(C, seed=0)
    old = {k: os.environ.get(k) for k in ("SGA_FUSED_GDN", "SGA_KEEP_U")}
    try:
        os.environ["SGA_FUSED_GDN"] = "1"; os.environ["SGA_KEEP_U"] = "1"
        fused = SGACodec(w, C, B, H, W, lab=True)
        os.environ["SGA_FUSED_GDN"] = "0"
        legacy = SGACodec(w, C, B, H, W, lab=True)
        os.environ.pop("SGA_FUSED_GDN"); os.environ.pop("SGA_KEEP_U")
        default = SGACodec(w, C, B, H, W, lab=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return fused, legacy, default


# shapes: every tile shape of gdn_fused.hip (general / 32-row "small", C = 64..256), ragged sizes,
# split and unsplit producers
SHAPES = [(64, 2, 64, 64), (64, 1, 50, 70), (128, 1, 37, 41), (192, 2, 64, 48), (192, 1, 256, 256),
          (256, 1, 96, 80), (192, 3, 128, 192), (256, 1, 200, 264),
          # >= 1024 tiles in gs2.fwd: the IGDN runs as the post-phase of the convolution launch (conv_mfma.hip POST)
          (192, 2, 512, 512), (192, 2, 520, 504),
          # the same at cfg 4's width (README.md:58-60): 256 x 256 tile, the post-phase in four 64-row parts
          (256, 1, 520, 504), (256, 2, 512, 512)]


@pytest.mark.parametrize("C,B,H,W", SHAPES)
def test_fused_equals_unfused_bitwise(C, B, H, W):
    fused, legacy, default = _pair(C, B, H, W)
    x = np.random.RandomState(C + H).rand(B, H, W, 3).astype(np.float32)
    ya, za = fused.encode(x)
    yb, zb = legacy.encode(x)
    assert torch.equal(ya, yb) and torch.equal(za, zb)            # GDN forward (analysis)
    ra = fused.step_grads(x, ya, za, 0.4, 0.01, seed=3, it=5)
    rb = legacy.step_grads(x, ya, za, 0.4, 0.01, seed=3, it=5)
    assert torch.equal(ra["gy"], rb["gy"]) and torch.equal(ra["gz"], rb["gz"])
    assert ra["rd_loss"] == rb["rd_loss"] and ra["train_mse"] == rb["train_mse"]
    assert float(ra["gy"].abs().max()) > 0
    # what ships: the forward pass stores s and v, the IGDN data-gradient forms u = v / s
    rc = default.step_grads(x, ya, za, 0.4, 0.01, seed=3, it=5)
    assert rc["rd_loss"] == ra["rd_loss"] and torch.equal(rc["gz"], ra["gz"])          # forward and hyper branch untouched
    err = float((rc["gy"] - ra["gy"]).abs().max() / ra["gy"].abs().max())
    assert 0 < err < 1e-5, err
    its = 12 if H * W > 40000 else 30
    a = fused.run(x, 0.01, its=its, seed=1)
    b = legacy.run(x, 0.01, its=its, seed=1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2][:, [0, 1, 4, 5, 6]], b[2][:, [0, 1, 4, 5, 6]])
    fused.close(); legacy.close(); default.close()


@pytest.mark.parametrize("C,B,H,W", [(192, 8, 256, 256), (192, 1, 496, 1000)])
def test_post_phase_in_the_64_row_instance_equals_the_separate_igdn(C, B, H, W, monkeypatch):
    """nn_models.py:52-55 (layer 1 of g_s at cfg 2: 512 unsplit 64-row tiles): the IGDN as the post-phase of the 4-wave
    64 x 192 convolution instance (default since the joint schedule sweep; SGA_FUSED_POST64=0 turns it off) against the
    separate tile-kernel launch -- same contraction order, bit for bit."""
    from sga_amd.codec import SGACodec
    w = sga_amd.make_synthetic_weights(C, seed=0)
    monkeypatch.setenv("SGA_FUSED_POST64", "1")
    on = SGACodec(w, C, B, H, W, lab=True)
    monkeypatch.setenv("SGA_FUSED_POST64", "0")
    off = SGACodec(w, C, B, H, W, lab=True)
    x = np.random.RandomState(7).rand(B, H, W, 3).astype(np.float32)
    y, z = off.encode(x)
    on.profile_begin(); ra = on.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5); names = [k["name"] for k in on.profile_end()]
    rb = off.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    assert any(n.replace(" ", "").startswith("conv_mfma_kernel<1,3,2,2,0,false,0,1>") for n in names), names
    assert torch.equal(ra["gy"], rb["gy"]) and torch.equal(ra["gz"], rb["gz"]) and ra["rd_loss"] == rb["rd_loss"]
    a = on.run(x, 0.01, its=6, seed=1); b = off.run(x, 0.01, its=6, seed=1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    on.close(); off.close()


def _codec_env(name, value, *args):
    from sga_amd.codec import SGACodec
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return SGACodec(*args, lab=True)
    finally:
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old


@pytest.mark.parametrize("C,B,H,W", [(64, 2, 64, 64), (64, 1, 50, 70), (128, 3, 37, 41), (192, 2, 256, 256)])
def test_distortion_in_the_gs3_epilogue_equals_the_separate_kernel(C, B, H, W):
    """sga.py:150,161,170-173: squared-error sums and d loss / d x_tilde computed in the epilogue of the C -> 3
    transposed convolution (deconv3.hip, MSE = true) vs the stand-alone k_mse launch (SGA_FUSED_MSE=0).  The
    gradient image is the same expression on the same values: gradients BIT-equal; the sums are f32 partials
    over different groupings accumulated in f64: equal to f32 rounding."""
    w = sga_amd.make_synthetic_weights(C, seed=0)
    fused = _codec_env("SGA_FUSED_MSE", "1", w, C, B, H, W)
    plain = _codec_env("SGA_FUSED_MSE", "0", w, C, B, H, W)
    x = np.random.RandomState(C + W).rand(B, H, W, 3).astype(np.float32)
    y, z = fused.encode(x)
    ra = fused.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    rb = plain.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    assert torch.equal(ra["gy"], rb["gy"]) and torch.equal(ra["gz"], rb["gz"])
    assert float(ra["gy"].abs().max()) > 0
    assert ra["train_mse"] == pytest.approx(rb["train_mse"], rel=2e-7)
    assert ra["rd_loss"] == pytest.approx(rb["rd_loss"], rel=2e-7)
    a = fused.run(x, 0.01, its=20, seed=1)
    b = plain.run(x, 0.01, its=20, seed=1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.allclose(a[2][:, :4], b[2][:, :4], rtol=1e-6, atol=0, equal_nan=True)
    fused.close(); plain.close()


@pytest.mark.parametrize("C,B,H,W", [(64, 2, 64, 64), (128, 3, 37, 41), (192, 2, 256, 256)])
def test_step_boundary_kernel_equals_the_three_launches(C, B, H, W):
    """Adam (adam.py:40-56), the relaxation of the next iteration (sga.py:86-98, 111-121) and the per-iteration
    scalars / context advance as ONE launch (k_step_boundary) vs three (SGA_FUSED_BOUNDARY=0): the same
    arithmetic per element, so latents, metrics and the per-iteration trace agree BIT FOR BIT -- also when the
    run is cut into pieces with other entry points using the workspace in between, and for a sibling relaxation."""
    w = sga_amd.make_synthetic_weights(C, seed=0)
    one = _codec_env("SGA_FUSED_BOUNDARY", "1", w, C, B, H, W)
    three = _codec_env("SGA_FUSED_BOUNDARY", "0", w, C, B, H, W)
    x = np.random.RandomState(C + H).rand(B, H, W, 3).astype(np.float32)
    a = one.run(x, 0.01, its=60, seed=5, trace=True)
    b = three.run(x, 0.01, its=60, seed=5, trace=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2].nan_to_num(), b[2].nan_to_num())
    assert torch.equal(a[3], b[3])                                   # [its, 4] per-iteration trace
    # in pieces, with an evaluation (which overwrites the relaxed latents in the workspace) in between
    one.run_begin(x, 0.01, its=60, seed=5)
    one.run_steps(17)
    y, z = one.run_latents()
    one.evaluate(x, torch.round(y), torch.round(z))
    one.run_steps(43)
    y, z, tr = one.run_latents(trace=True)
    assert torch.equal(torch.round(y), a[0]) and torch.equal(torch.round(z), a[1])
    assert torch.equal(tr[:60], a[3])
    one.close(); three.close()


@pytest.mark.parametrize("C,B,H,W", [(64, 2, 64, 64), (64, 1, 50, 70), (192, 2, 256, 256), (256, 1, 96, 80)])
def test_gs3_as_gemm_plus_col2im_equals_the_halo_kernel(C, B, H, W):
    """The C -> 3 transposed convolution (nn_models.py:60-63) as a plain GEMM over all 25 x 3 kernel columns + a col2im
    kernel with the distortion in it (deconv3_gemm.hip; the default since the joint schedule sweep, SGA_GS3_GEMM=0: halo kernel)
    against the default halo-tiled kernel: the same products summed in another order -- reconstruction, gradients and
    the distortion sums agree to float32 rounding, and ragged sizes crop identically."""
    w = sga_amd.make_synthetic_weights(C, seed=0)
    gemm = _codec_env("SGA_GS3_GEMM", "1", w, C, B, H, W)
    halo = _codec_env("SGA_GS3_GEMM", "0", w, C, B, H, W)
    x = np.random.RandomState(C + W).rand(B, H, W, 3).astype(np.float32)
    y, z = halo.encode(x)
    ra = gemm.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    rb = halo.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    err = float((ra["gy"] - rb["gy"]).abs().max() / rb["gy"].abs().max())
    assert 0 < err < 1e-5, err
    assert torch.equal(ra["gz"], rb["gz"])
    assert ra["train_mse"] == pytest.approx(rb["train_mse"], rel=1e-6) and ra["rd_loss"] == pytest.approx(rb["rd_loss"], rel=1e-6)
    (ma, xa), (mb, xb) = gemm.evaluate(x, torch.round(y), torch.round(z), want_x_hat=True), \
        halo.evaluate(x, torch.round(y), torch.round(z), want_x_hat=True)
    assert float((xa - xb).abs().max()) < 1e-5 * float(xb.abs().max())
    assert torch.allclose(ma[:, [0, 1, 4]], mb[:, [0, 1, 4]], rtol=1e-3)
    gemm.close(); halo.close()


def test_results_do_not_depend_on_the_schedule(monkeypatch):
    """where the hyper branch is forked (timed per geometry in runs of >= 100 iterations), whether it is forked
    at all, and whether the step graph is replayed or launched eagerly are schedule choices -- the same kernels with the same
    arguments -- so the latents and metrics of a run must agree BIT FOR BIT across them.  (Regression: the branch's split-K
    target used to be a property of the stream it ran on, so unforked execution summed in another order.)"""
    from sga_amd.codec import SGACodec
    C, B, H, W = 192, 2, 128, 128
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(3).rand(B, H, W, 3).astype(np.float32)

    def run(env):
        for k in ("SGA_FORK_AT", "SGA_FORK_NAME", "SGA_FORK2_NAME", "SGA_NO_OVERLAP", "SGA_NO_GRAPH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = SGACodec(w, C, B, H, W, lab=True)
        before = c.fork_point()
        out = c.run(x, 0.01, its=130, seed=4)            # >= 100 iterations: the timed fork point when nothing pins it
        after = c.fork_point()
        if not env:                                      # sga_get_fork_point: reporting only
            assert before == "untimed" and after in ("start", "gs2.fwd", "gs3.fwd"), (before, after)
        elif "SGA_FORK_AT" in env or "SGA_FORK_NAME" in env:
            assert before == after == "pinned", (before, after)
        g = c.step_grads(x, out[0], out[1], 0.3, 0.01, seed=2, it=7)
        c.close()
        return out, g

    ref, gref = run({})
    for env in ({"SGA_FORK_AT": "0"}, {"SGA_FORK_NAME": "gs2.fwd"}, {"SGA_FORK_NAME": "gs3.fwd"},
                {"SGA_FORK_NAME": "gs2.fwd", "SGA_FORK2_NAME": "gs2.bwd"}, {"SGA_NO_OVERLAP": "1"}, {"SGA_NO_GRAPH": "1"}):
        out, g = run(env)
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), env
        assert torch.equal(out[2][:, [0, 1, 4, 5, 6]], ref[2][:, [0, 1, 4, 5, 6]]), env
        assert torch.equal(g["gy"], gref["gy"]) and torch.equal(g["gz"], gref["gz"]) and g["rd_loss"] == gref["rd_loss"], env


def test_lab_build_equals_the_product_build():
    """libsga_hip_lab.so = libsga_hip.so + switches: with none of them set, a step and a run agree bit for bit."""
    from sga_amd.codec import SGACodec
    C, B, H, W = 192, 2, 128, 128
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(5).rand(B, H, W, 3).astype(np.float32)
    prod, lab = SGACodec(w, C, B, H, W), SGACodec(w, C, B, H, W, lab=True)
    assert prod.lib is not lab.lib
    y, z = prod.encode(x)
    y2, z2 = lab.encode(x)
    assert torch.equal(y, y2) and torch.equal(z, z2)
    ga, gb = prod.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5), lab.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    assert torch.equal(ga["gy"], gb["gy"]) and torch.equal(ga["gz"], gb["gz"]) and ga["rd_loss"] == gb["rd_loss"]
    a, b = prod.run(x, 0.01, its=120, seed=1), lab.run(x, 0.01, its=120, seed=1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    prod.close(); lab.close()
