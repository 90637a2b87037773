"""Model parameters of the mean-scale hyperprior (mbt2018) as the SGA hot path consumes them.

The hot path treats every weight as a constant (sga.py:164 differentiates w.r.t. the latents
only), so what crosses the C-ABI are the *effective* tensors, after tfc's
reparameterisations:

  conv kernels   HWIO  (kh, kw, C_in, C_out)                       nn_models.py:14-29,48-63,85-96,152-163
  GDN / IGDN     beta[C] and gamma[C_in(j), C_out(i)]: n_i = beta_i + sum_j gamma[j,i] x_j^2
  factorized prior (tfc EntropyBottleneck; restated in-tree at learned_prior.py:34-66)
                 m_k = softplus(matrix_k)  shapes (C,3,1) (C,3,3) (C,3,3) (C,1,3)
                 b_k                       shapes (C,3,1) (C,3,1) (C,3,1) (C,1,1)
                 f_k = tanh(factor_k)      shapes (C,3,1) x3

`make_synthetic_weights` is the deterministic "trained-like" generator of SURVEY.md 8(c):
the released checkpoints live on Google Drive (README.md:100-103) and are not available
offline, so benches and parity tests run on these.  A real checkpoint enters through
`tf_checkpoint.py` and produces the same dict.
"""
from __future__ import annotations

import hashlib
import numpy as np

def layer_shapes(C: int, bb: bool = False):
    """Shapes of every effective tensor for num_filters=C. bb=True: mbt2018_bb variant
    (hyper-analysis emits 2C channels: z_mean, z_logvar; bb_sga.py:69)."""
    C15 = int(C * 1.5)          # nn_models.py:157
    ha_out = 2 * C if bb else C
    s = {}
    for i, cin in enumerate([3, C, C, C]):
        s[f"ga.k{i}"] = (5, 5, cin, C)
        s[f"ga.b{i}"] = (C,)
    for i in range(3):
        s[f"ga.beta{i}"] = (C,)
        s[f"ga.gamma{i}"] = (C, C)
    for i, cout in enumerate([C, C, C, 3]):
        s[f"gs.k{i}"] = (5, 5, C, cout)
        s[f"gs.b{i}"] = (cout,)
    for i in range(3):
        s[f"gs.beta{i}"] = (C,)
        s[f"gs.gamma{i}"] = (C, C)
    s["ha.k0"] = (3, 3, C, C)
    s["ha.b0"] = (C,)
    s["ha.k1"] = (5, 5, C, C)
    s["ha.b1"] = (C,)
    s["ha.k2"] = (5, 5, C, ha_out)        # use_bias=False, nn_models.py:95
    s["hs.k0"] = (5, 5, C, C)
    s["hs.b0"] = (C,)
    s["hs.k1"] = (5, 5, C, C15)
    s["hs.b1"] = (C15,)
    s["hs.k2"] = (3, 3, C15, 2 * C)
    s["hs.b2"] = (2 * C,)
    dims = (1, 3, 3, 3, 1)
    for k in range(4):
        s[f"eb.m{k}"] = (C, dims[k + 1], dims[k])
        s[f"eb.b{k}"] = (C, dims[k + 1], 1)
        if k < 3:
            s[f"eb.f{k}"] = (C, dims[k + 1], 1)
    return s


def _softplus(x):
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)


def make_synthetic_weights(C: int = 192, seed: int = 0, bb: bool = False) -> dict:
    """Deterministic trained-like parameters (float32 numpy), SURVEY.md 8(c) last row.

    Scales are chosen so that on uniform-random images the latents y span a few integer
    bins (std ~ 2-3), predicted scales are O(1), and the reconstruction is O(1): the
    regime in which the SGA relaxation, both entropy models and every conv matter.
    """
    rng = np.random.RandomState(seed)
    shapes = layer_shapes(C, bb)
    w = {}

    def conv(name, gain=1.0, deconv=False):
        kh, kw, cin, cout = shapes[name]
        # a stride-2 transposed conv sums ~ (kh*kw/4)*cin products per output sample
        fan = kh * kw * cin / (4.0 if deconv else 1.0)
        w[name] = (rng.standard_normal(shapes[name]) * (gain / np.sqrt(fan))).astype(np.float32)

    def bias(name, scale=0.05, mean=0.0):
        w[name] = (mean + scale * rng.standard_normal(shapes[name])).astype(np.float32)

    def gdn(prefix, i):
        # tfc init: beta = 1, gamma = 0.1 * I; add small positive off-diagonals so the
        # channel contraction is a real C x C GEMM in tests.
        w[f"{prefix}.beta{i}"] = (1.0 + 0.1 * rng.rand(C)).astype(np.float32)
        g = 0.1 * np.eye(C) + (0.02 / C) * rng.rand(C, C) * 10.0
        w[f"{prefix}.gamma{i}"] = g.astype(np.float32)

    conv("ga.k0", 3.0); bias("ga.b0")
    conv("ga.k1", 2.0); bias("ga.b1")
    conv("ga.k2", 2.0); bias("ga.b2")
    conv("ga.k3", 2.0); bias("ga.b3", 0.3)
    for i in range(3):
        gdn("ga", i)
    conv("gs.k0", 0.27, deconv=True); bias("gs.b0")
    conv("gs.k1", 1.0, deconv=True); bias("gs.b1")
    conv("gs.k2", 1.0, deconv=True); bias("gs.b2")
    conv("gs.k3", 0.15, deconv=True); bias("gs.b3", 0.02, mean=0.5)
    for i in range(3):
        gdn("gs", i)
    conv("ha.k0", 1.0); bias("ha.b0")
    conv("ha.k1", 1.4); bias("ha.b1")
    conv("ha.k2", 3.0)
    conv("hs.k0", 1.0, deconv=True); bias("hs.b0", 0.1, mean=0.1)
    conv("hs.k1", 1.4, deconv=True); bias("hs.b1", 0.1, mean=0.1)
    conv("hs.k2", 1.0)
    w["hs.k2"][..., C:] *= np.float32(0.5)      # log-scale half: sigma = exp(.) within ~[0.3, 5]
    b2 = 0.05 * rng.standard_normal(2 * C)
    b2[C:] += 0.3                      # sigma = exp(.) ~ 1.3
    w["hs.b2"] = b2.astype(np.float32)

    # Factorized prior at its tfc initialisation (visible in-tree at learned_prior.py:35,44-65):
    # init_scale=10, filters=(3,3,3); factors perturbed away from 0 so tanh terms are exercised.
    dims = (1, 3, 3, 3, 1)
    scale = 10.0 ** (1.0 / 4.0)
    for k in range(4):
        init = np.log(np.expm1(1.0 / scale / dims[k + 1]))
        raw = init + 0.1 * rng.standard_normal(shapes[f"eb.m{k}"])
        w[f"eb.m{k}"] = _softplus(raw).astype(np.float32)
        w[f"eb.b{k}"] = rng.uniform(-0.5, 0.5, shapes[f"eb.b{k}"]).astype(np.float32)
        if k < 3:
            w[f"eb.f{k}"] = np.tanh(0.3 * rng.standard_normal(shapes[f"eb.f{k}"])).astype(np.float32)
    return w


def make_lowpass_images(B: int, H: int, W: int, seed: int = 0, passes: int = 2, k: int = 9) -> np.ndarray:
    """Deterministic "natural-ish" synthetic images: uniform noise, `passes` k x k box blurs with
    reflected borders, every image channel stretched back to [0, 1].  float32 [B,H,W,3].  Pure numpy (float64 sums), so
    the GPU box, the build container and the fixture generators produce the same pixels."""
    x = np.random.RandomState(seed).rand(B, H, W, 3)
    r = k // 2
    for _ in range(passes):
        for axis, n in ((1, H), (2, W)):
            pad = [(0, 0)] * 4
            pad[axis] = (r, r)
            c = np.cumsum(np.pad(x, pad, mode="reflect"), axis=axis)
            c = np.concatenate([np.zeros_like(np.take(c, [0], axis=axis)), c], axis=axis)
            x = (np.take(c, range(k, n + k), axis=axis) - np.take(c, range(0, n), axis=axis)) / k
    lo, hi = x.min(axis=(1, 2), keepdims=True), x.max(axis=(1, 2), keepdims=True)
    return ((x - lo) / (hi - lo)).astype(np.float32)


def load_weights_npz(path: str) -> dict:
    """Effective tensors stored by `save_weights_npz` (float16-representable values kept as float16 on disk)."""
    with np.load(path) as f:
        return {k: f[k].astype(np.float32) for k in f.files}


def save_weights_npz(path: str, w: dict) -> dict:
    """Rounds every tensor to float16-representable values (that IS the stored model), writes them compressed and returns
    the float32 dict a later `load_weights_npz` yields."""
    q = {k: np.asarray(v, np.float32).astype(np.float16) for k, v in w.items()}
    np.savez_compressed(path, **q)
    return {k: v.astype(np.float32) for k, v in q.items()}


def weights_digest(w: dict) -> str:
    """sha256 over all tensors in key order: committed in tests instead of the tensors."""
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k], dtype=np.float32).tobytes())
    return h.hexdigest()


def check_weights(w: dict, C: int, bb: bool = False) -> None:
    shapes = layer_shapes(C, bb)
    for k, shp in shapes.items():
        if k not in w:
            raise KeyError(f"missing weight tensor {k}")
        if tuple(w[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {tuple(w[k].shape)} != expected {shp}")
        if w[k].dtype != np.float32:
            raise TypeError(f"{k}: dtype {w[k].dtype} != float32")
