"""`SGACodec`: the host-side object behind which the reference's tf.Session interactions of
sga.py sit.  PyTorch-ROCm tensors own the device memory and the stream;
every arithmetic step is a call through the C ABI into libsga_hip.so.

Reference interaction                                              -> method
  tf.train.Saver().restore                     (sga.py:180-182)    -> SGACodec(weights, ...)
  sess.run([y_init, z_init], {x})              (sga.py:207)        -> encode
  sess.run([rd_gradients, rd_loss, ...])       (sga.py:212-214)    -> step_grads
  Adam.update                                  (sga.py:215)        -> adam
  the per-batch loop                           (sga.py:207-247)    -> run
  sess.run(eval_tensors, {y_tilde:.., ..})     (sga.py:244-245)    -> evaluate
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .weights import check_weights

EVAL_FIELDS = ["mse", "psnr", "msssim", "msssim_db", "est_bpp", "est_y_bpp", "est_z_bpp"]  # sga.py:183
BB_EVAL_FIELDS = EVAL_FIELDS + ["est_bpp_back"]                                             # bb_sga.py:181


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class SGACodec:
    def __init__(self, weights: dict, num_filters: int, max_batch: int, max_height: int,
                 max_width: int, device: str | torch.device = "cuda:0", bits_back: bool = False,
                 precision: str = "default", scale_bound: float = _lib.SCALE_BOUND_NONE, lab: bool = False):
        """scale_bound: lower bound on the conditional's sigma in step / run / evaluate.  0 (default) mirrors
        sga.py:130-133 and its siblings, which call `_likelihood` on a tfc layer that is never built (tfc 1.3 bounds
        the scale in build()); 0.11 mirrors a built layer, mbt2018.py:77-80 (`base_compress` switches to it for the
        call).  include/sga_hip.h, SGA_SCALE_BOUND_*."""
        if not torch.cuda.is_available():
            raise RuntimeError("SGACodec needs a ROCm GPU (gfx950); there is no CPU fallback")
        # lab=True: the laboratory build of the library (libsga_hip_lab.so), which honours the ablation switches of
        # DESIGN_EXPERIMENTS.md; tests compare the shipped kernels with the launches they replace through it
        self.lib = _lib.load_lab_library() if lab else _lib.load_library()
        self.device = torch.device(device)
        self.C = int(num_filters)
        self.bits_back = bool(bits_back)
        check_weights(weights, self.C, bits_back)
        self._weights_for_ec = {k: v for k, v in weights.items() if k.startswith("eb.")}
        # medians of the factorized prior (tfc `quantiles[:, 0, 1]`): centre of the rounding in
        # mbt2018.py:69 / map.py:83; None for synthetic weights (= 0)
        med = weights.get("eb.medians")
        self.medians = None if med is None else np.ascontiguousarray(med, dtype=np.float32).reshape(self.C)
        self._ec = None
        self.max_batch, self.max_height, self.max_width = int(max_batch), int(max_height), int(max_width)
        torch.cuda.set_device(self.device)
        # a dedicated non-null stream: hipGraph capture is not allowed on the legacy null stream
        # (SGA_MAIN_PRIORITY: experiment switch, scripts/side_priority.py)
        self.stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("SGA_MAIN_PRIORITY", "0")))
        self.precision = precision
        self.scale_bound = float(scale_bound)
        cfg = _lib.SgaConfig(self.C, self.max_batch, self.max_height, self.max_width, int(bits_back),
                             _lib.PRECISIONS[precision], self.scale_bound, 0)
        w = _lib.SgaWeights()
        keep = []

        def fp(name):
            a = np.ascontiguousarray(weights[name], dtype=np.float32)
            keep.append(a)
            return a.ctypes.data_as(_lib._FP)

        for i in range(4):
            w.ga_kernel[i] = fp(f"ga.k{i}"); w.ga_bias[i] = fp(f"ga.b{i}")
            w.gs_kernel[i] = fp(f"gs.k{i}"); w.gs_bias[i] = fp(f"gs.b{i}")
            w.eb_matrix[i] = fp(f"eb.m{i}"); w.eb_bias[i] = fp(f"eb.b{i}")
        for i in range(3):
            w.ga_beta[i] = fp(f"ga.beta{i}"); w.ga_gamma[i] = fp(f"ga.gamma{i}")
            w.gs_beta[i] = fp(f"gs.beta{i}"); w.gs_gamma[i] = fp(f"gs.gamma{i}")
            w.ha_kernel[i] = fp(f"ha.k{i}"); w.hs_kernel[i] = fp(f"hs.k{i}")
            w.hs_bias[i] = fp(f"hs.b{i}")
            w.eb_factor[i] = fp(f"eb.f{i}")
        w.ha_bias[0] = fp("ha.b0"); w.ha_bias[1] = fp("ha.b1")
        self.handle = C.c_void_p(0)
        st = self.lib.sga_create(C.byref(self.handle), C.byref(cfg), C.byref(w))
        _lib.check(self.lib, None, st, "sga_create")
        del keep

    def close(self):
        if getattr(self, "handle", None) and self.handle.value:
            self.lib.sga_destroy(self.handle)
            self.handle = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ---------------------------------------------------------------------------
    def _t(self, a, shape=None):
        t = torch.as_tensor(a, dtype=torch.float32, device=self.device).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return C.c_void_p(self.stream.cuda_stream)

    def _exit(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def _chk(self, st, what):
        _lib.check(self.lib, self.handle, st, what)

    def latent_shape(self, H, W):
        yh, yw, zh, zw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.sga_latent_shape(self.handle, H, W, C.byref(yh), C.byref(yw),
                                            C.byref(zh), C.byref(zw)), "sga_latent_shape")
        return yh.value, yw.value, zh.value, zw.value

    def _shapes(self, x):
        B, H, W, c3 = x.shape
        if c3 != 3:
            raise ValueError("x must be [B,H,W,3]")
        yh, yw, zh, zw = self.latent_shape(H, W)
        zc = 2 * self.C if self.bits_back else self.C
        return B, H, W, (B, yh, yw, self.C), (B, zh, zw, zc)

    def set_relaxation(self, relaxation="sga", schedule="exp0"):
        """Sibling inference methods on the same step (danneal.py / unoise.py / ste.py / map.py)."""
        self._chk(self.lib.sga_set_relaxation(self.handle, _lib.RELAXATIONS[relaxation],
                                              _lib.SCHEDULES[schedule]), "sga_set_relaxation")

    def set_scale_bound(self, scale_bound: float):
        """Change the sigma bound of the live handle (the bound is part of the graph cache's key: graphs captured under the
        other value stay cached; nothing is synchronised or dropped)."""
        self._chk(self.lib.sga_set_scale_bound(self.handle, float(scale_bound)), "sga_set_scale_bound")
        self.scale_bound = float(scale_bound)

    def set_image_ids(self, ids=None):
        """Positions of this codec's images in their reference batch (None: 0,1,2,...): the device
        noise of image b is then the noise position ids[b] draws in the un-sharded batch."""
        ids = [] if ids is None else [int(i) for i in ids]
        arr = (C.c_int32 * max(len(ids), 1))(*ids)
        self._chk(self.lib.sga_set_image_ids(self.handle, arr, len(ids)), "sga_set_image_ids")

    def set_image_seeds(self, seeds=None):
        """Per-image noise keys for a launch that pools images of several reference batches (None: every image uses
        the `seed` of the run): seeds[b] = the seed of the b-th image's reference batch."""
        seeds = [] if seeds is None else [int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds]
        arr = (C.c_uint64 * max(len(seeds), 1))(*seeds)
        self._chk(self.lib.sga_set_image_seeds(self.handle, arr, len(seeds)), "sga_set_image_seeds")

    # ---- the session interactions ------------------------------------------------------------
    def encode(self, x):
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        y, z = self._empty(*ys), self._empty(*zs)
        s = self._enter()
        self._chk(self.lib.sga_encode(self.handle, _ptr(x), B, H, W, _ptr(y), _ptr(z), s), "sga_encode")
        self._exit()
        return y, z

    def step_grads(self, x, y, z, T, lmbda, loss_scale=None, seed=0, it=0, u_y=None, u_z=None):
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        y, z = self._t(y, ys), self._t(z, zs)
        if loss_scale is None:
            loss_scale = 1.0 / B
        uy = self._t(u_y).reshape(*ys, 2) if u_y is not None else None
        uz = self._t(u_z).reshape(*zs, 2) if u_z is not None else None
        gy, gz = self._empty(*ys), self._empty(*zs)
        scal, psnr = self._empty(3), self._empty(B)
        s = self._enter()
        self._chk(self.lib.sga_step_grads(self.handle, _ptr(x), B, H, W, _ptr(y), _ptr(z), float(T),
                                          float(lmbda), float(loss_scale), int(seed), int(it),
                                          _ptr(uy), _ptr(uz), _ptr(gy), _ptr(gz), _ptr(scal),
                                          _ptr(psnr), s), "sga_step_grads")
        self._exit()
        sc = scal.cpu().numpy()
        return dict(gy=gy, gz=gz, rd_loss=float(sc[0]), train_mse=float(sc[1]),
                    train_bpp=float(sc[2]), psnr=psnr)

    def adam(self, p, g, m, v, t, lr=0.005, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        """In-place adam.py:20-59 update of one array (t = iterations + 1)."""
        for a in (p, g, m, v):
            if not (a.is_cuda and a.dtype == torch.float32 and a.is_contiguous()):
                raise ValueError("adam operands must be contiguous float32 CUDA tensors")
        s = self._enter()
        self._chk(self.lib.sga_adam(self.handle, _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(),
                                    int(t), lr, beta_1, beta_2, epsilon, s), "sga_adam")
        self._exit()
        return p

    def run(self, x, lmbda, its=2000, lr=0.005, annealing_rate=1e-3, t0=700, T_ub=0.5, seed=0,
            loss_scale=None, y0=None, z0=None, trace=False, metrics=True):
        """sga.py:207-247 entirely on the device. Returns (y_hat, z_hat, metrics[B,7], trace)."""
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        if loss_scale is None:
            loss_scale = 1.0 / B
        y0t = self._t(y0, ys) if y0 is not None else None
        z0t = self._t(z0, zs) if z0 is not None else None
        y_hat, z_hat = self._empty(*ys), self._empty(*zs)
        met = self._empty(B, 7) if metrics else None
        tr = self._empty(max(its, 1), 4) if trace else None
        s = self._enter()
        self._chk(self.lib.sga_run(self.handle, _ptr(x), B, H, W, float(lmbda), float(loss_scale),
                                   int(its), float(lr), float(annealing_rate), int(t0), float(T_ub),
                                   int(seed), _ptr(y0t), _ptr(z0t), _ptr(y_hat), _ptr(z_hat),
                                   _ptr(met), _ptr(tr), s), "sga_run")
        self._exit()
        return y_hat, z_hat, met, (tr[:its] if trace else None)

    # ---- the same loop in pieces (map.py / ste.py decide on the host every 10 iterations) ------
    def run_begin(self, x, lmbda, its=2000, lr=0.005, annealing_rate=1e-3, t0=700, T_ub=0.5, seed=0,
                  loss_scale=None, y0=None, z0=None):
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        if loss_scale is None:
            loss_scale = 1.0 / B
        y0t = self._t(y0, ys) if y0 is not None else None
        z0t = self._t(z0, zs) if z0 is not None else None
        self._run = (x, B, H, W, ys, zs, int(its))
        s = self._enter()
        self._chk(self.lib.sga_run_begin(self.handle, _ptr(x), B, H, W, float(lmbda), float(loss_scale),
                                         int(its), float(lr), float(annealing_rate), int(t0), float(T_ub),
                                         int(seed), _ptr(y0t), _ptr(z0t), s), "sga_run_begin")
        self._exit()

    def run_steps(self, n):
        s = self._enter()
        self._chk(self.lib.sga_run_steps(self.handle, int(n), s), "sga_run_steps")
        self._exit()

    def run_latents(self, trace=False):
        """Current continuous (y, z) [and the trace of the iterations done so far]."""
        _, B, H, W, ys, zs, its = self._run
        y, z = self._empty(*ys), self._empty(*zs)
        tr = self._empty(max(its, 1), 4) if trace else None
        s = self._enter()
        self._chk(self.lib.sga_run_state(self.handle, 0, _ptr(y), _ptr(z), _ptr(tr), s), "sga_run_state")
        self._exit()
        return (y, z, tr) if trace else (y, z)

    def run_set_latents(self, y, z):
        _, B, H, W, ys, zs, its = self._run
        y, z = self._t(y, ys), self._t(z, zs)
        s = self._enter()
        self._chk(self.lib.sga_run_state(self.handle, 1, _ptr(y), _ptr(z), None, s), "sga_run_state")
        self._exit()

    def quantize_centered(self, y, z, H, W, medians=None):
        """tfc `_quantize(., 'dequantize')`: z_hat = round(z - median) + median, y_hat = round(y - mu)
        + mu with mu = h_s(z)[..., :C] (map.py:83,101)."""
        y, z = self._t(y), self._t(z)
        B = y.shape[0]
        med = self._t(medians, (self.C,)) if medians is not None else None
        y_hat, z_hat = torch.empty_like(y), torch.empty_like(z)
        s = self._enter()
        self._chk(self.lib.sga_quantize_centered(self.handle, _ptr(y), _ptr(z), B, int(H), int(W), _ptr(med),
                                                 _ptr(y_hat), _ptr(z_hat), s), "sga_quantize_centered")
        self._exit()
        return y_hat, z_hat

    def evaluate(self, x, y_hat, z_hat, want_x_hat=False):
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        y_hat, z_hat = self._t(y_hat, ys), self._t(z_hat, zs)
        met = self._empty(B, 7)
        xh = self._empty(B, H, W, 3) if want_x_hat else None
        s = self._enter()
        self._chk(self.lib.sga_eval(self.handle, _ptr(x), B, H, W, _ptr(y_hat), _ptr(z_hat),
                                    _ptr(met), _ptr(xh), s), "sga_eval")
        self._exit()
        return (met, xh) if want_x_hat else met

    def base_compress(self, x, medians=None, scale_bound: float = _lib.SCALE_BOUND_BUILT):
        """mbt2018.py compress, estimated-rate path (cfg 1).  mbt2018.py:80 CALLS the conditional layer, so tfc builds
        it and bounds sigma below by scale_table[0] = 0.11: `scale_bound` is an argument of this call only
        (sga_base_compress_bound): the handle's own bound and its cached step graphs are left alone."""
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        med = self._t(medians, (self.C,)) if medians is not None else None
        y_hat, z_hat, met = self._empty(*ys), self._empty(*zs), self._empty(B, 7)
        s = self._enter()
        self._chk(self.lib.sga_base_compress_bound(self.handle, _ptr(x), B, H, W, _ptr(med), float(scale_bound), _ptr(y_hat),
                                                   _ptr(z_hat), _ptr(met), s), "sga_base_compress_bound")
        self._exit()
        return y_hat, z_hat, met

    # ---- bb_sga.py (cfg 5): SGA + bits-back; needs bits_back=True ---------------------------------
    def bb_init_z(self, y_tilde, H, W):
        """(z_mean | z_logvar) = h_a(y_tilde) -> [B,zh,zw,2C]  (bb_sga.py:93-94,203-204,247)."""
        yh, yw, zh, zw = self.latent_shape(H, W)
        y_tilde = self._t(y_tilde)
        B = y_tilde.shape[0]
        zml = self._empty(B, zh, zw, 2 * self.C)
        s = self._enter()
        self._chk(self.lib.sga_bb_init_z(self.handle, _ptr(y_tilde), B, H, W, _ptr(zml), s), "sga_bb_init_z")
        self._exit()
        return zml

    def bb_step_grads(self, x, y, zml, T, lmbda, loss_scale=None, seed=0, it=0, u_y=None, eps=None,
                      rate_only=False):
        """bb_sga.py:211-214 (stage 1) / :252-254 (rate_only)."""
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        y, zml = self._t(y, ys), self._t(zml, zs)
        if loss_scale is None:
            loss_scale = 1.0 / B
        uy = self._t(u_y).reshape(*ys, 2) if u_y is not None else None
        ep = self._t(eps).reshape(*zs[:-1], self.C) if eps is not None else None
        gy, gz = self._empty(*ys), self._empty(*zs)
        scal, psnr = self._empty(3), self._empty(B)
        s = self._enter()
        self._chk(self.lib.sga_bb_step_grads(self.handle, _ptr(x), B, H, W, _ptr(y), _ptr(zml), float(T),
                                             float(lmbda), float(loss_scale), int(seed), int(it),
                                             _ptr(uy), _ptr(ep), int(bool(rate_only)), _ptr(gy),
                                             _ptr(gz), _ptr(scal), _ptr(psnr), s), "sga_bb_step_grads")
        self._exit()
        sc = scal.cpu().numpy()
        return dict(gy=gy, gzml=gz, rd_loss=float(sc[0]), train_mse=float(sc[1]),
                    train_bpp=float(sc[2]), psnr=psnr)

    def bb_run(self, x, lmbda, its=2000, r_its=2000, lr=0.005, r_lr=0.003, annealing_rate=1e-3,
               t0=700, T_ub=0.5, seed=0, loss_scale=None, trace=False):
        """bb_sga.py:199-276 on the device. Returns (y_hat, zml, metrics[B,8], trace1, trace2)."""
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        if loss_scale is None:
            loss_scale = 1.0 / B
        y_hat, zml, met = self._empty(*ys), self._empty(*zs), self._empty(B, 8)
        tr1 = self._empty(max(its, 1), 4) if trace else None
        tr2 = self._empty(max(r_its, 1), 4) if trace else None
        s = self._enter()
        self._chk(self.lib.sga_bb_run(self.handle, _ptr(x), B, H, W, float(lmbda), float(loss_scale),
                                      int(its), int(r_its), float(lr), float(r_lr),
                                      float(annealing_rate), int(t0), float(T_ub), int(seed),
                                      _ptr(y_hat), _ptr(zml), _ptr(met), _ptr(tr1), _ptr(tr2), s),
                  "sga_bb_run")
        self._exit()
        return y_hat, zml, met, (tr1[:its] if trace else None), (tr2[:r_its] if trace else None)

    def bb_refine(self, y_hat, H, W, r_its=2000, r_lr=0.003, seed=0, loss_scale=None):
        """bb_sga.py:238-261 alone: y_hat -> refined (z_mean | z_logvar), exactly the stage 2 of bb_run (same kernels, same
        noise): what a receiver that has decoded y_hat runs to recover the sender's posterior."""
        y_hat = self._t(y_hat)
        B = y_hat.shape[0]
        yh, yw, zh, zw = self.latent_shape(H, W)
        if loss_scale is None:
            loss_scale = 1.0 / B
        zml = self._empty(B, zh, zw, 2 * self.C)
        s = self._enter()
        self._chk(self.lib.sga_bb_refine(self.handle, _ptr(y_hat), B, int(H), int(W), float(loss_scale), int(r_its),
                                         float(r_lr), int(seed), _ptr(zml), s), "sga_bb_refine")
        self._exit()
        return zml

    def bb_evaluate(self, x, y_hat, zml, eps=None, seed=0):
        x = self._t(x)
        B, H, W, ys, zs = self._shapes(x)
        y_hat, zml = self._t(y_hat, ys), self._t(zml, zs)
        ep = self._t(eps).reshape(*zs[:-1], self.C) if eps is not None else None
        met = self._empty(B, 8)
        s = self._enter()
        self._chk(self.lib.sga_bb_eval(self.handle, _ptr(x), B, H, W, _ptr(y_hat), _ptr(zml), _ptr(ep),
                                       int(seed), _ptr(met), s), "sga_bb_eval")
        self._exit()
        return met

    def factorized_density(self, v):
        v = self._t(v)
        p, dp = torch.empty_like(v), torch.empty_like(v)
        s = self._enter()
        self._chk(self.lib.sga_op_factorized_density(self.handle, _ptr(v), v.numel() // self.C,
                                                     _ptr(p), _ptr(dp), s), "sga_op_factorized_density")
        self._exit()
        return p, dp

    # ---- actual bitstreams (mbt2018.py:84-85,211-222) ------------------------------
    def hyper_synthesis(self, z_hat, yh, yw):
        """(mu, sigma) = split(h_s(z_hat)), sigma = exp(.), cropped to the y grid (sga.py:107-108,
        126-128); same device kernels on the encoder and decoder side."""
        t = self.layer_fwd("HS2", self.layer_fwd("HS1", self.layer_fwd("HS0", z_hat)))
        t = t[:, :yh, :yw, :]
        return t[..., :self.C].contiguous(), torch.exp(t[..., self.C:]).contiguous()

    def _entropy_coder(self, weights=None, device_tables=True, centred=False, medians=None):
        """The coder's quantised CDF tables come from the SAME device kernels that evaluate the entropy
        models in the SGA step (factorized mass, box-convolved Gaussian), so that what is coded is the
        model whose rate was optimised; device_tables=False builds them with numpy instead.  centred: the tables of
        mbt2018.py compress (entropy_coding.EntropyCoder: zero-offset tables for y, z tables on the grid median + k); medians
        default to the weights' `eb.medians` (else 0).  One coder per (device_tables, centred, medians) is kept."""
        if centred:
            med = medians if medians is not None else self.medians
            med = np.zeros(self.C, np.float32) if med is None else np.ascontiguousarray(
                med.detach().cpu().numpy() if torch.is_tensor(med) else med, np.float32).reshape(self.C)
        else:
            med = None
        key = (bool(device_tables), bool(centred), None if med is None else med.tobytes())
        cache = self.__dict__.setdefault("_ec_cache", {})
        if key not in cache:
            from .entropy_coding import EntropyCoder
            dm = None
            if device_tables:
                dm = (lambda v: self.factorized_likelihood(v)[0].cpu().numpy(),
                      # a FIXED bound (the scale table's own minimum), whatever the handle's mutable sigma bound is at the
                      # moment: an encoder and a decoder must build identical tables
                      lambda y, mu, sr: self.gaussian_likelihood(y, mu, sr, scale_bound=_lib.SCALE_BOUND_BUILT)[0].cpu().numpy())
            cache[key] = EntropyCoder(weights if weights is not None else self._weights_for_ec, device_models=dm,
                                      centred=centred, medians=med)
        return cache[key]

    # ---- the coder itself on the device (csrc/rans.hip) ---------------------------------------------------------
    def _ec_tables(self, coder):
        """The coder's CDF tables, lengths, offsets and the scale table as device tensors (cached per coder)."""
        if getattr(self, "_ec_dev_tabs", None) is None or self._ec_dev_tabs[0] is not coder:
            dev = self.device
            self._ec_dev_tabs = (coder,
                                 torch.as_tensor(coder.cdf.astype(np.int64), device=dev).to(torch.int32).contiguous(),
                                 torch.as_tensor(coder.lens, device=dev).contiguous(),
                                 torch.as_tensor(coder.offs, device=dev).contiguous(),
                                 torch.as_tensor(coder.scale_table, dtype=torch.float64, device=dev).contiguous())
        return self._ec_dev_tabs[1:]

    def _ec_encode_device(self, coder, sym, tab, block=None):
        """sym / tab (int32 device tensors) -> one blocked rANS stream (bytes), encoded by one lane per block.  block=None: a
        first pass at ec.BLOCK symbols per block, then the pass that is written at `ec.adapted_block` (the host coder's rule)."""
        from . import entropy_coding as ec
        cdf, lens, offs, _ = self._ec_tables(coder)
        n = sym.numel()

        def one_pass(blk):
            nb = -(-n // blk)
            cap = 16 + 8 * blk
            slots = torch.empty(nb * cap, dtype=torch.uint8, device=self.device)
            bb = torch.empty(nb, dtype=torch.int32, device=self.device)
            s = self._enter()
            _lib.check(self.lib, None, self.lib.sga_ec_encode(_ptr(sym), _ptr(tab), n, blk, _ptr(cdf), _ptr(lens),
                                                              _ptr(offs), coder.stride, _ptr(slots), cap, _ptr(bb), s),
                       "sga_ec_encode")
            self._exit()
            bbh = bb.cpu().numpy().astype(np.uint32)
            if (bbh == 0).any():
                raise RuntimeError("sga_ec_encode: output slot overflow")
            return nb, cap, slots, bb, bbh

        blk = int(block) if block else ec.BLOCK
        nb, cap, slots, bb, bbh = one_pass(blk)
        if not block:
            blk2 = ec.adapted_block(bbh, blk)
            if blk2 != blk:
                blk = blk2
                nb, cap, slots, bb, bbh = one_pass(blk)
        off = np.concatenate([[0], np.cumsum(bbh[:-1], dtype=np.uint64)]).astype(np.uint64)
        out = torch.empty(int(bbh.sum()), dtype=torch.uint8, device=self.device)
        offd = torch.as_tensor(off.astype(np.int64), device=self.device)
        s = self._enter()
        _lib.check(self.lib, None, self.lib.sga_ec_compact(_ptr(slots), cap, _ptr(bb), _ptr(offd), nb, _ptr(out), s),
                   "sga_ec_compact")
        self._exit()
        return ec.frame_blocks(bbh, out.cpu().numpy().tobytes(), blk)

    def _ec_decode_device(self, coder, data: bytes, tab):
        from . import entropy_coding as ec
        cdf, lens, offs, _ = self._ec_tables(coder)
        bbh, block, payload = ec.unframe_blocks(data)
        n = tab.numel()
        if bbh.size != -(-n // block):
            raise ValueError("rans_decode: corrupt stream (block count)")
        buf = torch.as_tensor(np.frombuffer(payload, np.uint8).copy() if payload else np.zeros(1, np.uint8), device=self.device)
        off = torch.as_tensor(np.concatenate([[0], np.cumsum(bbh[:-1], dtype=np.int64)]).astype(np.int64), device=self.device)
        bb = torch.as_tensor(bbh.astype(np.int64), device=self.device).to(torch.int32)
        sym = torch.empty(n, dtype=torch.int32, device=self.device)
        bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        s = self._enter()
        _lib.check(self.lib, None, self.lib.sga_ec_decode(_ptr(buf), _ptr(off), _ptr(bb), int(bbh.size), _ptr(tab), n, block,
                                                          _ptr(cdf), _ptr(lens), _ptr(offs), coder.stride, _ptr(sym),
                                                          _ptr(bad), s), "sga_ec_decode")
        self._exit()
        if int(bad.item()):
            raise ValueError("rans_decode: corrupt stream")
        return sym

    def _ec_symbols_device(self, coder, y_hat, mu, sigma, z_hat):
        """(sym, tab[, r0]) of y (y_hat may be None: decoder) on the device; a centred coder has no r0."""
        from . import entropy_coding as ec
        _, _, _, scales = self._ec_tables(coder)
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=self.device)
        bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        out = {}
        s = self._enter()
        if mu is not None:
            n = mu.numel()
            out["y_sym"], out["y_tab"] = (i32(n) if y_hat is not None else None), i32(n)
            if coder.centred:
                _lib.check(self.lib, None,
                           self.lib.sga_ec_y_symbols_centred(_ptr(y_hat), _ptr(mu), _ptr(sigma), n, _ptr(scales), ec.SCALES_LEVELS,
                                                             coder.y_tab0, _ptr(out["y_sym"]), _ptr(out["y_tab"]), _ptr(bad), s),
                           "sga_ec_y_symbols_centred")
            else:
                out["r0"] = i32(n)
                _lib.check(self.lib, None,
                           self.lib.sga_ec_y_symbols(_ptr(y_hat), _ptr(mu), _ptr(sigma), n, _ptr(scales), ec.SCALES_LEVELS,
                                                     ec.MEAN_BINS, coder.y_tab0, _ptr(out["y_sym"]), _ptr(out["y_tab"]),
                                                     _ptr(out["r0"]), _ptr(bad), s), "sga_ec_y_symbols")
        self._exit()
        out["bad"] = bad
        return out

    def _ec_z_symbols_device(self, coder, z_hat, nz, bad=None):
        """(sym, tab) of z on the device (z_hat None: the decoder's table indices only)."""
        z_sym = torch.empty(nz, dtype=torch.int32, device=self.device) if z_hat is not None else None
        z_tab = torch.empty(nz, dtype=torch.int32, device=self.device)
        s = self._enter()
        if coder.centred:
            med = self._t(coder.medians, (self.C,))
            _lib.check(self.lib, None, self.lib.sga_ec_z_symbols_centred(_ptr(z_hat), _ptr(med), nz, self.C, _ptr(z_sym), _ptr(z_tab),
                                                                         _ptr(bad), s), "sga_ec_z_symbols_centred")
        else:
            _lib.check(self.lib, None, self.lib.sga_ec_z_symbols(_ptr(z_hat), nz, self.C, _ptr(z_sym), _ptr(z_tab), _ptr(bad), s),
                       "sga_ec_z_symbols")
        self._exit()
        return z_sym, z_tab

    def compress_latents(self, x_shape, y_hat, z_hat, device_tables=True, on_device=True, centred=False, medians=None) -> bytes:
        """Entropy-code (y_hat, z_hat) of a batch into one byte string (cf. tfc.PackedTensors).  The stream records
        how its tables were built and their CRC32; device_tables=False gives the float64 host tables, which do
        not depend on the GPU's math library (the interchange mode).  on_device: (mu, sigma) -> table indices -> rANS
        bytes by the HIP kernels of csrc/rans.hip (one lane per block); False: the same bytes from the host
        coder (csrc_cpu/rans.c).  centred=False: integer latents, what an SGA run ends with (sga.py:240-241); centred=True:
        the mean- / median-centred latents of `base_compress` = mbt2018.py compress (mbt2018.py:69,80,211-222), `medians`
        as passed to it (default: the weights' `eb.medians`, else 0)."""
        from . import entropy_coding as ec
        coder = self._entropy_coder(device_tables=device_tables, centred=centred, medians=medians)
        y_hat, z_hat = self._t(y_hat), self._t(z_hat)
        mu, sigma = self.hyper_synthesis(z_hat, y_hat.shape[1], y_hat.shape[2])
        if on_device:
            ys = self._ec_symbols_device(coder, y_hat, mu, sigma, None)
            z_sym, z_tab = self._ec_z_symbols_device(coder, z_hat, z_hat.numel(), ys["bad"])
            if int(ys["bad"].item()):
                raise ValueError("y_hat / z_hat are not their centres + integers (coder centred: mu / medians)" if centred else
                                 "y_hat / z_hat must hold integers; centred latents need compress_latents(..., centred=True)")
            zb = self._ec_encode_device(coder, z_sym, z_tab)
            yb = self._ec_encode_device(coder, ys["y_sym"], ys["y_tab"])
            return ec.pack(tuple(x_shape), tuple(y_hat.shape), tuple(z_hat.shape), zb, yb, self._stream_mode(coder),
                           coder.table_crc())
        zb = coder.encode_z(z_hat.cpu().numpy())
        yb = coder.encode_y(y_hat.cpu().numpy(), mu.cpu().numpy(), sigma.cpu().numpy())
        return ec.pack(tuple(x_shape), tuple(y_hat.shape), tuple(z_hat.shape), zb, yb, self._stream_mode(coder),
                       coder.table_crc())

    def effective_precision(self) -> str:
        """The arithmetic this handle's convolutions run in ("default" resolves through SGA_PRECISION like the library does)."""
        p = self.precision
        if p == "default":
            p = os.environ.get("SGA_PRECISION", "f32")
        return p if p in ("bf16x3", "bf16x2") else "f32"

    def _stream_mode(self, coder) -> int:
        from . import entropy_coding as ec
        return coder.table_mode | (ec.MODE_PRECISIONS.index(self.effective_precision()) << ec.MODE_PRECISION_SHIFT)

    def decompress_latents(self, blob: bytes, on_device=True, medians=None):
        """-> (x_shape, y_hat, z_hat): z first, then (mu, sigma) = h_s(z_hat), then y.  The stream's mode byte says how its
        tables were built and whether it holds centred latents (then `medians` as at the encoder)."""
        from . import entropy_coding as ec
        x_shape, y_shape, z_shape, zb, yb, mode, crc = ec.unpack(blob, with_tables=True)
        if ec.mode_precision(mode) != self.effective_precision():
            raise ValueError("SGAC stream was coded by a handle in precision mode %r, this one runs in %r: its h_s would predict other "
                             "(mu, sigma) and the range decoder would return garbage -- create the SGACodec with precision=%r"
                             % (ec.mode_precision(mode), self.effective_precision(), ec.mode_precision(mode)))
        coder = self._entropy_coder(device_tables=bool(mode & 1), centred=bool(mode & 2), medians=medians)
        if coder.table_crc() != crc:
            raise ValueError("SGAC stream was coded with different CDF tables (mode %d, crc %08x; this decoder builds "
                             "%08x): other weights or medians, or device-built tables from another GPU / ROCm build -- encode "
                             "with device_tables=False for streams that must travel" % (mode, crc, coder.table_crc()))
        if on_device:
            nz = int(np.prod(z_shape))
            _, z_tab = self._ec_z_symbols_device(coder, None, nz)
            z_hat = self._ec_decode_device(coder, zb, z_tab).to(torch.float32).reshape(*z_shape)
            if coder.centred:
                z_hat = z_hat + self._t(coder.medians, (self.C,))      # float32 add: the operation that made z_hat (mbt2018.py:69)
            mu, sigma = self.hyper_synthesis(z_hat, y_shape[1], y_shape[2])
            ys = self._ec_symbols_device(coder, None, mu, sigma, None)
            sym = self._ec_decode_device(coder, yb, ys["y_tab"])
            if coder.centred:
                y_hat = sym.to(torch.float32).reshape(*y_shape) + mu      # round(y - mu) + mu (mbt2018.py:80)
            else:
                y_hat = (sym + ys["r0"]).to(torch.float32).reshape(*y_shape)
            return x_shape, y_hat, z_hat
        z_hat = self._t(coder.decode_z(zb, z_shape))
        mu, sigma = self.hyper_synthesis(z_hat, y_shape[1], y_shape[2])
        y_hat = self._t(coder.decode_y(yb, mu.cpu().numpy(), sigma.cpu().numpy()))
        return x_shape, y_hat, z_hat

    def reconstruct(self, y_hat, H, W):
        """x_hat = clip(g_s(y_hat)) cropped to HxW (mbt2018.py:283-288)."""
        t = self._t(y_hat)
        for name in ("GS0", "GS1", "GS2", "GS3"):
            t = self.layer_fwd(name, t)
        return torch.clamp(t[:, :H, :W, :], 0, 1)

    # ---- measurement ---------------------------------------------------------------------------
    def profile_begin(self):
        """Time every MFMA convolution launch with hipEvents until profile_end (sga_run goes eager)."""
        self._chk(self.lib.sga_profile_begin(self.handle), "sga_profile_begin")

    def profile_end(self):
        """-> list of dicts {name, launches, ms_total, flops_total} per kernel symbol."""
        arr = (_lib.SgaKernelStat * 64)()
        n = C.c_int(0)
        self._chk(self.lib.sga_profile_end(self.handle, arr, 64, C.byref(n)), "sga_profile_end")
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches),
                     ms_total=float(arr[i].ms_total), flops_total=float(arr[i].flops_total))
                for i in range(min(n.value, 64))]

    def profile_graph_begin(self, kernel_name: str):
        """Time one kernel symbol inside the hipGraph replay of the following run()s (sga_profile_graph_begin)."""
        self._chk(self.lib.sga_profile_graph_begin(self.handle, kernel_name.encode()), "sga_profile_graph_begin")

    def profile_graph_end(self):
        st = _lib.SgaKernelStat()
        self._chk(self.lib.sga_profile_graph_end(self.handle, C.byref(st)), "sga_profile_graph_end")
        return dict(name=st.name.decode(), launches=int(st.launches), ms_total=float(st.ms_total),
                    flops_total=float(st.flops_total))

    def fork_point(self):
        """Where the hyper branch of the cached step graph is forked (timed per geometry)."""
        buf = C.create_string_buffer(32)
        self._chk(self.lib.sga_get_fork_point(self.handle, buf, 32), "sga_get_fork_point")
        return buf.value.decode()

    def counter(self, which: str) -> int:
        """Graph-cache counters (sga_debug_counter): "captures", "cached", "evictions", "dropped" (destroyed in mid-life)."""
        v = C.c_longlong(0)
        idx = {"captures": 0, "cached": 1, "evictions": 2, "dropped": 3}[which]
        self._chk(self.lib.sga_debug_counter(self.handle, idx, C.byref(v)), "sga_debug_counter")
        return int(v.value)

    # ---- operator surface (unit parity) --------------------------------------------------------
    def layer_fwd(self, layer: str, inp):
        inp = self._t(inp)
        B, Hi, Wi, Ci = inp.shape
        C_, C15 = self.C, int(self.C * 1.5)
        up = {"GS0": C_, "GS1": C_, "GS2": C_, "GS3": 3, "HS0": C_, "HS1": C15}
        down = {"GA0": C_, "GA1": C_, "GA2": C_, "GA3": C_, "HA1": C_,
                "HA2": 2 * C_ if self.bits_back else C_}
        if layer in up:
            out = self._empty(B, 2 * Hi, 2 * Wi, up[layer])
        elif layer in down:
            out = self._empty(B, (Hi + 1) // 2, (Wi + 1) // 2, down[layer])
        elif layer == "HA0":
            out = self._empty(B, Hi, Wi, C_)
        elif layer == "HS2":
            out = self._empty(B, Hi, Wi, 2 * C_)
        else:
            raise ValueError(layer)
        s = self._enter()
        self._chk(self.lib.sga_op_layer_fwd(self.handle, _lib.LAYERS[layer], _ptr(inp), B, Hi, Wi,
                                            _ptr(out), s), f"sga_op_layer_fwd({layer})")
        self._exit()
        return out

    def layer_bwd(self, layer: str, inp, g_out):
        inp, g_out = self._t(inp), self._t(g_out)
        B, Hi, Wi, Ci = inp.shape
        g_in = torch.empty_like(inp)
        s = self._enter()
        self._chk(self.lib.sga_op_layer_bwd(self.handle, _lib.LAYERS[layer], _ptr(inp), _ptr(g_out),
                                            B, Hi, Wi, _ptr(g_in), s), f"sga_op_layer_bwd({layer})")
        self._exit()
        return g_in

    def sample(self, v, u, T):
        v = self._t(v)
        u = self._t(u).reshape(*v.shape, 2)
        vt, jac = torch.empty_like(v), torch.empty_like(v)
        s = self._enter()
        self._chk(self.lib.sga_op_sample(self.handle, _ptr(v), _ptr(u), v.numel(), float(T),
                                         _ptr(vt), _ptr(jac), s), "sga_op_sample")
        self._exit()
        return vt, jac

    def factorized_likelihood(self, v):
        v = self._t(v)
        if v.shape[-1] != self.C:
            raise ValueError("last dim must be num_filters")
        p, dp = torch.empty_like(v), torch.empty_like(v)
        s = self._enter()
        self._chk(self.lib.sga_op_factorized_likelihood(self.handle, _ptr(v), v.numel() // self.C,
                                                        _ptr(p), _ptr(dp), s),
                  "sga_op_factorized_likelihood")
        self._exit()
        return p, dp

    def gaussian_likelihood(self, y, mu, sigma_raw, scale_bound=None):
        """scale_bound=None: the handle's current bound; a number: that bound for this call only."""
        y, mu, sr = self._t(y), self._t(mu), self._t(sigma_raw)
        outs = [torch.empty_like(y) for _ in range(4)]
        s = self._enter()
        sb = self.scale_bound if scale_bound is None else float(scale_bound)
        self._chk(self.lib.sga_op_gaussian_likelihood_bound(self.handle, _ptr(y), _ptr(mu), _ptr(sr),
                                                            y.numel(), sb, *[_ptr(o) for o in outs], s),
                  "sga_op_gaussian_likelihood_bound")
        self._exit()
        return tuple(outs)


    def rate_terms(self, y_tilde, z_tilde, ms, H, W, loss_scale=None):
        """The step's own entropy-model kernels on fed intermediates (sga.py:100-104,126-146):
        -> dict(g_yt, g_ms, g_zt, est_y_bpp[B], est_z_bpp[B])."""
        yt, zt, ms = self._t(y_tilde), self._t(z_tilde), self._t(ms)
        B = yt.shape[0]
        if loss_scale is None:
            loss_scale = 1.0 / B
        g_yt, g_ms, g_zt, met = torch.empty_like(yt), torch.empty_like(ms), torch.empty_like(zt), self._empty(B, 7)
        s = self._enter()
        self._chk(self.lib.sga_op_rate_terms(self.handle, _ptr(yt), _ptr(zt), _ptr(ms), B, int(H), int(W),
                                             float(loss_scale), _ptr(g_yt), _ptr(g_ms), _ptr(g_zt), _ptr(met), s),
                  "sga_op_rate_terms")
        self._exit()
        return dict(g_yt=g_yt, g_ms=g_ms, g_zt=g_zt, est_y_bpp=met[:, 5], est_z_bpp=met[:, 6])


def metrics_to_dict(met) -> dict:
    """[B,7] metrics tensor -> dict keyed like sga.py:183 eval_fields (numpy arrays)."""
    m = met.detach().cpu().numpy()
    fields = BB_EVAL_FIELDS if m.shape[1] == 8 else EVAL_FIELDS
    return {k: m[:, i].copy() for i, k in enumerate(fields)}
