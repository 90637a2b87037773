"""Generate tests/golden/ fixtures by running the parts of the reference that run in the build
container.  Run in the build container only: /root/reference does not exist on
the GPU box.  Fixtures are data (inputs + the reference's outputs); no reference source is copied.

* adam.py, configs.py import as they are (pure numpy / pure Python).
* utils.py starts with `import tensorflow`, which is not installed, but four of its functions do
  not need TensorFlow: `annealed_temperature` and `log_normal_pdf` take `backend=np`,
  `get_runname` is plain Python, and `box_convolved_gaussian_pdf` only calls `tf.math.erfc`.
  Their function definitions are taken from the parsed module (ast) at run time and executed
  unmodified; for the last one the name `tf.math.erfc` is bound to `scipy.special.erfc`, so that
  fixture pins the reference's formula (the |x-mu| tail trick, the +-0.5 box, the division by
  sigma), not TensorFlow's erfc rounding.

* learned_prior.py / math_ops.py also start with `import tensorflow`; the class `BMSHJ2018Prior`
  (the CDF network `_logits_cdf`, `cdf`, the closed-form `cdf_pdf`) and the gradient functions of
  `lower_bound` / `upper_bound` are executed unmodified against a small numpy-backed stand-in for the
  TensorFlow names they use (matmul, softplus, tanh, sigmoid, transpose, reshape, logical_or, cast).

* nn_models.py is imported as it stands against recording stand-ins for `tf.keras.layers.Layer`,
  `tfc.SignalConv2D` and `tfc.GDN`: the layer table of the four transforms -> architecture_reference.json.

    python scripts/make_golden_from_reference.py
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def reference_utils_functions(names):
    """exec the named top-level definitions of /root/reference/utils.py (nothing else of it)."""
    import ast
    import types
    import scipy.special
    with open(os.path.join(REF, "utils.py")) as f:
        tree = ast.parse(f.read())
    keep = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in names)
            or (isinstance(n, ast.Assign) and any(getattr(t, "id", None) in names for t in n.targets))]
    found = {getattr(n, "name", None) or n.targets[0].id for n in keep}
    assert found == set(names), (found, names)
    tf_stub = types.SimpleNamespace(math=types.SimpleNamespace(erfc=scipy.special.erfc))
    ns = {"np": np, "tf": tf_stub}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "reference:utils.py", "exec"), ns)
    return ns


def utils_fixtures():
    ns = reference_utils_functions(["annealed_temperature", "log2pi", "log_normal_pdf", "get_runname",
                                    "gaussian_standardized_cumulative", "box_convolved_gaussian_pdf"])
    out = {}
    # ---- temperature schedule (utils.py:151-180) at the call sites' arguments
    its = np.arange(2000)
    # sga.py:206,211 / bb_sga.py:210: scheme exp0, r=1e-3, ub=0.5, t0=700 (+ two other settings)
    out["T_exp0_r1e-3_ub0.5_t0700"] = np.array(
        [ns["annealed_temperature"](int(t), r=1e-3, ub=0.5, scheme="exp0", t0=700) for t in its], np.float64)
    out["T_exp0_r2e-3_ub0.5_t0100"] = np.array(
        [ns["annealed_temperature"](int(t), r=2e-3, ub=0.5, scheme="exp0", t0=100) for t in its], np.float64)
    out["T_exp_r1e-3_ub1.0"] = np.array(          # the 'exp' scheme (danneal.py:188-193 restates it locally)
        [ns["annealed_temperature"](int(t), r=1e-3, ub=1.0, scheme="exp") for t in its], np.float64)
    # ---- log N(sample; mean, exp(logvar)) (utils.py:75-77), float32 like the graph
    rng = np.random.RandomState(99)
    sample = rng.standard_normal(512).astype(np.float32) * 3
    mean = rng.standard_normal(512).astype(np.float32)
    logvar = (rng.standard_normal(512) * 1.5).astype(np.float32)
    out.update(lnpdf_sample=sample, lnpdf_mean=mean, lnpdf_logvar=logvar,
               lnpdf_out=ns["log_normal_pdf"](sample, mean, logvar, backend=np))
    # ---- box-convolved Gaussian (utils.py:86-102), float64 inputs incl. tails and tiny sigma
    y = np.concatenate([rng.standard_normal(400) * 4, [-40.0, 40.0, 0.0, 0.5, -0.5, 12.25]])
    mu = np.concatenate([rng.standard_normal(400), [0.0, 0.0, 0.0, 0.0, 0.0, 12.0]])
    sigma = np.concatenate([np.exp(rng.standard_normal(400) * 1.5), [0.11, 0.11, 0.11, 1e-3, 64.0, 0.11]])
    out.update(box_y=y.copy(), box_mu=mu, box_sigma=sigma,
               box_out=ns["box_convolved_gaussian_pdf"](y.copy(), mu, sigma))   # (the function mutates its input)
    np.savez_compressed(os.path.join(OUT, "utils_reference.npz"), **out)
    # ---- run names (utils.py:51-69); sga.py:158 parses lambda back out of them
    cases = [dict(num_filters=192, num_hfilters=0, lmbda=0.01, last_step=2000000),
             dict(num_filters=128, num_hfilters=0, lmbda=0.0016, last_step=1000),
             dict(num_filters=192, num_hfilters=192, lmbda=0.08, last_step=5)]
    with open(os.path.join(OUT, "runnames.json"), "w") as f:
        json.dump([dict(args=c, runname=ns["get_runname"](c, prefix="mbt2018")) for c in cases], f, indent=1)


# ---------------------------------------------------------------------------------------------
# learned_prior.py / math_ops.py against a numpy-backed `tf` namespace
# ---------------------------------------------------------------------------------------------
class _Shape:
    """what `Tensor.get_shape()` must offer to learned_prior.py:137-142"""
    def __init__(self, shp):
        self._s, self.ndims = tuple(int(d) for d in shp), len(shp)

    def __getitem__(self, i):
        return self._s[i]


class _T(np.ndarray):
    def get_shape(self):
        return _Shape(self.shape)


def _t(a):
    return np.asarray(a, dtype=np.float64).view(_T)


def _numpy_tf(injected):
    """The handful of TensorFlow names the prior's CDF network and the bound gradients use, backed by
    numpy float64.  Variables come from `injected` (name -> array) instead of the initialisers."""
    import types

    class Model:                                  # stands in for tf.keras.Model
        dtype = "float64"

        def __init__(self, **kwargs):
            pass

        def add_weight(self, name, dtype=None, shape=None, initializer=None):
            v = _t(injected[name])
            assert tuple(v.shape) == tuple(shape), (name, v.shape, shape)
            return v

    def softplus(x):
        return _t(np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0))

    ns = types.SimpleNamespace
    init = ns(constant=lambda v: None, random_uniform=lambda a, b: None, zeros=lambda: None)
    return ns(
        keras=ns(Model=Model), initializers=init,
        nn=ns(softplus=softplus, sigmoid=lambda x: _t(1.0 / (1.0 + np.exp(-x)))),
        math=ns(tanh=lambda x: _t(np.tanh(x))),
        linalg=ns(matmul=lambda a, b: _t(np.matmul(a, b))),
        matmul=lambda a, b: _t(np.matmul(a, b)),
        executing_eagerly=lambda: False, stop_gradient=lambda x: x,
        transpose=lambda a, perm: _t(np.transpose(a, perm)), shape=lambda a: tuple(a.shape),
        reshape=lambda a, shp: _t(np.reshape(a, shp)), expand_dims=lambda a, ax: _t(np.expand_dims(a, ax)),
        RegisterGradient=lambda name: (lambda f: f),
        logical_or=np.logical_or, cast=lambda x, dt: np.asarray(x).astype(dt),
    )


def _exec_reference_defs(fname, names, ns):
    """exec the named top-level class / function definitions of /root/reference/<fname>, unmodified."""
    import ast
    with open(os.path.join(REF, fname)) as f:
        tree = ast.parse(f.read())
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in keep} == set(names), ({n.name for n in keep}, names)
    exec(compile(ast.Module(body=keep, type_ignores=[]), f"reference:{fname}", "exec"), ns)
    return ns


def prior_fixtures():
    """learned_prior.py:78-121 (`_logits_cdf`), :123-162 (`cdf`), :263-360 (closed-form `cdf_pdf`) and
    math_ops.py:45-76 (`_upper_bound_grad`, `_lower_bound_grad`), executed as they stand.  Pins the
    FORMULAS (layer order, softplus'd matrices, the `x + tanh(factor) * tanh(x)` nonlinearity, the
    Jacobian chain, the pass-through rule of the bounds); float64 numpy arithmetic, so not TF's
    float32 rounding."""
    C, dims = 64, (3, 3, 3)
    rng = np.random.RandomState(2024)
    d = (1,) + dims + (1,)
    scale = 10.0 ** (1.0 / 4.0)
    raw = {}
    for i in range(4):
        raw[f"matrix_{i}"] = np.log(np.expm1(1 / scale / d[i + 1])) + 0.3 * rng.standard_normal((C, d[i + 1], d[i]))
        raw[f"bias_{i}"] = rng.uniform(-0.5, 0.5, (C, d[i + 1], 1))
        if i < 3:
            raw[f"factor_{i}"] = 0.5 * rng.standard_normal((C, d[i + 1], 1))
    tf = _numpy_tf(raw)
    ns = _exec_reference_defs("learned_prior.py", ["BMSHJ2018Prior"], {"tf": tf, "np": np, "math_ops": None})
    prior = ns["BMSHJ2018Prior"](C, dims=dims, init_scale=10.0)
    out = {"channels": np.int64(C)}
    # The HIP path takes float32 EFFECTIVE parameters (softplus'd / tanh'd, include/sga_hip.h): let the
    # reference evaluate exactly those values (graph mode reads self._matrices / self._factors)
    for i in range(4):
        m32 = np.asarray(prior._matrices[i], np.float32)
        prior._matrices[i] = _t(m32)
        out[f"eb.m{i}"] = m32
        b32 = np.asarray(prior._biases[i], np.float32)
        prior._biases[i] = _t(b32)
        out[f"eb.b{i}"] = b32
        if i < 3:
            f32 = np.asarray(prior._factors[i], np.float32)
            prior._factors[i] = _t(f32)
            out[f"eb.f{i}"] = f32
        out[f"raw_matrix_{i}"] = raw[f"matrix_{i}"]
    # inputs [N, C] float32-representable: bulk, integers and half-integers, far tails
    v = np.concatenate([rng.standard_normal((80, C)) * 4.0, np.round(rng.standard_normal((8, C)) * 3.0),
                        np.round(rng.standard_normal((4, C)) * 3.0) + 0.5,
                        np.full((1, C), -30.0), np.full((1, C), 30.0), np.full((1, C), 12.0), np.zeros((1, C)),
                        np.full((1, C), -400.0), np.full((1, C), 400.0), np.full((1, C), -150.0), np.full((1, C), 150.0)])
    v = v.astype(np.float32).astype(np.float64)
    out["v"] = v.astype(np.float32)

    def logits(x):                                   # learned_prior.py:96-121 through the reshapes of :144-150
        order = [1, 0]
        t = tf.reshape(tf.transpose(_t(x), order), (C, 1, -1))
        return np.asarray(tf.transpose(tf.reshape(prior._logits_cdf(t, stop_gradient=False), (C, -1)), order))

    out["logits_cdf"] = logits(v)
    out["cdf"] = np.asarray(prior.cdf(_t(v), stop_gradient=False))
    cdf2, pdf = prior.cdf_pdf(_t(v), stop_gradient=False)
    out["cdf_closed_form"], out["pdf"] = np.asarray(cdf2), np.asarray(pdf)
    # box mass of tfc EntropyBottleneck._likelihood (sga.py:101) FROM THE REFERENCE'S LOGITS: the sign
    # trick itself is tfc's (un-vendored), the logits and the plain CDF difference are the reference's
    lo, up = logits(v - 0.5), logits(v + 0.5)
    sg = -np.sign(lo + up)
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))
    out["mass_sign_trick"] = np.abs(sig(sg * up) - sig(sg * lo))
    out["mass_cdf_difference"] = np.asarray(prior.cdf(_t(v + 0.5), False)) - np.asarray(prior.cdf(_t(v - 0.5), False))
    out["dmass_dv"] = np.asarray(prior.cdf_pdf(_t(v + 0.5))[1]) - np.asarray(prior.cdf_pdf(_t(v - 0.5))[1])
    # derivative of the density (bb_sga.py differentiates through learned_prior.pdf): Richardson-
    # extrapolated central differences of the reference's closed-form pdf (O(h^4), float64)
    pdf_at = lambda x: np.asarray(prior.cdf_pdf(_t(x))[1])
    h = 1e-2
    d1 = (pdf_at(v + h) - pdf_at(v - h)) / (2 * h)
    d2 = (pdf_at(v + h / 2) - pdf_at(v - h / 2)) / h
    out["dpdf_dv_fd"] = (4 * d2 - d1) / 3
    # ---- math_ops.py:45-76: the bounds' pass-through rule, every (side of the bound) x (gradient sign)
    import types
    ns2 = _exec_reference_defs("math_ops.py", ["_lower_bound_grad", "_upper_bound_grad"], {"tf": tf})
    x = np.array([0.05, 0.11, 0.5, 0.05, 0.11, 0.5, 0.05, 0.11, 0.5], np.float32)
    g = np.array([-2.0, -2.0, -2.0, 0.0, 0.0, 0.0, 3.0, 3.0, 3.0], np.float32)
    bound = np.float32(0.11)
    out["bound_x"], out["bound_g"], out["bound"] = x, g, bound
    out["lower_bound_grad"] = ns2["_lower_bound_grad"](types.SimpleNamespace(inputs=(x, bound)), g)[0]
    out["upper_bound_grad"] = ns2["_upper_bound_grad"](types.SimpleNamespace(inputs=(x, bound)), g)[0]
    np.savez_compressed(os.path.join(OUT, "prior_reference.npz"), **out)
    # self-checks of the fixture: the two routes to the CDF agree, the closed-form pdf is its derivative
    assert np.allclose(out["cdf"], out["cdf_closed_form"], rtol=0, atol=1e-15)
    cdf_at = lambda x: np.asarray(prior.cdf(_t(x), False))
    fd = (cdf_at(v + 1e-5) - cdf_at(v - 1e-5)) / 2e-5
    assert np.allclose(fd, out["pdf"], rtol=1e-6, atol=1e-10), np.abs(fd - out["pdf"]).max()


def architecture_fixture():
    """nn_models.py imported as it stands, with `tensorflow.compat.v1` and `tensorflow_compression` replaced by
    recording stand-ins: `tf.keras.layers.Layer` is an empty base class, `tfc.SignalConv2D` / `tfc.GDN` store
    their constructor arguments.  The four transforms are constructed exactly as sga.py:70-73 (and bb_sga.py:69
    for the bits-back hyper-analysis) construct them, `build()` runs, and every layer's arguments are written
    out: the layer table (order, filters, kernel support, corr, strides, padding, bias, activation) is then
    the reference's own, not a reading of it.  Pins the ARCHITECTURE; what `SignalConv2D` / `GDN` compute for
    these arguments remains tfc 1.3's (un-vendored)."""
    import importlib.util
    import types

    class Layer:
        def __init__(self, *a, **k):
            pass

        def build(self, input_shape):
            pass

    def relu(x):
        raise RuntimeError("not executed")

    class Rec:
        def __init__(self, *args, **kwargs):
            self.args, self.kwargs = args, kwargs

    class SignalConv2D(Rec):
        pass

    class GDN(Rec):
        pass

    tf1 = types.ModuleType("tensorflow.compat.v1")
    tf1.keras = types.SimpleNamespace(layers=types.SimpleNamespace(Layer=Layer))
    tf1.nn = types.SimpleNamespace(relu=relu)
    tfmod, compat = types.ModuleType("tensorflow"), types.ModuleType("tensorflow.compat")
    tfmod.compat, compat.v1 = compat, tf1
    tfc = types.ModuleType("tensorflow_compression")
    tfc.SignalConv2D, tfc.GDN = SignalConv2D, GDN
    stubs = {"tensorflow": tfmod, "tensorflow.compat": compat, "tensorflow.compat.v1": tf1,
             "tensorflow_compression": tfc}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("_reference_nn_models", os.path.join(REF, "nn_models.py"))
        nn = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(nn)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    def describe(t):
        t.build(None)
        rows = []
        for lay in t._layers:
            assert isinstance(lay, SignalConv2D)
            kw = dict(lay.kwargs)
            act = kw.pop("activation")
            if isinstance(act, GDN):
                assert not act.args and set(act.kwargs) <= {"name", "inverse"}, act.kwargs
                a = {"kind": "gdn", "inverse": bool(act.kwargs.get("inverse", False)), "name": act.kwargs["name"]}
            elif act is relu:
                a = {"kind": "relu"}
            else:
                assert act is None
                a = {"kind": "none"}
            rows.append({"filters": int(lay.args[0]), "kernel_support": [int(v) for v in lay.args[1]],
                         "activation": a, **kw})
        return rows

    out = {}
    for C in (192, 256, 64):
        out[str(C)] = {
            "analysis": describe(nn.AnalysisTransform(C)),                                           # sga.py:70
            "synthesis": describe(nn.SynthesisTransform(C)),                                         # sga.py:71
            "hyper_analysis": describe(nn.HyperAnalysisTransform(C)),                                # sga.py:72
            "hyper_synthesis": describe(nn.MBT2018HyperSynthesisTransform(C, num_output_filters=2 * C)),  # sga.py:73
            "hyper_analysis_bb": describe(nn.HyperAnalysisTransform(C, num_output_filters=2 * C)),   # bb_sga.py:69
        }
    with open(os.path.join(OUT, "architecture_reference.json"), "w") as f:
        json.dump({"source": "nn_models.py executed with recording stand-ins for tf.keras.layers.Layer, "
                             "tfc.SignalConv2D, tfc.GDN (scripts/make_golden_from_reference.py); "
                             "constructed as at sga.py:70-73, bb_sga.py:69",
                   "num_filters": out}, f, indent=1)


def cli_fixture():
    """tf_boilerplate.py:91-204 (`parse_args`), executed unmodified with `absl.flags.argparse_flags` bound to
    argparse (argparse_flags.ArgumentParser is an argparse.ArgumentParser that also knows absl's own flags):
    the namespaces the reference's command line produces for a set of `compress` invocations."""
    import argparse
    import types
    absl = types.ModuleType("absl")
    flags = types.ModuleType("absl.flags")
    af = types.ModuleType("absl.flags.argparse_flags")
    af.ArgumentParser = argparse.ArgumentParser
    absl.flags, flags.argparse_flags = flags, af
    stubs = {"absl": absl, "absl.flags": flags, "absl.flags.argparse_flags": af}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        ns = {"sys": sys}
        _exec_reference_defs("tf_boilerplate.py", ["parse_args"], ns)
        cases = [
            ["--num_filters", "192", "compress", "mbt2018-num_filters=192-lmbda=0.01", "kodak.npy"],
            ["--verbose", "--num_filters", "256", "--checkpoint_dir", "/ckpt", "compress", "--lambda", "0.08",
             "--annealing_rate", "0.002", "--t0", "100", "--sga_its", "500", "--results_dir", "",
             "mbt2018-num_filters=256-lmbda=0.08", "tecnick.npy", "out.tfci"],
            ["-V", "--num_filters", "128", "--num_hfilters", "64", "compress", "run-lmbda=0.04-x", "in.png"],
        ]
        out = [{"argv": argv, "namespace": vars(ns["parse_args"](["prog"] + argv))} for argv in cases]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    with open(os.path.join(OUT, "cli_reference.json"), "w") as f:
        json.dump({"source": "tf_boilerplate.py parse_args executed with argparse_flags.ArgumentParser = "
                             "argparse.ArgumentParser (scripts/make_golden_from_reference.py)", "cases": out},
                  f, indent=1)


def main():
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    utils_fixtures()
    architecture_fixture()
    cli_fixture()
    prior_fixtures()
    import adam as ref_adam          # /root/reference/adam.py
    import configs as ref_configs    # /root/reference/configs.py
    os.makedirs(OUT, exist_ok=True)

    # ---- Adam: fixed-seed p, g float32 -> p after 1, 2, 3, 700, 2000 updates (adam.py:20-59)
    rng = np.random.RandomState(1234)
    n, steps, lr = 64, 2000, 0.005
    p0 = rng.standard_normal(n).astype(np.float32)
    scale = np.where(np.arange(steps) % 3 == 0, 1e-3, 1.0).astype(np.float32)[:, None]
    grads = (rng.standard_normal((steps, n)).astype(np.float32) * scale).astype(np.float32)
    opt = ref_adam.Adam(lr=lr)
    checkpoints = [1, 2, 3, 700, 2000]
    out = dict(p0=p0, grads=grads, lr=np.float64(lr), steps=np.int64(steps),
               checkpoints=np.array(checkpoints))
    p = [p0.copy()]
    for t in range(1, steps + 1):
        p = opt.update(p, [grads[t - 1]])
        if t in checkpoints:
            out[f"p_after_{t}"] = np.asarray(p[0], dtype=np.float64)
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(OUT, "adam_reference.npz"), **out)

    # ---- eval batch size (configs.py:5-9)
    sizes = [256 * 256, 768 * 512, 512 * 768, 1200 * 1200, 64 * 64, 1000 * 1000, 2000 * 3000]
    with open(os.path.join(OUT, "eval_batch_sizes.json"), "w") as f:
        json.dump({str(s): int(ref_configs.get_eval_batch_size(s)) for s in sizes}, f, indent=1)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
