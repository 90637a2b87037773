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


def main():
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    utils_fixtures()
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
