"""Reference-EXECUTED fixtures for the operators that live in un-vendored wheels:
tfc.SignalConv2D, tfc.GDN, tfc.EntropyBottleneck._likelihood, tfc.GaussianConditional._likelihood (built and unbuilt),
tfp RelaxedOneHotCategorical.sample and tf.image.ssim_multiscale, as the reference composes them.

NOT RUNNABLE IN THE BUILD CONTAINER: it needs the reference's pinned environment (requirements.txt:19-22 -- TensorFlow
1.15, tensorflow-compression 1.3, tensorflow-probability 0.7.0) and the reference checkout.  On such a box:

    python scripts/make_golden_from_tf.py /path/to/reference tests/golden/tf_ops_reference.npz

It imports the reference's own nn_models.py, builds the four transforms at num_filters = 64 on a 64 x 64 input, overwrites
every variable with seeded random values (so that flips, transposes and gamma's index order all matter), and stores
  * the EFFECTIVE tensors read back from the layers (`layer.kernel`, `layer.bias`, `gdn.beta`, `gdn.gamma`: after tfc's
    RDFT / non-negative reparameterisations) under this build's weight names (weights.py::layer_shapes),
  * inputs and outputs of g_a, h_a, h_s, g_s, of both likelihoods, of the sampler and of MS-SSIM.
tests/test_tf_reference.py then runs the oracle (CPU) and the HIP path (GPU) on those tensors; until the file exists
those tests skip and DESIGN.md 4 keeps saying "tfc / tfp operators: parity unpinned".  Nothing here is imported by the
product or by any other test."""
import sys

import numpy as np


def main(ref_dir, out_path, C=64, H=64, W=64, seed=0):
    sys.path.insert(0, ref_dir)
    import tensorflow.compat.v1 as tf
    import tensorflow_compression as tfc
    import tensorflow_probability as tfp
    import nn_models                                     # the reference's file, unmodified

    tf.disable_eager_execution()
    rng = np.random.RandomState(seed)
    fx = {}
    x = tf.placeholder(tf.float32, [1, H, W, 3])
    ga = nn_models.AnalysisTransform(C)
    gs = nn_models.SynthesisTransform(C)
    ha = nn_models.HyperAnalysisTransform(C)
    hs = nn_models.MBT2018HyperSynthesisTransform(C, num_output_filters=2 * C)
    y = ga(x)                                            # sga.py:77
    z = ha(y)                                            # sga.py:78
    ms = hs(z)                                           # sga.py:107
    mu, sigma = ms[..., :C], tf.exp(ms[..., C:])         # sga.py:108 (split + exp)
    x_tilde = gs(y)                                      # sga.py:122
    eb = tfc.EntropyBottleneck()
    _ = eb(z, training=False)                            # the dummy call of sga.py:100 (creates the variables)
    z_t = tf.placeholder(tf.float32, z.shape)
    y_t = tf.placeholder(tf.float32, y.shape)
    eb_lik = eb._likelihood(z_t)                         # sga.py:101
    scale_table = np.exp(np.linspace(np.log(0.11), np.log(256), 64))          # sga.py:24-26
    cb_unbuilt = tfc.GaussianConditional(sigma, scale_table, mean=mu)          # sga.py:130-133: never called
    lik_unbuilt = cb_unbuilt._likelihood(y_t)
    cb_built = tfc.GaussianConditional(sigma, scale_table, mean=mu)            # mbt2018.py:77-80: called
    _ = cb_built(y_t, training=False)
    lik_built = cb_built._likelihood(y_t)

    sess = tf.Session()
    sess.run(tf.global_variables_initializer())
    # seeded random values for every variable (raw, i.e. before tfc's reparameterisations)
    for v in tf.global_variables():
        cur = sess.run(v)
        if "quantiles" in v.name:
            continue
        scale = 0.5 if ("matrix" in v.name or "factor" in v.name or "bias" in v.name) else 0.3
        v.load((cur + scale * np.abs(cur).mean() * rng.standard_normal(cur.shape) +
                0.02 * rng.standard_normal(cur.shape)).astype(np.float32), sess)

    def eff(prefix, layers, gdn_name):
        for i, layer in enumerate(layers):
            fx[f"{prefix}.k{i}"] = sess.run(layer.kernel)                    # HWIO, after the RDFT parameterizer
            if layer.use_bias:
                fx[f"{prefix}.b{i}"] = sess.run(layer.bias)
            act = layer.activation
            if isinstance(act, tfc.GDN):
                fx[f"{prefix}.beta{i}"] = sess.run(act.beta)
                fx[f"{prefix}.gamma{i}"] = sess.run(act.gamma)               # [C_in j, C_out i]: n_i = beta_i + sum_j gamma[j,i] x_j^2
    eff("ga", ga._layers, "gdn")
    eff("gs", gs._layers, "igdn")
    eff("ha", ha._layers, None)
    eff("hs", hs._layers, None)
    mats, biases, factors = sess.run([eb._matrices, eb._biases, eb._factors])
    for k in range(4):
        fx[f"eb.m{k}"] = np.log1p(np.exp(mats[k]))                           # softplus(matrix_k)
        fx[f"eb.b{k}"] = biases[k]
        if k < 3:
            fx[f"eb.f{k}"] = np.tanh(factors[k])

    xv = rng.rand(1, H, W, 3).astype(np.float32)
    yv, zv, muv, sgv, xt = sess.run([y, z, mu, sigma, x_tilde], {x: xv})
    fx.update(x=xv, y=yv, z=zv, mu=muv, sigma=sgv, x_tilde=xt)
    zt = (zv + rng.uniform(-0.5, 0.5, zv.shape)).astype(np.float32)
    yt = (yv + rng.uniform(-0.5, 0.5, yv.shape)).astype(np.float32)
    fx["z_tilde"], fx["y_tilde"] = zt, yt
    fx["eb_likelihood"] = sess.run(eb_lik, {z_t: zt})
    fx["gauss_likelihood_unbuilt"], fx["gauss_likelihood_built"] = sess.run([lik_unbuilt, lik_built], {x: xv, y_t: yt})
    fx["conditional_built_flags"] = np.array([int(cb_unbuilt.built), int(cb_built.built)])

    # tfp 0.7 RelaxedOneHotCategorical.sample with the uniforms INJECTED (sga.py:95-97): its one random op is replaced by
    # a constant for the duration of the call, so the fixture pins the deterministic transform (Gumbel, /T, softmax)
    logits = rng.standard_normal(yv.shape + (2,)).astype(np.float32)
    u = rng.uniform(1e-6, 1 - 1e-6, logits.shape).astype(np.float32)
    T = 0.37
    try:
        from unittest import mock
        with mock.patch.object(tf.random, "uniform", lambda *a, **k: tf.constant(u)), \
             mock.patch.object(tf, "random_uniform", lambda *a, **k: tf.constant(u), create=True):
            import tensorflow.compat.v2 as tf2
            with mock.patch.object(tf2.random, "uniform", lambda *a, **k: tf.constant(u)):
                s = tfp.distributions.RelaxedOneHotCategorical(T, logits=tf.constant(logits)).sample()
        fx["roc_logits"], fx["roc_u"], fx["roc_T"], fx["roc_sample"] = logits, u, np.float32(T), sess.run(s)
    except Exception as e:                                                    # keep the rest of the fixture
        print("RelaxedOneHotCategorical not pinned:", repr(e))

    a = np.round(255 * rng.rand(2, 192, 192, 3)).astype(np.float32)
    b = np.clip(a + 12 * rng.standard_normal(a.shape), 0, 255).round().astype(np.float32)
    fx["msssim_a"], fx["msssim_b"] = a, b
    fx["msssim"] = sess.run(tf.image.ssim_multiscale(tf.constant(a), tf.constant(b), 255))   # sga.py:175
    fx["versions"] = np.array([tf.__version__, tfc.__version__, tfp.__version__])
    np.savez_compressed(out_path, **fx)
    print("wrote", out_path, "with", len(fx), "arrays; conditional built flags", fx["conditional_built_flags"])


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    main(sys.argv[1], sys.argv[2])
