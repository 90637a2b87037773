"""Dry run of tests/test_tf_reference.py: writes a STAND-IN for tests/golden/tf_ops_reference.npz computed by the oracle itself
(same keys and shapes as scripts/make_golden_from_tf.py produces on a TF box).  NOT a reference fixture: it only proves that the
consuming tests run (SGA_TF_FIXTURE=/tmp/standin.npz python -m pytest tests/test_tf_reference.py).  Never commit its output."""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import sga_amd
from oracle.sga_oracle import SGAOracle
from oracle import msssim as om
C=64
w = sga_amd.make_synthetic_weights(C, seed=3)
o = SGAOracle(w)
rng = np.random.RandomState(0)
fx = {k: np.asarray(v) for k, v in w.items()}
x = rng.rand(1,64,64,3).astype(np.float32)
t = lambda a: torch.as_tensor(a, dtype=torch.float32)
y = o.analysis(t(x)); z = o.hyper_analysis(y); ms = o.hyper_synthesis(z).numpy()
fx.update(x=x, y=y.numpy(), z=z.numpy(), mu=ms[..., :C], sigma=np.exp(ms[..., C:]), x_tilde=o.synthesis(y).numpy())
zt = (z.numpy() + rng.uniform(-.5,.5,z.shape)).astype(np.float32); yt = (y.numpy() + rng.uniform(-.5,.5,y.shape)).astype(np.float32)
fx["z_tilde"], fx["y_tilde"] = zt, yt
fx["eb_likelihood"] = o.eb_likelihood(t(zt)).numpy()
fx["gauss_likelihood_unbuilt"] = o.gauss_likelihood(t(yt), t(fx["mu"]), t(fx["sigma"]), 0.0).numpy()
fx["gauss_likelihood_built"] = o.gauss_likelihood(t(yt), t(fx["mu"]), t(fx["sigma"]), 0.11).numpy()
fx["conditional_built_flags"] = np.array([0,1])
logits = rng.standard_normal(y.shape + (2,)).astype(np.float32); u = rng.uniform(1e-6, 1-1e-6, logits.shape).astype(np.float32)
fx["roc_logits"], fx["roc_u"], fx["roc_T"] = logits, u, np.float32(0.37)
fx["roc_sample"] = torch.softmax((t(logits) - torch.log(-torch.log(t(u)))) / 0.37, dim=-1).numpy()
a = np.round(255*rng.rand(2,192,192,3)).astype(np.float32); b = np.clip(a + 12*rng.standard_normal(a.shape),0,255).round().astype(np.float32)
fx["msssim_a"], fx["msssim_b"] = a, b
fx["msssim"] = np.asarray(om.ssim_multiscale(t(a), t(b), 255.0))
np.savez_compressed(sys.argv[1], **fx)
