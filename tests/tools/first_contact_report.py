"""(Test infrastructure: it runs the oracle as the checker.)  Last step of scripts/first_contact_tf.sh: ONE file that turns SURVEY.md 8(c) / `parity` from "partial" to pinned.

Runs in THIS repo's environment (no TensorFlow needed) on the fixture scripts/make_golden_from_tf.py wrote on a TF 1.15 /
tfc 1.3 / tfp 0.7.0 box (tests/golden/tf_ops_reference.npz) and writes profiles/first_contact_report.json:

  * the sigma-bound mode the reference's SGA scripts execute (sga.py:130-133 constructs the GaussianConditional layer and
    calls `_likelihood` without building it): the un-called layer's `.built` flag and whether its likelihood equals the raw-
    sigma or the 0.11-bounded formula -- settles the PROVISIONAL default of include/sga_hip.h:72;
  * per operator (g_a, h_a, h_s, g_s, EntropyBottleneck._likelihood, GaussianConditional._likelihood raw / bounded,
    RelaxedOneHotCategorical.sample, tf.image.ssim_multiscale): the maximum error, relative to the tensor's maximum, of the
    ORACLE against TensorFlow and -- when a GPU is present -- of the HIP path against TensorFlow;
  * with a checkpoint directory (argument 2 or SGA_TF_CHECKPOINT): the bundle's variable names and shapes next to what
    sga_amd/tf_checkpoint.py expects, and whether the effective weights load.

    python tests/tools/first_contact_report.py [fixture.npz] [checkpoint_dir]        # commit profiles/first_contact_report.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def main():
    fixture = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "tf_ops_reference.npz")
    ckpt = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("SGA_TF_CHECKPOINT", "")
    out_path = os.path.join(ROOT, "profiles", "first_contact_report.json")
    rep = dict(fixture=os.path.relpath(fixture, ROOT), fixture_present=os.path.exists(fixture))
    if not rep["fixture_present"]:
        rep["status"] = "NOT RUN: the fixture does not exist (scripts/first_contact_tf.sh on a TF 1.15 box writes it)"
        print(json.dumps(rep, indent=1))
        return 1
    import sga_amd
    from sga_amd.weights import layer_shapes
    from oracle.sga_oracle import SGAOracle
    from oracle import msssim as oracle_msssim
    fx = dict(np.load(fixture))
    C = fx["gs.k0"].shape[2]
    w = {}
    for name, shape in layer_shapes(C).items():
        a = np.asarray(fx[name], dtype=np.float32)
        w[name] = a.reshape(shape) if name.startswith("eb.") else a
    o = SGAOracle(w)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32)

    # ---- the sigma bound --------------------------------------------------------------------------------------------
    raw = o.gauss_likelihood(t(fx["y_tilde"]), t(fx["mu"]), t(fx["sigma"]), 0.0)
    bnd = o.gauss_likelihood(t(fx["y_tilde"]), t(fx["mu"]), t(fx["sigma"]), 0.11)
    e_raw, e_bnd = rel(raw, fx["gauss_likelihood_unbuilt"]), rel(bnd, fx["gauss_likelihood_unbuilt"])
    rep["sigma_bound"] = dict(
        built_flags_uncalled_called=[int(v) for v in fx["conditional_built_flags"]],
        uncalled_layer_vs_raw_sigma_formula=e_raw, uncalled_layer_vs_0p11_bounded_formula=e_bnd,
        fraction_of_sigma_below_0p11=float((fx["sigma"] < 0.11).mean()),
        decisive=bool(abs(e_raw - e_bnd) > 1e-6))
    rep["sigma_bound"]["observed_mode_of_the_sga_scripts"] = (
        "UNDECIDED: the two formulas agree on this fixture (no predicted sigma below 0.11?)" if not rep["sigma_bound"]["decisive"]
        else ("raw sigma (scale_bound = 0: the shipped default is confirmed)" if e_raw < e_bnd
              else "0.11 bound (make SGA_SCALE_BOUND_BUILT the default of the SGA path)"))

    # ---- oracle vs TensorFlow ------------------------------------------------------------------------------------------
    ms = o.hyper_synthesis(t(fx["z"])).numpy()
    ops = {
        "g_a (nn_models.py:5-36)": rel(o.analysis(t(fx["x"])), fx["y"]),
        "h_a (nn_models.py:73-103)": rel(o.hyper_analysis(t(fx["y"])), fx["z"]),
        "h_s mu (nn_models.py:140-170)": rel(ms[..., :C], fx["mu"]),
        "h_s sigma": rel(np.exp(ms[..., C:]), fx["sigma"]),
        "g_s (nn_models.py:39-70)": rel(o.synthesis(t(fx["y"])), fx["x_tilde"]),
        "EntropyBottleneck._likelihood (sga.py:101)": rel(o.eb_likelihood(t(fx["z_tilde"])), fx["eb_likelihood"]),
        "GaussianConditional._likelihood, un-called layer, raw sigma": e_raw,
        "GaussianConditional._likelihood, called layer, 0.11 bound": rel(bnd, fx["gauss_likelihood_built"]),
        "tf.image.ssim_multiscale (sga.py:175)": float(np.abs(np.asarray(oracle_msssim.ssim_multiscale(t(fx["msssim_a"]), t(fx["msssim_b"]), 255.0))
                                                              - fx["msssim"]).max()),
    }
    if "roc_sample" in fx:
        noisy = (t(fx["roc_logits"]) - torch.log(-torch.log(t(fx["roc_u"])))) / float(fx["roc_T"])
        ops["RelaxedOneHotCategorical.sample (sga.py:95-97)"] = rel(torch.softmax(noisy, dim=-1), fx["roc_sample"])
    rep["oracle_vs_tf_max_rel_error"] = ops

    # ---- HIP path vs TensorFlow ----------------------------------------------------------------------------------------
    if torch.cuda.is_available():
        from sga_amd.codec import SGACodec
        B, H, W, _ = fx["x"].shape
        codec = SGACodec(w, C, B, H, W)
        y, z = codec.encode(fx["x"])
        g = fx["y"]
        for i in range(4):
            g = codec.layer_fwd(f"GS{i}", g).cpu().numpy()
        hs = fx["z"]
        for i in range(3):
            hs = codec.layer_fwd(f"HS{i}", hs).cpu().numpy()
        p_eb, _ = codec.factorized_likelihood(fx["z_tilde"])
        codec.set_scale_bound(0.0)
        p_raw = codec.gaussian_likelihood(fx["y_tilde"], fx["mu"], np.log(fx["sigma"]))[0]
        codec.set_scale_bound(0.11)
        p_bnd = codec.gaussian_likelihood(fx["y_tilde"], fx["mu"], np.log(fx["sigma"]))[0]
        rep["hip_vs_tf_max_rel_error"] = {
            "g_a": rel(y.cpu().numpy(), fx["y"]), "h_a": rel(z.cpu().numpy(), fx["z"]), "g_s": rel(g, fx["x_tilde"]),
            "h_s mu": rel(hs[..., :C], fx["mu"]), "h_s sigma": rel(np.exp(hs[..., C:]), fx["sigma"]),
            "EntropyBottleneck._likelihood": rel(p_eb.cpu().numpy(), fx["eb_likelihood"]),
            "GaussianConditional._likelihood raw": rel(p_raw.cpu().numpy(), fx["gauss_likelihood_unbuilt" if e_raw < e_bnd else "gauss_likelihood_built"]),
            "GaussianConditional._likelihood 0.11": rel(p_bnd.cpu().numpy(), fx["gauss_likelihood_built"]),
        }
        codec.close()
    else:
        rep["hip_vs_tf_max_rel_error"] = "no GPU in this environment: run again on the GPU box (the fixture travels with the repo)"

    # ---- a checkpoint TensorFlow wrote ----------------------------------------------------------------------------------
    if ckpt:
        from sga_amd import tf_checkpoint
        c = dict(directory=ckpt)
        try:
            tensors = tf_checkpoint.read_checkpoint(tf_checkpoint.latest_checkpoint(ckpt))
            c["variables"] = {k: list(v.shape) for k, v in sorted(tensors.items())}
            for nf in (192, 256, 128, 64):
                try:
                    ww = tf_checkpoint.load_effective_weights(ckpt, nf)
                    sga_amd.check_weights(ww, nf)
                    c["loaded_as_num_filters"] = nf
                    c["effective_weights_digest"] = sga_amd.weights_digest(ww)
                    c["medians_present"] = "eb.medians" in ww or "medians" in ww
                    break
                except Exception as e:      # report, do not stop: the point is to see the names
                    c.setdefault("load_errors", {})[str(nf)] = str(e)[:300]
        except Exception as e:
            c["read_error"] = str(e)[:300]
        rep["checkpoint"] = c
    tol = 2e-4
    worst = max(v for v in ops.values())
    rep["status"] = ("PINNED: every operator of the oracle within %g of TensorFlow" % tol) if worst < tol else \
        ("MISMATCH: %s" % ", ".join(k for k, v in ops.items() if v >= tol))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))
    print("wrote", out_path)
    return 0 if worst < tol else 2


if __name__ == "__main__":
    sys.exit(main())
