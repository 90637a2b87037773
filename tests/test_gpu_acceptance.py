"""North-star acceptance (BASELINE.json: "results match ... within 1e-3 BPP / 0.01 dB PSNR"), applied
HIP path vs oracle on the COMPLETE 2000-step run (sga.py:210-247: encode, 2000 x (sample, forward,
backward, Adam), round, evaluate), on the GPU box, against committed golden vectors.

`tests/golden/full_run_oracle.json` holds the oracle's per-image end metrics for one synthetic
batch and 32 Philox seeds (generator: tests/tools/make_golden_full_run.py).  The HIP path consumes
identical noise, but the optimisation is chaotic in the last float32 bits: a rounding difference in
one convolution eventually flips one floor/ceil decision, after which the two runs are different
draws of the same stochastic optimiser (per-image seed-to-seed sigma: 1e-3..3e-3 bpp).  A per-image,
per-seed bound of 1e-3 bpp is therefore not a property even of the reference against itself; the
tolerance is asserted on the MEAN over images x seeds, together with the measured standard error,
and per-image deviations are bounded by the oracle's own seed-to-seed spread.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_BPP, TOL_PSNR = 1e-3, 0.01          # north_star tolerance


# the small set (C = 64, 4 x 64^2, 32 seeds), one at the north star's width (C = 192, 2 x 128^2, 64 seeds), and cfg 5:
# the complete two-stage bits-back run of bb_sga.py:199-276 (C = 64, 2 x 64^2, 2000 + 2000 iterations, 32 seeds)
# a ragged set (3 x 50 x 70: latents 4 x 5 and 1 x 2, every crop live in all 2000 steps) and one at cfg 4's rate
# point (lambda = 0.08)
# ... the small set again against the FLOAT64 oracle (the control of tests/test_oracle.py: float32-vs-float64 oracle runs
# differ from each other exactly as the HIP path differs from either), and -- when present -- the BENCHMARKED geometry
# (cfg 2: B = 8, 256^2, C = 192: POST / 256-row LDS-DMA / split gs2.bwd kernels and the two-stream graph over 2000 steps)
# ... and (round 4) a TRAINED-LIKE operating point: weights fitted on low-pass noise (tests/tools/fit_weights.py: 0.47 bpp /
# 33 dB, 55 % of y_hat at 0, predicted scales 0.09 .. 4.4 with 8 % below 0.11) in BOTH sigma-bound modes -- `fitted` is the
# resolvable full-run check of the SGA default (scale_bound = 0, raw sigma: sga.py:130-133), `fitted_b011` of a built layer
# ... and (round 5) the same at the NORTH STAR'S WIDTH: C = 192 fitted to 0.39 bpp / 33.5 dB one-shot (87 % of y_hat at 0, 78 % of
# the predicted scales below 0.11; tests/golden/fitted_weights_c192.npz), 2 x 128^2, 16 seeds, both sigma-bound modes: SGA takes
# the set's images to 0.3625 bpp / 36.10 dB, the regime of results/kodak/sga-psnr.csv; seed-to-seed sigma 3.5e-4 .. 4.7e-4 bpp
@pytest.mark.parametrize("golden", ["full_run_oracle.json", "full_run_oracle_c192.json", "full_run_oracle_bb.json",
                                    "full_run_oracle_ragged.json", "full_run_oracle_hirate.json",
                                    "full_run_oracle_f64.json", "full_run_oracle_fitted.json",
                                    "full_run_oracle_fitted_b011.json", "full_run_oracle_bb_fitted.json",
                                    "full_run_oracle_fitted_c192.json", "full_run_oracle_fitted_c192_b011.json",
                                    # ... and the fitted model AT THE BENCHMARKED GEOMETRY (B = 8, 256^2, C = 192; 8 seeds x 8 images):
                                    # the production launch plan over 2000 iterations on a set that RESOLVES 1e-3 bpp
                                    "full_run_oracle_cfg2_fitted.json"])
def test_full_run_matches_oracle_golden_within_north_star_tolerance(gpu_out_dir, golden):
    rep = _acceptance(gpu_out_dir, golden, "f32", "")
    assert rep["resolves_1e3_bpp"], rep          # these sets are large enough for the tolerance itself to be the bound


def test_full_run_at_the_benchmarked_geometry_is_consistent_with_the_oracle(gpu_out_dir):
    """cfg 2 (B = 8, 256^2, C = 192; 2.3 CPU-hours per oracle seed).  This set does NOT resolve the north-star tolerance:
    its runs end at 4.1 bpp with a seed-to-seed sigma of 7e-3 .. 1.3e-2, so the standard error of the mean over the
    affordable seeds (15 x 8 images: 6.3e-4) is of the order of the tolerance (1e-3) itself.  What is asserted is
    CONSISTENCY -- |mean| <= 3 standard errors, single runs inside the optimiser's own spread, PSNR within 0.01 dB (that one
    is resolved) -- and the report says `resolves_1e3_bpp: false`.  The 1e-3 criterion at this geometry is carried by the
    deterministic traces below (300 iterations of an accelerated schedule; all 2000 of the production schedule) and by
    the resolvable sets at the small geometries in the same sigma-bound mode (`fitted`)."""
    rep = _acceptance(gpu_out_dir, "full_run_oracle_cfg2.json", "f32", "")
    assert abs(rep["mean_d_psnr"]) <= TOL_PSNR


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x2"])
def test_full_run_at_the_benchmarked_geometry_bf16_modes(gpu_out_dir, precision):
    """The same consistency check (see above: this set does not resolve 1e-3 bpp; PSNR it does) for the two bf16-pipe modes
    at the geometry `alt_precision` / `fast_precision` of bench.py are timed on: the 256-row X3 / X2 instances with the IGDN
    post-phase and the two-stream graph over 2000 steps."""
    rep = _acceptance(gpu_out_dir, "full_run_oracle_cfg2.json", precision, "_" + precision)
    assert abs(rep["mean_d_psnr"]) <= TOL_PSNR


@pytest.mark.parametrize("golden", ["full_run_oracle.json", "full_run_oracle_fitted.json", "full_run_oracle_fitted_c192.json",
                                    "full_run_oracle_cfg2_fitted.json"])
def test_full_run_bf16x3_mode_within_north_star_tolerance(gpu_out_dir, golden):
    """The opt-in precision mode (exact 3 x bf16 operand split; `alt_precision` of bench.py, not the
    headline) on the small set and on the trained-like one."""
    rep = _acceptance(gpu_out_dir, golden, "bf16x3", "_bf16x3")
    assert rep["resolves_1e3_bpp"], rep


@pytest.mark.parametrize("golden", ["full_run_oracle.json", "full_run_oracle_fitted.json", "full_run_oracle_fitted_c192.json",
                                    "full_run_oracle_cfg2_fitted.json"])
def test_full_run_bf16x2_mode_within_north_star_tolerance(gpu_out_dir, golden):
    """The fast precision mode (two bf16 planes per convolution operand: 16 mantissa bits, include/sga_hip.h) is not
    f32-grade per step, but what the north star asks for is the END of a 2000-step run -- 1e-3 bpp / 0.01 dB against the
    reference -- and SGA is a stochastic optimiser whose own seed spread is 10x the tolerance: same sets, same criterion."""
    rep = _acceptance(gpu_out_dir, golden, "bf16x2", "_bf16x2")
    assert rep["resolves_1e3_bpp"], rep


def _inputs(cfg):
    if cfg.get("inputs") == "lowpass":
        return sga_amd.make_lowpass_images(cfg["B"], cfg["H"], cfg["W"], seed=cfg["x_seed"])
    return np.random.RandomState(cfg["x_seed"]).rand(cfg["B"], cfg["H"], cfg["W"], 3).astype(np.float32)


def _weights(cfg):
    if cfg.get("weights"):      # "fitted_c64" -> tests/golden/fitted_weights_c64.npz (tests/tools/fit_weights.py)
        return sga_amd.load_weights_npz(os.path.join(ROOT, "tests", "golden", "fitted_weights_%s.npz" % cfg["weights"].split("_")[1]))
    return sga_amd.make_synthetic_weights(cfg["C"], seed=cfg["weight_seed"], bb=bool(cfg.get("bb")))


def _acceptance(gpu_out_dir, golden, precision, tag):
    from sga_amd.codec import SGACodec, metrics_to_dict
    path = os.path.join(ROOT, "tests", "golden", golden)
    if not os.path.exists(path):
        pytest.skip(golden + " not generated yet (tests/tools/make_golden_full_run.py)")
    with open(path) as f:
        gold = json.load(f)
    cfg = gold["config"]
    C, B, H, W = cfg["C"], cfg["B"], cfg["H"], cfg["W"]
    x, w = _inputs(cfg), _weights(cfg)
    bb = bool(cfg.get("bb"))
    # the sets of rounds 1-2 were generated with the 0.11 sigma bound hard-coded; the cfg-2 set with the default
    # (0: sga.py:130-133 never builds the tfc layer).  The HIP path runs in the mode its golden set was made in.
    codec = SGACodec(w, C, B, H, W, precision=precision, bits_back=bb, scale_bound=cfg.get("scale_bound", 0.11))
    d_bpp, d_psnr, hip_bpp, hip_psnr, d_back = [], [], [], [], []
    for run in gold["runs"]:
        if bb:
            _, _, met, _, _ = codec.bb_run(x, cfg["lmbda"], its=cfg["its"], r_its=cfg["r_its"], seed=run["seed"])
        else:
            _, _, met, _ = codec.run(x, cfg["lmbda"], its=cfg["its"], seed=run["seed"])
        m = metrics_to_dict(met)
        if bb:
            d_back.append(m["est_bpp_back"].astype(np.float64) - np.array(run["est_bpp_back"]))
        d_bpp.append(m["est_bpp"].astype(np.float64) - np.array(run["est_bpp"]))
        d_psnr.append(m["psnr"].astype(np.float64) - np.array(run["psnr"]))
        hip_bpp.append(m["est_bpp"].astype(np.float64)); hip_psnr.append(m["psnr"].astype(np.float64))
    codec.close()
    d_bpp, d_psnr = np.array(d_bpp), np.array(d_psnr)           # [seed, image]
    n = d_bpp.size
    spread = gold["oracle_seed_spread"]
    rep = dict(n=n, seeds=len(gold["runs"]), images=B,
               mean_d_bpp=float(d_bpp.mean()), sem_d_bpp=float(d_bpp.std(ddof=1) / np.sqrt(n)),
               mean_d_psnr=float(d_psnr.mean()), sem_d_psnr=float(d_psnr.std(ddof=1) / np.sqrt(n)),
               max_abs_d_bpp=float(np.abs(d_bpp).max()), max_abs_d_psnr=float(np.abs(d_psnr).max()),
               frac_within_1e3_bpp=float((np.abs(d_bpp) <= TOL_BPP).mean()),
               frac_within_001_db=float((np.abs(d_psnr) <= TOL_PSNR).mean()),
               per_image_mean_d_bpp=d_bpp.mean(0).tolist(), per_image_mean_d_psnr=d_psnr.mean(0).tolist(),
               hip_seed_std_bpp=np.array(hip_bpp).std(0, ddof=1).tolist(),
               oracle_seed_std_bpp=spread["est_bpp_std_per_image"],
               hip_seed_std_psnr=np.array(hip_psnr).std(0, ddof=1).tolist(),
               oracle_seed_std_psnr=spread["psnr_std_per_image"])
    if bb:      # the bits-back refund (bb_sga.py:181 `est_bpp_back`): the net rate is est_bpp - est_bpp_back
        d_back = np.array(d_back)
        rep.update(mean_d_bpp_back=float(d_back.mean()), sem_d_bpp_back=float(d_back.std(ddof=1) / np.sqrt(n)),
                   max_abs_d_bpp_back=float(np.abs(d_back).max()))
    with open(os.path.join(gpu_out_dir, golden.replace("full_run_oracle", "acceptance_full_run").replace(".json", tag + ".json")), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    # the north-star tolerance, on the means (standard errors stated in the report: ~3e-4 bpp, ~1e-3 dB).
    # Where the set is too small to resolve 1e-3 bpp -- the set at the benchmarked geometry costs 1.9 CPU-hours per seed
    # and its runs end at 4.1 bpp with a seed-to-seed sigma of 7e-3..1.3e-2, so n = 40..80 runs give a standard error of
    # ~1e-3 -- the bound is 3 standard errors (a criterion tighter than the noise cannot be tested), the report states
    # both, and the deterministic trace test below carries the weight for that geometry.
    resolvable = rep["sem_d_bpp"] <= TOL_BPP / 3
    rep["resolves_1e3_bpp"] = bool(resolvable)
    rep["criterion_bpp"] = "abs(mean) <= 1e-3 (the north-star tolerance)" if resolvable else \
        "CONSISTENCY ONLY: abs(mean) <= 3 standard errors; this set cannot resolve 1e-3 and does not claim it"
    with open(os.path.join(gpu_out_dir, golden.replace("full_run_oracle", "acceptance_full_run").replace(".json", tag + ".json")), "w") as f:
        json.dump(rep, f, indent=1)
    assert abs(rep["mean_d_bpp"]) <= (TOL_BPP if resolvable else 3 * rep["sem_d_bpp"]), rep
    assert abs(rep["mean_d_psnr"]) <= TOL_PSNR, rep
    if bb:
        assert abs(rep["mean_d_bpp_back"]) <= TOL_BPP, rep
    # per image (mean over the seeds): no systematic offset of any image beyond the tolerance.  HARD bounds (1e-3 bpp, 0.01 dB)
    # for every set whose single runs scatter by less than the tolerance (all sets of rounds 1-3 with >= 16 seeds).  Only where
    # a per-image mean over k seeds carries a standard error that is not small against the tolerance -- fewer than 16 seeds, or
    # the trained-like sets (config key `weights`: fitted models, single runs scatter by 0.04 dB) -- the bound is 3.5 standard
    # errors of that image's mean (the allowance is no longer applied to the synthetic-weight sets)
    k = d_bpp.shape[0]
    noisy = k < 16 or bool(cfg.get("weights"))
    rep["per_image_criterion"] = "3.5 standard errors (noisy set)" if noisy else "hard 1e-3 bpp / 0.01 dB"
    # (k < 16: the standard error itself is estimated from few runs, so the multiplier is the Student-t quantile at the two-sided
    #  tail probability of 3.5 sigma -- 5.9 for k = 8 -- instead of the normal one; sets with >= 16 seeds keep 3.5)
    nse = 3.5
    if k < 16:
        from scipy import stats
        nse = float(stats.t.ppf(1.0 - stats.norm.sf(3.5), k - 1))
    rep["per_image_standard_errors"] = nse
    img_tol = np.maximum(TOL_BPP, nse * d_bpp.std(0, ddof=1) / np.sqrt(k)) if noisy else np.full(d_bpp.shape[1], TOL_BPP)
    img_tol_p = np.maximum(TOL_PSNR, nse * d_psnr.std(0, ddof=1) / np.sqrt(k)) if noisy else np.full(d_psnr.shape[1], TOL_PSNR)
    rep["per_image_tol_bpp"], rep["per_image_tol_psnr"] = img_tol.tolist(), img_tol_p.tolist()
    assert (np.abs(d_bpp.mean(0)) <= img_tol).all(), rep
    assert (np.abs(d_psnr.mean(0)) <= img_tol_p).all(), rep
    # single runs stay inside the optimiser's own noise: 5 sigma of the seed-to-seed spread of a
    # difference of two draws (sqrt(2) sigma), and the HIP path's spread equals the oracle's
    sig_b = np.sqrt(2.0) * np.array(spread["est_bpp_std_per_image"])
    sig_p = np.sqrt(2.0) * np.array(spread["psnr_std_per_image"])
    nsig = 5.0
    if len(gold["runs"]) < 16:      # a per-image sigma from < 16 seeds is itself uncertain by 20-35 %: pool the images, 6 sigma
        sig_b = np.maximum(sig_b, np.sqrt(np.mean(sig_b ** 2)))
        sig_p = np.maximum(sig_p, np.sqrt(np.mean(sig_p ** 2)))
        nsig = 6.0
    assert (np.abs(d_bpp) <= nsig * sig_b[None, :] + 1e-4).all(), rep
    assert (np.abs(d_psnr) <= nsig * sig_p[None, :] + 1e-3).all(), rep
    ratio = np.array(rep["hip_seed_std_bpp"]) / np.array(spread["est_bpp_std_per_image"])
    # (a standard deviation estimated from k seeds has a relative error of ~ 1 / sqrt(2 (k - 1)): wider bounds for the
    # 5-seed set at the benchmarked geometry)
    lo, hi = (0.5, 2.0) if len(gold["runs"]) >= 16 else (0.3, 3.3)
    assert (ratio > lo).all() and (ratio < hi).all(), rep
    return rep


def test_trace_at_the_benchmarked_geometry(gpu_out_dir):
    """cfg 2 (B = 8, 256^2, C = 192) step by step: the per-iteration trace (rd_loss, train_mse, train_bpp, mean PSNR;
    sga.py:216-236 logs these) of the first 300 iterations -- temperature annealing inside them -- against the oracle's
    committed trace (`full_run_oracle_cfg2trace.json`).  This is the production launch plan over hundreds of steps (IGDN
    post-phase, 256-row LDS-DMA tiles, split 256-row gs2.bwd, two-stream graph replay) while the two float32 trajectories
    are still the same trajectory -- measured: rd_loss within 2.3e-6 (relative) over the first 100 iterations and 8.3e-6
    over all 300 (it falls from 84.9 to 71.8), train_bpp within 2.5e-5 / 1.2e-4, mean PSNR within 5e-6 -- where the
    statistical sets can only compare end points.  Bounds asserted: 1e-4 over the first 100 iterations, 1e-3 over all."""
    from sga_amd.codec import SGACodec
    path = os.path.join(ROOT, "tests", "golden", "full_run_oracle_cfg2trace.json")
    if not os.path.exists(path):
        pytest.skip("full_run_oracle_cfg2trace.json not generated yet (GOLDEN=cfg2trace tests/tools/make_golden_full_run.py)")
    with open(path) as f:
        gold = json.load(f)
    cfg, run = gold["config"], gold["runs"][0]
    C, B, H, W = cfg["C"], cfg["B"], cfg["H"], cfg["W"]
    x = _inputs(cfg)
    codec = SGACodec(_weights(cfg), C, B, H, W, scale_bound=cfg["scale_bound"])
    want = np.array(run["trace"])
    outs = []
    for _ in range(2):
        y_hat, z_hat, met, tr = codec.run(x, cfg["lmbda"], its=cfg["its"], t0=cfg["t0"], annealing_rate=cfg["annealing_rate"],
                                          seed=run["seed"], trace=True)
        outs.append((y_hat, tr))
    import torch
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])          # bit-reproducible
    got = outs[0][1].cpu().numpy().astype(np.float64)
    rel = np.abs(got / want - 1)
    rep = dict(its=int(cfg["its"]), max_rel_first_100=rel[:100].max(0).tolist(), max_rel_all=rel.max(0).tolist(),
               rd_loss_first=float(want[0, 0]), rd_loss_last=float(want[-1, 0]), rd_loss_last_hip=float(got[-1, 0]),
               frac_nonzero_y_hat_oracle=run.get("frac_nonzero_y_hat", (1.0 - run["frac_zero_y_hat"]) if "frac_zero_y_hat" in run else None), frac_nonzero_y_hat_hip=float((y_hat != 0).float().mean()))
    with open(os.path.join(gpu_out_dir, "acceptance_trace_cfg2.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    assert want[-1, 0] < 0.9 * want[0, 0]                           # the run really optimises over these iterations
    assert (rel[:100, :3] < 1e-4).all(), rep
    assert (rel[:, :3] < 1e-3).all() and (np.abs(got[:, 3] - want[:, 3]) < 0.01).all(), rep
    codec.close()


@pytest.mark.parametrize("golden", ["cfg2trace2000", "cfg2_fitted_trace2000"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x2"])
def test_trace_2000_at_the_production_schedule(gpu_out_dir, precision, golden):
    """(round 5: also on the FITTED C = 192 model -- `cfg2_fitted_trace2000`: the same geometry and schedule at a codec's operating
    point, 0.35 bpp / 36.6 dB, where 89 % of y_hat end at 0 -- and in the two bf16-pipe modes, whose per-step error is larger -- same golden trace, the f32 test's bounds, the
    first-divergence iterations reported per mode in acceptance_trace2000_cfg2_<mode>.json; VERDICT r4 #3c.)
    cfg 2 under the PRODUCTION schedule (t0 = 700, rate 1e-3: sga.py:193-196), all 2000 iterations, step by step against
    the oracle's committed trace of the same run (`full_run_oracle_cfg2trace2000.json` = seed 0 of the cfg-2 golden set
    with its per-iteration scalars kept; 2.3 CPU-hours).  Both sides draw identical Philox noise, so the two float32
    trajectories ARE one trajectory until rounding differences have flipped enough floor / ceil decisions to show in the
    batch scalars; from then on they are two draws of the same stochastic optimiser.  Reported: the first iteration at
    which the relative rd_loss difference exceeds 1e-6 / 1e-5 / 1e-4 / 1e-3 and the maximum over four windows.  Asserted:
    rd_loss, train_mse and train_bpp within 1e-4 (relative) over the first 300 iterations -- T is at its upper bound for
    700 iterations under this schedule, the regime the accelerated 300-iteration trace test never sees -- and within
    3e-3 afterwards (3 sigma of the seed-to-seed spread of the batch-mean bpp at this geometry: sigma_image 7e-3 .. 1.3e-2
    of 4.1 bpp over 8 images), mean PSNR within 0.01 dB throughout, and the end point within the same bounds."""
    from sga_amd.codec import SGACodec
    path = os.path.join(ROOT, "tests", "golden", "full_run_oracle_%s.json" % golden)
    if not os.path.exists(path):
        pytest.skip("full_run_oracle_%s.json not generated yet (GOLDEN=%s tests/tools/make_golden_full_run.py)" % (golden, golden))
    with open(path) as f:
        gold = json.load(f)
    cfg, run = gold["config"], gold["runs"][0]
    C, B, H, W = cfg["C"], cfg["B"], cfg["H"], cfg["W"]
    codec = SGACodec(_weights(cfg), C, B, H, W, scale_bound=cfg["scale_bound"], precision=precision)
    want = np.array(run["trace"])
    y_hat, z_hat, met, tr = codec.run(_inputs(cfg), cfg["lmbda"], its=cfg["its"], seed=run["seed"], trace=True)
    got = tr.cpu().numpy().astype(np.float64)
    rel = np.abs(got / want - 1)

    def first_above(th):
        idx = np.nonzero(rel[:, 0] > th)[0]
        return int(idx[0]) if idx.size else None

    windows = [(0, 100), (100, 300), (300, 1000), (1000, cfg["its"])]
    from sga_amd.codec import metrics_to_dict
    m = metrics_to_dict(met)
    nwin = cfg["its"] // 100          # 100-iteration window means: the optimiser's state without the draw-to-draw noise
    win_rel = np.abs(got[:nwin * 100, :3].reshape(nwin, 100, 3).mean(1) / want[:nwin * 100, :3].reshape(nwin, 100, 3).mean(1) - 1)
    rep = dict(precision=precision, its=int(cfg["its"]), max_rel_window_means=win_rel.max(0).tolist(), first_iteration_rel_rd_loss_above={"1e-6": first_above(1e-6), "1e-5": first_above(1e-5),
                                                                       "1e-4": first_above(1e-4), "1e-3": first_above(1e-3)},
               max_rel_by_window={"%d-%d" % w: rel[w[0]:w[1], :3].max(0).tolist() for w in windows},
               max_abs_d_mean_psnr=float(np.abs(got[:, 3] - want[:, 3]).max()),
               rd_loss_first=float(want[0, 0]), rd_loss_last=float(want[-1, 0]), rd_loss_last_hip=float(got[-1, 0]),
               end_d_bpp_mean=float(m["est_bpp"].mean() - np.mean(run["est_bpp"])),
               end_d_psnr_mean=float(m["psnr"].mean() - np.mean(run["psnr"])),
               frac_nonzero_y_hat_oracle=run.get("frac_nonzero_y_hat", (1.0 - run["frac_zero_y_hat"]) if "frac_zero_y_hat" in run else None), frac_nonzero_y_hat_hip=float((y_hat != 0).float().mean()))
    with open(os.path.join(gpu_out_dir, "acceptance_trace2000_%s%s.json" % (golden.replace("trace2000", "").rstrip("_"),
                                                                             "" if precision == "f32" else "_" + precision)), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    if cfg.get("weights"):
        # The FITTED model: measured (profiles/r05_acceptance_trace2000_cfg2_fitted*.json), the two float32 trajectories separate ten
        # times EARLIER than on the synthetic weights -- 1e-5 at iteration 15 in all three modes, 1e-4 at 34-56, 1e-3 at ~190 (synthetic:
        # 345 / never / never): 89 % of the rounded latents are zero here (`frac_zero_y_hat`), the objective is 0.5-0.77 instead of
        # 72-85, and a last-bit difference flips a floor / ceil draw within tens of iterations.  From then on they are two draws of the
        # same optimiser: mid-run rd_loss differs by up to 0.9 %, the END agrees to 5e-5 bpp / 0.004 dB (the statistics of
        # `cfg2_fitted`: sigma 1.5-3.8e-4 bpp).  So: ONE trajectory to 1e-5 over the first 10 iterations (an implementation difference
        # shows at once) and to 1e-4 over the first 25; a sanity bound afterwards; the north-star tolerance itself at the end.
        assert (rel[:10, :3] < 1e-5).all() and (rel[:25, :3] < 1e-4).all(), rep
        assert (rel[:, :3] < 3e-2).all() and (np.abs(got[:, 3] - want[:, 3]) < 0.06).all(), rep
        assert (win_rel < 4e-3).all(), rep          # ... and the 100-iteration window means stay together all the way (round 6;
                                                   # measured 1.0e-3 / 4.6e-4 / 9.7e-4 in f32 / bf16x3 / bf16x2, pointwise up to 1 %)
        assert abs(rep["end_d_bpp_mean"]) < TOL_BPP and abs(rep["end_d_psnr_mean"]) < TOL_PSNR, rep
    else:
        assert (rel[:300, :3] < 1e-4).all(), rep
        assert (rel[:, :3] < 3e-3).all() and (np.abs(got[:, 3] - want[:, 3]) < 0.01).all(), rep
        assert abs(rep["end_d_bpp_mean"]) < 3e-3 * float(np.mean(run["est_bpp"])) and abs(rep["end_d_psnr_mean"]) < 0.01, rep
    codec.close()


# ---- round 6: COMPLETE runs at the REAL sizes of BASELINE.json's configs 3, 4 and 5 against the oracle ---------------
# One image, one seed, all 2000 (+ 2000) iterations, trained-like models, raw sigma.  Rounds 2-5 had oracle parity at these sizes for
# ONE step (tests/test_gpu_configs.py) and complete-run goldens only up to 256 x 256; the full runs at Kodak / Tecnick size asserted
# properties (finite, reproducible, objective falls).  These replace them: the oracle's own trace of the same run is the reference.
REAL_SIZE_SETS = {
    # name: (golden file stem, what BASELINE.json calls it)
    "cfg3_kodak": "cfg3_kodak_trace2000",          # sga.py:201-247, 512 x 768, C = 192, lambda = 0.01
    "cfg4_tecnick": "cfg4_tecnick_trace2000",      # 1200 x 1200 (75 x 75 latents, 76 -> 75 crop), C = 256 fitted at lambda = 0.08
    "cfg5_kodak": "cfg5_kodak_trace2000",          # bb_sga.py:199-276, 512 x 768, C = 192 bits-back model, 2000 + 2000
}


@pytest.mark.parametrize("name", sorted(REAL_SIZE_SETS))
def test_complete_run_at_the_real_size_follows_the_oracle(gpu_out_dir, name):
    """configs.py:5-9 / sga.py:201-247 / bb_sga.py:199-276 at full size.  Both sides draw identical Philox noise, so the HIP run and
    the oracle's run are ONE trajectory until float32 rounding has flipped a floor / ceil draw (trained-like models: within tens of
    iterations, see test_trace_2000_at_the_production_schedule), then two draws of one optimiser.  Asserted: the per-iteration
    scalars (rd_loss, train_mse, train_bpp) within 1e-5 over the first 10 iterations and 1e-4 over the first 25, a sanity bound to
    the end, and the END POINT -- est_bpp, PSNR (and the bits-back refund) of the rounded latents -- within the north-star
    tolerance, 1e-3 bpp / 0.01 dB.  Also the oracle-free properties the old full-run tests held: integer latents, the metrics
    are the evaluation of the returned latents, the rate fields add up."""
    import torch
    from sga_amd.codec import SGACodec, metrics_to_dict
    stem = REAL_SIZE_SETS[name]
    path = os.path.join(ROOT, "tests", "golden", "full_run_oracle_%s.json" % stem)
    if not os.path.exists(path):
        pytest.skip("full_run_oracle_%s.json not generated yet (GOLDEN=%s NTHREADS=8 tests/tools/make_golden_full_run.py)" % (stem, stem))
    with open(path) as f:
        gold = json.load(f)
    cfg, run = gold["config"], gold["runs"][0]
    C, B, H, W = cfg["C"], cfg["B"], cfg["H"], cfg["W"]
    bb = bool(cfg.get("bb"))
    x = _inputs(cfg)
    codec = SGACodec(_weights(cfg), C, B, H, W, scale_bound=cfg["scale_bound"], bits_back=bb)
    want = np.array(run["trace"])
    if bb:
        y_hat, zml, met, tr, tr2 = codec.bb_run(x, cfg["lmbda"], its=cfg["its"], r_its=cfg["r_its"], seed=run["seed"], trace=True)
    else:
        y_hat, z_hat, met, tr = codec.run(x, cfg["lmbda"], its=cfg["its"], seed=run["seed"], trace=True)
    got = tr.cpu().numpy().astype(np.float64)
    rel = np.abs(got / want - 1)
    m = metrics_to_dict(met)

    def first_above(th):
        idx = np.nonzero(rel[:, 0] > th)[0]
        return int(idx[0]) if idx.size else None

    # between the first tens of iterations (one trajectory) and the end point (the tolerance) the two runs are two draws of one
    # optimiser: pointwise they differ by up to ~1 % in the middle of the annealing, but their 100-iteration WINDOW MEANS -- the
    # optimiser's state without the draw-to-draw noise -- must stay together over the whole run
    nwin = cfg["its"] // 100
    wm_got = got[:nwin * 100, :3].reshape(nwin, 100, 3).mean(1)
    wm_want = want[:nwin * 100, :3].reshape(nwin, 100, 3).mean(1)
    win_rel = np.abs(wm_got / wm_want - 1)
    rep = dict(config=name, C=C, H=H, W=W, its=int(cfg["its"]),
               max_rel_window_means=win_rel.max(0).tolist(), worst_window=int(win_rel[:, 0].argmax()),
               first_iteration_rel_rd_loss_above={k: first_above(float(k)) for k in ("1e-6", "1e-5", "1e-4", "1e-3")},
               max_rel_first_10=rel[:10, :3].max(0).tolist(), max_rel_first_25=rel[:25, :3].max(0).tolist(),
               max_rel_all=rel[:, :3].max(0).tolist(), max_abs_d_psnr=float(np.abs(got[:, 3] - want[:, 3]).max()),
               rd_loss_first=float(want[0, 0]), rd_loss_last=float(want[-1, 0]), rd_loss_last_hip=float(got[-1, 0]),
               est_bpp_oracle=run["est_bpp"], est_bpp_hip=m["est_bpp"].tolist(), psnr_oracle=run["psnr"], psnr_hip=m["psnr"].tolist(),
               end_d_bpp=float(m["est_bpp"][0] - run["est_bpp"][0]), end_d_psnr=float(m["psnr"][0] - run["psnr"][0]),
               frac_zero_y_hat_oracle=run.get("frac_zero_y_hat"), frac_zero_y_hat_hip=float((y_hat == 0).float().mean()))
    if bb:
        want2, got2 = np.array(run["trace2"]), tr2.cpu().numpy().astype(np.float64)[:, 2]      # stage 2: the rate-only objective
        rel2 = np.abs(got2 / want2 - 1)
        rep.update(stage2_max_rel_first_25=float(rel2[:25].max()), stage2_max_rel_all=float(rel2.max()),
                   stage2_first=float(want2[0]), stage2_last=float(want2[-1]), stage2_last_hip=float(got2[-1]),
                   end_d_bpp_back=float(m["est_bpp_back"][0] - run["est_bpp_back"][0]))
    with open(os.path.join(gpu_out_dir, "acceptance_real_size_%s.json" % name), "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep))
    assert want[-1, 0] < want[0, 0]                                  # the run optimises
    assert (rel[:10, :3] < 1e-5).all() and (rel[:25, :3] < 1e-4).all(), rep
    assert (rel[:, :3] < 3e-2).all() and (np.abs(got[:, 3] - want[:, 3]) < 0.06).all(), rep
    assert (win_rel < 4e-3).all(), rep          # measured: 1.1e-3 (cfg 3), 3.8e-4 (cfg 4), 1.5e-3 (cfg 5); runs are bit-reproducible
    assert abs(rep["end_d_bpp"]) < TOL_BPP and abs(rep["end_d_psnr"]) < TOL_PSNR, rep
    if bb:
        assert abs(rep["end_d_bpp_back"]) < TOL_BPP and rep["stage2_max_rel_all"] < 3e-2, rep
        assert np.allclose(m["est_bpp"], m["est_y_bpp"] + m["est_z_bpp"] - m["est_bpp_back"], rtol=1e-5, atol=1e-6)
        assert got2[-50:].mean() < got2[:50].mean()                  # stage 2 lowers the rate objective it optimises
    else:
        assert torch.equal(z_hat, torch.round(z_hat))
        again = metrics_to_dict(codec.evaluate(x, y_hat, z_hat))
        for k in ("mse", "psnr", "est_bpp", "est_y_bpp", "est_z_bpp"):
            assert np.array_equal(m[k], again[k]), k
        assert np.allclose(m["est_bpp"], m["est_y_bpp"] + m["est_z_bpp"], rtol=1e-6)
        assert np.isfinite(m["msssim"]).all() and (m["msssim"] <= 1).all()
    assert torch.equal(y_hat, torch.round(y_hat))
    codec.close()
