"""Host driver: the counterpart of the reference's `sga.py:compress` (sga.py:37-295) on top of
`SGACodec`.  Same inputs (.npy [N,H,W,3] uint8 or one image file), same batching rule
(configs.py:5-9), same hyper-parameters and flags (tf_boilerplate.py:155-176), same result file
(`rd-sga-lmbda=<l>+<runname>-input=<file>.npz` with the fields of sga.py:183).

Multi-GPU: one process per GPU; the images of each reference batch are dealt
round-robin to the ranks, every rank runs its shard with loss_scale = 1/len(reference batch) so
per-image gradients equal the un-sharded ones, and the only collective is the final gather of
the [N_local, 7] metrics (torch.distributed: RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

EVAL_FIELDS = ["mse", "psnr", "msssim", "msssim_db", "est_bpp", "est_y_bpp", "est_z_bpp"]  # sga.py:183
BB_EVAL_FIELDS = EVAL_FIELDS + ["est_bpp_back"]                                             # bb_sga.py:181

# configs.py:5
eval_batch_num_pixels = 1e7


def get_eval_batch_size(num_pixels_per_image):
    """configs.py:8-9."""
    return round(eval_batch_num_pixels / num_pixels_per_image)


def annealed_temperature(t, r, ub, lb=1e-8, scheme="exp0", t0=700):
    """utils.py:151-180 (numpy backend semantics, float64)."""
    if scheme == "exp":
        tau = math.exp(-r * t)
    elif scheme == "exp0":
        tau = ub * math.exp(-r * (t - t0))
    elif scheme == "linear":
        tau = -r * (t - t0) + ub
    else:
        raise NotImplementedError
    return min(max(tau, lb), ub)


def load_images(input_file: str) -> np.ndarray:
    """sga.py:41-53: .npy of N same-shape images [N,H,W,3], or one image file -> float32 in [0,1]."""
    if input_file.endswith(".npy"):
        X = np.load(input_file)
    else:
        from PIL import Image
        X = np.asarray(Image.open(input_file).convert("RGB"))[None, ...]
    if X.ndim != 4 or X.shape[-1] != 3:
        raise ValueError(f"expected [N,H,W,3] images, got {X.shape}")
    X = X.astype("float32")
    X /= 255.0
    return X


def lambda_from_runname(runname: str) -> float:
    """sga.py:157-158: re-use the lmbda the model was trained with."""
    return float(runname.split("lmbda=")[1].split("-")[0])


def result_filename(prefix, script_name, lmbda, runname, input_file):
    """sga.py:258-268."""
    input_file = os.path.basename(input_file)
    trained_script_name = runname.split("-")[0]
    if script_name != trained_script_name:
        return "%s-%s-lmbda=%g+%s-input=%s.npz" % (prefix, script_name, lmbda, runname, input_file)
    return "%s-%s-input=%s.npz" % (prefix, runname, input_file)


def reference_batches(num_images: int, batch_size: int):
    """tf.data `.batch(batch_size)` (sga.py:56-57): consecutive chunks, last one ragged."""
    return [list(range(s, min(s + batch_size, num_images))) for s in range(0, num_images, batch_size)]


def shard_batch(indices, rank: int, world: int, batch_no: int = 0):
    """Round-robin deal of one reference batch to the ranks.  The rank that gets the first image
    rotates with the batch number so that ragged batches (len % world != 0) do not always load
    the low ranks (Tecnick: 100 images in batches of 7 on 8 GPUs)."""
    return indices[(rank - batch_no) % world::world]


def plan_launches(num_images, batch_size, rank, world, seed, max_batch, pool=True):
    """The launches of one rank: lists of (image, position in its reference batch, batch seed, loss_scale, batch number).
    One noise stream per reference batch (seed + 1000003 * batch number); an image's noise is keyed on its position in
    that batch (sga_set_image_ids) and on that batch's seed (sga_set_image_seeds), so results do not depend on world
    size, chunking or pooling.  A launch holds consecutive work items with the same loss_scale = 1 / len(reference
    batch) (the batch means of sga.py:147,150), up to the workspace size; items of DIFFERENT reference batches pool
    into one launch (a rank that holds one Tecnick image of each 7-image batch then runs B > 1 per launch) unless
    pool=False: the early-stopping scripts decide on their batch objective (map.py:188, ste.py:189)."""
    work = []
    for b_i, batch in enumerate(reference_batches(num_images, batch_size)):
        for i in shard_batch(batch, rank, world, b_i):
            work.append((i, i - batch[0], seed + 1000003 * b_i, 1.0 / len(batch), b_i))
    launches = []
    for item in work:
        cur = launches[-1] if launches else None
        if cur and len(cur) < max_batch and cur[-1][3] == item[3] and (pool or cur[-1][4] == item[4]):
            cur.append(item)
        else:
            launches.append([item])
    return launches


def gather_metrics(local_idx, local_met, num_images, dist=None, device=None, nfields=None):
    """All-gather [n_local, F] metrics + their image indices; returns [num_images, F] on every
    rank.  dist=None: single process."""
    import torch
    nfields = nfields or len(EVAL_FIELDS)
    out = np.full((num_images, nfields), np.nan, np.float32)
    if dist is None or not dist.is_initialized():
        if len(local_idx):
            out[np.asarray(local_idx)] = local_met
        return out
    world = dist.get_world_size()
    n = len(local_idx)
    # same padded size on every rank: the largest local count (a rank's share is not bounded by
    # ceil(N / world): every ragged reference batch can give it one image more than the others)
    cnt = torch.tensor([n], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
    cap = max(int(cnt.item()), 1)
    buf = torch.full((cap, nfields + 1), -1.0, dtype=torch.float32)
    if n:
        buf[:n, 0] = torch.as_tensor(np.asarray(local_idx, np.float32))
        buf[:n, 1:] = torch.as_tensor(np.asarray(local_met, np.float32))
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    for p in parts:
        p = p.cpu().numpy()
        keep = p[:, 0] >= 0
        out[p[keep, 0].astype(np.int64)] = p[keep, 1:]
    return out


# hyper-parameters of the sibling scripts: (relaxation, schedule, lr, annealing_rate, T_ub, t0, early stop)
SIBLINGS = {
    "danneal": ("danneal", "exp", 0.005, 4e-3, 0.2, 0, False),    # danneal.py:183-193
    "unoise": ("unoise", "exp0", 0.005, 1e-3, 0.5, 700, False),    # unoise.py:76,150-151 (no temperature)
    "ste": ("ste", "exp0", 0.0001, 1e-3, 0.5, 700, True),          # ste.py:32,163-164
    "map": ("none", "exp0", 0.005, 1e-3, 0.5, 700, True),          # map.py:32,153-154
}


def run_early_stop(codec, x, lmbda, *, method, its=2000, lr, seed=0, loss_scale=None, medians=None,
                   check_itv=10, log=None):
    """The early-stopping loops of map.py:167-199 and ste.py:177-203 on the device-resident run:
    every `check_itv` iterations (it % 10 == 0, and the last one) look at the objective;
    map: the objective AFTER centred rounding (y_hat = round(y - mu) + mu, z_hat = round(z - med) +
    med, fed back without relaxation) must not get worse; ste: the training objective must improve.
    Otherwise return to the latents of the previous check and stop.
    Returns (y_hat, z_hat, metrics, iterations done)."""
    import torch
    B, H, W, _ = x.shape
    if loss_scale is None:
        loss_scale = 1.0 / B
    codec.run_begin(x, lmbda, its=its, lr=lr, seed=seed, loss_scale=loss_scale)
    prev, y_prev, z_prev, done = math.inf, None, None, 0
    # the reference checks after iteration index it when it % 10 == 0 or it + 1 == its
    checks = [i + 1 for i in range(its) if i % check_itv == 0 or i + 1 == its]
    y_cur, z_cur = codec.run_latents()
    it = 0
    for nxt in checks:
        codec.run_steps(nxt - it)
        it = nxt
        y_cur, z_cur, tr = codec.run_latents(trace=True)
        if method == "map":
            y_hat, z_hat = codec.quantize_centered(y_cur, z_cur, H, W, medians)
            obj = codec.step_grads(x, y_hat, z_hat, 1.0, lmbda, loss_scale=loss_scale)["rd_loss"]
            improved = obj <= prev                                # map.py:188
        else:
            obj = float(tr[it - 1, 0])                            # this iteration's training rd_loss
            improved = obj < prev                                 # ste.py:189
        if log:
            log("it=%d obj=%.4f" % (it - 1, obj))
        if improved:
            prev, y_prev, z_prev, done = obj, y_cur, z_cur, it
        else:
            y_cur, z_cur = y_prev, z_prev
            break
    if method == "map":
        y_hat, z_hat = codec.quantize_centered(y_cur, z_cur, H, W, medians)   # map.py:201
    else:
        y_hat, z_hat = torch.round(y_cur), torch.round(z_cur)                  # ste.py:201-202
    return y_hat, z_hat, codec.evaluate(x, y_hat, z_hat), done


def run_verbose(codec, x, lmbda, *, its, lr, annealing_rate, t0, T_ub, seed, loss_scale, log_itv=100, log=print, after=None,
                latents=None):
    """sga.py:210-238 with --verbose: at every log point also feed the ROUNDED latents straight into
    the graph (sga.py:219-225) and print both objectives.  The run pauses at the log points
    (sga_run_steps); the rounded latents are evaluated with the relaxation switched off."""
    import torch
    codec.run_begin(x, lmbda, its=its, lr=lr, annealing_rate=annealing_rate, t0=t0, T_ub=T_ub, seed=seed,
                    loss_scale=loss_scale)
    done = 0
    y = z = None
    for it in [i for i in range(its) if i % log_itv == 0 or i + 1 == its]:
        codec.run_steps(it + 1 - done)
        done = it + 1
        y, z, tr = codec.run_latents(trace=True)
        T = annealed_temperature(it, annealing_rate, T_ub, scheme="exp0", t0=t0)
        # sga.py:219-225 feeds y_tilde / z_tilde directly, i.e. past the sampler
        codec.set_relaxation("none", "exp0")
        try:
            r = codec.step_grads(x, torch.round(y), torch.round(z), T, lmbda, loss_scale=loss_scale)
        finally:
            codec.set_relaxation("sga", "exp0")
        t = tr[it].tolist()
        if after is not None:
            after.append(r["rd_loss"])                                      # sga.py:231 opt_record['rd_loss_after_rounding']
        log("it=%d, T=%.3f rd_loss=%.4f mse=%.3f bpp=%.4f psnr=%.4f\t after rounding: rd_loss=%.4f, bpp=%.4f psnr=%.4f"
            % (it, T, t[0], t[1], t[2], t[3], r["rd_loss"], r["train_bpp"], float(r["psnr"].mean())))
    if y is None:
        y, z = codec.run_latents()
    if latents is not None:                                                 # the rounded latents, for sga.py:281-291
        latents.append(torch.round(y))
    return codec.evaluate(x, torch.round(y), torch.round(z))                # sga.py:240-245


def run_dataset(codec, X, lmbda, *, its=2000, lr=0.005, annealing_rate=1e-3, t0=700, T_ub=0.5,
                seed=0, rank=0, world=1, dist=None, verbose=False, log_itv=100, log=print,
                method="sga", r_its=2000, r_lr=0.003, medians=None, base_scale_bound=None, check_finite=False,
                opt_record=None, recon=None, output_file=None):
    """The per-batch loop of sga.py:201-253 (method "sga"), bb_sga.py:199-280 ("bb_sga") or the
    one-shot mbt2018.py:159-180 ("mbt2018") over a dataset X [N,H,W,3] float32.
    Returns dict field -> [N] array (on every rank).
    check_finite: test the logged rows (every `log_itv` iterations and the last one: the points sga.py:216 prints) of every
      launch's per-iteration trace on the host and raise FloatingPointError naming the first non-finite iteration (SURVEY.md
      5; the reference lets a NaN run silently for 2000 steps).
    opt_record: a dict to fill with the optimisation record of this rank's LAST launch -- `its`, `T`, `rd_loss` at the log
      points, as sga.py:209,234-236 keeps them for the last batch (+ `rd_loss_after_rounding` with --verbose).  rd_loss is
      the objective of the images in that launch (loss_scale = 1 / len(reference batch)): the reference's value when the
      launch holds a whole reference batch.
    recon: a list to receive (image index, x_hat [H,W,3] float32) of this rank's images (sga.py:281-291).
    Method "mbt2018" also CODES every launch's centred latents (mbt2018.py:84-85,211-222) and returns what the reference's
      compress() adds to its npz (mbt2018.py:218-232): `batch_actual_bpp` (bits of the launch's stream / pixels of ONE image,
      as the reference computes it), `batch_sizes`, `avg_batch_actual_bpp`; output_file: the last stream is written there."""
    want_trace = verbose or check_finite or opt_record is not None
    fields = BB_EVAL_FIELDS if method == "bb_sga" else EVAL_FIELDS
    N, H, W, _ = X.shape
    bs = get_eval_batch_size(H * W)
    local_idx, local_met, coded = [], [], []
    if method in ("mbt2018", "map") and medians is None:
        medians = getattr(codec, "medians", None)       # from the checkpoint's `quantiles`
    log_sched, log_rate, log_Tub, log_t0 = "exp0", annealing_rate, T_ub, t0
    launches = plan_launches(N, bs, rank, world, seed, codec.max_batch, pool=method not in ("map", "ste"))
    try:
        for chunk in launches:
            idx = [c[0] for c in chunk]
            loss_scale, sd = chunk[0][3], chunk[0][2]
            codec.set_image_ids([c[1] for c in chunk])
            codec.set_image_seeds([c[2] for c in chunk])
            if method == "bb_sga":
                _, _, met, tr, _ = codec.bb_run(X[idx], lmbda, its=its, r_its=r_its, lr=lr, r_lr=r_lr,
                                                annealing_rate=annealing_rate, t0=t0, T_ub=T_ub,
                                                seed=sd, loss_scale=loss_scale, trace=want_trace)
            elif method == "mbt2018":
                kw = {} if base_scale_bound is None else dict(scale_bound=base_scale_bound)
                y_hat_b, z_hat_b, met = codec.base_compress(X[idx], medians=medians, **kw)      # default 0.11: mbt2018.py:80
                tr = None
                blob = codec.compress_latents((len(idx), H, W), y_hat_b, z_hat_b, centred=True, medians=medians)
                coded.append((idx[0], 8.0 * len(blob) / (H * W), len(idx)))      # mbt2018.py:218-221
                if output_file and rank == 0:
                    with open(output_file, "wb") as f:                            # mbt2018.py:214-216 (overwritten per batch)
                        f.write(blob)
            elif method in SIBLINGS:
                relax, sched, s_lr, s_r, s_Tub, s_t0, early = SIBLINGS[method]
                log_sched, log_rate, log_Tub, log_t0 = sched, s_r, s_Tub, s_t0
                codec.set_relaxation(relax, sched)
                try:
                    if early:
                        _, _, met, _ = run_early_stop(codec, X[idx], lmbda, method=method, its=its, lr=s_lr,
                                                      seed=sd, loss_scale=loss_scale, medians=medians,
                                                      log=log if verbose else None)
                        tr = None
                    else:
                        _, _, met, tr = codec.run(X[idx], lmbda, its=its, lr=s_lr, annealing_rate=s_r,
                                                  t0=s_t0, T_ub=s_Tub, seed=sd, loss_scale=loss_scale,
                                                  trace=want_trace)
                finally:
                    codec.set_relaxation("sga", "exp0")
            elif verbose:
                after = [] if opt_record is not None else None
                lat = [] if recon is not None else None
                met = run_verbose(codec, X[idx], lmbda, its=its, lr=lr, annealing_rate=annealing_rate, t0=t0,
                                  T_ub=T_ub, seed=sd, loss_scale=loss_scale, log_itv=log_itv, log=log, after=after, latents=lat)
                tr = codec.run_latents(trace=True)[2] if want_trace else None
                if recon is not None:                                       # sga.py:281-291 writes it regardless of --verbose
                    for k, xh in zip(idx, codec.reconstruct(lat[0], H, W).cpu().numpy()):
                        recon.append((k, xh))
            else:
                y_hat_l, z_hat_l, met, tr = codec.run(X[idx], lmbda, its=its, lr=lr, annealing_rate=annealing_rate,
                                                      t0=t0, T_ub=T_ub, seed=sd, loss_scale=loss_scale,
                                                      trace=want_trace)
                after = None
                if output_file and rank == 0:      # beyond sga.py (which stops at the estimate): the run's integer latents as a
                    with open(output_file, "wb") as f:      # stream `driver decompress` reads (last launch wins, as mbt2018.py:214-216)
                        f.write(codec.compress_latents((len(idx), H, W), y_hat_l, z_hat_l))
                if recon is not None:
                    for k, xh in zip(idx, codec.reconstruct(y_hat_l, H, W).cpu().numpy()):
                        recon.append((k, xh))
            if tr is not None:
                tr = tr.cpu().numpy()
                pts = [it for it in range(min(its, len(tr))) if it % log_itv == 0 or it + 1 == its]      # sga.py:216,232-233
                if verbose and not (method == "sga"):
                    for it in pts:
                        T = annealed_temperature(it, log_rate, log_Tub, scheme=log_sched, t0=log_t0)
                        log("it=%d, T=%.3f rd_loss=%.4f mse=%.3f bpp=%.4f psnr=%.4f" %
                            (it, T, tr[it, 0], tr[it, 1], tr[it, 2], tr[it, 3]))
                if check_finite:
                    for it in pts:
                        if not np.isfinite(tr[it, :3]).all():
                            raise FloatingPointError(
                                "non-finite objective at iteration %d of the launch holding images %s: rd_loss=%r mse=%r bpp=%r"
                                % (it, idx, tr[it, 0], tr[it, 1], tr[it, 2]))
                if opt_record is not None:
                    opt_record.clear()
                    opt_record.update(its=np.asarray(pts), rd_loss=tr[pts, 0].copy(),
                                      T=np.asarray([annealed_temperature(it, log_rate, log_Tub, scheme=log_sched, t0=log_t0)
                                                    for it in pts]))
                    if method == "sga" and verbose and after is not None:
                        opt_record["rd_loss_after_rounding"] = np.asarray(after)
            local_idx += idx
            local_met.append(met.cpu().numpy())
    finally:
        # an exception in a run must not leave the handle drawing noise for stale batch positions
        codec.set_image_ids(None)
        codec.set_image_seeds(None)
    local_met = np.concatenate(local_met, 0) if local_met else np.zeros((0, len(fields)), np.float32)
    device = codec.device if (dist is not None and dist.is_initialized()
                              and dist.get_backend() == "nccl") else None
    allm = gather_metrics(local_idx, local_met, N, dist, device, len(fields))
    res = {k: allm[:, i].copy() for i, k in enumerate(fields)}
    if method == "mbt2018":
        if dist is not None and dist.is_initialized() and world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, coded)
            coded = [c for part in parts for c in part]
        coded.sort()
        res["batch_actual_bpp"] = np.asarray([c[1] for c in coded])
        res["batch_sizes"] = np.asarray([c[2] for c in coded])
        res["avg_batch_actual_bpp"] = np.asarray(res["batch_actual_bpp"].sum() / max(res["batch_sizes"].sum(), 1))      # mbt2018.py:229
    return res


def parse_args(argv):
    """The reference's flags for `compress` (tf_boilerplate.py:91-204) + build-specific ones."""
    p = argparse.ArgumentParser(description="SGA iterative inference on MI355X (sga.py drop-in)")
    p.add_argument("--verbose", "-V", action="store_true")
    p.add_argument("--num_filters", type=int, default=-1)
    p.add_argument("--num_hfilters", type=int, default=-1,
                   help="accepted like the reference's parser does; only its training run names use it (utils.py:65)")
    p.add_argument("--checkpoint_dir", default="./checkpoints")
    p.add_argument("--seed", type=int, default=0)
    sub = p.add_subparsers(dest="command")
    c = sub.add_parser("compress")
    c.add_argument("--results_dir", default="./results")
    c.add_argument("--lambda", type=float, default=-1, dest="lmbda")
    c.add_argument("--sga_its", type=int, default=2000,
                   help="number of SGA iterations.  DELIBERATE DEVIATION: the reference parses this flag and then ignores "
                        "it (sga.py:191-192 hard-codes 2000); here it is honoured -- leave it at 2000 to mirror the reference")
    c.add_argument("--annealing_rate", type=float, default=1e-3)
    c.add_argument("--t0", type=int, default=700)
    c.add_argument("--method", default="sga", choices=["sga", "bb_sga", "mbt2018", "danneal", "unoise", "ste", "map"],
                   help="which reference script to mirror: sga.py, bb_sga.py or mbt2018.py compress")
    c.add_argument("--synthetic_weights", action="store_true",
                   help="use the deterministic synthetic parameters instead of a checkpoint")
    c.add_argument("--max_batch", type=int, default=0, help="images per GPU launch (0 = reference batch)")
    c.add_argument("--scale_bound", type=float, default=None,
                   help="lower bound on the conditional's sigma; default: what the mirrored script executes -- none (0) "
                        "for sga.py and its siblings, which never build the tfc GaussianConditional layer "
                        "(sga.py:130-133), 0.11 for mbt2018.py, which calls it (mbt2018.py:77-80)")
    c.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16x2"],
                   help="arithmetic of the conv contractions (DESIGN.md 3.1b); f32 = fp32 MFMA")
    c.add_argument("--check_finite", action="store_true",
                   help="test the logged objective (every 100 iterations and the last) of every launch and abort with the "
                        "iteration index at the first NaN / Inf (the reference runs on silently)")
    c.add_argument("--save_opt_record", action="store_true",
                   help="also write opt-<...>.npz: its / T / rd_loss at the log points of the last batch (configs.py:12, "
                        "sga.py:209,234-236,271-278; a module constant in the reference, a flag here)")
    c.add_argument("--save_reconstruction", action="store_true",
                   help="write recon-<...>.png of the (single) input image (configs.py:13, sga.py:281-291)")
    c.add_argument("runname")
    c.add_argument("input_file")
    c.add_argument("output_file", nargs="?")
    d = sub.add_parser("decompress", description="reads a stream written by `compress --method mbt2018 ... output_file` (or by "
                                                 "SGACodec.compress_latents), reconstructs the image(s) and writes PNG "
                                                 "(mbt2018.py:248-295, tf_boilerplate.py:178-197)")
    d.add_argument("--synthetic_weights", action="store_true")
    d.add_argument("runname")
    d.add_argument("input_file")
    d.add_argument("output_file", nargs="?", help="default: input_file + '.png' (tf_boilerplate.py:194-197); image k > 0 of a "
                                                  "batch stream goes to <output minus .png>.<k>.png")
    args = p.parse_args(argv)
    return args


def compress(args, weights=None):
    """sga.py:37-295."""
    import torch
    from .codec import SGACodec
    from .weights import make_synthetic_weights
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one process per GPU, launched by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment); at
    # world size 1 the same path runs (RCCL group of one, device all_gather) when the launcher set the variables
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend=os.environ.get("SGA_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}")
                                    if os.environ.get("SGA_DIST_BACKEND", "nccl") == "nccl" else None)
    X = load_images(args.input_file)
    N, H, W, _ = X.shape
    if args.lmbda < 0:
        args.lmbda = lambda_from_runname(args.runname)
        if rank == 0:
            print("Defaulting lmbda (mse coefficient) to %g as used in model training." % args.lmbda)
    method = getattr(args, "method", "sga")
    bb = method == "bb_sga"
    if weights is None:
        if args.synthetic_weights:
            weights = make_synthetic_weights(args.num_filters, seed=0, bb=bb)
        else:
            from .tf_checkpoint import load_effective_weights
            weights = load_effective_weights(os.path.join(args.checkpoint_dir, args.runname),
                                             args.num_filters, bb=bb)
            if method in ("mbt2018", "map") and "eb.medians" not in weights:
                raise SystemExit(f"--method {method} rounds z around the prior's medians (mbt2018.py:69, "
                                 "map.py:83) but the checkpoint has no entropy_bottleneck/quantiles")
    bs = get_eval_batch_size(H * W)
    per_rank = -(-min(bs, N) // world)
    # a rank's share of one reference batch can be a single large image (Tecnick: batches of 7 on 8 GPUs): let its
    # launches pool images of consecutive batches up to ~6 Mpixel (4 Tecnick images: 0.78 instead of 0.73 of the roofline)
    pool = min(-(-N // world), max(1, int(6e6 // (H * W))))
    max_batch = args.max_batch or max(per_rank, pool)
    sb = getattr(args, "scale_bound", None)
    codec = SGACodec(weights, args.num_filters, max_batch, H, W, device=f"cuda:{local_rank}",
                     bits_back=bb, precision=getattr(args, "precision", "f32"),
                     scale_bound=0.0 if sb is None else sb)
    opt_record = {} if getattr(args, "save_opt_record", False) else None
    recon = [] if getattr(args, "save_reconstruction", False) else None
    if recon is not None:
        if not (N == 1 and method == "sga"):      # (not an assert: it must survive python -O)
            raise ValueError("--save_reconstruction: one image, --method sga (sga.py:282)")
    res = run_dataset(codec, X, args.lmbda, its=args.sga_its, annealing_rate=args.annealing_rate,
                      t0=args.t0, seed=args.seed, rank=rank, world=world, dist=dist,
                      verbose=args.verbose, method=method, base_scale_bound=sb,
                      check_finite=getattr(args, "check_finite", False), opt_record=opt_record, recon=recon,
                      output_file=getattr(args, "output_file", None) if method in ("mbt2018", "sga") else None)
    if rank == 0:
        if args.results_dir:
            os.makedirs(args.results_dir, exist_ok=True)
            f = result_filename("rd", method, args.lmbda, args.runname, args.input_file)
            np.savez(os.path.join(args.results_dir, f), **res)
            if opt_record:                                                   # sga.py:271-278
                np.savez(os.path.join(args.results_dir, result_filename("opt", method, args.lmbda, args.runname,
                                                                        args.input_file)), **opt_record)
            if recon:                                                        # sga.py:281-291
                from PIL import Image
                f = result_filename("recon", method, args.lmbda, args.runname, args.input_file)[:-4] + ".png"
                if method != args.runname.split("-")[0]:
                    f = f.replace("+" + args.runname, "-rd_opt_its=%d+%s" % (args.sga_its, args.runname))
                img = np.round(np.clip(recon[0][1], 0.0, 1.0) * 255.0).astype(np.uint8)
                print("Saving image reconstruction to ", os.path.join(args.results_dir, f))
                Image.fromarray(img).save(os.path.join(args.results_dir, f))
        for field in res:
            if field not in ("batch_actual_bpp", "batch_sizes"):            # mbt2018.py:230-245 prints the average only
                print("Avg {}: {:0.4f}".format(field, res[field].mean()))   # sga.py:293-295
    return res


def decompress(args, weights=None):
    """mbt2018.py:248-295: stream -> z_hat -> (mu, sigma) = h_s(z_hat) -> y_hat -> g_s -> crop -> PNG.  The stream's mode byte says
    whether it holds the centred latents of `compress --method mbt2018` or the integer latents of an SGA run."""
    from PIL import Image
    from .codec import SGACodec
    from .weights import make_synthetic_weights
    from . import entropy_coding as ec
    with open(args.input_file, "rb") as f:
        blob = f.read()
    (B, H, W), mode = ec.unpack(blob)[0], ec.unpack(blob, with_tables=True)[5]
    if weights is None:
        if args.synthetic_weights:
            weights = make_synthetic_weights(args.num_filters, seed=0)
        else:
            from .tf_checkpoint import load_effective_weights
            weights = load_effective_weights(os.path.join(args.checkpoint_dir, args.runname), args.num_filters)
    # the decoder's h_s must run in the arithmetic the encoder's ran in: the stream says which (entropy_coding.MODE_PRECISIONS)
    codec = SGACodec(weights, args.num_filters, B, H, W, device="cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")),
                     precision=ec.mode_precision(mode))
    try:
        _, y_hat, _ = codec.decompress_latents(blob)
        x_hat = codec.reconstruct(y_hat, H, W).cpu().numpy()
    finally:
        codec.close()
    out = args.output_file or args.input_file + ".png"
    stem = out[:-4] if out.lower().endswith(".png") else out
    for k in range(B):
        img = np.round(np.clip(x_hat[k], 0.0, 1.0) * 255.0).astype(np.uint8)
        Image.fromarray(img).save(out if k == 0 else "%s.%d.png" % (stem, k))
    return x_hat


def main(argv=None):
    args = parse_args(sys.argv[1:] if argv is None else argv)
    if args.command not in ("compress", "decompress"):                      # sga.py:303
        raise ValueError("Only compress / decompress are supported.")
    if args.num_filters <= 0:
        raise SystemExit("--num_filters is required")
    return compress(args) if args.command == "compress" else decompress(args)


if __name__ == "__main__":
    main()
