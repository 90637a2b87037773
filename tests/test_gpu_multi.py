"""Multi-process / sharding first contact on ONE GPU (sga.py:147,150 are the only
cross-image ops).  An 8-GPU node is not available to the tests, so the distributed host path runs
here with the REAL `SGACodec`:

* image ids: an image's Philox noise is keyed on its position in the reference batch, so any
  grouping of the batch into chunks gives the same per-image result;
* 2 ranks sharing cuda:0 under gloo through `driver.run_dataset` == the single-process run, bit for bit;
* `bench.py --gpus 2` (2 ranks sharing cuda:0, gloo) and `bench.py` through RCCL at world size 1
  (backend "nccl": init_process_group + device all_gather) == plain `bench.py`, bit for bit.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, H, W, ITS = 64, 64, 64, 25


def test_image_ids_make_results_independent_of_the_grouping():
    from sga_amd.codec import SGACodec
    w = sga_amd.make_synthetic_weights(C, seed=0)
    codec = SGACodec(w, C, 2, H, W)
    x = np.random.RandomState(11).rand(4, H, W, 3).astype(np.float32)
    res = {}
    for name, groups in (("a", [[0, 1], [2, 3]]), ("b", [[0, 2], [3, 1]])):
        out = {}
        for g in groups:
            codec.set_image_ids(g)
            y_hat, z_hat, met, _ = codec.run(x[g], 0.01, its=ITS, seed=4, loss_scale=0.25)
            for k, i in enumerate(g):
                out[i] = (y_hat[k].cpu().numpy(), z_hat[k].cpu().numpy(), met[k].cpu().numpy())
        res[name] = out
    for i in range(4):
        for a, b in zip(res["a"][i], res["b"][i]):
            assert np.array_equal(a, b, equal_nan=True), i
    # and the ids matter: position 0 and position 1 draw different noise
    codec.set_image_ids([1])
    y1, _, _, _ = codec.run(x[:1], 0.01, its=ITS, seed=4, loss_scale=0.25)
    codec.set_image_ids(None)
    assert not np.array_equal(y1[0].cpu().numpy(), res["a"][0][0])
    codec.close()


def _rank_main(rank, world, port, xfile, outdir, bs):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPU_MAX_HW_QUEUES="2")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import sga_amd as pkg
    from sga_amd import driver as d
    from sga_amd.codec import SGACodec
    d.eval_batch_num_pixels = bs * H * W
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = np.load(xfile)
    codec = SGACodec(pkg.make_synthetic_weights(C, seed=0), C, 1, H, W, device="cuda:0")
    res = d.run_dataset(codec, X, 0.01, its=ITS, seed=3, rank=rank, world=world, dist=dist)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()
    codec.close()


def test_two_ranks_on_one_device_equal_single_process(tmp_path, monkeypatch):
    """driver.run_dataset with the real codec: 2 processes sharing cuda:0 (gloo) vs one process."""
    import torch.multiprocessing as mp
    from sga_amd import driver
    from sga_amd.codec import SGACodec
    bs = 3
    monkeypatch.setattr(driver, "eval_batch_num_pixels", bs * H * W)
    X = np.random.RandomState(5).rand(5, H, W, 3).astype(np.float32)      # reference batches 3 + 2
    np.save(tmp_path / "x.npy", X)
    codec = SGACodec(sga_amd.make_synthetic_weights(C, seed=0), C, 1, H, W)
    single = driver.run_dataset(codec, X, 0.01, its=ITS, seed=3)
    codec.close()
    assert np.isfinite(single["est_bpp"]).all() and np.isfinite(single["psnr"]).all()
    port = 29700 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, str(tmp_path / "x.npy"), str(tmp_path), bs))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npz")
        for k in driver.EVAL_FIELDS:
            assert np.array_equal(got[k], single[k], equal_nan=True), (r, k, got[k], single[k])


BENCH = ["bench.py", "--steps", "1", "--warmup", "0", "--batch", "2", "--size", "64", "--num_filters", "64",
         "--its", "20", "--no-cpu-baseline", "--no-kernel-profile"]


def _run(cmd, env_extra, timeout=900):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2", **env_extra)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_and_rccl_world1_equal_plain_bench(tmp_path):
    plain = _run([sys.executable, *BENCH, "--gpus", "1", "--dump-metrics", str(tmp_path / "m1.npy")], {})
    m1 = np.load(tmp_path / "m1.npy")
    assert plain["n_gpus"] == 1 and plain["value"] > 0 and m1.shape == (2, 7)
    # RCCL path at world size 1: backend "nccl", init_process_group + all_gather of device tensors
    rccl = _run([sys.executable, *BENCH, "--gpus", "1", "--dump-metrics", str(tmp_path / "m1r.npy")],
                dict(SGA_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90),
                     RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert rccl["n_gpus"] == 1 and np.array_equal(np.load(tmp_path / "m1r.npy"), m1, equal_nan=True)
    # the driver's N>1 launch line, two ranks sharing cuda:0 over gloo
    port = 29950 + os.getpid() % 40
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                "--master-addr", "127.0.0.1", "--master-port", str(port), *BENCH, "--gpus", "2",
                "--dump-metrics", str(tmp_path / "m2.npy")],
               dict(SGA_BENCH_BACKEND="gloo", SGA_BENCH_SHARE_DEVICE="1"))
    m2 = np.load(tmp_path / "m2.npy")
    assert two["n_gpus"] == 2 and two["config"]["images_per_step"] == 4 and m2.shape == (4, 7)
    assert np.array_equal(m2[:2], m1, equal_nan=True)      # rank 0's images: same inputs, same result
    assert two["scaling"] == "weak" and two["value"] > 0


def test_bench_launch_line_at_world_size_8(tmp_path):
    """The driver's N = 8 line (`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`) with all
    eight ranks on the one GPU (gloo): rendezvous, the [8 x B, 7] gather, the max-over-ranks timing, and rank r's rows
    equal what a single process computes for rank r's inputs (x is seeded with 1000 + rank)."""
    port = 29800 + os.getpid() % 90
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                  "--master-addr", "127.0.0.1", "--master-port", str(port), *BENCH, "--gpus", "8",
                  "--dump-metrics", str(tmp_path / "m8.npy")],
                 dict(SGA_BENCH_BACKEND="gloo", SGA_BENCH_SHARE_DEVICE="1"))
    m8 = np.load(tmp_path / "m8.npy")
    assert eight["n_gpus"] == 8 and eight["config"]["images_per_step"] == 16 and m8.shape == (16, 7)
    assert eight["scaling"] == "weak" and eight["value"] > 0 and np.isfinite(m8[:, [1, 4]]).all()
    plain = _run([sys.executable, *BENCH, "--gpus", "1", "--dump-metrics", str(tmp_path / "m1.npy")], {})
    assert plain["n_gpus"] == 1
    assert np.array_equal(m8[:2], np.load(tmp_path / "m1.npy"), equal_nan=True)
    assert not np.array_equal(m8[2:4], m8[:2])          # other ranks work on other images


def test_pooled_launch_keeps_each_images_own_batch_noise():
    """sga_set_image_seeds: one launch holding images of two DIFFERENT reference batches (Tecnick on 8 GPUs: a rank has
    one image of each 7-image batch) gives each image the noise of its own batch -- image A (batch 0, position 2,
    seed s0) pooled with image B (batch 1, position 0, seed s1) ends exactly where it ends when it shares the launch
    with another image of batch 0 under the plain run seed s0, and likewise for B."""
    from sga_amd.codec import SGACodec
    codec = SGACodec(sga_amd.make_synthetic_weights(C, seed=0), C, 2, H, W)
    x = np.random.RandomState(21).rand(4, H, W, 3).astype(np.float32)       # A, A2 (batch 0), B2, B (batch 1)
    s0, s1 = 3, 3 + 1000003
    kw = dict(its=ITS, loss_scale=1.0 / 7)
    codec.set_image_ids([2, 5]); codec.set_image_seeds(None)
    ref_a = codec.run(x[[0, 1]], 0.01, seed=s0, **kw)                      # A in slot 0, its batch-mate beside it
    codec.set_image_ids([4, 0])
    ref_b = codec.run(x[[2, 3]], 0.01, seed=s1, **kw)                      # B in slot 1
    codec.set_image_ids([2, 0]); codec.set_image_seeds([s0, s1])
    pooled = codec.run(x[[0, 3]], 0.01, seed=12345, **kw)                  # the run seed is not used for keyed images
    assert torch.equal(pooled[0][0], ref_a[0][0]) and torch.equal(pooled[1][0], ref_a[1][0])
    assert torch.equal(pooled[0][1], ref_b[0][1]) and torch.equal(pooled[1][1], ref_b[1][1])
    assert torch.equal(pooled[2][:, [0, 1, 4, 5, 6]], torch.stack([ref_a[2][0], ref_b[2][1]])[:, [0, 1, 4, 5, 6]])
    codec.set_image_seeds([s0, s0])                                        # B under the wrong batch's seed: other noise
    wrong = codec.run(x[[0, 3]], 0.01, seed=12345, **kw)
    assert torch.equal(wrong[0][0], ref_a[0][0]) and not torch.equal(wrong[0][1], ref_b[0][1])
    codec.set_image_ids(None); codec.set_image_seeds(None)
    codec.close()


def test_driver_compress_under_rccl_at_world_size_1(tmp_path):
    """`python -m sga_amd.driver ... compress` (not only bench.py) as torch.distributed.run starts it: the launcher's
    RANK / WORLD_SIZE / MASTER_* make `compress` call init_process_group("nccl") and gather the metrics with a device
    all_gather (RCCL) -- at world size 1, the only size a 1-GPU box can run; results equal the plain command's."""
    X = (np.random.RandomState(9).rand(5, 48, 64, 3) * 255).astype(np.uint8)
    np.save(tmp_path / "x.npy", X)
    runname = "mbt2018-num_filters=64-lmbda=0.02"
    base = [sys.executable, "-m", "sga_amd.driver", "--num_filters", "64", "compress", "--sga_its", "15", "--t0", "4",
            "--synthetic_weights", "--max_batch", "2", runname, str(tmp_path / "x.npy")]
    outs = {}
    for tag, env in (("plain", {}), ("rccl", dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                                  MASTER_PORT=str(29600 + os.getpid() % 90)))):
        out = tmp_path / tag
        cmd = base[:6] + ["--results_dir", str(out)] + base[6:]
        p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, GPU_MAX_HW_QUEUES="2", **env), capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        files = os.listdir(out)
        assert len(files) == 1
        outs[tag] = dict(np.load(out / files[0]))
        assert "Avg est_bpp" in p.stdout
    for k in outs["plain"]:
        assert outs["plain"][k].shape == (5,)
        assert np.array_equal(outs["plain"][k], outs["rccl"][k], equal_nan=True), k
