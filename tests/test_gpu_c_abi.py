"""The drop-in boundary is a C ABI: a plain-C program (tests/c_client/sga_client.c, compiled with gcc
against include/sga_hip.h, no Python and no torch in that process, device memory from hipMalloc) runs
the complete sga_run and must produce bit-for-bit what the Python host gets from the same library."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import sga_amd  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "improving-inference-for-neural-image-compression_amd")


def _write_weights(path, w):
    keys = []
    for pre in ("ga", "gs"):
        for k in range(4):
            keys += [f"{pre}.k{k}", f"{pre}.b{k}"]
            if k < 3:
                keys += [f"{pre}.beta{k}", f"{pre}.gamma{k}"]
    for k in range(3):
        keys += [f"ha.k{k}"] + ([f"ha.b{k}"] if k < 2 else [])
    for k in range(3):
        keys += [f"hs.k{k}", f"hs.b{k}"]
    for k in range(4):
        keys += [f"eb.m{k}", f"eb.b{k}"] + ([f"eb.f{k}"] if k < 3 else [])
    assert sorted(keys) == sorted(w), set(w) ^ set(keys)
    with open(path, "wb") as f:
        for k in keys:
            a = np.ascontiguousarray(w[k], np.float32)
            f.write(np.int64(a.size).tobytes())
            f.write(a.tobytes())


def test_plain_c_client_matches_python_host(tmp_path):
    from sga_amd.codec import SGACodec
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path / "sga_client")
    subprocess.run(["gcc", "-O2", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(rocm, "include"), os.path.join(ROOT, "tests", "c_client", "sga_client.c"),
                    "-o", exe, "-L", PKG, "-lsga_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                    f"-Wl,-rpath,{PKG}:{os.path.join(rocm, 'lib')}"], check=True)
    C, B, H, W, its, lmbda, seed = 64, 2, 80, 48, 25, 0.01, 12345
    w = sga_amd.make_synthetic_weights(C, seed=0)
    x = np.random.RandomState(5).rand(B, H, W, 3).astype(np.float32)
    _write_weights(tmp_path / "w.bin", w)
    x.tofile(tmp_path / "x.bin")
    env = dict(os.environ, SGA_PRECISION="f32")
    r = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "x.bin"), str(tmp_path / "out.bin"),
                        str(C), str(B), str(H), str(W), str(its), repr(lmbda), str(seed)],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    codec = SGACodec(w, C, B, H, W, precision="f32")
    y_hat, z_hat, met, _ = codec.run(x, lmbda, its=its, seed=seed)
    want = np.concatenate([met.cpu().numpy().ravel(), y_hat.cpu().numpy().ravel(), z_hat.cpu().numpy().ravel()])
    assert out.shape == want.shape
    nm = B * 7
    assert np.array_equal(out[nm:], want[nm:]), "latents from the C client differ from the Python host"
    assert np.allclose(out[:nm], want[:nm], rtol=1e-6, atol=0, equal_nan=True)
    codec.close()
