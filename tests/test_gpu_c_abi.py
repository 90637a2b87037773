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


def test_error_codes_of_the_round3_entry_points():
    """Bad arguments come back as negative sga_status codes, never as a crash or a silent no-op (include/sga_hip.h:
    "return 0 on success, a negative sga_status on error")."""
    import ctypes as C
    from sga_amd import _lib
    from sga_amd.codec import SGACodec
    w = sga_amd.make_synthetic_weights(64, seed=0)
    codec = SGACodec(w, 64, 2, 64, 64)
    lib, h = codec.lib, codec.handle
    BAD_ARG, UNSUPPORTED = -1, -3
    assert lib.sga_set_scale_bound(h, C.c_float(-0.5)) == BAD_ARG
    assert lib.sga_set_scale_bound(h, C.c_float(float("nan"))) == BAD_ARG
    assert lib.sga_set_scale_bound(None, C.c_float(0.11)) == BAD_ARG
    assert lib.sga_set_scale_bound(h, C.c_float(0.11)) == 0 and lib.sga_set_scale_bound(h, C.c_float(0.0)) == 0
    seeds = (C.c_uint64 * 3)(1, 2, 3)
    assert lib.sga_set_image_seeds(h, seeds, 3) == BAD_ARG            # more than max_batch
    assert lib.sga_set_image_seeds(h, None, 1) == BAD_ARG
    assert lib.sga_set_image_seeds(h, seeds, 2) == 0 and lib.sga_set_image_seeds(h, None, 0) == 0
    z = torch.zeros(2, 1, 1, 128, device="cuda")
    y = torch.zeros(2, 4, 4, 64, device="cuda")
    assert lib.sga_bb_refine(h, C.c_void_p(y.data_ptr()), 2, 64, 64, C.c_float(0.5), 5, C.c_double(0.003), 0,
                             C.c_void_p(z.data_ptr()), None) == UNSUPPORTED      # not a bits-back handle
    assert lib.sga_profile_graph_begin(h, b"") == BAD_ARG
    i32 = torch.zeros(16, dtype=torch.int32, device="cuda")
    u8 = torch.zeros(16 * 64, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.sga_ec_encode(None, p(i32), 16, 1024, p(i32), p(i32), p(i32), 4, p(u8), 64, p(i32), None) == BAD_ARG
    assert lib.sga_ec_encode(p(i32), p(i32), 16, 1024, p(i32), p(i32), p(i32), 4, p(u8), 8, p(i32), None) == BAD_ARG   # slot too small
    assert lib.sga_ec_decode(p(u8), p(i32), p(i32), 3, p(i32), 16, 1024, p(i32), p(i32), p(i32), 4, p(i32), p(i32), None) == BAD_ARG
    cfg = _lib.SgaConfig(64, 1, 64, 64, 0, 0, -1.0, 0)                # negative bound in the config
    hh = C.c_void_p(0)
    assert lib.sga_create(C.byref(hh), C.byref(cfg), C.byref(_lib.SgaWeights())) == BAD_ARG
    codec.close()
