"""CPU tests of the TF tensor-bundle importer.  No TensorFlow and no real checkpoint exist
offline, so a minimal writer of the same container format (LevelDB table + BundleEntryProto,
uncompressed and snappy blocks) produces the fixture, with variable names as tfc 1.3 / Keras
scope them; the reader + the inverse reparameterisations must round-trip the effective weights."""
import os
import struct

import numpy as np
import pytest

import sga_amd
from sga_amd import tf_checkpoint as tfc

MAGIC = 0xDB4775248B80FB57


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _entry_proto(shape, offset, size):
    dims = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in shape))
    return (b"\x08" + _vi(1) + b"\x12" + _vi(len(dims)) + dims + b"\x20" + _vi(offset) +
            b"\x28" + _vi(size) + b"\x35" + struct.pack("<I", 0))


def _block(items, restart_interval=4):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", max(len(restarts), 1))
    return bytes(out)


def _snappy_literal(data):     # valid snappy stream made only of literals
    out = bytearray(_vi(len(data)))
    for i in range(0, len(data), 60):
        chunk = data[i:i + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
    return bytes(out)


def write_bundle(prefix, tensors, snappy=False, per_block=5):
    names = sorted(tensors)
    data = bytearray()
    items = [(b"", b"\x08\x01")]          # header: num_shards = 1
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype="<f4")
        items.append((n.encode(), _entry_proto(a.shape, len(data), a.nbytes)))
        data += a.tobytes()
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    table, index_items = bytearray(), []
    for i in range(0, len(items), per_block):
        chunk = items[i:i + per_block]
        blk = _block(chunk)
        payload = _snappy_literal(blk) if snappy else blk
        index_items.append((chunk[-1][0] + b"\xff", _vi(len(table)) + _vi(len(payload))))
        table += payload + bytes([1 if snappy else 0]) + b"\0\0\0\0"
    meta = _block([])
    meta_h = _vi(len(table)) + _vi(len(meta)); table += meta + b"\0" + b"\0\0\0\0"
    idx = _block(index_items, restart_interval=1)
    idx_h = _vi(len(table)) + _vi(len(idx)); table += idx + b"\0" + b"\0\0\0\0"
    footer = meta_h + idx_h
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(table) + footer)


def raw_variables(w, C, bb=False):
    """Invert the reparameterisations: effective weights -> checkpoint variables (tfc 1.3 names)."""
    ped = 2.0 ** -36
    scope = {"ga": "analysis_transform", "gs": "synthesis_transform",
             "ha": "hyper_analysis_transform", "hs": "mbt2018_hyper_synthesis_transform"}
    t = {}
    for p, n in (("ga", 4), ("gs", 4), ("ha", 3), ("hs", 3)):
        for i in range(n):
            k = w[f"{p}.k{i}"]
            kh, kw, ci, co = k.shape
            if p == "hs":
                t[f"{scope[p]}/layer_{i}/kernel"] = k
            else:
                M = tfc.irdft_matrix((kh, kw)).astype(np.float64)
                t[f"{scope[p]}/layer_{i}/kernel_rdft"] = (M.T @ k.reshape(kh * kw, ci * co)).astype(np.float32)
            if f"{p}.b{i}" in w:
                t[f"{scope[p]}/layer_{i}/bias"] = w[f"{p}.b{i}"]
    for p, g in (("ga", "gdn"), ("gs", "igdn")):
        for i in range(3):
            t[f"{scope[p]}/layer_{i}/{g}_{i}/reparam_beta"] = np.sqrt(w[f"{p}.beta{i}"].astype(np.float64) + ped)
            t[f"{scope[p]}/layer_{i}/{g}_{i}/reparam_gamma"] = np.sqrt(w[f"{p}.gamma{i}"].astype(np.float64) + ped)
    for k in range(4):
        t[f"entropy_bottleneck/matrix_{k}"] = np.log(np.expm1(w[f"eb.m{k}"].astype(np.float64)))
        t[f"entropy_bottleneck/bias_{k}"] = w[f"eb.b{k}"]
        if k < 3:
            t[f"entropy_bottleneck/factor_{k}"] = np.arctanh(w[f"eb.f{k}"].astype(np.float64))
    med = w.get("eb.medians", np.zeros(C, np.float32))
    t["entropy_bottleneck/quantiles"] = np.stack([med - 9.0, med, med + 9.0], -1).reshape(C, 1, 3).astype(np.float32)
    t["analysis_transform/layer_0/bias/Adam"] = np.zeros(C, np.float32)      # optimizer slot: ignored
    return t


def test_irdft_matrix_is_orthonormal():
    for shp in ((5, 5), (3, 3), (4, 4)):
        M = tfc.irdft_matrix(shp).astype(np.float64)
        assert np.allclose(M @ M.T, np.eye(M.shape[0]), atol=1e-6)


def test_snappy_decoder():
    data = bytes(range(256)) * 3
    assert tfc.snappy_decompress(_snappy_literal(data)) == data
    # copy with 2-byte offset: "abcd" then copy 8 bytes from offset 4 (overlapping run)
    stream = _vi(12) + bytes([(4 - 1) << 2]) + b"abcd" + bytes([((8 - 1) << 2) | 2]) + struct.pack("<H", 4)
    assert tfc.snappy_decompress(stream) == b"abcd" * 3


@pytest.mark.parametrize("snappy", [False, True])
def test_bundle_round_trip(tmp_path, snappy):
    C = 64
    w = sga_amd.make_synthetic_weights(C, seed=3)
    d = tmp_path / "mbt2018-num_filters=64-lmbda=0.01"
    d.mkdir()
    prefix = str(d / "model.ckpt-2000000")
    write_bundle(prefix, raw_variables(w, C), snappy=snappy)
    (d / "checkpoint").write_text('model_checkpoint_path: "model.ckpt-2000000"\n')
    assert tfc.latest_checkpoint(str(d)) == prefix
    idx = tfc.read_index(prefix + ".index")
    assert "analysis_transform/layer_0/kernel_rdft" in idx and "" not in idx
    assert idx["synthesis_transform/layer_3/kernel_rdft"]["shape"] == [25, C * 3]
    got = tfc.load_effective_weights(str(d), C)
    for k, v in w.items():
        assert got[k].shape == v.shape and got[k].dtype == np.float32
        assert np.allclose(got[k], v, rtol=2e-5, atol=2e-6), k


def test_missing_variable_is_a_clear_error(tmp_path):
    C = 64
    t = raw_variables(sga_amd.make_synthetic_weights(C, seed=3), C)
    del t["synthesis_transform/layer_1/igdn_1/reparam_gamma"]
    with pytest.raises(KeyError, match="reparam_gamma"):
        tfc.effective_weights_from_tensors({k: np.asarray(v, np.float32) for k, v in t.items()}, C)


def test_npz_weights(tmp_path):
    C = 64
    w = sga_amd.make_synthetic_weights(C, seed=1)
    np.savez(tmp_path / "w.npz", **w)
    got = tfc.load_effective_weights(str(tmp_path / "w.npz"), C)
    assert all(np.array_equal(got[k], w[k]) for k in w)


def test_medians_come_from_quantiles():
    """tfc 1.3: medians = quantiles[:, 0, 1]; mbt2018.py:69 / map.py:83 round z around them."""
    C = 64
    w = sga_amd.make_synthetic_weights(C, seed=3)
    w["eb.medians"] = np.linspace(-0.4, 0.45, C).astype(np.float32)
    t = {k: np.asarray(v, np.float32) for k, v in raw_variables(w, C).items()}
    got = tfc.effective_weights_from_tensors(t, C)
    assert np.array_equal(got["eb.medians"], w["eb.medians"])
    del t["entropy_bottleneck/quantiles"]            # e.g. the bits-back prior: no medians, no error
    assert "eb.medians" not in tfc.effective_weights_from_tensors(t, C)


@pytest.mark.parametrize("shape", [(5, 5), (3, 3), (4, 6)])
def test_irdft_matrix_columns_are_separable_real_fourier_modes(shape):
    """`kernel = irdft_matrix(support) @ kernel_rdft` (tfc RDFTParameterizer): the basis must be the
    orthonormal, separable REAL DFT basis of the kernel support (the rfft is applied axis by axis with
    real and imaginary parts kept as separate real coefficients).  Checked against numpy's FFT: every
    column, reshaped to the support, has its whole 2-D spectrum on ONE |frequency| pair (+-u, +-v)
    (np.fft.fftn); every such class is covered exactly as often as it has real degrees of freedom;
    column 0 is the constant (DC) mode; analysis is the transpose."""
    kh, kw = shape
    M = tfc.irdft_matrix(shape).astype(np.float64)              # [kh*kw (space), kh*kw (coefficients)]
    assert np.allclose(M @ M.T, np.eye(kh * kw), atol=1e-6)
    assert np.allclose(M[:, 0], 1.0 / np.sqrt(kh * kw), atol=1e-6)
    seen = {}
    for k in range(kh * kw):
        p = np.abs(np.fft.fftn(M[:, k].reshape(kh, kw), norm="ortho")) ** 2
        assert abs(p.sum() - 1.0) < 1e-5                          # unit energy
        (u, v) = np.unravel_index(np.argmax(p), p.shape)
        cls = {((su * u) % kh, (sv * v) % kw) for su in (1, -1) for sv in (1, -1)}
        assert sum(p[a] for a in cls) > 1 - 1e-5, (k, p.round(3))
        key = (min(u, (-u) % kh), min(v, (-v) % kw))
        seen[key] = seen.get(key, 0) + 1
    dof = lambda f, n: 1 if (f == 0 or 2 * f == n) else 2        # a self-conjugate 1-D frequency is real
    for (u, v), n in seen.items():
        assert n == dof(u, kh) * dof(v, kw), (u, v, n)
    assert sum(seen.values()) == kh * kw
    rng = np.random.RandomState(0)
    c = rng.standard_normal(kh * kw)
    x = M @ c
    assert np.allclose(M.T @ x, c, atol=1e-5)                     # analysis = transpose (orthonormal)
    # a pure DC coefficient synthesises the constant kernel, as np.fft.irfftn of a DC-only spectrum does
    spec = np.zeros((kh, kw // 2 + 1), complex)
    spec[0, 0] = np.sqrt(kh * kw)
    assert np.allclose(M[:, 0].reshape(kh, kw), np.fft.irfftn(spec, s=shape), atol=1e-6)
