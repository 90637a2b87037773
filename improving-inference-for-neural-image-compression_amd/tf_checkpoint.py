"""TensorFlow checkpoint (tensor-bundle V2) importer -> effective weights of the hot path.

The reference restores its models with `tf.train.Saver().restore(sess, latest)` (sga.py:180-182)
from checkpoints that are only distributed via Google Drive (README.md:100-103).  This module
reads such a checkpoint WITHOUT TensorFlow:

  <prefix>.index                 an SSTable (LevelDB table format) name -> BundleEntryProto
  <prefix>.data-00000-of-00001   raw little-endian tensor bytes

and applies the reparameterisations of tensorflow-compression 1.3 so that the result is the dict
of *effective* tensors `weights.layer_shapes()` describes (8(f)-1):

  conv kernels of g_a / g_s / h_a  : `kernel_rdft` [prod(support), C_in*C_out]
                                     -> kernel = irdft_matrix(support) @ kernel_rdft   (RDFTParameterizer)
  conv kernels of h_s              : `kernel` as stored (kernel_parameterizer=None, nn_models.py:154-162)
  GDN beta / gamma                 : `reparam_beta`, `reparam_gamma` (NonnegativeParameterizer)
                                     beta  = max(r, sqrt(1e-6 + 2^-36))^2 - 2^-36
                                     gamma = max(r, 2^-18)^2 - 2^-36
  factorized prior                 : softplus(matrix_k), bias_k, tanh(factor_k)

Variable names expected in the checkpoint (Keras layer names of nn_models.py under tfc 1.3; matched by
regular expression on scope + suffix, optimizer slots `/Adam*` and moving averages ignored):

  analysis_transform/layer_{0..3}/kernel_rdft, .../bias          layer_{0..2}/gdn_{0..2}/reparam_beta|reparam_gamma
  synthesis_transform/layer_{0..3}/kernel_rdft, .../bias         layer_{0..2}/igdn_{0..2}/reparam_beta|reparam_gamma
  hyper_analysis_transform/layer_{0..2}/kernel_rdft, layer_{0,1}/bias            (layer_2: use_bias=False, nn_models.py:95)
  mbt2018_hyper_synthesis_transform/layer_{0..2}/kernel, .../bias                (kernel_parameterizer=None)
  entropy_bottleneck/matrix_{0..3}, bias_{0..3}, factor_{0..2}, quantiles        (the scope the dummy call at
      sga.py:100 / danneal.py:102-111 exists to create: without it the variables would be named "matrix_0", ... and
      `Saver.restore` fails with "Key bias_0 not found in checkpoint"); the bits-back models keep the same
      tensors under the scope of learned_prior.BMSHJ2018Prior (no `quantiles`)

STATUS: the container format code is exercised by tests/test_host.py against a minimal writer
(uncompressed and snappy-compressed blocks).  The variable naming and the RDFT basis follow
tfc 1.3 as published; they are UNVERIFIED here -- no real checkpoint exists offline and TF/tfc
cannot be installed.  `load_effective_weights` matches variables by suffix so
that scope-name differences surface as a clear error listing the keys it found.
"""
from __future__ import annotations

import glob
import os
import re
import struct

import numpy as np

from .weights import check_weights, layer_shapes

_TABLE_MAGIC = 0xDB4775248B80FB57
_DT_FLOAT = 1


# ---------------------------------------------------------------------------------------------
# low-level decoding: varints, snappy, LevelDB table blocks, the two protos we need
# ---------------------------------------------------------------------------------------------
def _varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def snappy_decompress(data: bytes) -> bytes:
    """Raw snappy block format (the LevelDB tables TF writes may use it for index blocks)."""
    n, pos = _varint(data, 0)
    out = bytearray()
    while pos < len(data):
        tag = data[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                    # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += data[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 2], "little")
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy stream")
        for _ in range(ln):                              # may overlap: byte by byte
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


def _read_block(f_bytes: bytes, offset: int, size: int) -> bytes:
    """Block contents (without the 5-byte trailer: 1 byte compression type + masked crc32c)."""
    raw = f_bytes[offset:offset + size]
    ctype = f_bytes[offset + size]
    if ctype == 0:
        return raw
    if ctype == 1:
        return snappy_decompress(raw)
    raise ValueError(f"unsupported block compression {ctype}")


def _block_entries(block: bytes):
    """Yield (key, value) of a LevelDB block (prefix-compressed keys, restart array at the end)."""
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _parse_proto(buf: bytes):
    """Minimal protobuf wire decoder -> list of (field, wire_type, value)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        out.append((field, wt, v))
    return out


def _parse_entry(buf: bytes):
    """BundleEntryProto: dtype=1, shape=2 (TensorShapeProto.dim=2{size=1}), shard_id=3, offset=4,
    size=5, crc32c=6, slices=7."""
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, sliced=False)
    for field, wt, v in _parse_proto(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e["shape"].append(size)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 7:
            e["sliced"] = True
    return e


def read_index(index_path: str) -> dict:
    """name -> entry dict for every tensor in a `<prefix>.index` file (the "" header is skipped)."""
    data = open(index_path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != _TABLE_MAGIC:
        raise ValueError(f"{index_path}: not a TensorFlow bundle index (bad table magic)")
    footer = data[-48:]
    _, p = _varint(footer, 0)            # metaindex handle (offset, size): unused
    _, p = _varint(footer, p)
    idx_off, p = _varint(footer, p)
    idx_size, p = _varint(footer, p)
    entries = {}
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        off, q = _varint(handle, 0)
        size, _ = _varint(handle, q)
        for key, value in _block_entries(_read_block(data, off, size)):
            if key:
                entries[key.decode()] = _parse_entry(value)
    return entries


def latest_checkpoint(checkpoint_dir: str) -> str:
    """tf.train.latest_checkpoint: the `checkpoint` state file's model_checkpoint_path, else the
    newest *.index."""
    state = os.path.join(checkpoint_dir, "checkpoint")
    if os.path.exists(state):
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if m:
            p = m.group(1)
            return p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)
    idx = sorted(glob.glob(os.path.join(checkpoint_dir, "*.index")), key=os.path.getmtime)
    if not idx:
        raise FileNotFoundError(f"no checkpoint found in {checkpoint_dir}")
    return idx[-1][:-len(".index")]


def read_checkpoint(prefix: str) -> dict:
    """All float32 tensors of a V2 checkpoint as numpy arrays."""
    entries = read_index(prefix + ".index")
    shards = {}
    out = {}
    for name, e in entries.items():
        if e["dtype"] != _DT_FLOAT or e["sliced"]:
            continue
        sid = e["shard_id"]
        if sid not in shards:
            cands = glob.glob(f"{prefix}.data-{sid:05d}-of-*")
            if not cands:
                raise FileNotFoundError(f"data shard {sid} of {prefix} is missing")
            shards[sid] = np.memmap(cands[0], dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        out[name] = np.frombuffer(bytes(raw), dtype="<f4").reshape(e["shape"]).copy()
    return out


# ---------------------------------------------------------------------------------------------
# tensorflow-compression 1.3 reparameterisations
# ---------------------------------------------------------------------------------------------
def irdft_matrix(shape) -> np.ndarray:
    """tfc.python.ops.spectral_ops.irdft_matrix (UNVERIFIED restatement): orthonormal real-DFT
    basis, one axis at a time; returns [prod(shape), prod(shape)] with rows = spatial positions,
    so that kernel = irdft_matrix @ kernel_rdft."""
    shape = tuple(int(s) for s in shape)
    size = int(np.prod(shape))
    rank = len(shape)
    matrix = np.identity(size, dtype=np.float64).reshape((size,) + shape)
    for axis in range(rank):
        matrix = np.fft.rfft(matrix, axis=axis + 1)
        slices = (rank + 1) * [slice(None)]
        slices[axis + 1] = slice(1, None) if shape[axis] % 2 == 1 else slice(1, -1)
        matrix[tuple(slices)] *= np.sqrt(2)
        matrix /= np.sqrt(shape[axis])
        matrix = np.concatenate([matrix.real, matrix.imag[tuple(slices)]], axis=axis + 1)
    return matrix.reshape(size, size).astype(np.float32)


_PEDESTAL = 2.0 ** -36


def gdn_beta(reparam):   # NonnegativeParameterizer(minimum=1e-6), reparam_offset = 2^-18
    return (np.maximum(reparam, np.sqrt(1e-6 + _PEDESTAL)) ** 2 - _PEDESTAL).astype(np.float32)


def gdn_gamma(reparam):  # NonnegativeParameterizer(minimum=0)
    return (np.maximum(reparam, 2.0 ** -18) ** 2 - _PEDESTAL).astype(np.float32)


def _softplus(x):
    return (np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)).astype(np.float32)


def _find(tensors: dict, scope_pat: str, suffix: str):
    """The unique variable whose name matches `<scope>.../<suffix>` (optimizer slots excluded)."""
    rx = re.compile(scope_pat + r".*/" + re.escape(suffix) + r"$")
    hits = [k for k in tensors if rx.search(k) and "/Adam" not in k and "ExponentialMovingAverage" not in k]
    if len(hits) != 1:
        raise KeyError(f"expected exactly one variable matching /{scope_pat}.*{suffix}/, found {hits}; "
                       f"checkpoint holds: {sorted(tensors)[:40]} ...")
    return tensors[hits[0]]


def effective_weights_from_tensors(tensors: dict, num_filters: int, bb: bool = False) -> dict:
    """Map raw checkpoint variables to the effective-weight dict (see module docstring)."""
    C = int(num_filters)
    shapes = layer_shapes(C, bb)
    scopes = {"ga": r"(^|/)analysis_transform", "gs": r"(^|/)synthesis_transform",
              "ha": r"hyper_analysis_transform", "hs": r"hyper_synthesis_transform"}
    w = {}

    def kernel(prefix, i):
        name = f"{prefix}.k{i}"
        kh, kw, cin, cout = shapes[name]
        scope = scopes[prefix]
        try:
            rdft = _find(tensors, scope, f"layer_{i}/kernel_rdft")
            k = irdft_matrix((kh, kw)) @ rdft.reshape(kh * kw, cin * cout)
        except KeyError:
            k = _find(tensors, scope, f"layer_{i}/kernel")          # kernel_parameterizer=None
        w[name] = np.ascontiguousarray(k.reshape(kh, kw, cin, cout), dtype=np.float32)

    for prefix, n in (("ga", 4), ("gs", 4), ("ha", 3), ("hs", 3)):
        for i in range(n):
            kernel(prefix, i)
            if f"{prefix}.b{i}" in shapes:
                w[f"{prefix}.b{i}"] = _find(tensors, scopes[prefix], f"layer_{i}/bias").astype(np.float32)
    for prefix, gname in (("ga", "gdn"), ("gs", "igdn")):
        for i in range(3):
            w[f"{prefix}.beta{i}"] = gdn_beta(_find(tensors, scopes[prefix], f"{gname}_{i}/reparam_beta"))
            w[f"{prefix}.gamma{i}"] = gdn_gamma(_find(tensors, scopes[prefix], f"{gname}_{i}/reparam_gamma"))
    prior = r"(entropy_bottleneck|bmshj2018_prior|hyper_prior)"
    for k in range(4):
        w[f"eb.m{k}"] = _softplus(_find(tensors, prior, f"matrix_{k}"))
        w[f"eb.b{k}"] = _find(tensors, prior, f"bias_{k}").astype(np.float32)
        if k < 3:
            w[f"eb.f{k}"] = np.tanh(_find(tensors, prior, f"factor_{k}")).astype(np.float32)
    for name, shp in shapes.items():
        w[name] = np.ascontiguousarray(w[name].reshape(shp), dtype=np.float32)
    # tfc 1.3 EntropyBottleneck keeps `quantiles` (C,1,3) = (lower tail, MEDIAN, upper tail); the
    # median centres the rounding of `_quantize(z, 'dequantize')` (mbt2018.py:69, map.py:83).  The
    # bits-back prior (learned_prior.BMSHJ2018Prior) has no such variable.
    try:
        q = _find(tensors, prior, "quantiles")
        w["eb.medians"] = np.ascontiguousarray(np.asarray(q, np.float32).reshape(C, -1)[:, 1])
    except KeyError:
        pass
    check_weights(w, C, bb)
    return w


def load_effective_weights(model_dir: str, num_filters: int, bb: bool = False) -> dict:
    """Counterpart of sga.py:180-182: latest checkpoint of `<checkpoint_dir>/<runname>` ->
    effective weights.  Also accepts an `.npz` of effective tensors (`weights.layer_shapes` keys)."""
    if model_dir.endswith(".npz") or os.path.isfile(model_dir + ".npz"):
        path = model_dir if model_dir.endswith(".npz") else model_dir + ".npz"
        w = {k: np.asarray(v, dtype=np.float32) for k, v in np.load(path).items()}
        check_weights(w, num_filters, bb)
        return w
    return effective_weights_from_tensors(read_checkpoint(latest_checkpoint(model_dir)), num_filters, bb)
