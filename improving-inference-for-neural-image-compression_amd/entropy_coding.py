"""Actual entropy coding of the quantised latents (y_hat, z_hat) -- SURVEY.md 8(f)-4.

The reference turns latents into bytes only in `mbt2018.py` (`entropy_bottleneck.compress(z)`,
`conditional_bottleneck.compress(y)`, `tfc.PackedTensors`; mbt2018.py:84-85,211-222) through
tensorflow-compression's C++ range coder; `sga.py` reports estimated rates only.  This module
provides the counterpart for this build: quantised CDF tables from the same two entropy models,
a rANS core in C (`csrc_cpu/rans.c` -> librans.so) and a small container.  The byte format is
this build's own (tfc's `.tfci` cannot be reproduced without tfc).

  z_hat : per-channel factorized prior mass p_c(k) (tfc EntropyBottleneck._likelihood, restated
          from learned_prior.py:96-121), one table per channel, integers outside the table's
          range escape-coded.
  y_hat : N(mu, sigma) conv U(-.5,.5) (utils.py:80-102).  Tables are indexed by the scale level
          (the reference's scale_table: 64 log-spaced levels in [0.11, 256], sga.py:24-26,129) and
          by the fractional part of mu quantised to 1/8; the coded symbol is y_hat - round(mu).
          (SGA's y_hat = round(y) is not mean-centred, unlike tfc's round(y - mu) + mu.)

Probabilities use 16 bits, every in-range symbol keeps frequency >= 1, tail mass goes to the
escape symbol (tfc: tail_mass = 1e-9 + Golomb overflow codes).  Decoding needs bit-identical
(mu, sigma) on both sides: compute them with the same kernels (SGACodec.hyper_synthesis).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import struct

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "librans.so")
PRECISION = 16
TOTAL = 1 << PRECISION
SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64          # sga.py:24-26
MEAN_BINS = 8
MAGIC = b"SGAC"
BLOCK = 1024            # symbols per independent rANS stream (one device lane each; 8 bytes of overhead per block): the FIRST pass
BLOCK_MAX = 1 << 13     # ... and the largest block the second pass may choose.  A block is ONE device lane (csrc/rans.hip): at the
                        # trained-like operating point 65536-symbol blocks left ~5 lanes and saved 2.4 % of the bytes of 4096-symbol
                        # blocks for 3-4 x the device time (profiles/r05_coder_timing.txt: encode 5.1 -> 16.7 ms, decode 6.9 -> 27.7 ms;
                        # 16384 -> 65536 saves 0.03 %).  8192 keeps the framing under ~1 % there
BLOCK_TARGET_BYTES = 512


def adapted_block(block_bytes, block: int = BLOCK) -> int:
    """Symbols per block for the stream that is written, from the byte counts of a first pass at `block`.  Every block costs
    8 bytes of framing (its length + the final rANS state).  At 4 bpp a 1024-symbol block holds ~700 bytes and that is 1 %; at a
    trained codec's operating point (0.35 bpp, 87 % of the symbols 0 at ~0.05 bit each) it holds SEVEN bytes and the framing is
    15 % of the file (measured: tests/test_gpu_entropy.py::test_real_bytes_at_the_trained_like_operating_point_c192).  So: double
    the block until the mean payload per block reaches BLOCK_TARGET_BYTES (framing <= 1.6 %), up to BLOCK_MAX.  Host and device
    encoders produce the same first-pass counts, hence the same choice; the decoders read the block size from the frame."""
    bb = np.asarray(block_bytes, np.int64)
    mean_payload = max(float(bb.mean()) - 4.0, 1.0) if bb.size else float(BLOCK_TARGET_BYTES)
    out = int(block)
    while mean_payload < BLOCK_TARGET_BYTES and out < BLOCK_MAX:
        out *= 2
        mean_payload *= 2
    return out


def frame_blocks(block_bytes: np.ndarray, payload: bytes, block: int = BLOCK) -> bytes:
    """One blocked stream: <u32 number of blocks> <u32 symbols per block> <u32 bytes of block b>* <the blocks' bytes>."""
    bb = np.ascontiguousarray(block_bytes, dtype="<u4")
    return struct.pack("<II", bb.size, block) + bb.tobytes() + payload


def unframe_blocks(data: bytes):
    if len(data) < 8:
        raise ValueError("rans stream: truncated header")
    nb, block = struct.unpack("<II", data[:8])
    if nb > (len(data) - 8) // 4 or block <= 0:
        raise ValueError("rans stream: corrupt header")
    bb = np.frombuffer(data, "<u4", nb, 8).astype(np.uint32)
    payload = data[8 + 4 * nb:]
    if int(bb.sum()) != len(payload):
        raise ValueError("rans stream: block lengths do not add up to the payload")
    return bb, block, payload

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run __graft_entry__.build()")
        lib = C.CDLL(LIB_PATH)
        i32p, u32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
        lib.rans_encode.restype = C.c_size_t
        lib.rans_encode.argtypes = [i32p, i32p, C.c_size_t, u32p, i32p, i32p, C.c_int, u8p, C.c_size_t]
        lib.rans_decode.restype = C.c_int
        lib.rans_decode.argtypes = [u8p, C.c_size_t, i32p, C.c_size_t, u32p, i32p, i32p, C.c_int, i32p]
        lib.rans_encode_blocked.restype = C.c_size_t
        lib.rans_encode_blocked.argtypes = [i32p, i32p, C.c_size_t, C.c_size_t, u32p, i32p, i32p, C.c_int, u8p, C.c_size_t,
                                            u32p, u8p, C.c_size_t]
        lib.rans_decode_blocked.restype = C.c_int
        lib.rans_decode_blocked.argtypes = [u8p, u32p, C.c_size_t, i32p, C.c_size_t, C.c_size_t, u32p, i32p, i32p, C.c_int,
                                            i32p]
        _lib = lib
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---------------------------------------------------------------------------------------------
# probability models (numpy, float64) -> quantised CDF tables
# ---------------------------------------------------------------------------------------------
def factorized_mass(w: dict, ks: np.ndarray, half: float = 0.5) -> np.ndarray:
    """p_c(k) for grid points ks [K] and every channel: the mass of [k - half, k + half], returns [K, C]
    (learned_prior.py:96-121 + the box mass with the sign trick of tfc EntropyBottleneck._likelihood; half = 0.5: the
    integer bins of z_hat, smaller: the fine grid of bits_back.py)."""
    Cn = w["eb.m0"].shape[0]

    def logits(v):                       # v [K] (one grid for every channel) or [K, C] -> [K, C]
        v = np.asarray(v, np.float64)
        t = (np.broadcast_to(v[None, None, :], (Cn, 1, v.size)) if v.ndim == 1 else v.T[:, None, :]).astype(np.float64)
        for k in range(4):
            t = np.matmul(w[f"eb.m{k}"].astype(np.float64), t) + w[f"eb.b{k}"].astype(np.float64)
            if k < 3:
                t = t + w[f"eb.f{k}"].astype(np.float64) * np.tanh(t)
        return t[:, 0, :].T

    lo, up = logits(ks - half), logits(ks + half)
    sg = -np.sign(lo + up)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    return np.abs(sig(sg * up) - sig(sg * lo))


def _phi(x):
    return 0.5 * np.vectorize(math.erfc)(-x / math.sqrt(2.0))


def quantise_pmf(pmf: np.ndarray) -> np.ndarray:
    """pmf over regular symbols (sum <= 1; the remainder is the escape mass) -> CDF of len+2 uint32
    entries (regular symbols + escape), total exactly 1<<16, every frequency >= 1."""
    n = pmf.size
    esc = max(1.0 - float(pmf.sum()), 0.0)
    p = np.concatenate([pmf, [esc]])
    f = np.maximum(np.round(p * TOTAL).astype(np.int64), 1)
    # fix the total by adjusting the largest entries
    diff = int(TOTAL - f.sum())
    order = np.argsort(-f)
    i = 0
    while diff != 0:
        j = order[i % (n + 1)]
        step = 1 if diff > 0 else -1
        if f[j] + step >= 1:
            f[j] += step
            diff -= step
        i += 1
    cdf = np.zeros(n + 2, np.uint32)
    cdf[1:] = np.cumsum(f)
    assert cdf[-1] == TOTAL
    return cdf


class EntropyCoder:
    def __init__(self, weights: dict, z_max_abs: int = 96, tail: float = 2.0 ** -14, device_models=None, centred: bool = False,
                 medians=None):
        """centred=False: INTEGER latents (SGA's y_hat = round(y), sga.py:240-241): y symbol = y_hat - rint(mu) under the table
        of (scale level, bin of mu - rint(mu)); z symbol = z_hat under its channel's table on the integer grid.
        centred=True: what mbt2018.py compress codes (mbt2018.py:69,80): y_hat = round(y - mu) + mu, symbol round(y_hat - mu) under
        the ZERO-offset table of its scale level (one per level); z_hat = round(z - median_c) + median_c, symbol round(z_hat -
        median_c) under the channel's table on the grid median_c + k (`medians` [C], default weights["eb.medians"] or 0).
        device_models: optional pair (mass_fn, box_fn) evaluating the two entropy models with the
        HIP kernels of the SGA step instead of numpy -- `mass_fn(v [K, C]) -> p [K, C]` (factorized
        mass, sga_op_factorized_likelihood) and `box_fn(y, mu, sigma_raw) -> p` (box-convolved
        Gaussian, sga_op_gaussian_likelihood); see SGACodec._entropy_coder(device_tables=True).
        Encoder and decoder must build their tables the same way."""
        self.C = weights["eb.m0"].shape[0]
        self.centred = bool(centred)
        self.mean_bins = 1 if self.centred else MEAN_BINS
        if medians is None:
            medians = weights.get("eb.medians")
        self.medians = (np.zeros(self.C, np.float32) if medians is None or not self.centred
                        else np.ascontiguousarray(medians, np.float32).reshape(self.C))
        self.scale_table = np.exp(np.linspace(math.log(SCALES_MIN), math.log(SCALES_MAX), SCALES_LEVELS))
        tables, lens, offs = [], [], []
        # ---- z: one table per channel (on the grid median_c + k when centred) ----------------
        ks = np.arange(-z_max_abs, z_max_abs + 1, dtype=np.float64)
        grid = ks[:, None] + self.medians[None, :].astype(np.float64)            # [K, C]
        if device_models is not None:
            mass = np.asarray(device_models[0](np.ascontiguousarray(grid, np.float32)), np.float64)
        else:
            mass = factorized_mass(weights, grid if self.centred else ks)      # [K, C]
        for c in range(self.C):
            m = mass[:, c]
            keep = np.nonzero(m >= tail / 8)[0]
            lo, hi = (keep[0], keep[-1]) if keep.size else (z_max_abs, z_max_abs)
            tables.append(quantise_pmf(m[lo:hi + 1]))
            lens.append(hi - lo + 2)
            offs.append(int(ks[lo]))
        self.z_tab0 = 0
        # ---- y: scale level x mean-fraction bin -----------------------------------------------
        self.y_tab0 = len(tables)
        specs = []                                               # (R, f, s) of every y table, in table order
        for s in self.scale_table:
            R = int(math.ceil(6.0 * s + 1.0))                    # +-6 sigma, rest escapes
            for j in range(self.mean_bins):
                specs.append((R, 0.0 if self.centred else (j + 0.5) / MEAN_BINS - 0.5, s))
        dev = None
        if device_models is not None:                            # all tables' grids in ONE device call
            ys = np.concatenate([np.arange(-R, R + 1, dtype=np.float32) for R, _, _ in specs])
            mus = np.concatenate([np.full(2 * R + 1, f, np.float32) for R, f, _ in specs])
            srs = np.concatenate([np.full(2 * R + 1, math.log(s), np.float32) for R, _, s in specs])
            dev = np.asarray(device_models[1](ys, mus, srs), np.float64)
        pos = 0
        for R, f, s in specs:
            r = np.arange(-R, R + 1, dtype=np.float64)
            if dev is not None:
                pm = dev[pos:pos + r.size]
                pos += r.size
            else:
                pm = _phi((r + 0.5 - f) / s) - _phi((r - 0.5 - f) / s)
            tables.append(quantise_pmf(pm))
            lens.append(r.size + 1)
            offs.append(-R)
        self.stride = max(t.size for t in tables)
        self.cdf = np.zeros((len(tables), self.stride), np.uint32)
        for i, t in enumerate(tables):
            self.cdf[i, :t.size] = t
            self.cdf[i, t.size:] = TOTAL
        self.lens = np.asarray(lens, np.int32)
        self.offs = np.asarray(offs, np.int32)
        self.table_mode = (1 if device_models is not None else 0) | (2 if self.centred else 0)      # the container's mode byte
        self._crc = None

    def table_crc(self) -> int:
        """CRC32 over every quantised CDF table: the fingerprint a stream carries (`pack`) and a decoder checks."""
        if self._crc is None:
            import zlib
            self._crc = zlib.crc32(np.ascontiguousarray(self.cdf).tobytes() + self.lens.tobytes() + self.offs.tobytes())
        return self._crc

    # ---- symbol/table preparation ---------------------------------------------------------------
    def _levels(self, sigma):
        sg = np.maximum(np.asarray(sigma, np.float64), SCALES_MIN)
        return np.clip(np.searchsorted(self.scale_table, sg, side="left"), 0, SCALES_LEVELS - 1).astype(np.int32)

    @staticmethod
    def _centred_symbols(v, centre, what):
        """round(v - centre) in float32 (the arithmetic that made v), refusing anything that is not an integer to 1e-3"""
        d = np.asarray(v, np.float32) - np.asarray(centre, np.float32)
        r = np.rint(d)
        if d.size and float(np.abs(d - r).max()) > 1e-3:
            raise ValueError(f"{what} is not its centre + an integer (max deviation {float(np.abs(d - r).max()):.3g}): "
                             "a centred coder codes round(v - centre)")
        return r.astype(np.int32)

    def _y_symbols(self, y_hat, mu, sigma):
        mu = np.asarray(mu, np.float32)
        if self.centred:
            return np.zeros(mu.shape, np.int32), (self.y_tab0 + self._levels(sigma)).astype(np.int32)
        r0 = np.rint(mu)
        frac = (mu - r0).astype(np.float64)
        jbin = np.clip(np.floor((frac + 0.5) * MEAN_BINS), 0, MEAN_BINS - 1).astype(np.int32)
        sg = np.maximum(np.asarray(sigma, np.float64), SCALES_MIN)
        lvl = np.clip(np.searchsorted(self.scale_table, sg, side="left"), 0, SCALES_LEVELS - 1).astype(np.int32)
        tab = (self.y_tab0 + lvl * MEAN_BINS + jbin).astype(np.int32)
        return r0.astype(np.int32), tab

    def _run_encode(self, sym, tab, block=None):
        """One blocked stream.  block=None: a first pass at BLOCK symbols per block, then -- if its blocks came out nearly empty --
        the pass that is written, at `adapted_block` symbols per block."""
        lib = _load()
        sym = np.ascontiguousarray(sym.reshape(-1), np.int32)
        tab = np.ascontiguousarray(tab.reshape(-1), np.int32)

        def one_pass(blk):
            nb = -(-sym.size // blk)
            cap = 16 * nb + 8 * sym.size
            out, scratch = np.zeros(cap, np.uint8), np.zeros(16 + 8 * blk, np.uint8)
            bb = np.zeros(nb, np.uint32)
            n = lib.rans_encode_blocked(_ptr(sym, C.c_int32), _ptr(tab, C.c_int32), sym.size, blk,
                                        _ptr(self.cdf, C.c_uint32), _ptr(self.lens, C.c_int32), _ptr(self.offs, C.c_int32),
                                        self.stride, _ptr(out, C.c_uint8), cap, _ptr(bb, C.c_uint32),
                                        _ptr(scratch, C.c_uint8), scratch.size)
            if n == 0:
                raise RuntimeError("rans_encode: output buffer overflow")
            return bb, out[:n].tobytes()

        blk = int(block) if block else BLOCK
        bb, payload = one_pass(blk)
        if not block:
            blk2 = adapted_block(bb, blk)
            if blk2 != blk:
                blk = blk2
                bb, payload = one_pass(blk)
        return frame_blocks(bb, payload, blk)

    def _run_decode(self, data: bytes, tab):
        lib = _load()
        tab = np.ascontiguousarray(tab.reshape(-1), np.int32)
        bb, block, payload = unframe_blocks(data)
        if bb.size != -(-tab.size // block):
            raise ValueError("rans_decode: corrupt stream (block count)")
        buf = np.frombuffer(payload, np.uint8).copy() if payload else np.zeros(1, np.uint8)
        sym = np.zeros(tab.size, np.int32)
        rc = lib.rans_decode_blocked(_ptr(buf, C.c_uint8), _ptr(bb, C.c_uint32), bb.size, _ptr(tab, C.c_int32), tab.size,
                                     block, _ptr(self.cdf, C.c_uint32), _ptr(self.lens, C.c_int32),
                                     _ptr(self.offs, C.c_int32), self.stride, _ptr(sym, C.c_int32))
        if rc != 0:
            raise ValueError("rans_decode: corrupt stream")
        return sym

    # ---- public API -----------------------------------------------------------------------------
    @staticmethod
    def _integers(v, what):
        """The coder codes INTEGER latents (SGA's y_hat = round(y), sga.py:240-241).  Mean- or median-
        centred latents (round(y - mu) + mu, mbt2018.py:69,80) are not integers: refuse them instead of
        silently coding something else than what was passed in."""
        v = np.asarray(v)
        r = np.rint(v)
        if not np.array_equal(r, v):
            raise ValueError(f"{what} must hold integers (got max |v - rint(v)| = "
                             f"{float(np.abs(v - r).max()):.3g}); centred latents are not supported by this coder")
        return r

    def encode_z(self, z_hat) -> bytes:
        if self.centred:
            z = self._centred_symbols(z_hat, self.medians, "z_hat")
        else:
            z = self._integers(z_hat, "z_hat").astype(np.int32)
        tab = np.broadcast_to(np.arange(self.C, dtype=np.int32), z.shape)
        return self._run_encode(z, tab)

    def decode_z(self, data: bytes, shape) -> np.ndarray:
        tab = np.broadcast_to(np.arange(self.C, dtype=np.int32), shape)
        z = self._run_decode(data, tab).reshape(shape).astype(np.float32)
        return z + self.medians if self.centred else z      # float32 add: the operation that made z_hat (mbt2018.py:69)

    def encode_y(self, y_hat, mu, sigma) -> bytes:
        r0, tab = self._y_symbols(y_hat, mu, sigma)
        if self.centred:
            return self._run_encode(self._centred_symbols(y_hat, mu, "y_hat"), tab)
        return self._run_encode(self._integers(y_hat, "y_hat").astype(np.int32) - r0, tab)

    def decode_y(self, data: bytes, mu, sigma) -> np.ndarray:
        r0, tab = self._y_symbols(None, mu, sigma)
        sym = self._run_decode(data, tab).reshape(r0.shape)
        if self.centred:
            return sym.astype(np.float32) + np.asarray(mu, np.float32)      # round(y - mu) + mu in float32 (mbt2018.py:80)
        return (sym + r0).astype(np.float32)

    def ideal_bits_y(self, y_hat, mu, sigma) -> float:
        """-sum log2 of the QUANTISED model probabilities actually used (for tests)."""
        r0, tab = self._y_symbols(y_hat, mu, sigma)
        sym = self._centred_symbols(y_hat, mu, "y_hat").astype(np.int64) if self.centred else np.rint(np.asarray(y_hat)).astype(np.int64) - r0
        idx = (sym - self.offs[tab]).reshape(-1)
        tab = tab.reshape(-1)
        ln = self.lens[tab]
        esc = (idx < 0) | (idx >= ln - 1)
        idx = np.where(esc, ln - 1, idx)
        f = self.cdf[tab, idx + 1].astype(np.float64) - self.cdf[tab, idx]
        return float(-np.log2(f / TOTAL).sum() + 32.0 * esc.sum())


FORMAT_VERSION = 3      # 1 (round 2, magic only) had no table fingerprint; 2 (rounds 3-5) did not say in which arithmetic (mu, sigma) were
                        # computed, and its second pass chose blocks of up to 65536 symbols (same framing: format 3 reads format-2 bodies)
# the container's mode byte: bit 0 tables built on the device, bit 1 centred latents, bits 2-3 the precision mode of the handle whose
# h_s produced (mu, sigma) -- a decoder in another mode computes other sigma levels and decodes garbage --, bits 4-7 zero
MODE_PRECISION_SHIFT, MODE_PRECISION_MASK, MODE_KNOWN_BITS = 2, 3, 0x0F
MODE_PRECISIONS = ("f32", "bf16x3", "bf16x2")


def mode_precision(table_mode: int) -> str:
    return MODE_PRECISIONS[(table_mode >> MODE_PRECISION_SHIFT) & MODE_PRECISION_MASK]


def pack(x_shape, y_shape, z_shape, z_bytes: bytes, y_bytes: bytes, table_mode: int = 0, table_crc: int = 0) -> bytes:
    """Container (cf. tfc.PackedTensors, mbt2018.py:211-214): magic, format version, how the coder's CDF tables were
    built (bit 0: 0 = host float64 numpy, 1 = device float32 kernels; bit 1: centred latents, mbt2018.py compress; bits 2-3: the
    arithmetic of the h_s that gave (mu, sigma): 0 f32, 1 bf16x3, 2 bf16x2) and their CRC32, shapes, two length-prefixed streams.
    A range coder needs bit-identical tables on both sides: the decoder refuses a stream whose fingerprint is not its own."""
    head = MAGIC + struct.pack("<BBHI", FORMAT_VERSION, table_mode, 0, table_crc & 0xFFFFFFFF)
    head += struct.pack("<3I4I4I", *x_shape, *y_shape, *z_shape)
    return head + struct.pack("<I", len(z_bytes)) + z_bytes + struct.pack("<I", len(y_bytes)) + y_bytes


def unpack(blob: bytes, with_tables: bool = False):
    """Inverse of `pack`.  Every malformed input -- wrong magic, another format version, a truncated header, stream lengths
    that run past the blob -- raises ValueError (never struct.error, never silently short streams)."""
    if len(blob) < 4 or blob[:4] != MAGIC:
        raise ValueError("not an SGAC stream")
    if len(blob) < 12:
        raise ValueError("corrupt stream: truncated SGAC header")
    version, table_mode, _, table_crc = struct.unpack("<BBHI", blob[4:12])
    if version != FORMAT_VERSION:
        raise ValueError(f"SGAC stream format {version}, this build reads format {FORMAT_VERSION} "
                         + ("(format 1, round 2, carried no table fingerprint: re-encode the latents)" if version == 1 else
                            "(format 2 did not record the precision mode of its encoder: re-encode the latents)" if version == 2 else ""))
    if (table_mode & ~MODE_KNOWN_BITS) or ((table_mode >> MODE_PRECISION_SHIFT) & MODE_PRECISION_MASK) >= len(MODE_PRECISIONS):
        raise ValueError(f"SGAC stream: unknown bits in the mode byte ({table_mode:#04x}); written by a newer build?")
    if len(blob) < 56 + 8:
        raise ValueError("corrupt stream: truncated SGAC header")
    v = struct.unpack("<3I4I4I", blob[12:56])
    x_shape, y_shape, z_shape = v[:3], v[3:7], v[7:11]
    p = 56
    (nz,) = struct.unpack("<I", blob[p:p + 4]); p += 4
    if p + nz + 4 > len(blob):
        raise ValueError("corrupt stream: z stream length runs past the end")
    z_bytes = blob[p:p + nz]; p += nz
    (ny,) = struct.unpack("<I", blob[p:p + 4]); p += 4
    if p + ny != len(blob):
        raise ValueError("corrupt stream: y stream length does not match the blob")
    y_bytes = blob[p:p + ny]
    if with_tables:
        return x_shape, y_shape, z_shape, z_bytes, y_bytes, table_mode, table_crc
    return x_shape, y_shape, z_shape, z_bytes, y_bytes
