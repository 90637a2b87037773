"""Bits-back coding for the mbt2018_bb model (cfg 5).  `bb_sga.py:133-139` only ESTIMATES the refund (`bpp_back = -sum
log q(z_tilde | y) / (ln 2 * pixels)`, subtracted from the estimated rate); the reference has no coder for it (its one
real coder, mbt2018.py:84-85, is for the base model).  This module is the coder that earns the estimate, on the ANS stack
of csrc_cpu/rans.c (rANS is a stack: what `push` writes, `pop` reads back):

  sender                                            receiver
  1. POP  z_bar ~ Q(. | y)   (bits got back)        1. POP  z_bar under the prior P
  2. (mu, sigma) = h_s(z_bar);                      2. (mu, sigma) = h_s(z_bar);  POP y_hat under p(y | z_bar)
     PUSH y_hat under p(y | z_bar)                  3. zml = stage 2 on y_hat (sga_bb_refine: the sender's q, bit for bit);
  3. PUSH z_bar under the prior P                      PUSH z_bar under Q  -> the stack is the sender's initial stack again

z is continuous: it is coded on a grid of width delta (z_bar = delta * k).  Q and P are the bin masses of q and of the prior's
density, so the net cost -log2 P(k) + log2 Q(k) = -log2 p(z_bar) + log2 q(z_bar) up to the discretisation: delta
cancels, as it does in the estimate.  Q reuses the conditional's table family on the variable z / delta (a unit-bin
Gaussian with mean z_mean / delta and scale sigma_q / delta); P gets per-channel tables of the factorized prior on the grid.
The first pop needs bits on the stack: `init_bytes` seeded bytes (in a real stream: previously coded data); they
are returned exactly by the receiver and are not charged to the image.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from . import entropy_coding as ec


def _stack_lib():
    lib = ec._load()
    if not getattr(lib, "_stack_typed", False):
        i32p, u32p, u8p, szp = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)
        lib.rans_stack_push.restype = C.c_int
        lib.rans_stack_push.argtypes = [u32p, u8p, C.c_size_t, szp, i32p, i32p, C.c_size_t, u32p, i32p, i32p, C.c_int]
        lib.rans_stack_pop.restype = C.c_int
        lib.rans_stack_pop.argtypes = [u32p, u8p, szp, i32p, C.c_size_t, u32p, i32p, i32p, C.c_int, i32p]
        lib._stack_typed = True
    return lib


class AnsStack:
    """32-bit rANS state + byte stack (top = end of the buffer)."""

    def __init__(self, initial: bytes, capacity: int):
        self.buf = np.zeros(capacity, np.uint8)
        self.buf[:len(initial)] = np.frombuffer(initial, np.uint8)
        self.len = C.c_size_t(len(initial))
        self.x = C.c_uint32(1 << 23)
        self.lib = _stack_lib()

    def _tabs(self, coder):
        return (ec._ptr(coder.cdf, C.c_uint32), ec._ptr(coder.lens, C.c_int32), ec._ptr(coder.offs, C.c_int32), coder.stride)

    def push(self, coder, sym, tab):
        sym, tab = np.ascontiguousarray(sym.reshape(-1), np.int32), np.ascontiguousarray(tab.reshape(-1), np.int32)
        cdf, lens, offs, stride = self._tabs(coder)
        rc = self.lib.rans_stack_push(C.byref(self.x), ec._ptr(self.buf, C.c_uint8), self.buf.size, C.byref(self.len),
                                      ec._ptr(sym, C.c_int32), ec._ptr(tab, C.c_int32), sym.size, cdf, lens, offs, stride)
        if rc:
            raise RuntimeError("ANS stack overflow")

    def pop(self, coder, tab):
        tab = np.ascontiguousarray(tab.reshape(-1), np.int32)
        sym = np.zeros(tab.size, np.int32)
        cdf, lens, offs, stride = self._tabs(coder)
        rc = self.lib.rans_stack_pop(C.byref(self.x), ec._ptr(self.buf, C.c_uint8), C.byref(self.len),
                                     ec._ptr(tab, C.c_int32), tab.size, cdf, lens, offs, stride, ec._ptr(sym, C.c_int32))
        if rc == -2:
            raise RuntimeError("ANS stack ran dry: more initial bits are needed to sample z from q(z | y)")
        if rc:
            raise ValueError("ANS stack: corrupt table or stream")
        return sym

    def tobytes(self) -> bytes:
        return struct.pack("<I", self.x.value) + self.buf[:self.len.value].tobytes()

    @classmethod
    def frombytes(cls, data: bytes, capacity: int):
        s = cls(data[4:], capacity)
        s.x = C.c_uint32(struct.unpack("<I", data[:4])[0])
        return s

    def bits(self) -> float:
        """Information held: the byte stack plus the state's significant bits."""
        return 8.0 * self.len.value + float(np.log2(max(self.x.value, 1)))


class _PriorGrid:
    """Quantised CDF tables of the factorized prior on the grid delta * k (one table per channel), in the layout
    EntropyCoder's tables have (cdf / lens / offs / stride), so that the same rANS primitives code with them."""

    def __init__(self, weights, delta, k_max, tail=2.0 ** -17):
        Cn = weights["eb.m0"].shape[0]
        ks = np.arange(-k_max, k_max + 1, dtype=np.float64)
        mass = ec.factorized_mass(weights, ks * delta, half=0.5 * delta)          # [K, C]
        tables, lens, offs = [], [], []
        for c in range(Cn):
            m = mass[:, c]
            keep = np.nonzero(m >= tail)[0]
            lo, hi = (keep[0], keep[-1]) if keep.size else (k_max, k_max)
            tables.append(ec.quantise_pmf(m[lo:hi + 1]))
            lens.append(hi - lo + 2)
            offs.append(int(ks[lo]))
        self.stride = max(t.size for t in tables)
        self.cdf = np.zeros((Cn, self.stride), np.uint32)
        for i, t in enumerate(tables):
            self.cdf[i, :t.size] = t
            self.cdf[i, t.size:] = ec.TOTAL
        self.lens, self.offs = np.asarray(lens, np.int32), np.asarray(offs, np.int32)


def quantise_pmf_closed(pmf: np.ndarray, tail_lo: float, tail_hi: float) -> np.ndarray:
    """A CDF WITHOUT escape mass: the probability outside the table is folded into its two edge bins, every regular symbol
    keeps frequency >= 1 and the regular symbols sum to exactly 1 << 16; the escape entry the rANS primitives expect at
    the end of a table has frequency 0 (never popped: no state value maps to it)."""
    p = np.array(pmf, np.float64)
    p[0] += max(tail_lo, 0.0)
    p[-1] += max(tail_hi, 0.0)
    p = p / p.sum()
    f = np.maximum(np.round(p * ec.TOTAL).astype(np.int64), 1)
    diff = int(ec.TOTAL - f.sum())
    order = np.argsort(-f)
    i = 0
    while diff != 0:
        j = order[i % f.size]
        step = 1 if diff > 0 else -1
        if f[j] + step >= 1:
            f[j] += step
            diff -= step
        i += 1
    cdf = np.zeros(f.size + 2, np.uint32)
    cdf[1:-1] = np.cumsum(f)
    cdf[-1] = cdf[-2]
    assert cdf[-1] == ec.TOTAL
    return cdf


class _PosteriorTables:
    """The tables Q(z | y) is sampled from and coded back with: the conditional's family (scale level x mean-fraction bin,
    unit bins, entropy_coding.EntropyCoder) WITHOUT escape symbols.  A `pop` draws a symbol from whatever bits are on the
    stack; with the conditional's own tables it lands in the escape symbol (frequency >= 1 of 65536 in all 512 tables) once
    in ~2^16 elements and then reads 32 raw stack bits as the value -- |z_bar| ~ 2^30 / 8, one image in four at Kodak size
    (18 432 elements).  Here every state value maps to a regular symbol.  Built on the host in float64 (both sides must
    build identical tables; they do not depend on the weights)."""

    def __init__(self, yq: ec.EntropyCoder):
        import math
        tables, lens, offs = [], [], []
        for s in yq.scale_table:
            R = int(math.ceil(6.0 * s + 1.0))
            for j in range(ec.MEAN_BINS):
                f = (j + 0.5) / ec.MEAN_BINS - 0.5
                r = np.arange(-R, R + 1, dtype=np.float64)
                up, lo = ec._phi((r + 0.5 - f) / s), ec._phi((r - 0.5 - f) / s)
                tables.append(quantise_pmf_closed(up - lo, float(lo[0]), float(1.0 - up[-1])))
                lens.append(r.size + 1)
                offs.append(-R)
        self.stride = max(t.size for t in tables)
        self.cdf = np.zeros((len(tables), self.stride), np.uint32)
        for i, t in enumerate(tables):
            self.cdf[i, :t.size] = t
            self.cdf[i, t.size:] = ec.TOTAL
        self.lens, self.offs = np.asarray(lens, np.int32), np.asarray(offs, np.int32)
        self.tab0 = yq.y_tab0

    def in_range(self, sym, tab) -> bool:
        i = sym.astype(np.int64) - self.offs[tab]
        return bool(((i >= 0) & (i < self.lens[tab] - 1)).all())


class BitsBackCoder:
    def __init__(self, codec, delta: float = 1.0 / 8, k_max: int = 1024):
        """codec: an SGACodec created with bits_back=True (its HIP layers give h_s, its sga_bb_refine the posterior)."""
        if not codec.bits_back:
            raise ValueError("bits-back coding needs a codec of the mbt2018_bb model (bits_back=True)")
        self.codec, self.delta = codec, float(delta)
        self.yq = codec._entropy_coder()                    # unit-bin Gaussian tables: p(y | z) and, on z / delta, Q(z | y)
        self.q = _PosteriorTables(self.yq)                  # the same family without escape symbols (see _PosteriorTables)
        self.prior = _PriorGrid(codec._weights_for_ec, self.delta, k_max)
        self.C = codec.C

    # ---- the three distributions as (symbol, table) -----------------------------------------------------------
    def _q_tables(self, zml):
        mean, logvar = zml[..., :self.C], zml[..., self.C:]
        u_mean = (mean / self.delta).astype(np.float32)
        u_sigma = (np.exp(0.5 * logvar.astype(np.float64)) / self.delta).astype(np.float32)
        r0, tab = self.yq._y_symbols(None, u_mean, u_sigma)
        return r0, tab - self.q.tab0                        # (r0 = rint(u_mean), table index in self.q)

    def _y_tables(self, z_bar, yh, yw):
        mu, sigma = self.codec.hyper_synthesis(self.codec._t(z_bar), yh, yw)
        return self.yq._y_symbols(None, mu.cpu().numpy(), sigma.cpu().numpy())

    def _chan(self, shape):
        return np.broadcast_to(np.arange(self.C, dtype=np.int32), shape)

    # ---- sender / receiver --------------------------------------------------------------------------------------
    def encode(self, y_hat, zml, init_bytes: int = 0, seed: int = 0):
        """y_hat [B,yh,yw,C] integers, zml [B,zh,zw,2C] the refined posterior (bb_run / bb_refine).
        -> (message bytes, info dict incl. the z_bar that was coded)."""
        y_hat = np.asarray(y_hat.cpu() if hasattr(y_hat, "cpu") else y_hat, np.float32)
        zml = np.asarray(zml.cpu() if hasattr(zml, "cpu") else zml, np.float32)
        nz = zml[..., :self.C].size
        if init_bytes <= 0:
            init_bytes = 8 * nz // 8 + 64          # ~8 bits per element of z: far more than -log2 Q needs
        initial = np.random.RandomState(seed).bytes(init_bytes)
        st = AnsStack(initial, capacity=init_bytes + 16 + 8 * (y_hat.size + nz))
        bits0 = st.bits()
        r0q, tabq = self._q_tables(zml)
        k = st.pop(self.q, tabq).reshape(r0q.shape) + r0q                               # 1. z_bar ~ Q (escape-free tables)
        bits1 = st.bits()
        z_bar = (k.astype(np.float32) * np.float32(self.delta)).reshape(zml[..., :self.C].shape)
        r0y, taby = self._y_tables(z_bar, y_hat.shape[1], y_hat.shape[2])
        st.push(self.yq, ec.EntropyCoder._integers(y_hat, "y_hat").astype(np.int32) - r0y, taby)   # 2. y | z_bar
        bits2 = st.bits()
        st.push(self.prior, k, self._chan(k.shape))                                      # 3. z_bar under the prior
        bits3 = st.bits()
        info = dict(z_bar=z_bar, init_bits=bits0, bits_back=bits0 - bits1, y_bits=bits2 - bits1, z_bits=bits3 - bits2,
                    net_bits=bits3 - bits0)
        head = struct.pack("<4I4I", *y_hat.shape, *z_bar.shape)
        return head + st.tobytes(), info

    def decode(self, blob: bytes, H: int, W: int, r_its=2000, r_lr=0.003, seed=0, loss_scale=None):
        """-> (y_hat, z_bar, remaining stack bytes = the sender's initial bytes)."""
        v = struct.unpack("<4I4I", blob[:32])
        y_shape, z_shape = v[:4], v[4:]
        st = AnsStack.frombytes(blob[32:], capacity=len(blob) + 16 + 8 * int(np.prod(z_shape)))
        k = st.pop(self.prior, self._chan(z_shape)).reshape(z_shape)                     # 1. z_bar under the prior
        z_bar = k.astype(np.float32) * np.float32(self.delta)
        r0y, taby = self._y_tables(z_bar, y_shape[1], y_shape[2])
        y_hat = (st.pop(self.yq, taby).reshape(y_shape) + r0y).astype(np.float32)         # 2. y | z_bar
        zml = self.codec.bb_refine(y_hat, H, W, r_its=r_its, r_lr=r_lr, seed=seed, loss_scale=loss_scale)
        r0q, tabq = self._q_tables(zml.cpu().numpy())
        if not self.q.in_range((k - r0q).reshape(-1), tabq.reshape(-1)):
            raise ValueError("bits-back decode: z_bar lies outside the posterior's table -- the receiver's q(z | y) is not "
                             "the sender's (other weights, seed, r_its or r_lr?)")
        st.push(self.q, k - r0q, tabq)                                                    # 3. give the bits back
        return y_hat, z_bar, st.tobytes()[4:], st.x.value
