"""TEST INFRASTRUCTURE (oracle) -- not a product path.

Philox4x32-10 counter RNG in numpy, bit-identical to the device generator in
`csrc/sga_common.h` (`philox4x32_10`).  The reference draws its Gumbel noise
from TF's graph-seeded RandomUniform (sga.py:15-17,97,120), a stream that cannot
be reproduced outside TF; the build therefore defines its own counter RNG and
the oracle restates it so the GPU path and the oracle consume identical noise.

Counter layout (one Philox call per latent element):
    ctr = (elem_idx_lo, elem_idx_hi, step, stream)   stream: 0 = y, 1 = z
    key = (seed_lo, seed_hi)
Outputs [0], [1] become the two uniforms (down, up) of the element via
    u = ((bits >> 9) + 0.5) * 2^-23      in [2^-24, 1 - 2^-24]
(the counterpart of tfp's U(tiny, 1) draw in RelaxedOneHotCategorical.sample).
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable uint32 arrays / ints. Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32),
            c2.astype(np.uint32), c3.astype(np.uint32))


def bits_to_uniform(bits):
    """uint32 -> float32 in (0,1): ((bits >> 9) + 0.5) * 2^-23.  23 random bits so that the
    +0.5 is exact in f32 (24-bit mantissa): u in [2^-24, 1 - 2^-24], never 0 or 1."""
    return ((bits >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)


def sga_uniforms(n_elems, step, stream, seed):
    """Uniform pairs for `n_elems` latent elements (flat NHWC order) at SGA step `step`.

    Returns float32 [n_elems, 2]: [:,0] feeds the DOWN (floor) logit, [:,1] the UP (ceil) logit.
    """
    idx = np.arange(n_elems, dtype=np.uint64)
    lo = (idx & _MASK).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    seed = int(seed)
    r0, r1, _, _ = philox4x32_10(lo, hi, np.uint32(step), np.uint32(stream),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack([bits_to_uniform(r0), bits_to_uniform(r1)], axis=-1)


def sga_normals(n_elems, step, stream, seed):
    """Standard normals (Box-Muller on Philox outputs [2],[3]) for the bits-back path
    (bb_sga.py:99-100: eps ~ N(0,1)).  float32 [n_elems]."""
    idx = np.arange(n_elems, dtype=np.uint64)
    lo = (idx & _MASK).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    seed = int(seed)
    _, _, r2, r3 = philox4x32_10(lo, hi, np.uint32(step), np.uint32(stream),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = bits_to_uniform(r2)
    u2 = bits_to_uniform(r3)
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    return (rad * np.cos(np.float32(2.0 * np.pi) * u2)).astype(np.float32)
