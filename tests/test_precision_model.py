"""CPU: the arithmetic model of the two bf16-pipe precision modes (include/sga_hip.h: SGA_PRECISION_BF16X3 / BF16X2), restated in
numpy, against float64 -- so that the claims in the header and in DESIGN.md 3.6 ("exact three-plane split", "f32-grade",
"2^-16 per product, 32 x finer than TF32") are checked facts, independent of the GPU.  The kernels do exactly this: planes by
round-to-nearest-even bf16 conversion of the running remainder (conv_mfma.hip split3 / split2), plane products exact in float32
(8 x 8 significant bits), float32 accumulation."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def planes(x, n):
    out, rem = [], np.asarray(x, np.float32)
    for _ in range(n):
        p = bf16_rne(rem)
        out.append(p)
        rem = (rem - p).astype(np.float32)          # exact: p carries the leading bits of rem
    return out


def test_three_planes_are_exact_and_two_planes_keep_16_bits():
    rng = np.random.RandomState(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    h, m, l = planes(x, 3)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)          # 8 + 8 + 8 bits = the float32 mantissa
    h2, m2 = planes(x, 2)
    rel = np.abs((h2.astype(np.float64) + m2) - x) / np.abs(x)
    assert rel.max() <= 2.0 ** -17 * 1.0001            # round to nearest at 16 significant bits
    tf32 = np.abs((np.asarray(x).view(np.uint32) & 0xFFFFE000).view(np.float32).astype(np.float64) - x) / np.abs(x)
    assert tf32.max() > 30 * rel.max()                   # TF32 keeps 10 explicit mantissa bits (truncation shown; RNE: 2^-11)


def _dot(a_planes, b_planes, pairs):
    acc = np.zeros(a_planes[0].shape[0], np.float32)
    K = a_planes[0].shape[1]
    for k0 in range(0, K, 16):                           # one v_mfma_f32_32x32x16_bf16 per 16 k's and plane pair, float32 accumulate
        for (i, j) in pairs:
            prod = (a_planes[i][:, k0:k0 + 16].astype(np.float64) * b_planes[j][:, k0:k0 + 16]).sum(1)
            acc = (acc + prod.astype(np.float32)).astype(np.float32)
    return acc


def test_dot_product_error_of_the_two_modes():
    """K = 4800 (5x5 taps x 192 channels): bf16x3's six products give float32-GEMM accuracy; bf16x2's three stay within a few
    2^-16 of the result scale -- the bounds the GPU parity tests assert per layer (2e-5 / 5e-5)."""
    rng = np.random.RandomState(1)
    M, K = 256, 4800
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    ref = (a.astype(np.float64) * b).sum(1)
    scale = np.abs(ref).max()
    f32 = np.zeros(M, np.float32)
    for k in range(K):
        f32 = (f32 + a[:, k] * b[:, k]).astype(np.float32)            # an f32 FMA-chain GEMM (up to fused rounding)
    e_f32 = np.abs(f32 - ref).max() / scale
    a3, b3 = planes(a, 3), planes(b, 3)
    x3 = _dot(a3, b3, [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])      # smallest products first, as in the kernels
    x2 = _dot(a3[:2], b3[:2], [(1, 0), (0, 1), (0, 0)])
    e3, e2 = np.abs(x3 - ref).max() / scale, np.abs(x2 - ref).max() / scale
    assert e3 <= max(2 * e_f32, 1e-6), (e3, e_f32)      # f32-grade
    assert e2 <= 2e-5, e2                               # 16-bit operands: within the per-layer bound of the GPU tests
    assert e2 > e3                                      # ... and really the coarser mode
