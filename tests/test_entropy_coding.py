"""CPU tests of the entropy coder: exact decode round trip, coded size close to
the model's ideal code length, escape path, corrupt-stream detection."""
import numpy as np
import pytest

import sga_amd
from sga_amd import entropy_coding as ec


@pytest.fixture(scope="module")
def coder():
    return ec.EntropyCoder(sga_amd.make_synthetic_weights(64, seed=0))


def test_tables_are_valid(coder):
    for t in range(coder.cdf.shape[0]):
        c = coder.cdf[t, :coder.lens[t] + 1].astype(np.int64)
        assert c[0] == 0 and c[-1] == ec.TOTAL and (np.diff(c) >= 1).all()


def test_z_round_trip_and_size(coder):
    rng = np.random.RandomState(0)
    z = np.rint(rng.standard_normal((2, 3, 4, 64)) * 6).astype(np.float32)
    z[0, 0, 0, :4] = [500, -321, 97, -97]                        # outside every table: escapes
    blob = coder.encode_z(z)
    assert np.array_equal(coder.decode_z(blob, z.shape), z)
    w = sga_amd.make_synthetic_weights(64, seed=0)
    ks = np.arange(-96, 97, dtype=np.float64)
    mass = ec.factorized_mass(w, ks)
    zi = z.astype(np.int64).reshape(-1, 64)
    inr = np.abs(zi) <= 96
    ideal = -np.log2(mass[np.clip(zi, -96, 96) + 96, np.arange(64)[None, :]])[inr].sum() + 32 * (~inr).sum()
    assert 8 * len(blob) < ideal * 1.03 + 64


def test_y_round_trip_and_size(coder):
    rng = np.random.RandomState(1)
    shape = (2, 5, 7, 64)
    mu = (rng.standard_normal(shape) * 2).astype(np.float32)
    sigma = np.exp(rng.standard_normal(shape) * 1.2).astype(np.float32)
    sigma.reshape(-1)[:50] = 0.02                                 # below the 0.11 bound
    sigma.reshape(-1)[50:60] = 300.0                              # above the table
    y = np.rint(mu + sigma * rng.standard_normal(shape)).astype(np.float32)
    y.reshape(-1)[100:104] = [4000, -4000, 1e5, -1e5]             # escapes
    blob = coder.encode_y(y, mu, sigma)
    assert np.array_equal(coder.decode_y(blob, mu, sigma), y)
    ideal = coder.ideal_bits_y(y, mu, sigma)
    nblocks = -(-y.size // ec.BLOCK)                              # 8 bytes per independent block + the 8-byte frame header
    assert ideal <= 8 * len(blob) <= ideal * 1.01 + 64 + 64 * (nblocks + 1)      # rANS overhead is tiny
    # quantising (sigma, frac(mu)) costs only a few % over the exact model
    from math import erfc, sqrt
    sb = np.maximum(sigma.astype(np.float64), 0.11)
    d = np.abs(y - mu).astype(np.float64)
    phi = np.vectorize(lambda t: 0.5 * erfc(-t / sqrt(2)))
    p = np.maximum(phi((0.5 - d) / sb) - phi((-0.5 - d) / sb), 1e-9)
    keep = np.ones(y.size, bool); keep[:60] = False; keep[100:104] = False
    exact = -np.log2(p.reshape(-1)[keep]).sum()
    yk, mk, sk = (a.reshape(-1)[keep] for a in (y, mu, sigma))
    assert coder.ideal_bits_y(yk, mk, sk) < exact * 1.08 + 100


def test_block_size_follows_the_rate(coder):
    """A trained codec's y_hat is ~90 % zeros under sigma ~ 0.1: a 1024-symbol block then holds a handful of bytes and its 8
    bytes of framing dominate.  The encoder's second pass doubles the block until the mean payload is >= 512 bytes
    (`ec.adapted_block`); dense latents keep 1024.  Exact round trip either way, the block size travels in the frame."""
    rng = np.random.RandomState(5)
    shape = (4, 16, 16, 64)
    mu = (rng.standard_normal(shape) * 0.05).astype(np.float32)
    sigma = np.full(shape, 0.1, np.float32)
    y = np.where(rng.rand(*shape) < 0.02, rng.randint(-3, 4, shape), 0).astype(np.float32)
    r0, tab = coder._y_symbols(y, mu, sigma)
    sym = y.astype(np.int32) - r0
    fixed = coder._run_encode(sym, tab, block=ec.BLOCK)
    auto = coder.encode_y(y, mu, sigma)
    bb_f, blk_f, _ = ec.unframe_blocks(fixed)
    bb_a, blk_a, _ = ec.unframe_blocks(auto)
    assert blk_f == ec.BLOCK and blk_a > ec.BLOCK and blk_a == ec.adapted_block(bb_f) and bb_a.size < bb_f.size
    assert np.array_equal(coder.decode_y(auto, mu, sigma), y) and np.array_equal(coder.decode_y(fixed, mu, sigma), y)
    ideal = coder.ideal_bits_y(y, mu, sigma)
    assert 8 * len(auto) < ideal * 1.02 + 64 + 64 * (bb_a.size + 1) and len(auto) < 0.9 * len(fixed)
    # dense latents: the first pass is the stream
    sig2 = np.full(shape, 8.0, np.float32)
    y2 = np.rint(mu + sig2 * rng.standard_normal(shape)).astype(np.float32)
    _, blk2, _ = ec.unframe_blocks(coder.encode_y(y2, mu, sig2))
    assert blk2 == ec.BLOCK
    assert ec.adapted_block(np.array([11, 12, 10])) == ec.BLOCK_MAX and ec.adapted_block(np.array([600, 700])) == ec.BLOCK


def test_centred_latents_round_trip_and_size():
    """What mbt2018.py compress codes (mbt2018.py:69,80): y_hat = round(y - mu) + mu under the zero-offset table of its scale
    level, z_hat = round(z - median) + median on the grid median + k.  Exact round trip INCLUDING the float32 re-centring, size at
    the quantised model's ideal, table CRC distinct from the integer coder's, integer latents refused (and vice versa)."""
    w = sga_amd.make_synthetic_weights(64, seed=0)
    med = np.linspace(-0.45, 0.45, 64).astype(np.float32)
    cc = ec.EntropyCoder(w, centred=True, medians=med)
    ic = ec.EntropyCoder(w)
    assert cc.table_mode == 2 and ic.table_mode == 0 and cc.table_crc() != ic.table_crc()
    rng = np.random.RandomState(4)
    shape = (2, 6, 5, 64)
    mu = (rng.standard_normal(shape) * 3).astype(np.float32)
    sigma = np.exp(rng.standard_normal(shape) * 1.2).astype(np.float32)
    y = (mu + sigma * rng.standard_normal(shape).astype(np.float32)).astype(np.float32)
    y_hat = np.rint(y - mu) + mu                                       # float32, as the device computes it
    y_hat.reshape(-1)[7] = mu.reshape(-1)[7] + 5000.0                  # an escape
    z = (rng.standard_normal((2, 2, 2, 64)) * 4).astype(np.float32)
    z_hat = np.rint(z - med) + med
    zb, yb = cc.encode_z(z_hat), cc.encode_y(y_hat, mu, sigma)
    assert np.array_equal(cc.decode_z(zb, z_hat.shape), z_hat) and np.array_equal(cc.decode_y(yb, mu, sigma), y_hat)
    ideal = cc.ideal_bits_y(y_hat, mu, sigma)
    assert ideal <= 8 * len(yb) <= ideal * 1.01 + 64 + 64 * 5
    # the exact model: no mean-bin quantisation in this mode, only the 64 scale levels
    from math import erfc, sqrt
    phi = np.vectorize(lambda t: 0.5 * erfc(-t / sqrt(2)))
    sb = np.maximum(sigma.astype(np.float64), 0.11)
    k = np.rint(y_hat - mu).astype(np.float64)
    p = np.maximum(phi((k + 0.5) / sb) - phi((k - 0.5) / sb), 1e-9)
    keep = np.ones(y.size, bool); keep[7] = False
    exact = -np.log2(p.reshape(-1)[keep]).sum()
    assert cc.ideal_bits_y(y_hat.reshape(-1)[keep], mu.reshape(-1)[keep], sigma.reshape(-1)[keep]) < exact * 1.03 + 100
    with pytest.raises(ValueError):
        cc.encode_y(np.rint(y), mu, sigma)                             # integers are not mu + integers
    with pytest.raises(ValueError):
        ic.encode_y(y_hat, mu, sigma)
    with pytest.raises(ValueError):
        cc.encode_z(np.rint(z))


def test_container_and_corruption(coder):
    z = np.rint(np.random.RandomState(2).standard_normal((1, 2, 2, 64)) * 3).astype(np.float32)
    zb = coder.encode_z(z)
    blob = ec.pack((1, 64, 64), (1, 4, 4, 64), z.shape, zb, b"abc")
    xs, ys, zs, zb2, yb2 = ec.unpack(blob)
    assert xs == (1, 64, 64) and zs == z.shape and zb2 == zb and yb2 == b"abc"
    with pytest.raises(ValueError):
        ec.unpack(b"nope" + blob[4:])
    # format version and table fingerprint travel with the stream (a range coder needs identical tables)
    blob = ec.pack((1, 64, 64), (1, 4, 4, 64), z.shape, zb, b"abc", coder.table_mode, coder.table_crc())
    *_, mode, crc = ec.unpack(blob, with_tables=True)
    assert mode == 0 and crc == coder.table_crc() and crc != 0
    import sga_amd
    w2 = sga_amd.make_synthetic_weights(64, seed=1)                    # another model: another fingerprint
    assert ec.EntropyCoder({k: v for k, v in w2.items() if k.startswith("eb.")}).table_crc() != crc
    with pytest.raises(ValueError):
        ec.unpack(blob[:4] + bytes([1]) + blob[5:])            # a round-2 (format 1) stream is refused
    with pytest.raises(ValueError):
        coder.decode_z(zb[:3], z.shape)


def test_ans_stack_push_and_pop_are_inverse(coder):
    """rANS as a stack (bits_back.AnsStack over csrc_cpu/rans.c): pop after push returns the symbols and the stack;
    pop FIRST (sampling from the tables with the stack's bits, the bits-back move) then push restores the stack."""
    from sga_amd.bits_back import AnsStack
    rng = np.random.RandomState(4)
    n = 5000
    tab = rng.randint(coder.y_tab0, coder.cdf.shape[0], n).astype(np.int32)
    sym = (coder.offs[tab] + rng.randint(0, 3, n) * (coder.lens[tab] - 2) // 2).astype(np.int32)
    sym[::97] = 30000                                                   # escapes
    init = rng.bytes(6000)
    st = AnsStack(init, 6000 + 8 * n + 64)
    st.push(coder, sym, tab)
    assert st.len.value > 6000
    assert np.array_equal(st.pop(coder, tab), sym)
    assert st.tobytes()[4:] == init and st.x.value == 1 << 23
    # sample first, then put the sample back
    got = st.pop(coder, tab)
    assert st.len.value < 6000                                          # bits were consumed
    idx = got - coder.offs[tab]
    assert ((idx >= 0) & (idx < coder.lens[tab])).mean() > 0.99         # samples land inside their tables
    st.push(coder, got, tab)
    assert st.tobytes()[4:] == init and st.x.value == 1 << 23
    with pytest.raises(RuntimeError):
        AnsStack(b"", 64).pop(coder, tab)                               # nothing to sample from


def test_unpack_refuses_malformed_blobs(coder):
    """a truncated or inconsistent container is a ValueError, never struct.error or silently short streams."""
    blob = ec.pack((1, 64, 64), (1, 4, 4, 64), (1, 1, 1, 64), b"zz", b"yyyy", 1, 0x1234)
    assert ec.unpack(blob, with_tables=True)[3:] == (b"zz", b"yyyy", 1, 0x1234)
    for bad in (b"", b"SGA", blob[:8], blob[:40], blob[:57], blob[:-1], blob + b"\0", blob[:56] + b"\xff\xff\xff\x7f" + blob[60:]):
        with pytest.raises(ValueError):
            ec.unpack(bad)
    with pytest.raises(ValueError, match="format 1"):
        ec.unpack(blob[:4] + b"\x01" + blob[5:])
    with pytest.raises(ValueError, match="format 2"):              # rounds 3-5: no precision mode in the mode byte
        ec.unpack(blob[:4] + b"\x02" + blob[5:])
    # the mode byte: bits 2-3 name the arithmetic of the encoder's h_s, the other bits must be zero
    for prec, name in enumerate(ec.MODE_PRECISIONS):
        b2 = ec.pack((1, 64, 64), (1, 4, 4, 64), (1, 1, 1, 64), b"zz", b"yyyy", 1 | (prec << ec.MODE_PRECISION_SHIFT), 0x1234)
        assert ec.mode_precision(ec.unpack(b2, with_tables=True)[5]) == name
    for bad_mode in (0x10, 0x80, 0x0C | 1):                        # unknown bits; precision code 3
        with pytest.raises(ValueError, match="mode byte"):
            ec.unpack(blob[:5] + bytes([bad_mode]) + blob[6:])
    assert ec.adapted_block(np.array([3, 4, 5])) <= 8192           # a block is one device lane: never fewer, bigger lanes than this


def test_host_coder_equals_the_rans_oracle_byte_for_byte(coder):
    """csrc_cpu/rans.c against oracle/rans_ref.py (the published rANS recurrences in Python integers): identical bytes for
    identical tables, escapes included, and the oracle decodes what the host coder wrote (and vice versa)."""
    from oracle import rans_ref
    rng = np.random.RandomState(5)
    n = 3000
    tab = rng.randint(0, coder.cdf.shape[0], n).astype(np.int32)
    span = coder.lens[tab] - 1
    sym = (coder.offs[tab] + (rng.rand(n) * span).astype(np.int32)).astype(np.int32)
    sym[::97] = rng.randint(-70000, 70000, sym[::97].size)               # far outside every table: escapes
    sym[5], sym[6] = -2 ** 31 + 1, 2 ** 31 - 1
    data = coder._run_encode(sym, tab)
    bb, block, payload = ec.unframe_blocks(data)
    sizes, ref_payload = rans_ref.encode_blocked(sym.tolist(), tab.tolist(), block, coder.cdf, coder.lens, coder.offs)
    assert sizes == bb.tolist() and ref_payload == payload
    assert rans_ref.decode_blocked(payload, bb.tolist(), tab.tolist(), block, coder.cdf, coder.lens, coder.offs) == sym.tolist()
    assert np.array_equal(coder._run_decode(ec.frame_blocks(np.asarray(sizes, np.uint32), ref_payload, block), tab), sym)
    # code length: within 1 % + the per-block flush of the ideal length of the quantised tables
    idx = sym.astype(np.int64) - coder.offs[tab]
    esc = (idx < 0) | (idx >= coder.lens[tab] - 1)
    idx = np.where(esc, coder.lens[tab] - 1, idx)
    f = coder.cdf[tab, idx + 1].astype(np.float64) - coder.cdf[tab, idx]
    ideal = -np.log2(f / ec.TOTAL).sum() + 32.0 * esc.sum()
    assert ideal <= 8 * len(payload) <= ideal * 1.01 + 64 * len(sizes)


def test_bits_back_posterior_tables_have_no_escape(coder):
    """ADVICE r3 (bits_back.py): sampling z_bar ~ Q by POPPING from the stack must never land in an escape symbol -- with
    the conditional's own tables (escape frequency >= 1 / 65536 in every table) it does once in ~2^16 elements, i.e. in one
    Kodak-size image (18 432 elements of z) out of four, and then reads 32 raw stack bits as the value.  The posterior's
    tables fold the tails into the edge bins: every 16-bit state slot maps to a regular symbol, and a pop / push round
    trip over 10 Kodak images' worth of elements returns the stack bit for bit."""
    from sga_amd import bits_back as bb
    q = bb._PosteriorTables(coder)
    for t in range(q.cdf.shape[0]):
        c = q.cdf[t, :q.lens[t] + 1].astype(np.int64)
        d = np.diff(c)
        assert c[0] == 0 and c[-1] == ec.TOTAL and (d[:-1] >= 1).all() and d[-1] == 0
    esc = coder.cdf[coder.y_tab0:, :].astype(np.int64)
    assert all(esc[t, coder.lens[coder.y_tab0 + t]] - esc[t, coder.lens[coder.y_tab0 + t] - 1] >= 1 for t in range(512))
    n = 184320
    init = np.random.RandomState(0).bytes(2 * n)
    st = bb.AnsStack(init, 2 * n + 16 + 8 * n)
    tab = np.random.RandomState(1).randint(0, 512, n).astype(np.int32)
    before = st.tobytes()
    sym = st.pop(q, tab)
    assert q.in_range(sym, tab) and np.abs(sym).max() <= 1537          # never a raw 32-bit value
    # the same draw through the conditional's OWN tables does hit escapes at this size
    st2 = bb.AnsStack(init, 2 * n + 16 + 8 * n)
    sym2 = st2.pop(coder, tab + coder.y_tab0)
    assert np.abs(sym2).max() > 1 << 20
    st.push(q, sym, tab)
    assert st.tobytes() == before
