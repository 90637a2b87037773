"""TEST INFRASTRUCTURE (oracle) -- range-variant asymmetric numeral system (rANS), pure Python.

Restated from the published algorithm (J. Duda, "Asymmetric numeral systems", arXiv:1311.2540, sec. 4 "range variants";
byte-wise renormalisation as in F. Giesen, "Interleaved entropy coders", arXiv:1402.3392): state x in [L, 256 L) with
L = 2^23, probabilities of n = 16 bits.  For a symbol with cumulative start c and frequency f

    encode:  while x >= ((L >> n) << 8) f:  emit x & 255, x >>= 8          (bytes are emitted BACKWARDS)
             x = (x // f << n) + (x % f) + c
    decode:  s = x & (2^n - 1);  symbol = the one with c <= s < c + f
             x = f (x >> n) + s - c;  while x < L:  x = x << 8 | next byte

The reference turns latents into bytes with tensorflow-compression's C++ range coder (mbt2018.py:84-85,211-222); this
build's coders (csrc_cpu/rans.c on the host, csrc/rans.hip on the device) are checked against THIS file -- same stream
layout: a stream is the final state as 4 big-endian bytes followed by the renormalisation bytes in decoding order; a
value outside its table is the table's last ("escape") symbol followed by the zig-zag code of the value as two raw
16-bit symbols (frequency 1, low half decoded first); `blocked` streams are independent streams of `block` symbols.
Only tests/ may import this module.  Python integers: no overflow to reason about; seconds for ~1e5 symbols.
"""
L = 1 << 23
N = 16
MASK = (1 << N) - 1


def _put(x, out, c, f):
    x_max = ((L >> N) << 8) * f
    while x >= x_max:
        out.append(x & 0xFF)
        x >>= 8
    return ((x // f) << N) + (x % f) + c


def encode(sym, tab, cdf, lens, offs):
    """sym[i] coded with table tab[i]: cdf[t] = prefix sums (lens[t] + 1 entries, total 2^16), lens[t] - 1 regular
    symbols for the values offs[t] .. offs[t] + lens[t] - 2, then the escape symbol.  Returns bytes."""
    out = []                       # emitted backwards
    x = L
    for k in range(len(sym) - 1, -1, -1):
        t, v = int(tab[k]), int(sym[k])
        c, n = cdf[t], int(lens[t])
        i = v - int(offs[t])
        if 0 <= i < n - 1:
            x = _put(x, out, int(c[i]), int(c[i + 1]) - int(c[i]))
        else:
            z = ((v << 1) ^ (v >> 31)) & 0xFFFFFFFF
            x = _put(x, out, z >> 16, 1)
            x = _put(x, out, z & 0xFFFF, 1)
            x = _put(x, out, int(c[n - 1]), int(c[n]) - int(c[n - 1]))
    head = [(x >> 24) & 0xFF, (x >> 16) & 0xFF, (x >> 8) & 0xFF, x & 0xFF]
    return bytes(head + out[::-1])


def decode(data, tab, cdf, lens, offs):
    """Inverse of `encode`: len(tab) symbols.  Raises ValueError on a stream that cannot have come from `encode`."""
    if len(data) < 4:
        raise ValueError("rans_ref: truncated stream")
    x = int.from_bytes(data[:4], "big")
    p = 4

    def renorm(x, p):
        while x < L and p < len(data):
            x = (x << 8) | data[p]
            p += 1
        return x, p

    out = []
    for k in range(len(tab)):
        t = int(tab[k])
        c, n = cdf[t], int(lens[t])
        s = x & MASK
        lo, hi = 0, n                       # largest index with c[lo] <= s
        while hi - lo > 1:
            mid = (lo + hi) >> 1
            if int(c[mid]) <= s:
                lo = mid
            else:
                hi = mid
        start, f = int(c[lo]), int(c[lo + 1]) - int(c[lo])
        if f <= 0:
            raise ValueError("rans_ref: zero-frequency symbol")
        x, p = renorm(f * (x >> N) + s - start, p)
        if lo < n - 1:
            out.append(int(offs[t]) + lo)
        else:
            zl = x & MASK
            x, p = renorm(x >> N, p)
            zh = x & MASK
            x, p = renorm(x >> N, p)
            z = (zh << 16) | zl
            out.append((z >> 1) ^ -(z & 1))
    return out


def encode_blocked(sym, tab, block, cdf, lens, offs):
    """(list of per-block byte counts, payload) of independent streams of `block` symbols each."""
    sizes, payload = [], b""
    for s in range(0, len(sym), block):
        b = encode(sym[s:s + block], tab[s:s + block], cdf, lens, offs)
        sizes.append(len(b))
        payload += b
    return sizes, payload


def decode_blocked(payload, sizes, tab, block, cdf, lens, offs):
    out, off = [], 0
    for b, s in enumerate(range(0, len(tab), block)):
        out += decode(payload[off:off + sizes[b]], tab[s:s + block], cdf, lens, offs)
        off += sizes[b]
    return out
