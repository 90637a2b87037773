"""Entropy coder throughput at Tecnick size: one 1200 x 1200 image at num_filters = 256 has
75 x 75 x 256 = 1.44 M y symbols (+ 92 K of z).  Times the device symbol / table kernels, the device rANS encode (one lane per
1024-symbol block), the compaction and the decode with hipEvents, and the host coder (csrc_cpu/rans.c) on the same symbols.
    python scripts/rans_throughput.py        (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
from sga_amd import entropy_coding as ec

C, H, W = 256, 1200, 1200
w = sga_amd.make_synthetic_weights(C, 0)
codec = SGACodec(w, C, 1, H, W)
coder = codec._entropy_coder()
yh, yw, zh, zw = codec.latent_shape(H, W)
rng = np.random.RandomState(0)
n = yh * yw * C
mu = (rng.standard_normal(n) * 2).astype(np.float32)
sigma = np.exp(rng.standard_normal(n) * 1.0).astype(np.float32)
y = np.rint(mu + sigma * rng.standard_normal(n)).astype(np.float32)
yt, mt, st = codec._t(y), codec._t(mu), codec._t(sigma)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


t_sym, o = timed(lambda: codec._ec_symbols_device(coder, yt, mt, st, None))
sym, tab = o["y_sym"], o["y_tab"]
t_enc, data = timed(lambda: codec._ec_encode_device(coder, sym, tab))
t_dec, back = timed(lambda: codec._ec_decode_device(coder, data, tab))
assert torch.equal(back, sym)
# kernel-only times (hipEvents around the launches)
cdf, lens, offs, _ = codec._ec_tables(coder)
nb = -(-n // ec.BLOCK); cap = 16 + 8 * ec.BLOCK
slots = torch.empty(nb * cap, dtype=torch.uint8, device="cuda"); bb = torch.empty(nb, dtype=torch.int32, device="cuda")
from sga_amd import _lib
from sga_amd.codec import _ptr
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s = codec._enter()
with torch.cuda.stream(codec.stream):
    e0.record(codec.stream)
    for _ in range(10):
        codec.lib.sga_ec_encode(_ptr(sym), _ptr(tab), n, ec.BLOCK, _ptr(cdf), _ptr(lens), _ptr(offs), coder.stride, _ptr(slots), cap, _ptr(bb), s)
    e1.record(codec.stream)
codec._exit(); torch.cuda.synchronize()
k_enc = e0.elapsed_time(e1) / 10 * 1e-3
sy, ty = sym.cpu().numpy(), tab.cpu().numpy()
t0 = time.perf_counter(); host = coder._run_encode(sy, ty); t_host = time.perf_counter() - t0
t0 = time.perf_counter(); hb = coder._run_decode(host, ty); t_hdec = time.perf_counter() - t0
assert host == data and np.array_equal(hb, sy)
mb = len(data) / 1e6
print("symbols %d (%d blocks of %d), stream %.3f MB (%.3f bits/symbol)" % (n, nb, ec.BLOCK, mb, 8 * len(data) / n))
print("device: symbol/table kernels %.3f ms; encode kernel alone %.3f ms = %.1f Msym/s = %.1f MB/s; encode incl. compaction + D2H %.2f ms; decode incl. H2D %.2f ms"
      % (t_sym * 1e3, k_enc * 1e3, n / k_enc / 1e6, mb / k_enc, t_enc * 1e3, t_dec * 1e3))
print("host (csrc_cpu/rans.c, 1 thread): encode %.2f ms = %.1f Msym/s = %.1f MB/s; decode %.2f ms" % (t_host * 1e3, n / t_host / 1e6, mb / t_host, t_hdec * 1e3))
codec.close()
