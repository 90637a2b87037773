"""bf16x3: the pipelined 16-wide K loop (conv_mfma.hip X3P) against a reference build of the single-stage loop
(SGA_X3_REF=<path to the older libsga_hip.so>): same operands, same plane order, same k order -> bit-identical gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sga_amd
from sga_amd import _lib
from sga_amd.codec import SGACodec
ref_path = os.environ.get("SGA_X3_REF", os.path.join(os.path.dirname(_lib.LIB_PATH), "libsga_hip_x3ref.so"))
ok = True
for (C, B, H, W) in [(192, 8, 256, 256), (192, 1, 200, 264), (64, 2, 64, 80), (256, 1, 520, 504)]:
    w = sga_amd.make_synthetic_weights(C, 0)
    x = np.random.RandomState(C + H).rand(B, H, W, 3).astype(np.float32)
    new = SGACodec(w, C, B, H, W, precision="bf16x3")
    old = SGACodec(w, C, B, H, W, precision="bf16x3")
    old.close(); old.lib = _lib.load_library(ref_path)          # re-create the handle in the reference build
    old.__init__(w, C, B, H, W, precision="bf16x3") if False else None
    # (SGACodec binds its library in __init__: build a second codec object by hand)
    import types
    ref = SGACodec.__new__(SGACodec)
    _orig = _lib.load_library
    _lib.load_library = lambda path=None: _orig(ref_path)
    try:
        ref.__init__(w, C, B, H, W, precision="bf16x3")
    finally:
        _lib.load_library = _orig
    y, z = new.encode(x); y2, z2 = ref.encode(x)
    a = new.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5); b = ref.step_grads(x, y, z, 0.4, 0.01, seed=3, it=5)
    ra = new.run(x, 0.01, its=30, seed=1); rb = ref.run(x, 0.01, its=30, seed=1)
    e = dict(enc=torch.equal(y, y2) and torch.equal(z, z2), gy=torch.equal(a["gy"], b["gy"]), gz=torch.equal(a["gz"], b["gz"]),
             run=torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]))
    print((C, B, H, W), e, "max|gy diff|", float((a["gy"] - b["gy"]).abs().max()), flush=True)
    ok = ok and all(e.values())
    new.close(); ref.close()
print("x3 loops bit-identical:", ok)
