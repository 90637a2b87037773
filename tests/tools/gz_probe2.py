"""GPU box: where the 4.8e-4 gz deviation of the f32 path at B = 4, 256x256 sits and which launch knob moves it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from oracle import philox
from oracle.sga_oracle import SGAOracle
from sga_amd.codec import SGACodec
C, B, H, W = 192, 4, 256, 256
w = sga_amd.make_synthetic_weights(C, seed=0)
seed, it, T, lm = 5, 7, 0.25, 0.02
x = np.random.RandomState(100 + B).rand(B, H, W, 3).astype(np.float32)
o32, o64 = SGAOracle(w), SGAOracle(w, dtype=torch.float64)
yo, zo = o32.encode(x)
u_y = philox.sga_uniforms(yo.numel(), it, 0, seed); u_z = philox.sga_uniforms(zo.numel(), it, 1, seed)
want = o64.step(x, yo, zo, T, u_y, u_z, lm)
gz64 = want["gz"].numpy(); mx = np.abs(gz64).max()
first = True
for env in ["", "SGA_SIDE_TARGET=0", "SGA_SIDE_TARGET=128", "SGA_SIDE_TARGET=256", "SGA_SIDE_TARGET=512", "SGA_SIDE_TARGET=768",
            "SGA_BM64_MAX=0", "SGA_REDUCE_BATCH=0", "SGA_PLAN_TILES=0", "SGA_XCD_REMAP=0", "SGA_FUSED_BOUNDARY=0", "SGA_BN96_AS_192=0"]:
    kv = dict(e.split("=") for e in env.split()) if env else {}
    os.environ.update(kv)
    c = SGACodec(w, C, B, H, W, lab=True)
    got = c.step_grads(x, yo.numpy(), zo.numpy(), T, lm, seed=seed, it=it)
    gz = got["gz"].cpu().numpy().astype(np.float64)
    e = np.abs(gz - gz64) / mx
    print("%-24s gz err per image %s" % (env or "(default)", " ".join("%.1e" % e[b].max() for b in range(B))), flush=True)
    if first:
        first = False
        idx = np.argsort(-e.ravel())[:12]
        for i in idx:
            b, zy, zx, ch = np.unravel_index(i, e.shape)
            print("   worst: img %d pos (%d,%d) ch %3d  got % .6e want % .6e  err/max %.1e" % (b, zy, zx, ch, gz[b, zy, zx, ch], gz64[b, zy, zx, ch], e[b, zy, zx, ch]))
        print("   elements with err > 5e-5:", int((e > 5e-5).sum()), "of", e.size, "; by image", [(int((e[b] > 5e-5).sum())) for b in range(B)])
        print("   by channel (count > 5e-5):", np.nonzero((e > 5e-5).sum(axis=(0, 1, 2)))[0][:40])
        print("   by position img0:", (e[0] > 5e-5).sum(axis=2).tolist())
    c.close()
    for k in kv: del os.environ[k]
