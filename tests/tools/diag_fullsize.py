"""GPU-box diagnostic: parity of one SGA step and of the encoder at the full bench size, and the
GPU trace of a long run.  (tests/-style use of the oracle as checker.)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sga_amd
from sga_amd.codec import SGACodec, metrics_to_dict
from oracle.sga_oracle import SGAOracle
from oracle import philox

torch.set_num_threads(32)
C, B, H, W = 192, int(os.environ.get("DIAG_B", 8)), 256, 256
w = sga_amd.make_synthetic_weights(C, 0)
codec = SGACodec(w, C, B, H, W)
orc = SGAOracle(w)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).numpy()

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

t = time.time(); yo, zo = orc.encode(x); print("oracle encode s", time.time() - t, flush=True)
y, z = codec.encode(x)
print("encode rel err", rel(y.cpu().numpy(), yo.numpy()), rel(z.cpu().numpy(), zo.numpy()), flush=True)
for it, T in ((0, 0.5), (1500, 0.2)):
    u_y = philox.sga_uniforms(yo.numel(), it, 0, 5); u_z = philox.sga_uniforms(zo.numel(), it, 1, 5)
    t = time.time(); ref = orc.step(x, yo, zo, T, u_y, u_z, 0.01); print("oracle step s", time.time() - t, flush=True)
    got = codec.step_grads(x, yo.numpy(), zo.numpy(), T, 0.01, seed=5, it=it)
    print("it", it, "gy", rel(got["gy"].cpu().numpy(), ref["gy"].numpy()), "gz", rel(got["gz"].cpu().numpy(), ref["gz"].numpy()),
          "loss", got["rd_loss"], ref["rd_loss"], "mse", got["train_mse"], ref["train_mse"], "bpp", got["train_bpp"], ref["train_bpp"], flush=True)
    # per-image breakdown of gradient error
    gy, ry = got["gy"].cpu().numpy(), ref["gy"].numpy()
    print("  per-image gy err", [round(rel(gy[b], ry[b]), 8) for b in range(B)], flush=True)
for its in (100, 2000):
    torch.cuda.synchronize(); t = time.time()
    y_hat, z_hat, met, tr = codec.run(x, 0.01, its=its, seed=100, trace=True)
    torch.cuda.synchronize(); print("run its", its, "s", time.time() - t)
    tr = tr.cpu().numpy()
    for i in sorted(set([0, 1, 2, 5, 10, 20, 50, 99, 200, 400, 700, 1000, 1300, 1600, 1999]) & set(range(its))):
        print("  ", i, tr[i])
    m = metrics_to_dict(met)
    print("  final bpp", m["est_bpp"], "psnr", m["psnr"], "yhat absmax", float(y_hat.abs().max()), "finite", bool(torch.isfinite(y_hat).all()), flush=True)
