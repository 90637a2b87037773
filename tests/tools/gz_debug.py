import sys, os; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
from oracle.sga_oracle import SGAOracle
from oracle import philox
C,B,H,W = 192,1,512,768
w = sga_amd.make_synthetic_weights(C, 0)
x = np.random.RandomState(1).rand(B,H,W,3).astype(np.float32)
orc = SGAOracle(w); orc64 = SGAOracle(w, dtype=torch.float64)
yo, zo = orc.encode(x)
u_y = philox.sga_uniforms(yo.numel(), 3, 0, 9); u_z = philox.sga_uniforms(zo.numel(), 3, 1, 9)
w32 = orc.step(x, yo, zo, 0.3, u_y, u_z, 0.05)
w64 = orc64.step(x, yo, zo, 0.3, u_y, u_z, 0.05)
c = SGACodec(w, C, B, H, W)
got = c.step_grads(x, yo.numpy(), zo.numpy(), 0.3, 0.05, seed=9, it=3)
g = got["gz"].cpu().numpy().astype(np.float64); a = w32["gz"].numpy().astype(np.float64); b = w64["gz"].numpy()
print("shape", g.shape, "max|ref|", np.abs(b).max())
print("gpu vs f64", np.abs(g-b).max()/np.abs(b).max(), " oracle32 vs f64", np.abs(a-b).max()/np.abs(b).max())
d = np.abs(g-b); idx = np.unravel_index(np.argsort(-d.ravel())[:8], d.shape)
for k in range(8):
    i = tuple(ix[k] for ix in idx); print(i, "gpu", g[i], "f64", b[i], "f32", a[i])
