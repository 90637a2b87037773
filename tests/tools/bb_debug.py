import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
from oracle.sga_oracle import SGAOracle
C=192
w = sga_amd.make_synthetic_weights(C, seed=0, bb=True)
for (H,W) in [(64,64),(256,256),(512,768)]:
    codec = SGACodec(w, C, 1, H, W, bits_back=True)
    orc, orc64 = SGAOracle(w), SGAOracle(w, dtype=torch.float64)
    x = np.random.RandomState(2).rand(1, H, W, 3).astype(np.float32)
    y = orc.analysis(torch.tensor(x)).numpy()
    zml = orc.bb_init_z(y).numpy()
    rng = np.random.RandomState(3)
    u_y = rng.uniform(1e-4, 1 - 1e-4, (y.size, 2)).astype(np.float32)
    eps = rng.standard_normal(zml.size // 2).astype(np.float32)
    ref = orc64.bb_step(x, y, zml, 0.35, u_y, eps, 0.01)
    got = codec.bb_step_grads(x, y, zml, 0.35, 0.01, u_y=u_y, eps=eps)
    g = got["gzml"].cpu().numpy(); r = ref["gzml"].numpy()
    print(H, W, "zml range", zml[..., :C].min(), zml[..., :C].max(), "logvar range", zml[..., C:].min(), zml[..., C:].max())
    print("  nan gpu", np.isnan(g).sum(), "nan ref", np.isnan(r).sum(), "inf ref", np.isinf(r).sum(), "loss gpu", got["rd_loss"], "ref", ref["rd_loss"], "bpp", got["train_bpp"], ref["train_bpp"])
    gy, ry = got["gy"].cpu().numpy(), ref["gy"].numpy()
    print("  gy err", np.abs(gy-ry).max()/np.abs(ry).max(), "nan gy gpu", np.isnan(gy).sum())
    codec.close()
