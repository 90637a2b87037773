"""GPU box: the gz error of one SGA evaluation vs the float64 oracle at 256x256, C = 192, by batch size and launch knob --
is a 5e-4 deviation (tests/test_gpu_configs.py::test_step_across_launch_plans[192-4-256-256]) the plan or the conditioning?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, sga_amd
from oracle import philox
from oracle.sga_oracle import SGAOracle
from sga_amd.codec import SGACodec

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

C, H, W = 192, 256, 256
w = sga_amd.make_synthetic_weights(C, seed=0)
seed, it, T, lm = 5, 7, 0.25, 0.02
for B in [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,5,8").split(",")]:
    x = np.random.RandomState(100 + B).rand(B, H, W, 3).astype(np.float32)
    o32, o64 = SGAOracle(w), SGAOracle(w, dtype=torch.float64)
    yo, zo = o32.encode(x)
    u_y = philox.sga_uniforms(yo.numel(), it, 0, seed); u_z = philox.sga_uniforms(zo.numel(), it, 1, seed)
    want = o64.step(x, yo, zo, T, u_y, u_z, lm)
    w32 = o32.step(x, yo, zo, T, u_y, u_z, lm)
    gz64 = want["gz"].numpy()
    line = "B=%d  f32-oracle gz %.2e gy %.2e |" % (B, rel(w32["gz"].numpy(), gz64), rel(w32["gy"].numpy(), want["gy"].numpy()))
    for prec, env in [("f32", {}), ("f32", {"SGA_NO_SPLITK": "1"}), ("f32", {"SGA_NO_OVERLAP": "1"}), ("bf16x3", {}), ("bf16x2", {})]:
        for k, v in env.items(): os.environ[k] = v
        try:
            c = SGACodec(w, C, B, H, W, precision=prec)
            got = c.step_grads(x, yo.numpy(), zo.numpy(), T, lm, seed=seed, it=it)
            gz = got["gz"].cpu().numpy()
            e = np.abs(gz.astype(np.float64) - gz64) / np.abs(gz64).max()
            line += " %s%s gz %.2e (per image %s) gy %.2e |" % (prec, "+" + ",".join(env) if env else "", e.max(),
                      " ".join("%.1e" % e[b].max() for b in range(B)), rel(got["gy"].cpu().numpy(), want["gy"].numpy()))
            c.close()
        except Exception as ex:
            line += " %s FAILED %s |" % (prec, str(ex)[:80])
        for k in env: del os.environ[k]
    print(line, flush=True)
