import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch, sga_amd, time
from sga_amd.codec import SGACodec, metrics_to_dict
from oracle.sga_oracle import SGAOracle
from oracle import philox
def rel(a,b): a=np.asarray(a,np.float64); b=np.asarray(b,np.float64); return float(np.abs(a-b).max()/(np.abs(b).max()+1e-30))
for (C,B,H,W) in [(256,1,96,80),(256,1,1200,1200),(192,2,512,768)]:
    w = sga_amd.make_synthetic_weights(C, 0)
    x = np.random.RandomState(1).rand(B,H,W,3).astype(np.float32)
    orc = SGAOracle(w)
    for prec in ("f32","bf16x3"):
        c = SGACodec(w, C, B, H, W, precision=prec)
        y, z = c.encode(x)
        if prec == "f32":
            t=time.time(); yo, zo = orc.encode(x); te=time.time()-t
            u_y = philox.sga_uniforms(yo.numel(), 3, 0, 9); u_z = philox.sga_uniforms(zo.numel(), 3, 1, 9)
            t=time.time(); want = orc.step(x, yo, zo, 0.3, u_y, u_z, 0.05); ts=time.time()-t
        got = c.step_grads(x, yo.numpy(), zo.numpy(), 0.3, 0.05, seed=9, it=3)
        print(C,B,H,W,prec, "enc", rel(y.cpu().numpy(), yo.numpy()), rel(z.cpu().numpy(), zo.numpy()),
              "gy", rel(got["gy"].cpu().numpy(), want["gy"].numpy()), "gz", rel(got["gz"].cpu().numpy(), want["gz"].numpy()),
              "loss", abs(got["rd_loss"]/want["rd_loss"]-1), "oracle s", round(te,1), round(ts,1), flush=True)
        torch.cuda.synchronize(); t=time.time(); yh, zh, met, _ = c.run(x, 0.05, its=50, seed=1); torch.cuda.synchronize()
        m = metrics_to_dict(met)
        print("   run 50 its: %.1f ms/it" % ((time.time()-t)*20), "bpp", m["est_bpp"], "psnr", m["psnr"], flush=True)
        c.close()
