"""ms per SGA iteration and fraction of the fp32-MFMA roofline at the shapes of BASELINE.json's configs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import gflop_per_image_step
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
for (name, C, B, H, W) in [("cfg2", 192, 8, 256, 256), ("cfg2 B=1", 192, 1, 256, 256), ("cfg2 B=32", 192, 32, 256, 256),
                           ("cfg3 Kodak 3/GPU", 192, 3, 512, 768), ("cfg3 Kodak B=1", 192, 1, 512, 768),
                           ("cfg4 Tecnick C=256 B=1", 256, 1, 1200, 1200), ("cfg4 Tecnick C=256 B=4", 256, 4, 1200, 1200)]:
    w = sga_amd.make_synthetic_weights(C, 0)
    x = torch.rand(B, H, W, 3).cuda()
    c = SGACodec(w, C, B, H, W, precision=prec)
    c.run(x, 0.01, its=110, metrics=False); torch.cuda.synchronize()      # >= 100 iterations: the fork point of the hyper branch is timed here
    its = 150
    t = time.time(); c.run(x, 0.01, its=its, metrics=False); torch.cuda.synchronize()
    dt = (time.time() - t) / its
    fl = gflop_per_image_step(H, W, C) * 1e9 * B
    print(f"{prec} {name:26s} {dt*1e3:8.2f} ms/it  {fl/dt/1e12:6.1f} TFLOP/s  frac {fl/dt/157.3e12:.3f}  img/s(2000 its) {B/(dt*2000):.3f}", flush=True)
    c.close()
