"""Potential of batch-level pipelining: two B/2 handles on two streams vs one B handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = sga_amd.make_synthetic_weights(192, 0)
x = torch.rand(B, 256, 256, 3).cuda()
its = 400
def timeit(fn):
    fn(50); torch.cuda.synchronize()
    t = time.time(); fn(its); torch.cuda.synchronize()
    return (time.time() - t) / its * 1e3
one = SGACodec(w, 192, B, 256, 256, precision=prec)
print(prec, f"one handle B={B}: %.3f ms/it" % timeit(lambda n: one.run(x, 0.01, its=n, metrics=False)))
for nsplit in (2, 4):
    hs = [SGACodec(w, 192, B // nsplit, 256, 256, precision=prec) for _ in range(nsplit)]
    ss = [torch.cuda.Stream() for _ in range(nsplit)]
    xs = [x[i * (B // nsplit):(i + 1) * (B // nsplit)].contiguous() for i in range(nsplit)]
    def run(n):
        for c, s, xx in zip(hs, ss, xs):
            with torch.cuda.stream(s):
                c.run(xx, 0.01, its=n, metrics=False, loss_scale=1.0 / B)
    print(prec, f"{nsplit} handles B={B//nsplit} each, concurrent: %.3f ms/it" % timeit(run), flush=True)
