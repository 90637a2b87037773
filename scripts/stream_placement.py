"""Which caller stream does a handle run fastest on?  (HIP maps streams onto a few hardware queues.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
w = sga_amd.make_synthetic_weights(192, 0)
x = torch.rand(8, 256, 256, 3).cuda()
c = SGACodec(w, 192, 8, 256, 256, precision=sys.argv[1] if len(sys.argv) > 1 else "f32")
def wall(its=150):
    c.run(x, 0.01, its=20, seed=7, metrics=False); torch.cuda.synchronize()
    t = time.time(); c.run(x, 0.01, its=its, seed=7, metrics=False); torch.cuda.synchronize()
    return 1e3 * (time.time() - t) / its
print("own stream  %.3f" % wall(), flush=True)
for prio in (0, -1):
    for i in range(6):
        c.stream = torch.cuda.Stream(priority=prio)
        print("prio %d stream %d  %.3f ms/it" % (prio, i, wall()), flush=True)
