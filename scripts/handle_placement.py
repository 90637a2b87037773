import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
w = sga_amd.make_synthetic_weights(192, 0)
x = torch.rand(8,256,256,3).cuda()
def wall(c, its=200):
    c.run(x, 0.01, its=20, seed=7, metrics=False); torch.cuda.synchronize()
    t=time.time(); c.run(x, 0.01, its=its, seed=7, metrics=False); torch.cuda.synchronize()
    return 1e3*(time.time()-t)/its
def prof(c, its=30):
    c.profile_begin(); c.run(x, 0.01, its=its, seed=7, metrics=False); ks = c.profile_end()
    return sum(k["ms_total"] for k in ks)/its
hs=[]
for i in range(8):
    g = SGACodec(w, 192, 8, 256, 256, precision=sys.argv[1])
    hs.append(g)
    print("handle", i, "graph wall %.3f  profile conv %.3f" % (wall(g), prof(g)), flush=True)
