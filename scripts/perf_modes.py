import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, sga_amd
from sga_amd.codec import SGACodec
w = sga_amd.make_synthetic_weights(192, 0)
x = torch.rand(8,256,256,3).cuda()
prec = sys.argv[1]
c = SGACodec(w, 192, 8, 256, 256, precision=prec)
c.run(x, 0.01, its=100, metrics=False); torch.cuda.synchronize()
t=time.time(); c.run(x, 0.01, its=1000, metrics=False); torch.cuda.synchronize()
print(prec, "ms/it %.4f" % ((time.time()-t)))
