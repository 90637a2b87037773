"""Throughput of n independent batches of B images in flight at once (n handles, n streams, one process)
against one batch at a time: us per (batch-of-B iteration).  usage: python scripts/concurrent_batches.py [B] [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ns = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
w = sga_amd.make_synthetic_weights(192, 0)
its = 400
for n in ns:
    hs = [SGACodec(w, 192, B, 256, 256) for _ in range(n)]
    ss = [torch.cuda.Stream() for _ in range(n)]
    xs = [torch.rand(B, 256, 256, 3, generator=torch.Generator().manual_seed(1000 + i)).cuda() for i in range(n)]
    def run(k):
        for c, s, xx in zip(hs, ss, xs):
            with torch.cuda.stream(s):
                c.run(xx, 0.01, its=k, metrics=False)
    run(40); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.time(); run(its); torch.cuda.synchronize()
        best = min(best, (time.time() - t) / its / n)
    print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} B={B} n={n} batches in flight: {best*1e6:.1f} us per batch-iteration", flush=True)
    for c in hs:
        c.close()
