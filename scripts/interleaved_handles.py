"""GPU box: n independent batches in flight in ONE process with their iterations enqueued ALTERNATELY (run_begin +
run_steps(chunk) round-robin over the handles), so that the host never fills one hardware queue before it feeds the
next.  scripts/concurrent_batches.py enqueues whole runs one after the other and measured exactly the single-handle rate;
eight PROCESSES sharing the GPU measured 6 % more throughput.  us per (batch-of-B iteration).
usage: python scripts/interleaved_handles.py B n chunk"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sga_amd
from sga_amd.codec import SGACodec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w = sga_amd.make_synthetic_weights(192, 0)
hs = [SGACodec(w, 192, B, 256, 256) for _ in range(n)]
xs = [torch.rand(B, 256, 256, 3, generator=torch.Generator().manual_seed(1000 + i)).cuda() for i in range(n)]
its = 400
def run(k):
    for c, x in zip(hs, xs):
        c.run_begin(x, 0.01, its=k, loss_scale=1.0 / 8)
    for _ in range(0, k, chunk):
        for c in hs:
            c.run_steps(chunk)
run(40); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.time(); run(its); torch.cuda.synchronize()
    best = min(best, (time.time() - t) / its / n)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} B={B} n={n} chunk={chunk}: {best*1e6:.1f} us per batch-iteration "
      f"({best*1e6*n:.1f} us per round of all handles)", flush=True)
