#!/bin/bash
# GPU box: sample sclk / power with rocm-smi while the SGA loop runs (steady state for several seconds)
rm -f /tmp/ready
python - <<'PY' &
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, sga_amd
from sga_amd.codec import SGACodec
c = SGACodec(sga_amd.make_synthetic_weights(192, 0), 192, 8, 256, 256)
x = torch.rand(8, 256, 256, 3, generator=torch.Generator().manual_seed(1000)).cuda()
c.run(x, 0.01, its=40, metrics=False); torch.cuda.synchronize()
open("/tmp/ready", "w").write("1")
t = time.time(); c.run(x, 0.01, its=4000, metrics=False); torch.cuda.synchronize()
print("us/it", (time.time() - t) / 4000 * 1e6)
PY
PID=$!
while [ ! -f /tmp/ready ]; do sleep 0.2; done
sleep 1
rocm-smi --showmaxpower 2>/dev/null | grep -i "max"
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.3
done
amd-smi metric -g 0 --clock --power 2>/dev/null | head -60
wait $PID
