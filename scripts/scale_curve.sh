#!/bin/bash
# the 1/2/4/8-GPU curve of the headline benchmark on ONE node, one process per GPU over RCCL/xGMI (the only
# collective is the final all_gather of the [8,7] metrics).  For whoever has a multi-GPU MI355X node:
#     scripts/scale_curve.sh [steps=5] [warmup=2] [gpu counts="1 2 4 8"]
# Prints one bench line per N and a table with the weak-scaling efficiency value(N) / (N * value(1)).
set -eu
cd "$(dirname "$0")/.."
STEPS=${1:-5}; WARM=${2:-2}; NS=${3:-"1 2 4 8"}
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
OUT=${OUT:-gpurun_out/scale}; mkdir -p "$OUT"
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
for N in $NS; do
  if [ "$N" -gt "$HAVE" ]; then echo "skipping N=$N: only $HAVE GPU(s) visible"; continue; fi
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-other-configs --no-alt-precision > "$OUT/n$N.json"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" > "$OUT/n$N.json"
  fi
  tail -n 1 "$OUT/n$N.json"
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for f in glob.glob(os.path.join(sys.argv[1], "n*.json")):
    lines = [l for l in open(f) if l.startswith("{")]
    if lines:
        d = json.loads(lines[-1]); rows[d["n_gpus"]] = d
if 1 in rows:
    v1 = rows[1]["value"]
    print("%6s %12s %14s %12s" % ("GPUs", "images/sec", "ms per step", "efficiency"))
    for n in sorted(rows):
        print("%6d %12.4f %14.2f %12.3f" % (n, rows[n]["value"], rows[n]["ms_per_step"], rows[n]["value"] / (n * v1)))
PY
