#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/call41; mkdir -p $O
{
echo "=== bf16x2 with the activation planes 128 bytes apart as well"
for p in bf16x2 bf16x3; do python scripts/perf_modes.py $p 2>&1 | tail -1; done
PREC=bf16x2 python scripts/profile_layers.py 2>&1 | grep -v amdgpu.ids | head -8
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -q -k bf16x2 2>&1 | tail -2
echo "=== LDS counters (bf16x2, eager single stream)"
( cd /tmp && SGA_NO_GRAPH=1 SGA_NO_OVERLAP=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/$O/lds --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --precision bf16x2 --its 6 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-profile --no-alt-precision --no-other-configs --no-other-input > /dev/null 2>&1 )
python - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/call41/lds/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*$", "", n).replace("void ", "").replace(" ", "")
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:10]:
    a = c.get("SQ_LDS_IDX_ACTIVE", 0); b = c.get("SQ_LDS_BANK_CONFLICT", 0)
    print("%-50s bank conflict %5.1f %% of LDS active cycles" % (n, 100 * b / a if a else 0))
PY
rm -rf $O/lds
} > $O/summary.log 2>&1
