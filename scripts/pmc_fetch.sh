#!/bin/bash
# GPU box: FETCH_SIZE per launch of the biggest kernels of the eager roofline leg (bench.py --roofline-only) under the
# environment given as arguments, e.g.  bash scripts/pmc_fetch.sh SGA_XCD_REMAP=0
export TMPDIR=/tmp
R=$(pwd); D=$R/gpurun_out/_pmc_fetch; rm -rf $D
( cd /tmp && env "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $D --output-format csv -- python $R/bench.py --roofline-only > /dev/null 2>&1 )
python - <<P
import csv, glob, re, collections
f = glob.glob("$D/**/*counter_collection.csv", recursive=True)[0]
t = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    m = re.search(r"(conv_mfma_kernel<[^>]*>|gdn_tile_kernel<[^>]*>|deconv3_halo_kernel)", r["Kernel_Name"])
    if not m: continue
    k = m.group(1).replace(" ", ""); t[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(t, key=lambda k: -t[k])[:6]:
    print("%-50s %4d launches %8.1f MB fetched per launch" % (k, len(n[k]), 2 * t[k] * 1024 / len(n[k]) / 1e6))
P
rm -rf $D
