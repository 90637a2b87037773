"""Average selected PMC counters over the big conv launches (>= 256 workgroups of conv_mfma_kernel
<2,3,2,2,0,...>) of a rocprofv3 --pmc run; prints one line per (grid, counter set)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "conv_mfma_kernel<2, 3, 2, 2, 0, false" not in n: continue
    blocks = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    if blocks < 256: continue
    key = blocks
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    agg[key]["_ns"] += 0
    cnt[key].add(r["Dispatch_Id"])
    if r["Counter_Name"] == sorted(set([r["Counter_Name"]]))[0]: pass
for key in sorted(agg):
    n = len(cnt[key])
    print("blocks", key, "n", n, " ".join(f"{c}={v/n:.4g}" for c, v in sorted(agg[key].items()) if c != "_ns"))
