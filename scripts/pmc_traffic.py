"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the
bench command run eagerly: average HBM bytes per launch for each conv kernel symbol.
Units per MI355X_MICROARCH.md: counters x 1024 bytes; FETCH_SIZE x 2 on gfx950 (wide coalesced
reads are tallied at half their size)."""
import csv, sys, json, glob, os, re, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        m = re.search(r"(conv_mfma_kernel<[^>]*>|gdn_tile_kernel<[^>]*>|deconv3_halo_kernel)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1).replace(" ", "")
        tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(n[k]) for k in tot}, {k: len(n[k]) for k in tot}
fe, nf = load(fetch_dir, "FETCH_SIZE")
wr, _ = load(write_dir, "WRITE_SIZE")
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = ""
res = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --roofline-only (eager), B=8 256x256 C=192; "
                  "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the symbol's launches",
       "_commit": commit if len(sys.argv) < 5 else sys.argv[4]}
# optional second pair of passes taken with SGA_NO_OVERLAP=1 (single stream: no hyper-branch kernel shares L2 / MALL with it)
fe1, wr1 = ({}, {})
if len(sys.argv) >= 7:
    fe1, _ = load(sys.argv[5], "FETCH_SIZE")
    wr1, _ = load(sys.argv[6], "WRITE_SIZE")
for k in fe:
    res[k] = {"launches": nf[k], "fetch_bytes_per_launch": round(2 * fe[k] * 1024), "write_bytes_per_launch": round(wr.get(k, 0) * 1024),
              "hbm_bytes_per_launch": round((2 * fe[k] + wr.get(k, 0)) * 1024)}
    if k in fe1:
        res[k]["single_stream"] = {"fetch_bytes_per_launch": round(2 * fe1[k] * 1024),
                                   "write_bytes_per_launch": round(wr1.get(k, 0) * 1024)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
