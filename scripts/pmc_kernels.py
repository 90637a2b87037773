"""Per-kernel PMC summary (eager, single stream, bench shape).  Run on the GPU box:
    python scripts/pmc_kernels.py collect gpurun_out/pmc      # 4 rocprofv3 --pmc passes + 1 kernel trace
    python scripts/pmc_kernels.py report gpurun_out/pmc       # table (also works offline on the merged CSVs)
Units per MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE x 1024 bytes, FETCH_SIZE x 2 on gfx950."""
import csv, glob, os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "sq": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE",
    "lds": "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU",
    "fetch": "FETCH_SIZE", "write": "WRITE_SIZE",
}
CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--its", "6", "--steps", "1", "--warmup", "0",
       "--no-cpu-baseline", "--no-kernel-profile"]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace(" ", "")
    return n


def collect(out):
    env = dict(os.environ, SGA_NO_GRAPH="1", SGA_NO_OVERLAP="1", TMPDIR="/tmp")
    os.makedirs(out, exist_ok=True)
    for name, ctrs in PASSES.items():
        subprocess.run(["rocprofv3", "--pmc", *ctrs.split(), "--kernel-trace", "-d", os.path.join(out, name),
                        "--output-format", "csv", "--", *CMD], env=env, check=False, cwd="/tmp",
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.run(["rocprofv3", "--kernel-trace", "-d", os.path.join(out, "trace"), "--output-format", "csv", "--", *CMD],
                   env=env, check=False, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def report(out):
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r["Grid_Size_Y"]))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in PASSES:
        for f in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(float); meta = {}
            for r in csv.DictReader(open(f)):
                k = (r["Dispatch_Id"], r["Counter_Name"])
                per[k] += float(r["Counter_Value"])
                meta[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            for (d, c), v in per.items():
                ctr[meta[d]][c].append(v)
    print(f"{'kernel':44s} {'wgs':>5s} {'n':>3s} {'us':>7s} {'mfma%':>6s} {'busy%':>6s} {'wait%':>6s} {'winst%':>6s} {'ldsW%':>6s} {'bank%':>6s} {'rdMB':>7s} {'wrMB':>7s} {'TB/s':>5s}")
    for key in sorted(dur, key=lambda k: -sum(dur[k])):
        if sum(dur[key]) < 30: continue
        c = {k: sum(v) / len(v) for k, v in ctr.get(key, {}).items()}
        us = sorted(dur[key])[len(dur[key]) // 2]
        act = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        wc = c.get("SQ_WAVE_CYCLES", 0)
        mf = 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * act) if act else 0
        pct = lambda x: 100 * c.get(x, 0) / wc if wc else 0
        bank = 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0
        rd = c.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6; wr = c.get("WRITE_SIZE", 0) * 1024 / 1e6
        print(f"{key[0][:44]:44s} {key[1]:5d} {len(dur[key]):3d} {us:7.1f} {mf:6.1f} {pct('SQ_BUSY_CYCLES'):6.1f} {pct('SQ_WAIT_ANY'):6.1f} {pct('SQ_WAIT_INST_ANY'):6.1f} "
              f"{100 * c.get('SQ_WAIT_INST_LDS', 0) / c['SQ_WAVE_CYCLES'] if False else 0:6.1f} {bank:6.1f} {rd:7.1f} {wr:7.1f} {(rd + wr) / us / 1e0 / 1e3 * 1e0:5.2f}")


if __name__ == "__main__":
    sys.argv[2] = os.path.abspath(sys.argv[2])
    (collect if sys.argv[1] == "collect" else report)(sys.argv[2])
    if sys.argv[1] == "collect":
        report(sys.argv[2])
