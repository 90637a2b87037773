"""Per-iteration timeline of a graph-replay run from a rocprofv3 kernel trace: busy time of the
critical (main) chain, gaps between consecutive kernels of it, overlap with the side branch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# find iteration boundaries: k_advance_ctx launches
starts = [i for i, k in enumerate(ks) if "k_advance_ctx" in k[2]]
print("kernels", len(ks), "iterations", len(starts))
its = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)][5:-2]
tot = collections.Counter(); n = 0
for a, b in its:
    seg = ks[a:b]
    t0, t1 = seg[0][0], ks[b][0]
    busy = sorted((s, e) for s, e, _ in seg)
    # union of busy intervals
    u = 0; cs, ce = busy[0]
    for s, e in busy[1:]:
        if s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    u += ce - cs
    tot["iter_ns"] += t1 - t0; tot["union_busy_ns"] += u; tot["sum_kernel_ns"] += sum(e - s for s, e, _ in seg); tot["nk"] += len(seg)
    n += 1
print("per iteration: wall %.1f us, GPU busy (union) %.1f us, idle %.1f us, sum of kernel durations %.1f us, kernels %d"
      % (tot["iter_ns"] / n / 1e3, tot["union_busy_ns"] / n / 1e3, (tot["iter_ns"] - tot["union_busy_ns"]) / n / 1e3, tot["sum_kernel_ns"] / n / 1e3, tot["nk"] // n))
# gaps where nothing runs, by following kernel
a, b = its[len(its) // 2]
seg = ks[a:b]
end = seg[0][1]
print("one iteration, idle gaps > 1.5 us (before kernel):")
for s, e, name in seg[1:]:
    if s - end > 1500: print("  %.1f us before %s" % ((s - end) / 1e3, name[:70]))
    end = max(end, e)
