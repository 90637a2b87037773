"""Timeline of one SGA iteration from a rocprofv3 --kernel-trace CSV (graph replay, two streams)."""
import csv, re, sys
f = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*$', '', n); n = n.replace('void ', '')
    return n[:58]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i + 1 for i, r in enumerate(rows) if 'k_step_boundary' in r['Kernel_Name']]      # the launch after the last one of an iteration
if len(idx) < 20:
    idx = [i for i, r in enumerate(rows) if 'k_sample_yz' in r['Kernel_Name']]      # SGA_FUSED_BOUNDARY=0: first launch of an iteration
idx = [i for i in idx if i < len(rows)]
its = [(int(rows[idx[k + 1]]['Start_Timestamp']) - int(rows[idx[k]]['Start_Timestamp'])) / 1e3 for k in range(10, len(idx) - 2)]
print("iterations", len(idx), "mean us/it", sum(its) / len(its), "min", min(its))
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]['Start_Timestamp'])
prev = {}
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    q = r['Queue_Id']; gap = s - prev.get(q, 0); prev[q] = e
    print(f"q{q} {s/1e3:8.1f} {e/1e3:8.1f} dur {(e-s)/1e3:7.1f} gap {gap/1e3:6.1f} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{r['Grid_Size_Y']} v{r['VGPR_Count']:>3s} {short(r['Kernel_Name'])}")
