"""GPU box, measurement build (make -C <pkg>/csrc clean; make -C <pkg>/csrc PROBE=1): shader clock and per-workgroup
K-loop time inside the convolution launches of the graph-replayed SGA iteration (bench shape).
    python scripts/clock_probe.py [its] > profiles/rNN_clock_probe.txt
Every conv workgroup records readcyclecounter / wall_clock64 (100 MHz) ticks around its K loop plus HW_ID / XCC_ID;
the values of the LAST replay are read back when the handle is destroyed (csrc/sga_api.hip, SGA_CLOCK_PROBE=2)."""
import os, sys, glob, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dump = tempfile.mkdtemp()
os.environ["SGA_CLOCK_PROBE"] = "2"; os.environ["SGA_CLOCK_PROBE_DUMP"] = dump
import numpy as np, torch, sga_amd
from sga_amd.codec import SGACodec
its = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
C, B, H, W = 192, 8, 256, 256
c = SGACodec(sga_amd.make_synthetic_weights(C, 0), C, B, H, W)
x = torch.rand(B, H, W, 3, generator=torch.Generator().manual_seed(1000)).cuda()
c.run(x, 0.01, its=its, seed=7, metrics=False); torch.cuda.synchronize()
c.close()      # prints one "clock_probe(graph) <layer> ..." line per captured conv launch to stderr
print("per-workgroup detail (launch order of the captured iteration; first = older workgroup of a CU pair):")
for f in sorted(glob.glob(os.path.join(dump, "clk_slot_*.bin"))):
    t = np.fromfile(f, dtype=np.uint64).reshape(-1, 6)
    pro = (t[:, 3].astype(float) - t[:, 4].astype(float)) / 100.0                          # entry -> K loop
    epi = (t[:, 5].astype(float) - t[:, 3].astype(float) - t[:, 1].astype(float)) / 100.0  # K loop end -> exit
    span = (t[:, 5].max() - t[:, 4].min()) / 100.0                                          # first entry -> last exit
    hwc = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    cuid = xc * 1000 + ((hwc >> 13) & 7) * 100 + ((hwc >> 12) & 1) * 16 + ((hwc >> 8) & 0xF)
    cg = []
    for c_ in np.unique(cuid):
        sel = np.where(cuid == c_)[0]
        if len(sel) == 2:
            a_, b_ = sorted(sel, key=lambda i: t[i, 4])
            if t[b_, 4] >= t[a_, 5]: cg.append((float(t[b_, 4]) - float(t[a_, 5])) / 100.0)
    cgap = float(np.mean(cg)) if cg else float("nan")
    wall = t[:, 1].astype(float) / 100.0
    mhz = 100.0 * t[:, 0].astype(float).sum() / t[:, 1].astype(float).sum()
    xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
    n = len(wall); h = n // 2
    print(f"{os.path.basename(f)} n={n:5d} {mhz:5.0f} MHz  K loop us: mean {wall.mean():7.1f} min {wall.min():7.1f} max {wall.max():7.1f}"
          f"  blocks[:n/2] {wall[:h].mean():7.1f}  blocks[n/2:] {wall[h:].mean():7.1f}  per XCC "
          + " ".join(f"{wall[xcc == k].mean():.0f}" for k in range(8) if (xcc == k).any())
          + f"  | exit->next entry on the CU {cgap:.1f}  prologue mean {pro.mean():.1f} max {pro.max():.1f}  after-K-loop mean {epi.mean():.1f} max {epi.max():.1f}  launch span {span:.1f}")

print("GDN tile kernel phases (wall clock per workgroup, us): prologue | fill (loads -> operand in LDS) | contraction | epilogue (stores)")
for f in sorted(glob.glob(os.path.join(dump, "gdn_slot_*.bin"))):
    t = np.fromfile(f, dtype=np.uint64).reshape(-1, 8)
    st = t[:, :5].astype(float) / 100.0
    st -= st[:, 0].min()
    d = np.diff(st, axis=1)
    ids = t[:, 5]; hw = (ids & np.uint64(0xffffffff)).astype(np.int64); xcc = (ids >> np.uint64(32)).astype(np.int64) & 0xF
    cu = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
    span = st[:, 4].max()
    # per CU: time with >= 1 / 2 workgroups inside the contraction phase
    one = two = 0.0; ncu = 0
    for c in np.unique(cu):
        sel = cu == c
        ev = sorted([(a, 1) for a in st[sel, 2]] + [(b, -1) for b in st[sel, 3]])
        k = 0; last = 0.0
        for x, dlt in ev:
            if k >= 1: one += x - last
            if k >= 2: two += x - last
            k += dlt; last = x
        ncu += 1
    # gap between a workgroup's exit and the entry of the next one in the same CU slot (greedy slot assignment)
    gaps = []
    for c in np.unique(cu):
        sel = np.where(cu == c)[0]
        order = sel[np.argsort(st[sel, 0])]
        free = []
        for i in order:
            cand = [x for x in free if x <= st[i, 0] + 0.5]
            if cand:
                x = max(cand); free.remove(x); gaps.append(st[i, 0] - x)
            free.append(st[i, 4])
    gap = float(np.mean(gaps)) if gaps else 0.0
    print(f"{os.path.basename(f)} n={len(t):5d} CUs {ncu}  slot gap exit->entry {gap:5.1f}  span {span:7.1f}  phases mean {d[:,0].mean():6.1f} | {d[:,1].mean():6.1f} | {d[:,2].mean():6.1f} | {d[:,3].mean():6.1f}"
          f"  (max {d[:,0].max():.1f} {d[:,1].max():.1f} {d[:,2].max():.1f} {d[:,3].max():.1f})  per CU: in contraction {one/ncu:6.1f} us ({100*one/ncu/span:.0f} % of span), two at once {two/ncu:6.1f} us")
