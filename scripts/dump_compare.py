import sys, numpy as np, glob
names = ["yt", "zt", "u0", "v0", "u1", "u2", "g_yt_dist", "g_yt_rate", "g_zt_hs", "g_zt_eb", "y", "z"]
files = sorted(glob.glob(sys.argv[1] + ".*"), key=lambda f: int(f.rsplit(".", 1)[1]))
runs = [np.fromfile(f, np.uint64).reshape(-1, 16) for f in files]
ref = runs[0]
for i, r in enumerate(runs[1:], 1):
    d = (r != ref)[:, :12]
    if d.any():
        it = int(np.argmax(d.any(1)))
        print("run", i, "first differing iteration", it, "buffers:", [n for n, x in zip(names, d[it]) if x],
              "| next:", [n for n, x in zip(names, d[min(it + 1, len(d) - 1)]) if x])
    else:
        print("run", i, "identical")
