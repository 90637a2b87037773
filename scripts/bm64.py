import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1:] or ("0", "64", "128", "256", "511"):
    env = dict(os.environ, SGA_BM64_MAX=v)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "profile_layers.py")], env=env, capture_output=True, text=True)
    print("SGA_BM64_MAX =", v)
    for ln in out.stdout.splitlines():
        if any(t in ln for t in ("gs0", "gs1", "hs", "igdn1", "igdn0", "total")): print("  ", ln)
    if out.returncode: print(out.stderr[-1500:])
