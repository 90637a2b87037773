"""GPU-box: A/B of the GDN tile shapes (SGA_GDN_SHAPE: 0 auto, 1 32-row tiles everywhere, 2 64-row)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for shape in ("0", "1", "2"):
    env = dict(os.environ, SGA_GDN_SHAPE=shape)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "profile_layers.py")], env=env, capture_output=True, text=True).stdout
    print("SGA_GDN_SHAPE =", shape)
    for ln in out.splitlines():
        if "gdn_tile" in ln or "total" in ln:
            print("  ", ln)
