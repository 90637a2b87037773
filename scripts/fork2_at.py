"""GPU-box: us per SGA iteration as a function of the two fork points of the hyper branch."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = open(os.path.join(ROOT, "scripts", "fork_at.py")).read().split("code = r'''")[1].split("''' % ROOT")[0] % ROOT
for fa, fb in [tuple(a.split(",")) for a in sys.argv[1:]] or [("0", "0"), ("0", "7"), ("0", "8"), ("0", "9"), ("0", "10"), ("0", "11"), ("0", "12")]:
    env = dict(os.environ, SGA_FORK_AT=fa, SGA_FORK2_AT=fb)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("fork", fa, "fork2", fb, "us/it:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
