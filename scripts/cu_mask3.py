import os, subprocess, sys
sys.argv = sys.argv[:1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exec(open(os.path.join(ROOT, "scripts", "cu_mask.py")).read().split('print("graph replay')[0])
allm = words(range(256))
print("graph replay (production)", run({}))
print("hybrid, unmasked-equivalent (all CUs)", run({"SGA_SIDE_CU_MASK": allm}))
print("hybrid, main chain ONLY (hyper branch skipped: timing experiment)", run({"SGA_SIDE_CU_MASK": allm, "SGA_SKIP_SIDE": "1"}))
print("graph, single stream (SGA_NO_OVERLAP=1)", run({"SGA_NO_OVERLAP": "1"}))
