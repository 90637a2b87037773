"""GPU box: hybrid replay (graph main chain + eager hyper branch on a CU-masked stream) under different masks.
Mask bit i = XCD (i % 8), CU (i / 8) of that XCD on this chip (the layout that explains scripts/cu_mask.py's numbers:
whole-XCD masks run at the production rate, masks that touch a few CUs of every XCD are much slower)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cu_mask import run, words
def xcd(x, ncu=32, first=0):
    return [8 * c + x for c in range(first, first + ncu)]
print("graph replay (production)", run({}))
for x in range(8):
    print("hybrid, whole XCD %d" % x, run({"SGA_SIDE_CU_MASK": words(xcd(x))}))
print("hybrid, 24 CUs of XCD 0", run({"SGA_SIDE_CU_MASK": words(xcd(0, 24))}))
print("hybrid, 16 CUs of XCD 0", run({"SGA_SIDE_CU_MASK": words(xcd(0, 16))}))
print("hybrid, XCD 0 + 16 CUs of XCD 1", run({"SGA_SIDE_CU_MASK": words(xcd(0) + xcd(1, 16))}))
print("hybrid, XCDs 0 and 1", run({"SGA_SIDE_CU_MASK": words(xcd(0) + xcd(1))}))
for tgt in (96, 192, 256):
    print("hybrid, whole XCD 0, SGA_SIDE_TARGET=%d" % tgt, run({"SGA_SIDE_CU_MASK": words(xcd(0)), "SGA_SIDE_TARGET": str(tgt)}))
print("hybrid, all 256 CUs in the mask", run({"SGA_SIDE_CU_MASK": words(range(256))}))
