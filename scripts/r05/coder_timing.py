"""GPU box, round 5: what the rate-following block size costs in coding time (fitted C = 192 model, 8 x 256^2, one-shot latents)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import sga_amd
from sga_amd.codec import SGACodec
from sga_amd import entropy_coding as ec
C, B, H, W = 192, 8, 256, 256
w = sga_amd.load_weights_npz("tests/golden/fitted_weights_c192.npz")
codec = SGACodec(w, C, B, H, W)
x = sga_amd.make_lowpass_images(B, H, W, seed=77)
y, z = codec.encode(x)
y_hat, z_hat = torch.round(y), torch.round(z)
out = open("gpurun_out/r05_coder_timing.txt", "w")
def say(s):
    print(s); out.write(s + "\n"); out.flush()
for bmax in (1024, 4096, 16384, 65536):
    ec.BLOCK_MAX = bmax
    for on_device in (True, False):
        codec.compress_latents((B, H, W), y_hat, z_hat, on_device=on_device)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        blob = codec.compress_latents((B, H, W), y_hat, z_hat, on_device=on_device)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        codec.decompress_latents(blob, on_device=on_device)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        say("BLOCK_MAX %6d %s: %6d bytes = %.4f bpp, encode %.1f ms, decode %.1f ms" %
            (bmax, "device" if on_device else "host  ", len(blob), 8.0 * len(blob) / (B * H * W), 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
