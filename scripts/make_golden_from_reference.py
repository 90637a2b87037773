"""Generate tests/golden/ fixtures by importing the two reference modules that import in the
build container (adam.py, configs.py: pure numpy / pure Python).  Run in the
build container only: /root/reference does not exist on the GPU box.  Fixtures are data
(inputs + the reference's outputs); no reference source is copied.

    python scripts/make_golden_from_reference.py
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    sys.path.insert(0, REF)
    import adam as ref_adam          # /root/reference/adam.py
    import configs as ref_configs    # /root/reference/configs.py
    os.makedirs(OUT, exist_ok=True)

    # ---- Adam: fixed-seed p, g float32 -> p after 1, 2, 3, 700, 2000 updates (adam.py:20-59)
    rng = np.random.RandomState(1234)
    n, steps, lr = 64, 2000, 0.005
    p0 = rng.standard_normal(n).astype(np.float32)
    scale = np.where(np.arange(steps) % 3 == 0, 1e-3, 1.0).astype(np.float32)[:, None]
    grads = (rng.standard_normal((steps, n)).astype(np.float32) * scale).astype(np.float32)
    opt = ref_adam.Adam(lr=lr)
    checkpoints = [1, 2, 3, 700, 2000]
    out = dict(p0=p0, grads=grads, lr=np.float64(lr), steps=np.int64(steps),
               checkpoints=np.array(checkpoints))
    p = [p0.copy()]
    for t in range(1, steps + 1):
        p = opt.update(p, [grads[t - 1]])
        if t in checkpoints:
            out[f"p_after_{t}"] = np.asarray(p[0], dtype=np.float64)
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(os.path.join(OUT, "adam_reference.npz"), **out)

    # ---- eval batch size (configs.py:5-9)
    sizes = [256 * 256, 768 * 512, 512 * 768, 1200 * 1200, 64 * 64, 1000 * 1000, 2000 * 3000]
    with open(os.path.join(OUT, "eval_batch_sizes.json"), "w") as f:
        json.dump({str(s): int(ref_configs.get_eval_batch_size(s)) for s in sizes}, f, indent=1)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    main()
