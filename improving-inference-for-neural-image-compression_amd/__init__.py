"""MI355X-native SGA iterative inference for the mean-scale hyperprior codec.

Host side (Python) of the drop-in for the hot path of the reference's `sga.py`
(SURVEY.md 8): PyTorch-ROCm tensors own device memory, every arithmetic step runs in
hand-written HIP kernels behind the C ABI declared in include/sga_hip.h.
"""
import os as _os

# HIP spreads a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues.  On MI355X
# the 3rd/4th queue dispatch each kernel 7-20 us later than the first two (measured with
# scripts/stream_placement.py: the same handle runs 2.33 ms/iteration on some streams and 2.6-3.2 on
# others); with two queues every stream and every handle measures the same.  The HIP runtime reads
# the variable when it initialises, i.e. at the first device call, so this only takes effect if the
# package is imported before that; an explicit setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

from .weights import (make_synthetic_weights, layer_shapes, weights_digest, check_weights,  # noqa: F401
                      make_lowpass_images, load_weights_npz, save_weights_npz)
