"""MI355X-native SGA iterative inference for the mean-scale hyperprior codec.

Host side (Python) of the drop-in for the hot path of the reference's `sga.py`
(SURVEY.md 8): PyTorch-ROCm tensors own device memory, every arithmetic step runs in
hand-written HIP kernels behind the C ABI declared in include/sga_hip.h.
"""
from .weights import make_synthetic_weights, layer_shapes, weights_digest, check_weights  # noqa: F401
