#!/bin/bash
# GPU box, round 5: cfg 5 at Kodak size on the fitted C = 192 bits-back model; the bench line's new legs
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "cfg5 or bits_back" 2>&1 | tail -4
