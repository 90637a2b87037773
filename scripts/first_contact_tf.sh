#!/bin/bash
# FIRST CONTACT with the reference's own stack -- the one action that turns `parity` from "partial" to pinned (SURVEY.md
# 8(c); VERDICT r3 next #9c).  Needs what the build container lacks: network access, Python 3.7, and (for the checkpoint
# part) one of the authors' checkpoints (README.md:100-103 of the reference).
#     scripts/first_contact_tf.sh /path/to/reference [/path/to/checkpoint_dir/<runname>]
# 1. creates a venv with the reference's pinned wheels (requirements.txt:19-22: tensorflow 1.15, tensorflow-compression 1.3,
#    tensorflow-probability 0.7.0);
# 2. runs scripts/make_golden_from_tf.py there: imports the reference's nn_models.py, builds the four transforms as
#    sga.py:70-73 does, writes EFFECTIVE tensors + outputs of g_a, h_a, h_s, g_s, EntropyBottleneck._likelihood,
#    GaussianConditional._likelihood (on an UN-called and on a called layer), RelaxedOneHotCategorical.sample with injected
#    uniforms and tf.image.ssim_multiscale -> tests/golden/tf_ops_reference.npz;
# 3. prints which sigma-bound mode the un-called layer executed (the PROVISIONAL default of include/sga_hip.h);
# 5. LAST: tests/tools/first_contact_report.py writes profiles/first_contact_report.json (sigma-bound mode observed, per-operator
#    maximum error oracle-vs-TF and HIP-vs-TF, the checkpoint's variable names against the importer's) -- the ONE file to
#    commit: it turns SURVEY.md 8(c) and `parity` from "partial" to pinned.
# 4. runs the consumers in THIS repo's environment: tests/test_tf_reference.py (oracle on CPU; HIP path with -m gpu) and, when
#    a checkpoint is given, tests/test_tf_checkpoint.py against it (SGA_TF_CHECKPOINT).
set -eu
REF=${1:?path to a checkout of mandt-lab/improving-inference-for-neural-image-compression}
CKPT=${2:-}
cd "$(dirname "$0")/.."
VENV=${VENV:-/tmp/sga_tf115_venv}
PY37=${PY37:-python3.7}
if [ ! -x "$VENV/bin/python" ]; then
  "$PY37" -m venv "$VENV"
  "$VENV/bin/pip" install --upgrade "pip<21"
  "$VENV/bin/pip" install "tensorflow==1.15.0" "tensorflow-compression==1.3" "tensorflow-probability==0.7.0" "numpy<1.19" "scipy<1.6" "protobuf<3.21" absl-py
fi
"$VENV/bin/python" scripts/make_golden_from_tf.py "$REF" tests/golden/tf_ops_reference.npz
"$VENV/bin/python" - <<'PY'
import numpy as np
f = np.load("tests/golden/tf_ops_reference.npz")
keys = set(f.files)
print("fixture keys:", len(keys))
print("conditional layers' .built flags [un-called, called]:", f["conditional_built_flags"].tolist())
same = np.allclose(f["gauss_likelihood_unbuilt"], f["gauss_likelihood_built"], rtol=1e-6)
print("un-called layer's _likelihood == built layer's:", same, "->",
      "the SGA scripts run WITH the 0.11 bound: make SGA_SCALE_BOUND_BUILT their default (driver --scale_bound 0.11)" if same else
      "the SGA scripts evaluate the RAW sigma: scale_bound = 0 (the current default) is confirmed")
PY
python -m pytest tests/test_tf_reference.py -q -m "not gpu"
if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)"; then python -m pytest tests/test_tf_reference.py -q -m gpu; fi
if [ -n "$CKPT" ]; then
  # a bundle TensorFlow wrote: list its variables against the names tf_checkpoint.py expects and load the effective tensors
  python - "$CKPT" <<'PY'
import sys
sys.path.insert(0, ".")
import sga_amd
from sga_amd import tf_checkpoint
path = sys.argv[1]
tensors = tf_checkpoint.read_checkpoint(tf_checkpoint.latest_checkpoint(path))
print("%d variables in the bundle:" % len(tensors))
for k in sorted(tensors):
    print("   %-70s %s" % (k, tuple(tensors[k].shape)))
for C in (192, 256, 128):
    try:
        w = tf_checkpoint.load_effective_weights(path, C)
        sga_amd.check_weights(w, C)
        print("loaded as num_filters = %d: %d tensors, digest %s" % (C, len(w), sga_amd.weights_digest(w)[:16]))
        break
    except Exception as e:
        print("num_filters = %d: %s" % (C, e))
PY
fi
python tests/tools/first_contact_report.py tests/golden/tf_ops_reference.npz ${CKPT:+"$CKPT"}
echo "commit tests/golden/tf_ops_reference.npz and profiles/first_contact_report.json"
