"""Regression test of "defect (a)" (rounds 3-5: host SIGSEGV inside a later hipGraphLaunch after mid-life hipGraphExecDestroy calls;
root cause found in round 6, DESIGN_EXPERIMENTS.md A.13): the library must survive every stream-creation / -destruction history of
its process, because it launches its step graphs on a stream whose hardware-queue class the graphs' internal streams cannot share.
The reference has no such state (one tf.Session per process, sga.py:178-182)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_graphs_survive_adversarial_stream_histories():
    """tests/tools/graph_stream_stress.py in a subprocess (a crash must not take pytest down): 64 histories of six ballast streams,
    freed-memory poisoning on (MALLOC_PERTURB_: an out-of-bounds read of the runtime's stream list returns garbage, not a lucky
    NULL or a stale valid pointer), a fresh handle each, the three fork-point candidates instantiated and launched, two of them
    destroyed in mid-life, bit-equal results.  With graphs on the CALLER's stream (rounds 1-5; `--control`) the same harness dies
    with SIGSEGV on the runtime bundled with PyTorch-ROCm 2.10 (profiles/r06_defect_a_fix.txt)."""
    env = dict(os.environ, MALLOC_PERTURB_="165")
    env.pop("SGA_LAUNCH_STREAM", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "graph_stream_stress.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-600:], r.stderr[-1500:])
    assert "survived 64 stream histories" in r.stdout, r.stdout[-600:]
