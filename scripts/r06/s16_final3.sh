#!/bin/bash
# GPU box, round 6, final: the measurement set at the final product commit, then what the driver runs at round end (GPU suite, smoke())
set -u
cd "$GRAFT_REPO_ROOT"
bash scripts/collect_profiles.sh r06 097d3fc > /dev/null 2>&1
OUT=$PWD/gpurun_out/r06_s16; mkdir -p $OUT
timeout 3000 python -X faulthandler -m pytest tests/ -q -m gpu --durations=6 > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -14 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
head -c 300 gpurun_out/r06/bench_n1.json
