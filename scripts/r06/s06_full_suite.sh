#!/bin/bash
# GPU box, round 6: what the driver runs at round end -- the GPU suite (with durations), smoke(), the default bench line
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s06; mkdir -p $OUT
timeout 3000 python -X faulthandler -m pytest tests/ -q -m gpu --durations=40 > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -70 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cp gpurun_out/acceptance_real_size_*.json gpurun_out/acceptance_*trace2000*.json $OUT/ 2>/dev/null
