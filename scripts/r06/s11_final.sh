#!/bin/bash
# GPU box, round 6: what the driver runs at round end, at the final commit -- GPU suite (third green run; now with the cfg-4 real-size
# golden), smoke(), the default bench line
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s11; mkdir -p $OUT
timeout 3000 python -X faulthandler -m pytest tests/ -q -m gpu --durations=15 > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -30 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4; head -c 600 $OUT/bench.json
cp gpurun_out/acceptance_real_size_*.json gpurun_out/acceptance_trace2000_cfg2_fitted*.json $OUT/ 2>/dev/null
