#!/bin/bash
# GPU box, round 5: what the driver runs at round end -- the GPU suite, smoke(), the default bench line
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r05_s20; mkdir -p $OUT
{
echo "=== pytest -m gpu"
timeout 3000 python -X faulthandler -m pytest tests/ -x -q -m gpu > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -6 $OUT/tests.log
echo "=== smoke"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "=== bench"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 4000 $OUT/bench.json
} > $OUT/summary.log 2>&1
tail -40 $OUT/summary.log
