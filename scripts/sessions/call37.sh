#!/bin/bash
# GPU box: what the driver runs at round end -- the GPU suite, smoke(), the default bench line
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/call37; mkdir -p $OUT
{
echo "=== pytest -m gpu"
timeout 2700 python -X faulthandler -m pytest tests/ -x -q -m gpu > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -6 $OUT/tests.log
echo "=== smoke"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "=== bench"
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
} > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
