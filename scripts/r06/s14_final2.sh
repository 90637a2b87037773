#!/bin/bash
# GPU box, round 6: after the side-target change (256 beside the synthesis chain, 384 alone) -- A/B against the previous value, the GPU
# suite, smoke(), the default bench line, bits-back timing
set -u
cd "$GRAFT_REPO_ROOT"; OUT=$PWD/gpurun_out/r06_s14; mkdir -p $OUT
python scripts/ab_iter.py --rounds 3 "" "LAB=1 SGA_SIDE_TARGET=384" 2>&1 | tee $OUT/ab_side_target.txt
python scripts/perf_bb.py 2>&1 | grep -v amdgpu.ids | tee $OUT/perf_bb.txt
timeout 3000 python -X faulthandler -m pytest tests/ -q -m gpu --durations=8 > $OUT/tests.log 2>&1; echo "rc $?" >> $OUT/tests.log; tail -16 $OUT/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $OUT/bench.json 2> $OUT/bench.err; head -c 400 $OUT/bench.json
