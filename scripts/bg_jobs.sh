#!/bin/bash
# (re)start the background CPU jobs of this round: resumable oracle runs + fits.  Safe to call repeatedly.
cd /root/repo; mkdir -p /tmp/golden_logs
start() { # tag, cmd...
  local tag=$1; shift
  local pf=/tmp/golden_logs/$tag.pid
  if [ -f $pf ] && kill -0 "$(cat $pf)" 2>/dev/null; then echo "$tag already running"; return; fi
  "$@" >> /tmp/golden_logs/$tag.log 2>&1 < /dev/null &
  echo $! > $pf; echo "started $tag"
}
[ -f tests/golden/full_run_oracle_cfg2trace2000.json ] || start trace2000 nohup setsid env BGTAG=trace2000 GOLDEN=cfg2trace2000 NPROC=1 PROGRESS=1 python tests/tools/make_golden_full_run.py
start cfg2seeds nohup setsid env BGTAG=cfg2seeds GOLDEN=cfg2 NSEEDS=15 NPROC=3 PROGRESS=1 python tests/tools/make_golden_full_run.py
python -c "import json;import sys;sys.exit(0 if len(json.load(open('tests/golden/full_run_oracle_fitted.json'))['runs'])>=64 else 1)" 2>/dev/null || start fitted nohup setsid env BGTAG=fitted GOLDEN=fitted NSEEDS=64 NPROC=1 python tests/tools/make_golden_full_run.py
python -c "import json;import sys;sys.exit(0 if len(json.load(open('tests/golden/full_run_oracle_fitted_b011.json'))['runs'])>=64 else 1)" 2>/dev/null || start fitted_b011 nohup setsid env BGTAG=fitted_b011 GOLDEN=fitted_b011 NSEEDS=64 NPROC=1 python tests/tools/make_golden_full_run.py
[ -f tests/golden/fitted_weights_c64bb.npz ] || start fit_bb nohup setsid env BGTAG=fit_bb NTHREADS=1 python tests/tools/fit_weights.py 4000 bb
python -c "import json;import sys;sys.exit(0 if len(json.load(open('tests/golden/full_run_oracle_bb_fitted.json'))['runs'])>=32 else 1)" 2>/dev/null || start bb_fitted nohup setsid env BGTAG=bb_fitted GOLDEN=bb_fitted NSEEDS=32 NPROC=1 python tests/tools/make_golden_full_run.py
