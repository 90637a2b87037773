#!/bin/bash
# GPU box, round 6: the two soaks of round 5 on the shipped library (own low-priority launch stream, every drop = hipGraphExecDestroy),
# under freed-memory poisoning: one handle under eviction pressure (24 keys on the 16-entry cache: every miss destroys three graphs),
# one handle under a service's pattern (seven keys).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06_s07
MALLOC_PERTURB_=165 timeout 900 python scripts/r05/soak_evictions.py 5 > gpurun_out/r06_s07/soak_evictions.log 2>&1; echo "soak_evictions rc $?"; tail -2 gpurun_out/r06_s07/soak_evictions.log
MALLOC_PERTURB_=165 timeout 900 python scripts/r05/soak_graph_cache.py 5 > gpurun_out/r06_s07/soak_graph_cache.log 2>&1; echo "soak_graph_cache rc $?"; tail -3 gpurun_out/r06_s07/soak_graph_cache.log
cp gpurun_out/r05_soak_evictions.txt gpurun_out/r06_s07/ 2>/dev/null; cp gpurun_out/r05_soak_graph_cache.txt gpurun_out/r06_s07/ 2>/dev/null
