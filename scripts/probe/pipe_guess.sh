#!/bin/bash
# filtered throughput at the headline size against the exploring rounds' guess rate (VS_HNSW_FILTER_GUESS, % of the observed selectivity)
cd ${GRAFT_REPO_ROOT:-.}
for g in ${GUESSES:-50 75 100}; do
  echo "== guess $g"
  VS_HNSW_FILTER_GUESS=$g PIPE_PROBE_MODS=${MODS:-10} timeout 600 python scripts/probe/pipe_probe.py ${1:-10000000} 200 ${2:-1,17,64} 1.5 2>&1 | grep -a "pipe:\|rror"
done
