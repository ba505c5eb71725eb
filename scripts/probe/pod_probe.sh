#!/bin/bash
# pods (csrc/pipe_pod.hpp) against launches: blocking plain callers and blocking filtered callers, same box, same index
#   bash scripts/probe/pod_probe.sh [vectors] [plain legs] [filtered threads]
N=${1:-10000000}
for pods in ${PODS:-1 0}; do
  echo "== VS_HNSW_PODS=$pods"
  VS_HNSW_PODS=$pods timeout 600 python scripts/probe/callers_probe.py $N 200 2 f32 ${2:-1x1,17x1,33x1,65x1} 2>&1 | grep -a "threads\|posted\|rror"
  if [ -z "$NOFILT" ]; then VS_HNSW_PODS=$pods PIPE_PROBE_FAST=1 PIPE_PROBE_MODS=${MODS:-10} timeout 900 python scripts/probe/pipe_probe.py $N 200 ${3:-1,17,64,128} 3 2>&1 | grep -a "pipe:\|rror"; fi
done
