#!/bin/bash
# A/B of the team search kernels' speculative evaluation on one box: VS_HNSW_SPEC=0 switches it off.
#   scripts/probe/spec_ab.sh [vectors]
N=${1:-1000000}
for s in 1 0 1 0; do echo "VS_HNSW_SPEC=$s"; VS_HNSW_SPEC=$s python3 scripts/latency_probe.py $N 2>/dev/null | grep -A3 "team of 8" | head -4; done
