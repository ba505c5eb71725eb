#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for t in random newest; do
  echo "== VS_HNSW_TIE=$t"
  VS_HNSW_TIE=$t python3 scripts/probe/duplicates_probe.py 2>&1 | grep -v amdgpu.ids
  VS_HNSW_TIE=$t python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "identical_vectors or build_recall or sequential_adds" 2>&1 | tail -2
done
