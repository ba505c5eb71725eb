#!/bin/bash
# Phase clocks of the usearch-order walk: rebuilds the walk objects with -DVS_WALK_PROFILE ON THE GPU BOX (the box is
# discarded afterwards; the in-tree library stays the product build) and prints the per-query phase distribution.
#   KINDS="i8 b1" EF=128 N=1000000 bash scripts/probe/walk_profile.sh
cd ${GRAFT_REPO_ROOT:-.}
touch vector_store_amd/csrc/walk_device.hpp
make -C vector_store_amd/csrc -j16 EXTRA=-DVS_WALK_PROFILE >/dev/null 2>&1 || { echo build failed; exit 1; }
for q in ${KINDS:-i8 b1}; do
  echo "== $q ef ${EF:-128} n ${N:-1000000}"
  VS_HNSW_WALK_DEBUG=1 timeout 600 python3 bench.py --vectors ${N:-1000000} --quantization $q --ef ${EF:-128} --cpu-seconds 0 --boundary-seconds 0 --no-side-records --steps 2 --warmup 1 2>&1 | grep -a "^\[walk\]" | tail -13
done
