#!/bin/bash
# builds the sources under scratch_old/ (an earlier commit, extracted with git archive) ON THE GPU BOX and runs the lone-caller probe
# against it and against the in-tree library, same box, same index parameters
cd ${GRAFT_REPO_ROOT:-.}
( cd scratch_old/vector_store_amd/csrc && make -j32 ../libvs_hnsw.so >/dev/null 2>&1 ) || { echo old build failed; exit 1; }
for lib in scratch_old/vector_store_amd/libvs_hnsw.so vector_store_amd/libvs_hnsw.so; do
  echo "== $lib"
  VS_HNSW_LIB=$lib timeout 300 python scripts/probe/callers_probe.py ${1:-10000000} 200 1.5 f32 1x1,17x1 2>&1 | grep -a "threads\|rror"
done
