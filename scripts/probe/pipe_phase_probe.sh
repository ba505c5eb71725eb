#!/bin/bash
# Where a lone PLAIN pipelined walk spends its clocks: a -DVS_WALK_PROFILE build of the library in a scratch copy (the tree's objects stay
# as they are), launches instead of pods (the dispatcher prints the per-query phase clocks with VS_HNSW_WALK_DEBUG=1).
#   scripts/probe/pipe_phase_probe.sh [vectors]      VS_PROFILE_LEVEL=1 phases of a hop; 2: "pushes, top" split in who passes / merge / pushes
#   (printed in the places of atomics / verdicts / schedule); 3: the urgent job parts (row / list / distances / report, count, neighbours x 100)
set -u
N=${1:-10000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/csrc_prof && mkdir -p /tmp/csrc_prof/csrc /tmp/csrc_prof/include && cp $R/vector_store_amd/csrc/*.h* $R/vector_store_amd/csrc/*.cpp $R/vector_store_amd/csrc/Makefile /tmp/csrc_prof/csrc/ && cp $R/include/*.h /tmp/csrc_prof/include/
mkdir -p /tmp/csrc_prof/vsa && sed -i 's|\.\./\.\./include|../include|g' /tmp/csrc_prof/csrc/*.h* /tmp/csrc_prof/csrc/*.cpp /tmp/csrc_prof/csrc/Makefile
cd /tmp/csrc_prof/csrc && make -j16 EXTRA=-DVS_WALK_PROFILE=${VS_PROFILE_LEVEL:-1} ../libvs_hnsw.so > /tmp/csrc_prof/build.log 2>&1 || { tail -20 /tmp/csrc_prof/build.log; exit 1; }
cd $R && VS_HNSW_LIB=/tmp/csrc_prof/libvs_hnsw.so VS_HNSW_PODS=0 VS_HNSW_WALK_DEBUG=1 python3 - "$N" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import vector_store_amd as vs
from bench import make_data
n = int(sys.argv[1]); dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev)
q = make_data(64, dim, "lowrank", 4321, dev).cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=200)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
for i in range(12):
    ix.search(q[i], k)
PY
