#!/usr/bin/env python3
"""Development aid: where a lone filtered walk spends its time.  Needs a library built with -DVS_WALK_PROFILE for the f32 arithmetic
 and prints the per-phase shader clocks of every round of a few filtered queries:
    VS_HNSW_LIB=vector_store_amd/libvs_hnsw_dbg.so VS_HNSW_WALK_DEBUG=1 python scripts/probe/filtered_phase_probe.py [vectors] [modulus]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vector_store_amd as vs
from bench import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
mod = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev)
q = make_data(8, dim, "lowrank", 4321, dev).cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=200)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
for i in range(4):
    print(f"--- query {i}, predicate key % {mod} == 0", file=sys.stderr, flush=True)
    keys, _ = ix.filtered_search(q[i], k, lambda key: key % mod == 0)
    print(f"    -> {len(keys)} results, filter stats {ix.filter_stats()}", file=sys.stderr, flush=True)
