#!/usr/bin/env python3
"""Development aid: where a lone UNFILTERED walk spends its time (team form of the usearch-order walk, LDS instance), on a library
built with -DVS_WALK_PROFILE for the f32 arithmetic:
    VS_HNSW_LIB=vector_store_amd/libvs_hnsw_dbg.so VS_HNSW_ORDER=usearch VS_HNSW_WALK_DEBUG=1 python scripts/probe/lone_phase_probe.py [vectors] [ef]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vector_store_amd as vs
from bench import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev)
q = make_data(64, dim, "lowrank", 4321, dev).cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
for nq in (1, 1, 1, 16):
    print(f"--- {nq} queries in one launch", file=sys.stderr, flush=True)
    ix.search_batch(q[:nq], k)
