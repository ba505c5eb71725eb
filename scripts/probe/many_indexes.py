#!/usr/bin/env python3
"""Development probe: thousands of small per-partition indexes (reference usearch.rs:704-705, 766-778: one index object
per partition of a local index, reserve increment 1,000): creation, first adds, a search each; time and HBM."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vector_store_amd as vs
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dim, per = 64, 200
rng = np.random.default_rng(0)
data = rng.standard_normal((per, dim)).astype(np.float32)
free0, _ = torch.cuda.mem_get_info()
t = time.perf_counter()
ixs = [vs.HipUsearchIndex(dim, vs.COS) for _ in range(P)]
t_create = time.perf_counter() - t
t = time.perf_counter()
for ix in ixs:
    ix.reserve(1000)
t_reserve = time.perf_counter() - t
t = time.perf_counter()
for p, ix in enumerate(ixs):
    ix.add_batch(np.arange(per, dtype=np.uint64) + np.uint64(p << 20), data)
t_add = time.perf_counter() - t
t = time.perf_counter()
hits = 0
for p, ix in enumerate(ixs):
    k, d = ix.search(data[p % per], 5)
    hits += int(k[0] == (p << 20) + p % per)
t_search = time.perf_counter() - t
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
print(f"{P} indexes: create {t_create:.2f}s, reserve(1000) {t_reserve:.2f}s, add {per} each {t_add:.2f}s ({P*per/t_add:.0f} vec/s), "
      f"one search each {t_search:.2f}s ({P/t_search:.0f} qps), self-hits {hits}/{P}, HBM {(free0-free1)/2**20:.0f} MiB "
      f"({(free0-free1)/P/1024:.0f} KiB per index)")
