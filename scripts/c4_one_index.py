#!/usr/bin/env python3
"""BASELINE.json configs[3]'s data set (100M x 768 cosine) in ONE index handle on ONE MI355X: f16 storage, 154 GB of
vectors + 14 GB of graph in HBM, 100M > 2^26 slots, i.e. beyond the plain visited tags -- the wide-tag instances of the
insert and search kernels take over from member 67,108,865 on (DESIGN.md "Limits").  Prints one JSON line.

    python scripts/c4_one_index.py [vectors=100000000] [quantization=f16]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vector_store_amd as vs
from bench import make_data, recall_at_k

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
quant = sys.argv[2] if len(sys.argv) > 2 else "f16"
dim, k, nq, chunk = 768, 10, 10_000, 1_000_000
dev = torch.device("cuda:0")
q = make_data(nq, dim, "lowrank", 4321, dev, 24)
ix = vs.HipUsearchIndex(dim, vs.COS, quantization=vs.SCALARS[quant])
ix.reserve(n)
t0 = time.time()
marks = []
for c0 in range(0, n, chunk):
    m = min(chunk, n - c0)
    data = make_data(m, dim, "lowrank", 1234 + c0 // chunk, dev, 24)
    ix.add_batch_device(np.arange(c0, c0 + m, dtype=np.uint64), data.data_ptr(), m, dim)
    del data
    if (c0 // chunk) % 10 == 9:
        torch.cuda.synchronize()
        marks.append((c0 + m, time.time() - t0))
torch.cuda.synchronize()
build_s = time.time() - t0
free_b, total_b = torch.cuda.mem_get_info()
st = torch.cuda.current_stream().cuda_stream
keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
found = torch.empty((nq,), dtype=torch.int32, device=dev)
nt = 2000  # ground truth for the first 2,000 queries: 100M x 768 x 2,000 = 0.3 PFLOP through the MFMA path
t0 = time.time()
ix.exact_search_batch_device(q.data_ptr(), nt, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
torch.cuda.synchronize()
exact_s = time.time() - t0
truth = keys[:nt].cpu().numpy().copy()
out = {"workload": f"ONE index: {n} x {dim} cos {quant} on one MI355X, {nq} queries/step, top-{k}",
       "build": {"seconds": build_s, "vectors_per_s": n / build_s, "progress": [(a, round(b, 1)) for a, b in marks]},
       "exact_ground_truth_seconds": exact_s, "hbm_used_gb": (total_b - free_b) / 1e9, "memory_info": ix.memory_info(), "sweep": []}
for ef in (128, 160, 200, 256, 320, 384):
    ix.set_expansion_search(ef)
    ix.search_batch_device(q.data_ptr(), nq, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
    torch.cuda.synchronize()
    rec = recall_at_k(truth, keys[:nt].cpu().numpy())
    ix.stats(reset=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ix.search_batch_device(q.data_ptr(), nq, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    s = ix.stats()
    eq, hq = s["search_evals"] / s["queries"], s["search_hops"] / s["queries"]
    bq = eq * ix.bytes_per_vector() + hq * 132 + dim * 4
    out["sweep"].append({"ef": ef, "recall_at_10": round(rec, 4), "ms_per_step": ms, "queries_per_s": nq / ms * 1e3,
                         "evals_per_query": eq, "hbm_frac": bq * nq / (ms * 1e-3) / 8e12, "visited_overflow": s["visited_overflow"]})
    if rec >= 0.95:
        break
print(json.dumps(out))
