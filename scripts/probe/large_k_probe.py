#!/usr/bin/env python3
"""limit 1,000..10,000 (reference httproutes.rs:842-847; benchmark CLI up to 10,000) at the headline size: the wide walk
(global visited bitmap, `top` of 2,048 / 10,240 entries in LDS) against the exhaustive ranking it replaces.
    python scripts/probe/large_k_probe.py [vectors=10000000]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, nq = 768, 1000
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
q = make_data(nq, dim, "lowrank", 4321, dev, 24)
ix = vs.HipUsearchIndex(dim, vs.COS)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
qh = q.cpu().numpy()
out = {"workload": f"{n}x{dim} cos, {nq} queries per launch", "points": []}
st = torch.cuda.current_stream().cuda_stream
for k, ef in ((1000, 1000), (2000, 2000), (5000, 5000), (10000, 10000), (10, 2000)):
    ix.set_expansion_search(ef)
    keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    found = torch.empty((nq,), dtype=torch.int32, device=dev)
    ix.stats(reset=True)
    ix.search_batch_device(q.data_ptr(), nq, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        ix.search_batch_device(q.data_ptr(), nq, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    s = ix.stats()
    eq = s["search_evals"] / max(s["queries"], 1)
    got = keys.cpu().numpy()
    # exhaustive ranking (the path a limit beyond 10,240 takes) for 20 of the queries: exact top-k
    rec, t0 = [], time.time()
    ix.set_expansion_search(20000)
    for i in range(20):
        ek, ed = ix.search(qh[i], 12000)
        rec.append(len(set(ek[:k].tolist()) & set(got[i].tolist())) / k)
    exhaustive_ms = (time.time() - t0) / 20 * 1e3
    out["points"].append({"k": k, "ef": ef, "queries_per_s": nq / ms * 1e3, "ms_per_launch": ms, "evals_per_query": eq,
                          "hbm_frac": (eq * dim * 4) * nq / (ms * 1e-3) / 8e12, "recall_vs_exhaustive": float(np.mean(rec)),
                          "exhaustive_ms_per_query": exhaustive_ms})
    print(out["points"][-1], flush=True)
print(json.dumps(out))
