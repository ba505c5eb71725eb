#!/usr/bin/env python3
"""BASELINE.json configs[3] (100M x 768 cosine, key-range shards, per-shard top-k merged) with all eight 12.5M-vector
shards resident on ONE MI355X -- what 288 GB of HBM allows with f16 storage (154 GB of vectors + 16 GB of graph).
The same steps as `bench.py --mode shard` runs across ranks (independent graphs, every shard answers every query,
vs_topk_merge_device picks the global top-k), executed shard after shard on one device.  Prints one JSON line.

    python scripts/c4_one_gpu.py [shards=8] [vectors_per_shard=12500000] [quantization=f16]
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

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 12_500_000
quant = sys.argv[3] if len(sys.argv) > 3 else "f16"
dim, k, nq, chunk = 768, 10, 10_000, 500_000
dev = torch.device("cuda:0")
q = make_data(nq, dim, "lowrank", 4321, dev, 24)
shards = []
t0 = time.time()
for s in range(S):
    ix = vs.HipUsearchIndex(dim, vs.COS, quantization=vs.SCALARS[quant])
    ix.reserve(per)
    for c0 in range(0, per, chunk):
        m = min(chunk, per - c0)
        data = make_data(m, dim, "lowrank", 1234 + 7919 * s + c0 // chunk, dev, 24)
        ix.add_batch_device(np.arange(s * per + c0, s * per + c0 + m, dtype=np.uint64), data.data_ptr(), m, dim)
        del data
    shards.append(ix)
torch.cuda.synchronize()
build_s = time.time() - t0
free_b, total_b = torch.cuda.mem_get_info()

pk = torch.empty((S, nq, k), dtype=torch.int64, device=dev)
pd = torch.empty((S, nq, k), dtype=torch.float32, device=dev)
pf = torch.empty((nq,), dtype=torch.int32, device=dev)
ok = torch.empty((nq, k), dtype=torch.int64, device=dev)
od = torch.empty((nq, k), dtype=torch.float32, device=dev)
of = torch.empty((nq,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def merged(fn):
    for s, ix in enumerate(shards):
        fn(ix, pk[s].data_ptr(), pd[s].data_ptr())
    vs.topk_merge_device(pk.data_ptr(), pd.data_ptr(), S, nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), st)


t0 = time.time()
merged(lambda ix, kp, dp: ix.exact_search_batch_device(q.data_ptr(), nq, k, kp, dp, pf.data_ptr(), st))
torch.cuda.synchronize()
exact_s = time.time() - t0
truth = ok.cpu().numpy().copy()
out = {"workload": f"{S} shards x {per} x {dim} cos {quant} on one MI355X = {S * per} vectors, {nq} queries/step, top-{k}",
       "build": {"seconds": build_s, "vectors_per_s": S * per / build_s}, "exact_ground_truth_seconds": exact_s,
       "hbm_used_gb": (total_b - free_b) / 1e9, "sweep": []}
walk = lambda ix, kp, dp: ix.search_batch_device(q.data_ptr(), nq, k, kp, dp, pf.data_ptr(), st)
for ef in (96, 128, 160, 200):
    for ix in shards:
        ix.set_expansion_search(ef)
    merged(walk)
    torch.cuda.synchronize()
    rec = recall_at_k(truth, ok.cpu().numpy())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        merged(walk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    out["sweep"].append({"ef": ef, "recall_at_10": round(rec, 4), "ms_per_step": ms, "queries_per_s": nq / ms * 1e3})
    if rec >= 0.95:
        break
print(json.dumps(out))
