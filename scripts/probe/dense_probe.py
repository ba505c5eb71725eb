#!/usr/bin/env python3
"""Development aid: the usearch-order walk on a quantised index, per-launch time and the instance that ran.
    python scripts/probe/dense_probe.py [vectors] [ef] [quantization]     (VS_HNSW_WALK_DENSE=0 / VS_HNSW_WALK_DEBUG=1 for A/B and sizes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import vector_store_amd as vs
from bench import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 208
quant = {"f32": vs.F32, "f16": vs.F16, "i8": vs.I8, "b1": vs.B1}[sys.argv[3] if len(sys.argv) > 3 else "i8"]
dim, k, nq = 768, 10, 10000
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev)
q = make_data(nq, dim, "lowrank", 4321, dev)
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef, quantization=quant)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
ok = torch.empty((nq, k), dtype=torch.int64, device=dev)
od = torch.empty((nq, k), dtype=torch.float32, device=dev)
of = torch.empty((nq,), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
reps = 2 if os.environ.get("VS_HNSW_WALK_DEBUG") else 12
for r in range(reps + 3):
    if r == 3:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ix.search_batch_device(q.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
    if r < 3:
        torch.cuda.synchronize()
        print("launch", r, "instance", ix.walk_info()["last_instance"], flush=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"n {n} ef {ef} dense {os.environ.get('VS_HNSW_WALK_DENSE', 'default')}: {ms:.3f} ms per {nq} queries = {nq / ms / 1e3:.1f}k QPS, instance {ix.walk_info()['last_instance']}, stats {ix.stats(reset=True)}", flush=True)
