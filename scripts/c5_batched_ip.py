#!/usr/bin/env python3
"""BASELINE.json configs[4]: batched search, q = 256, 10M x 768 inner product (base L2-normalised).
Times (a) the exact block-distance path (split-bf16 MFMA nomination + f32 re-score + certificate; VS_HNSW_EXACT=f32: the f32-input MFMA
path of round 1) and (b) the HNSW walk on
the same 256-query batches, with recall of (b) against (a).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vector_store_amd as vs
from bench import make_data, recall_at_k

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dim, k, nq, batches = 768, 10, 256, 8
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
base /= base.norm(dim=1, keepdim=True)
q = make_data(nq * batches, dim, "lowrank", 4321, dev, 24)
q /= q.norm(dim=1, keepdim=True)
ix = vs.HipUsearchIndex(dim, vs.IP, expansion_search=200)
ix.reserve(n)
t = time.time()
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
build_s = time.time() - t
ok = torch.empty((nq * batches, k), dtype=torch.int64, device=dev)
od = torch.empty((nq * batches, k), dtype=torch.float32, device=dev)
of = torch.empty((nq * batches,), dtype=torch.int32, device=dev)
tk = torch.empty_like(ok)
s = torch.cuda.current_stream().cuda_stream


def timed(fn):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in range(batches):
        fn(b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / batches


def exact(b):
    o = b * nq
    ix.exact_search_batch_device(q[o:].data_ptr(), nq, k, tk[o:].data_ptr(), od[o:].data_ptr(), of[o:].data_ptr(), s)


def walk(b):
    o = b * nq
    ix.search_batch_device(q[o:].data_ptr(), nq, k, ok[o:].data_ptr(), od[o:].data_ptr(), of[o:].data_ptr(), s)


exact_ms = timed(exact)
walk_ms = timed(walk)
rec = recall_at_k(tk.cpu().numpy(), ok.cpu().numpy())
flops = 2.0 * nq * n * dim
print(json.dumps({"workload": f"{n}x{dim} ip (unit vectors), batches of {nq} queries, top-{k}",
                  "exact_mfma": {"ms_per_batch": exact_ms, "queries_per_s": nq / exact_ms * 1e3, "tflops": flops / exact_ms / 1e9,
                                 "frac_of_157_tflops": flops / exact_ms / 1e9 / 157.0},
                  "hnsw_walk_ef200": {"ms_per_batch": walk_ms, "queries_per_s": nq / walk_ms * 1e3, "recall_at_10_vs_exact": rec},
                  "build_vectors_per_s": n / build_s}))
