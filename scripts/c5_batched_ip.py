#!/usr/bin/env python3
"""BASELINE.json configs[4]: batched search, q = 256, 10M x 768 inner product (base L2-normalised).
Times (a) the exact block-distance path (round 3: ONE bf16 MFMA product per score over the index's bf16 plane + f32 re-score +
certificate; VS_HNSW_EXACT=bf16x3: round 2's three split-bf16 products over the f32 rows; VS_HNSW_EXACT=f32: round 1's f32-input
MFMA path) and (b) the HNSW walk on the same 256-query batches, with recall of (b) against (a).  Prints one JSON line.
Fractions are against the dense bf16 MFMA peak (2,500 TFLOP/s) for the products actually issued and against the 8 TB/s HBM peak
for the bytes the nomination pass has to stream (the bench.py `configs` record of configs[4] reports the same)."""
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


x0 = ix.exact_stats()
exact_ms = timed(exact)
x1 = ix.exact_stats()
walk_ms = timed(walk)
mode = os.environ.get("VS_HNSW_EXACT", "")
plane8 = x1.get("plane8_batches", 0) > x0.get("plane8_batches", 0) and x1.get("plane8_fallbacks", 0) == x0.get("plane8_fallbacks", 0)
plane = not plane8 and x1["plane_batches"] > x0["plane_batches"] and x1["plane_fallbacks"] == x0["plane_fallbacks"]
split = not plane8 and not plane and x1["block_batches"] > x0["block_batches"] and x1["block_fallbacks"] == x0["block_fallbacks"]
# (int8 plane, round 6: one byte per element + one f32 scale per row; the int8 matrix pipe's dense peak is twice the bf16 one)
products, row_bytes, peak = ((1, (dim + 127) // 128 * 128 + 4, 5000.0) if plane8 else (1, 2 * ((dim + 63) // 64 * 64), 2500.0) if plane else
                             (3, 4 * dim, 2500.0) if split else (1, 4 * dim, 157.3))
rec = recall_at_k(tk.cpu().numpy(), ok.cpu().numpy())
flops = 2.0 * nq * n * dim
print(json.dumps({"workload": f"{n}x{dim} ip (unit vectors), batches of {nq} queries, top-{k}",
                  "exact_mfma": {"path": "int8 plane, 1 product" if plane8 else "bf16 plane, 1 product" if plane else "split bf16, 3 products" if split else "f32-input MFMA",
                                 "ms_per_batch": exact_ms, "queries_per_s": nq / exact_ms * 1e3, "f32_equivalent_tflops": flops / exact_ms / 1e9,
                                 "issued_tflops": products * flops / exact_ms / 1e9, "mfma_peak_tflops": peak,
                                 "frac_of_mfma_peak": products * flops / exact_ms / 1e9 / peak,
                                 "streamed_gb_per_batch": n * row_bytes / 1e9, "frac_of_8tbs": n * row_bytes / (exact_ms * 1e-3) / 8e12,
                                 "stats": {k_: x1[k_] - x0[k_] for k_ in x1}},
                  "hnsw_walk_ef200": {"ms_per_batch": walk_ms, "queries_per_s": nq / walk_ms * 1e3, "recall_at_10_vs_exact": rec},
                  "build_vectors_per_s": n / build_s}))
