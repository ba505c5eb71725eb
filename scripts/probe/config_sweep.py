#!/usr/bin/env python3
"""Development probe: GPU search == oracle search on the GPU-built graph over a sweep of connectivity, expansion_add,
dimension, metric and storage type (the unit tests fix M = 16)."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vector_store_amd as vs, oracle
rng = np.random.default_rng(11)
bad = 0
for M, efa, dim, metric, kind in itertools.product((2, 5, 8, 24, 32), (16, 200), (7, 100, 260), ("cos", "l2sq", "ip"), ("f32", "f16")):
    n = 4000
    r = min(16, dim)
    w = rng.standard_normal((r, dim)).astype(np.float32) / np.sqrt(r)
    base = (rng.standard_normal((n, r)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim)).astype(np.float32))
    q = (rng.standard_normal((100, r)).astype(np.float32) @ w + 0.05 * rng.standard_normal((100, dim)).astype(np.float32))
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], connectivity=M, expansion_add=efa, expansion_search=80, quantization=vs.SCALARS[kind])
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    k, d, f = ix.search_batch(q, 10)
    g = ix.export_graph()
    assert (g["adj0"][g["adj0"] != 0xFFFFFFFF] < n).all()
    o = oracle.OracleIndex(dim, oracle.METRICS[metric], M, efa, 80, quantization=vs.SCALARS[kind])
    o.import_graph(g)
    ko, do, fo = o.search_batch(q, 10, threads=8)
    same = sum(np.array_equal(k[i, :f[i]], ko[i, :fo[i]]) for i in range(len(q)))
    tk, td, tf = ix.exact_search_batch(q, 10)
    rec = np.mean([len(set(tk[i].tolist()) & set(k[i].tolist())) / 10 for i in range(len(q))])
    close = np.allclose(d[:, :5], do[:, :5], rtol=2e-3 if kind == "f16" else 1e-5, atol=1e-5)
    flag = "" if (same >= 90 and close) else "  <-- CHECK"
    bad += bool(flag)
    print(f"M={M:2d} ef_add={efa:3d} dim={dim:3d} {metric:4s} {kind}: identical rows {same}/100, recall {rec:.3f}{flag}", flush=True)
print("configs needing a look:", bad)
