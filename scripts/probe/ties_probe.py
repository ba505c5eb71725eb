import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs, oracle
rng = np.random.default_rng(3)
dim, n = 40, 12000
same = np.tile(rng.standard_normal(dim).astype(np.float32), (n, 1))
for metric in ("cos", "l2sq"):
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], expansion_search=128); ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), same)
    k, d, f = ix.search_batch(same[:300], 100)
    dup = sum(len(set(r[:fi].tolist())) != fi for r, fi in zip(k, f))
    print(metric, "gpu found min/max", f.min(), f.max(), "rows with duplicates", dup, "dist max", d[d < 1e30].max())
    o = oracle.OracleIndex(dim, oracle.METRICS[metric], 16, 128, 128); o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64), same, threads=1)
    ko, do, fo = o.search_batch(same[:300], 100, threads=4)
    print(metric, "oracle found min/max", fo.min(), fo.max())
    o2 = oracle.OracleIndex(dim, oracle.METRICS[metric], 16, 128, 128); o2.import_graph(ix.export_graph())
    k2, d2, f2 = o2.search_batch(same[:300], 100, threads=4)
    print(metric, "oracle on gpu graph found min/max", f2.min(), f2.max(), "same rows", sum(np.array_equal(a[:x], b[:y]) for a, x, b, y in zip(k, f, k2, f2)))
