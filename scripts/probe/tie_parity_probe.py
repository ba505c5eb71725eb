import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs, oracle
rng = np.random.default_rng(9)
for kind, metric, dim in (("b1", "hamming", 256), ("b1", "hamming", 64), ("i8", "l2sq", 32), ("i8", "ip", 64)):
    n = 30000
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((1000, dim)).astype(np.float32)
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind]); ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = oracle.OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[kind]); o.import_graph(ix.export_graph())
    for ef, k in ((64, 10), (100, 50), (128, 100)):
        ix.set_expansion_search(ef); o.set_expansion_search(ef)
        gk, gd, gf = ix.search_batch(q, k)
        ok, od, of = o.search_batch(q, k, threads=8)
        bad = sum(not (gf[i] == of[i] and np.array_equal(gd[i, :gf[i]], od[i, :of[i]])) for i in range(len(q)))
        bad_ids = sum(not (gf[i] == of[i] and np.array_equal(gk[i, :gf[i]], ok[i, :of[i]])) for i in range(len(q)))
        print(f"{kind} {metric} dim {dim} ef {ef} k {k}: queries whose distance list / id list differs from the oracle's: {bad} / {bad_ids} of 1000", flush=True)
