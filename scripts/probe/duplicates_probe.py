import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
rng = np.random.default_rng(5)
dim = 64
w = rng.standard_normal((16, dim)).astype(np.float32) / 4
for copies in (1, 5, 50, 500):
    uniq = 100000 // copies
    u = (rng.standard_normal((uniq, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((uniq, dim)).astype(np.float32))
    base = np.repeat(u, copies, axis=0)
    perm = rng.permutation(len(base)); base = base[perm]
    q = (rng.standard_normal((500, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((500, dim)).astype(np.float32))
    ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=128); ix.reserve(len(base))
    ix.add_batch(np.arange(len(base), dtype=np.uint64), base)
    k, d, f = ix.search_batch(q, 10)
    tk, td, tf = ix.exact_search_batch(q, 10)
    # distance-level recall: the 10th exact distance bounds what a correct answer may return
    ok = np.mean([(d[i, :f[i]] <= td[i, 9] * (1 + 1e-5) + 1e-7).sum() / 10 for i in range(len(q))])
    print(f"copies={copies}: found min {f.min()}, distance-recall@10 {ok:.3f}")
    if copies in (50,):
        import oracle
        for threads in (1, 8):
            o = oracle.OracleIndex(dim, oracle.COS, 16, 128, 128); o.reserve(len(base))
            o.add_batch(np.arange(len(base), dtype=np.uint64), base, threads=threads)
            ko, do, fo = o.search_batch(q, 10, threads=8)
            oko = np.mean([(do[i, :fo[i]] <= td[i, 9] * (1 + 1e-5) + 1e-7).sum() / 10 for i in range(len(q))])
            print(f"   oracle build ({threads} thread) + oracle search: distance-recall@10 {oko:.3f}")
        o2 = oracle.OracleIndex(dim, oracle.COS, 16, 128, 128); o2.import_graph(ix.export_graph())
        k2, d2, f2 = o2.search_batch(q, 10, threads=8)
        print(f"   oracle search on the GPU-built graph: {np.mean([(d2[i, :f2[i]] <= td[i, 9] * (1 + 1e-5) + 1e-7).sum() / 10 for i in range(len(q))]):.3f}")
