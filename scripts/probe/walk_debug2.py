import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs, oracle
def _dataset(n, dim, seed):
    rng = np.random.default_rng(seed)
    r = min(16, dim)
    w = rng.standard_normal((r, dim)).astype(np.float32) / np.sqrt(r)
    return (rng.standard_normal((n, r)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)
n, dim = 20000, 24
base = _dataset(n + 600, dim, 29)
for metric in ("cos", "l2sq"):
    for poison in (False, True):
        ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], expansion_search=200)
        ix.reserve(n + 64)
        poisoned = base[:n].copy()
        if poison:
            poisoned[::97, 5] = np.nan
            poisoned[50::97, 7] = np.inf
        ix.add_batch(np.arange(n, dtype=np.uint64), poisoned)
        q = base[n:n + 600].copy()
        q[::3, 1] = np.nan
        q[1::3, 2] = np.inf
        o = oracle.OracleIndex(dim, oracle.METRICS[metric]); o.import_graph(ix.export_graph()); o.set_expansion_search(200)
        for qi in (0, 1, 2):
            for k in (600, 2000):
                bk, bd = ix.search(q[qi], k)
                ok, od = o.search(q[qi], k)
                st = ix.stats(reset=True)
                print(metric, "poison", poison, "q", qi, "k", k, "gpu", len(bk), len(set(bk.tolist())), "cpu", len(ok), "dist head", bd[:3], bd[-3:], "evals", st["search_evals"], "hops", st["search_hops"], flush=True)
