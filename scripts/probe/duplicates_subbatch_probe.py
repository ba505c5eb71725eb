import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
rng = np.random.default_rng(5)
dim = 64
w = rng.standard_normal((16, dim)).astype(np.float32) / 4
copies = 50
uniq = 40000 // copies
u = (rng.standard_normal((uniq, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((uniq, dim)).astype(np.float32))
base = np.repeat(u, copies, axis=0)
base = base[rng.permutation(len(base))]
q = (rng.standard_normal((500, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((500, dim)).astype(np.float32))
for sb in ("32768", "256", "16", "1"):
    os.environ["VS_HNSW_MAX_SUBBATCH"] = sb
    ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=128); ix.reserve(len(base))
    ix.add_batch(np.arange(len(base), dtype=np.uint64), base)
    k, d, f = ix.search_batch(q, 10)
    tk, td, tf = ix.exact_search_batch(q, 10)
    ok = np.mean([(d[i, :f[i]] <= td[i, 9] * (1 + 1e-5) + 1e-7).sum() / 10 for i in range(len(q))])
    print(f"max sub-batch {sb}: distance-recall@10 {ok:.3f}", flush=True)
import oracle
o = oracle.OracleIndex(dim, oracle.COS, 16, 128, 128); o.reserve(len(base))
o.add_batch(np.arange(len(base), dtype=np.uint64), base, threads=1)
ko, do, fo = o.search_batch(q, 10, threads=8)
print("oracle:", np.mean([(do[i, :fo[i]] <= td[i, 9] * (1 + 1e-5) + 1e-7).sum() / 10 for i in range(len(q))]))
