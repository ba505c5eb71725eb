import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs, oracle
rng = np.random.default_rng(9)
n, dim = 30000, 256
base = rng.standard_normal((n, dim)).astype(np.float32); q = rng.standard_normal((1000, dim)).astype(np.float32)
ix = vs.HipUsearchIndex(dim, vs.HAMMING, quantization=vs.B1); ix.reserve(n)
ix.add_batch(np.arange(n, dtype=np.uint64), base)
o = oracle.OracleIndex(dim, oracle.HAMMING, quantization=oracle.B1); o.import_graph(ix.export_graph())
tk, td, tf = ix.exact_search_batch(q, 10)
for ef in (64, 128):
    ix.set_expansion_search(ef); o.set_expansion_search(ef)
    gk, gd, gf = ix.search_batch(q, 10); ok, od, of = o.search_batch(q, 10, threads=8)
    # distance-level recall: results within the exact 10th distance
    rg = np.mean([(gd[i] <= td[i, 9]).sum() / 10 for i in range(1000)]); ro = np.mean([(od[i] <= td[i, 9]).sum() / 10 for i in range(1000)])
    print(f"b1 ef {ef}: distance-recall@10 gpu {rg:.4f} oracle {ro:.4f}; mean 10th distance gpu {gd[:,9].mean():.3f} oracle {od[:,9].mean():.3f} exact {td[:,9].mean():.3f}")
