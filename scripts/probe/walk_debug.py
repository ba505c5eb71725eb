import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs, oracle
kind, metric, dim = "b1", "hamming", 256
n, nq = 100000, 300
rng = np.random.default_rng(9)
base = rng.standard_normal((n, dim)).astype(np.float32)
q = rng.standard_normal((nq, dim)).astype(np.float32)
for stress in (0, 32):
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind], _stress=stress)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = oracle.OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[kind])
    o.import_graph(ix.export_graph())
    for ef, k in ((256, 10), (300, 100), (400, 100), (512, 100), (512, 512)):
        ix.set_expansion_search(ef); o.set_expansion_search(ef)
        o.stats(reset=True); ix.stats(reset=True)
        gk, gd, gf = ix.search_batch(q, k)
        ok, od, of = o.search_batch(q, k, threads=8)
        so, sg = o.stats(), ix.stats()
        bad = [i for i in range(nq) if not (gf[i] == of[i] and np.array_equal(gk[i, :gf[i]], ok[i, :of[i]]))]
        print(f"stress {stress} ef {ef} k {k}: differing {len(bad)}/{nq}; evals gpu {sg['search_evals']} cpu {so['computed_distances']} hops gpu {sg['search_hops']} cpu {so['node_expansions']} q {sg['queries']}", flush=True)
        for i in bad[:3]:
            g, w = gk[i, :gf[i]].tolist(), ok[i, :of[i]].tolist()
            print("   query", i, "dups in gpu:", len(g) - len(set(g)), "only gpu:", [(x, float(gd[i, g.index(x)])) for x in g if x not in w][:5],
                  "only cpu:", [(x, float(od[i, w.index(x)])) for x in w if x not in g][:5])
