"""Tiny index, k above its size, lone searches between adds: pods against the batch path."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
rng = np.random.default_rng(1)
dim = 20
ix = vs.HipUsearchIndex(dim, vs.L2SQ)
ix.reserve(400)
ix.set_expansion_search(64)
nxt = 0
bad = 0
for phase in range(30):
    n = int(rng.integers(1, 30))
    keys = np.arange(nxt, nxt + n, dtype=np.uint64)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    nxt += n
    if phase % 2:
        ix.add_batch(keys, vecs)
    else:
        for i in range(n):
            ix.add(int(keys[i]), vecs[i])
    if phase % 5 == 4 and nxt > 10:
        assert ix.remove(nxt - 3)
    size = ix.size()
    k = min(size + 3, 250)
    for r in range(4):
        q = rng.standard_normal(dim).astype(np.float32)
        gk, gd = ix.search(q, k)
        bk, bd, bf = ix.search_batch(q[None, :], k)
        if len(gk) != int(bf[0]) or sorted(gk.tolist()) != sorted(bk[0][: bf[0]].tolist()):
            bad += 1
            missing = sorted(set(bk[0][: bf[0]].tolist()) - set(gk.tolist()))
            print(f"phase {phase} query {r}: size {size} lone found {len(gk)} batch found {int(bf[0])} missing {missing[:10]} pods {ix.pod_stats()}", flush=True)
print("mismatches", bad, "pod stats", ix.pod_stats(), "pipe", ix.pipe_stats(), flush=True)
