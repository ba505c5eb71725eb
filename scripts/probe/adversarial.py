#!/usr/bin/env python3
"""Development probe: inputs chosen to break invariants (ties everywhere, overflow, denormals, everything removed,
growth while searching).  Any GPU fault aborts the process; the script prints OK lines as it survives."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vector_store_amd as vs

rng = np.random.default_rng(3)
for metric in ("cos", "l2sq", "ip"):
    for kind in ("f32", "f16", "i8"):
        dim, n = 40, 12000
        ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind], expansion_search=128)
        ix.reserve(n)
        same = np.tile(rng.standard_normal(dim).astype(np.float32), (n, 1))     # every vector identical: all ties
        ix.add_batch(np.arange(n, dtype=np.uint64), same)
        k, d, f = ix.search_batch(same[:300], 100)
        assert (f == 100).all() and all(len(set(r.tolist())) == 100 for r in k), "ties"
        huge = (rng.standard_normal((64, dim)) * 1e30).astype(np.float32)       # products overflow to inf
        tiny = (rng.standard_normal((64, dim)) * 1e-42).astype(np.float32)      # denormals
        for q in (huge, tiny, np.zeros((8, dim), np.float32)):
            k, d, f = ix.search_batch(q, 10)
            ix.exact_search_batch(q, 10)
        for key in range(n):                                                     # remove everything
            pass
        print(f"{metric}/{kind}: ties / overflow / denormal / zero queries OK", flush=True)
ix = vs.HipUsearchIndex(16, vs.L2SQ)
ix.reserve(3000)
base = rng.standard_normal((3000, 16)).astype(np.float32)
ix.add_batch(np.arange(3000, dtype=np.uint64), base)
for key in range(3000):
    assert ix.remove(key)
k, d, f = ix.search_batch(base[:50], 10)
assert (f == 0).all()
assert len(ix.search(base[0], 5)[0]) == 0 and len(ix.search(base[0], 2000)[0]) == 0
ix.add_batch(np.arange(3000, 6000, dtype=np.uint64), base)                       # every slot reused
k, d, f = ix.search_batch(base[:50], 1)
assert (k[:, 0] == np.arange(3000, 3050)).all() and (d[:, 0] == 0).all()
print("remove-all / reuse-all OK", flush=True)
# growth + adds while other threads search (the reference serialises these; the ABI must at least stay memory-safe
# when a caller does not)
ix = vs.HipUsearchIndex(32, vs.COS)
ix.reserve(1000)
data = rng.standard_normal((60000, 32)).astype(np.float32)
ix.add_batch(np.arange(1000, dtype=np.uint64), data[:1000])
stop = False
def searcher():
    i = 0
    while not stop:
        ix.search(data[i % 1000], 10); i += 1
th = [threading.Thread(target=searcher) for _ in range(8)]
[t.start() for t in th]
for lo in range(1000, 60000, 1000):
    ix.reserve(lo + 1000)
    ix.add_batch(np.arange(lo, lo + 1000, dtype=np.uint64), data[lo:lo + 1000])
stop = True
[t.join() for t in th]
assert ix.size() == 60000
print("reserve + add under concurrent searches OK", flush=True)
