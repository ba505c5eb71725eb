"""A few lone plain queries with VS_HNSW_WALK_DEBUG=1: the pipelined walk's per-query counters (hops, early posts, misses, redone).
python scripts/probe/lone_pipe_debug.py [vectors] [ef]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("VS_HNSW_WALK_DEBUG", "1")
import vector_store_amd as vs
from bench import make_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
q = np.ascontiguousarray(make_data(64, dim, "lowrank", 4321, dev, 24).cpu().numpy())
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
del base
for i in range(12):
    t = time.perf_counter()
    ix.search(q[i], k)
    print(f"query {i}: {1e3 * (time.perf_counter() - t):.3f} ms", file=sys.stderr, flush=True)
