"""How long one launch of the pipelined walk takes as a function of the queries (workgroups, one CU each) it carries:
python scripts/probe/pipe_concurrency.py [vectors] [ef]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
q = np.ascontiguousarray(make_data(4096, dim, "lowrank", 4321, dev, 24).cpu().numpy())
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
del base
for nq in (1, 2, 4, 8, 16, 32, 48, 64, 128):
    for rep in range(3):
        ix.search_batch(q[:nq], k)
    p0 = ix.pipe_stats()["pipe_launches"]
    t0 = time.perf_counter()
    reps = 0
    off = 0
    while time.perf_counter() - t0 < 1.0:
        ix.search_batch(q[off:off + nq], k)
        off = (off + nq) % (4096 - nq)
        reps += 1
    dt = time.perf_counter() - t0
    print(f"nq {nq:4d}: {dt / reps * 1e3:.3f} ms per launch, {nq * reps / dt:.0f} QPS, pipe launches {ix.pipe_stats()['pipe_launches'] - p0} of {reps}", flush=True)
