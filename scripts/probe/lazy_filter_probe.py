import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = 768
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
q = make_data(16, dim, "lowrank", 4321, dev, 24).cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
for mod in (2, 10, 100, 1000):
    calls = [0]
    def pred(key, mod=mod):
        calls[0] += 1
        return key % mod == 1
    s0 = ix.filter_stats()
    t0 = time.time()
    for i in range(8):
        fk, fd = ix.filtered_search(q[i], 10, pred)
    dt = (time.time() - t0) / 8
    s1 = ix.filter_stats()
    print(f"n {n} selectivity 1/{mod}: {dt*1e3:.1f} ms per query (python predicate), rounds/query {(s1['lazy_rounds']-s0['lazy_rounds'])/8:.1f}, predicate calls/query {calls[0]/8:.0f}, found {len(fk)}", flush=True)
