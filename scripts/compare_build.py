#!/usr/bin/env python3
"""Development aid: recall of GPU-built vs oracle-built graphs on the same data (needs a GPU)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vector_store_amd as vs
import oracle
from scripts.quick_perf import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nq, k = 1000, 10
print("cpus", os.cpu_count())
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, rank)
q = make_data(nq, dim, "lowrank", 4321, dev, rank)
hb, hq = base.cpu().numpy(), q.cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS)
ix.reserve(n)
t = time.time(); ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim); print("gpu build", time.time() - t)
tk, _, _ = ix.exact_search_batch(hq, k)
thr = os.cpu_count()
o = oracle.OracleIndex(dim, oracle.COS)
o.reserve(n)
t = time.time(); o.add_batch(np.arange(n, dtype=np.uint64), hb, threads=thr); tb = time.time() - t
print(f"cpu build {tb:.1f}s = {n/tb:.0f} vec/s on {thr} threads")
o1 = None
if n <= 50000:
    o1 = oracle.OracleIndex(dim, oracle.COS); o1.reserve(n)
    t = time.time(); o1.add_batch(np.arange(n, dtype=np.uint64), hb, threads=1); print("cpu 1-thread build", time.time() - t)
og = oracle.OracleIndex(dim, oracle.COS)
og.import_graph(ix.export_graph())
def rec(keys): return np.mean([len(set(tk[i].tolist()) & set(keys[i].tolist())) / k for i in range(nq)])
for ef in (64, 128, 256):
    ix.set_expansion_search(ef); o.set_expansion_search(ef); og.set_expansion_search(ef)
    gk, _, _ = ix.search_batch(hq, k)
    t = time.time(); ck, _, _ = o.search_batch(hq, k, threads=thr); ts = time.time() - t
    cgk, _, _ = og.search_batch(hq, k, threads=thr)
    line = f"ef={ef}: recall gpu-graph/gpu-search {rec(gk):.4f} | gpu-graph/cpu-search {rec(cgk):.4f} | cpu-graph/cpu-search {rec(ck):.4f} (cpu {nq/ts:.0f} qps)"
    if o1 is not None:
        o1.set_expansion_search(ef); k1, _, _ = o1.search_batch(hq, k, threads=thr); line += f" | cpu-1thread-graph {rec(k1):.4f}"
    print(line, flush=True)
