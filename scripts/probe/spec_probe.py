import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data
n, dim, k = 1000000, 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev); q = make_data(64, dim, "lowrank", 4321, dev)
ix = vs.HipUsearchIndex(dim, vs.COS, _stress=4); ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
for ef in (128, 200):
    ix.set_expansion_search(ef)
    hq = q.cpu().numpy()
    for i in range(3):
        ix.search_batch(hq[i:i+1], k)
    print("ef", ef, ix.stats(reset=True), flush=True)
