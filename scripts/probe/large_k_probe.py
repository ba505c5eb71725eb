import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data
for n in (1_000_000, 10_000_000):
    dev = torch.device("cuda:0")
    base = make_data(n, 768, "lowrank", 1234, dev, 24)
    q = make_data(8, 768, "lowrank", 4321, dev, 24).cpu().numpy()
    ix = vs.HipUsearchIndex(768, vs.COS); ix.reserve(n)
    ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, 768)
    ix.search(q[0], 2000)
    t = time.perf_counter()
    for i in range(8): k, d = ix.search(q[i], 2000)
    t1 = (time.perf_counter() - t) / 8
    t = time.perf_counter()
    for i in range(8): fk, fd = ix.filtered_search(q[i], 100, lambda key: key % 1000 == 7)
    t2 = (time.perf_counter() - t) / 8
    print(f"n={n}: limit 2000 -> {t1*1e3:.1f} ms per query; filter passing 0.1% of the keys, limit 100 -> {t2*1e3:.1f} ms per query ({len(fk)} found)", flush=True)
    del ix, base; torch.cuda.empty_cache()
