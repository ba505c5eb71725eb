import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data
n, dim, k = 1000000, 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev); q = make_data(4096, dim, "lowrank", 4321, dev)
for mode in ("never", "mid", "always"):
    os.environ["VS_HNSW_TEAM"] = mode
    ix = vs.HipUsearchIndex(dim, vs.COS); ix.reserve(n)
    ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
    ix.set_expansion_search(128)
    ok = torch.empty((4096, k), dtype=torch.int64, device=dev); od = torch.empty((4096, k), dtype=torch.float32, device=dev); of = torch.empty((4096,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    out = []
    for nq in (128, 256, 384, 512, 768, 1024):
        for _ in range(3): ix.search_batch_device(q.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ix.search_batch_device(q.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        out.append(f"{nq}: {e0.elapsed_time(e1)/20:.3f}")
    print(mode, " ms per batch ->", "  ".join(out), flush=True)
    del ix
