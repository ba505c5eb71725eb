#!/usr/bin/env python3
"""Development aid: kernel-only and host-API latency of small query batches."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vector_store_amd as vs
from bench import make_data
n, dim, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev); q = make_data(4096, dim, "lowrank", 4321, dev)
for mode, stress in (("one wave per query", 8), ("team of 8 waves per query", 4)):
    print(mode, flush=True)
    ix = vs.HipUsearchIndex(dim, vs.COS, _stress=stress); ix.reserve(n)
    ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
    ix.set_expansion_search(128)
    ok = torch.empty((4096, k), dtype=torch.int64, device=dev); od = torch.empty((4096, k), dtype=torch.float32, device=dev); of = torch.empty((4096,), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    hq = q.cpu().numpy()
    for nq in (1, 4, 16, 64, 256, 1024, 4096):
        for _ in range(3): ix.search_batch_device(q.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): ix.search_batch_device(q.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        kms = e0.elapsed_time(e1) / reps
        t = time.perf_counter()
        for _ in range(reps): ix.search_batch(hq[:nq], k)
        hms = (time.perf_counter() - t) / reps * 1e3
        t = time.perf_counter()
        if nq == 1:
            for _ in range(reps): ix.search(hq[0], k)
            sms = (time.perf_counter() - t) / reps * 1e3
        else: sms = float("nan")
        print(f"nq={nq}: kernel {kms:.3f} ms, host batch API {hms:.3f} ms, single-query API {sms:.3f} ms", flush=True)
