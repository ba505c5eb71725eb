#!/usr/bin/env python3
"""Development aid: what blocking / non-blocking callers get through the C ABI on one index, for A/B runs of the
dispatcher's settings (VS_HNSW_SERVICE_SLOTS, VS_HNSW_SERVICE_SPIN) -- the loop is libvs_callers' (bench.py's `boundary`).
    python scripts/probe/callers_probe.py [vectors] [ef] [seconds] [quantization] [threadsxinflight,...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import numpy as np
import torch
import vector_store_amd as vs
from bench import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
quant = {"f32": vs.F32, "f16": vs.F16, "i8": vs.I8, "b1": vs.B1}[sys.argv[4] if len(sys.argv) > 4 else "f32"]
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev)
q = make_data(10000, dim, "lowrank", 4321, dev).cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef, quantization=quant)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
truth, _, _ = ix.search_batch(q, k)  # recall column = agreement with the batch path
truth = np.ascontiguousarray(truth, dtype=np.uint64)


class Res(C.Structure):
    _fields_ = [("seconds", C.c_double), ("queries", C.c_uint64), ("qps", C.c_double), ("latency_min_ns", C.c_int64),
                ("latency_max_ns", C.c_int64)] + [(f"p{p:02d}_ns", C.c_int64) for p in (1, 10, 25, 50, 75, 90, 99)] + [
                ("recall_avg", C.c_double), ("errors", C.c_uint64), ("launches", C.c_uint64), ("team_launches", C.c_uint64)]


L = C.CDLL(os.path.join(ROOT, "vector_store_amd", "libvs_callers.so"))
L.vs_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint,
                             C.c_double, C.POINTER(Res)]
print(f"n {n} ef {ef} slots {os.environ.get('VS_HNSW_SERVICE_SLOTS', 'default')} spin {os.environ.get('VS_HNSW_SERVICE_SPIN', 'default')}", flush=True)
legs = [tuple(int(x) for x in leg.split("x")) for leg in sys.argv[5].split(",")] if len(sys.argv) > 5 else ((1, 1), (4, 1), (17, 1), (33, 1), (65, 1), (16, 16), (16, 256))
for threads, inflight in legs:
    r = Res()
    ps0 = ix.pod_stats() if hasattr(ix, "pod_stats") else None
    rc = L.vs_callers_run(ix.h, q.ctypes.data, q.shape[0], dim, k, truth.ctypes.data, threads, inflight, seconds, C.byref(r))
    print(f"  threads {threads:3d} x {inflight:3d} in flight: {r.qps:10.0f} QPS  min {r.latency_min_ns / 1e6:.3f} ms  agreement {r.recall_avg:.4f}  "
          f"launches {r.launches} (team {r.team_launches})  queries/launch {r.queries / max(r.launches, 1):.1f}  rc {rc} errors {r.errors}  p50 {r.p50_ns / 1e6:.3f} p99 {r.p99_ns / 1e6:.3f} ms", flush=True)
    if ps0 is not None:
        ps1 = ix.pod_stats()
        nq = ps1["plain_queries"] - ps0["plain_queries"]
        if nq:
            print(f"      posted to pods: {nq} queries; per query: {(ps1['plain_ns'] - ps0['plain_ns']) / nq / 1e3:.0f} us in the library, "
                  f"{(ps1['plain_wait_ns'] - ps0['plain_wait_ns']) / nq / 1e3:.0f} us waiting, {(ps1['plain_device_ns'] - ps0['plain_device_ns']) / nq / 1e3:.0f} us on the device; "
                  f"pods opened {ps1['pods_opened'] - ps0['pods_opened']}", flush=True)
