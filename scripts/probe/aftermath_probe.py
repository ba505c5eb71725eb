#!/usr/bin/env python3
"""Development aid: does a leg of single-query / filtered callers leave the device slower for the batch kernel that follows?
    python scripts/probe/aftermath_probe.py [vectors]"""
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
dim, k, nq, ef = 768, 10, 10000, 128
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev)
qd = make_data(nq, dim, "lowrank", 4321, dev)
q = qd.cpu().numpy()
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
ok = torch.empty((nq, k), dtype=torch.int64, device=dev)
od = torch.empty((nq, k), dtype=torch.float32, device=dev)
of = torch.empty((nq,), dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
truth, _, _ = ix.search_batch(q, k)
truth = np.ascontiguousarray(truth, dtype=np.uint64)


def batch_ms(tag):
    for _ in range(2):
        ix.search_batch_device(qd.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ix.search_batch_device(qd.data_ptr(), nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:46s} batch kernel {e0.elapsed_time(e1) / 10:.3f} ms per {nq} queries", flush=True)


class Res(C.Structure):
    _fields_ = [("seconds", C.c_double), ("queries", C.c_uint64), ("qps", C.c_double), ("latency_min_ns", C.c_int64),
                ("latency_max_ns", C.c_int64)] + [(f"p{p:02d}_ns", C.c_int64) for p in (1, 10, 25, 50, 75, 90, 99)] + [
                ("recall_avg", C.c_double), ("errors", C.c_uint64), ("launches", C.c_uint64), ("team_launches", C.c_uint64)]


L = C.CDLL(os.path.join(ROOT, "vector_store_amd", "libvs_callers.so"))
L.vs_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_double, C.POINTER(Res)]
L.vs_callers_run_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint, C.c_double, C.POINTER(Res), C.POINTER(C.c_uint64)]
batch_ms("fresh")
for threads, inflight in ((17, 1), (16, 256)):
    r = Res()
    L.vs_callers_run(ix.h, q.ctypes.data, nq, dim, k, truth.ctypes.data, threads, inflight, 1.5, C.byref(r))
    batch_ms(f"after {threads} x {inflight} callers ({r.qps:.0f} QPS)")
for threads in (17, 64):
    r, extra = Res(), (C.c_uint64 * 4)()
    L.vs_callers_run_filtered(ix.h, q.ctypes.data, nq, dim, k, 10, threads, 1.5, C.byref(r), extra)
    batch_ms(f"after {threads} filtered callers ({r.qps:.0f} QPS)")
import time
time.sleep(3)
batch_ms("3 s later")
torch.cuda.empty_cache()
batch_ms("after empty_cache")
