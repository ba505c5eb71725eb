"""Filtered search through the C ABI with the reference's call pattern (blocking callers, native counted predicate):
QPS, latency, predicate calls, walk launches, evaluations and hops per query.   python scripts/probe/filtered_probe.py [vectors] [ef] [threads,threads,...] [pre]"""
import ctypes as C, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
q = np.ascontiguousarray(make_data(1000, dim, "lowrank", 4321, dev, 24).cpu().numpy())
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
del base


class Res(C.Structure):
    _fields_ = [("seconds", C.c_double), ("queries", C.c_uint64), ("qps", C.c_double), ("latency_min_ns", C.c_int64), ("latency_max_ns", C.c_int64)] + \
               [(f"p{p:02d}_ns", C.c_int64) for p in (1, 10, 25, 50, 75, 90, 99)] + [("recall_avg", C.c_double), ("errors", C.c_uint64), ("launches", C.c_uint64), ("team_launches", C.c_uint64)]


L = C.CDLL(os.path.join("vector_store_amd", "libvs_callers.so"))
L.vs_callers_run_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint, C.c_double, C.POINTER(Res), C.POINTER(C.c_uint64)]
if len(sys.argv) > 4 and sys.argv[4] == "pre":  # as bench.py does: the unfiltered legs first (the dispatcher's streams exist before the callers' own)
    L.vs_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_double, C.POINTER(Res)]
    truth = np.zeros((q.shape[0], k), dtype=np.uint64)
    for threads, inflight in ((17, 1), (16, 256)):
        r = Res()
        L.vs_callers_run(ix.h, q.ctypes.data, q.shape[0], dim, k, truth.ctypes.data, threads, inflight, 1.0, C.byref(r))
        print(f"pre: {threads} x {inflight}: {r.qps:.0f} QPS", flush=True)
for threads in ([int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else (1, 17)):
    for mod in (2, 10, 100):
        r, extra = Res(), (C.c_uint64 * 4)()
        ix.stats(reset=True)
        f0 = ix.filter_stats()
        rc = L.vs_callers_run_filtered(ix.h, q.ctypes.data, q.shape[0], dim, k, mod, threads, 2.0, C.byref(r), extra)
        f1, st = ix.filter_stats(), ix.stats(reset=True)
        nq = max(int(r.queries), 1)
        print(f"n {n} ef {ef} threads {threads} selectivity 1/{mod}: {r.qps:.1f} QPS, latency min {r.latency_min_ns/1e6:.1f} ms max {r.latency_max_ns/1e6:.1f} ms, "
              f"predicate calls/query {extra[0]/nq:.0f}, walk launches/query {(f1['lazy_rounds']-f0['lazy_rounds'])/nq:.1f}, "
              f"evals/query {st['search_evals']/nq:.0f}, hops/query {st['search_hops']/nq:.0f} (walks counted: {st['queries']/nq:.1f}), errors {r.errors} rc {rc}", flush=True)
