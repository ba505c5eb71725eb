"""The pipelined walk (kernels_pipe.hip) against the team form of the usearch-order walk on the same graph: ids and distance
bits must be identical on tie-free data; then filtered-search throughput through the C ABI both ways.
python scripts/probe/pipe_probe.py [vectors] [ef] [threads,threads] [seconds]"""
import ctypes as C, os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs
from bench import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
threads_list = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 17]
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
dim, k = 768, 10
dev = torch.device("cuda:0")
base = make_data(n, dim, "lowrank", 1234, dev, 24)
q = np.ascontiguousarray(make_data(1000, dim, "lowrank", 4321, dev, 24).cpu().numpy())
ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef)
ix.reserve(n)
t0 = time.time()
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
print(f"built {n} in {time.time() - t0:.1f} s", flush=True)
del base
big = n > 2_000_000 or os.environ.get('PIPE_PROBE_FAST') == '1'  # (no second copy of a large index: run again with VS_HNSW_PIPE=0 for the other side)
old = None
if not big:
    old = vs.HipUsearchIndex(dim, vs.COS, expansion_search=ef, _stress=256)  # never the pipelined walk
    old.import_graph(ix.export_graph())

# 1. parity: same graph, same queries, same predicate
bad = 0
for mod in (() if big else (2, 10, 100)):
    pred = lambda key, m=mod: key % m == 0
    for i in range(12):
        a = ix.filtered_search(q[i], k, pred)
        b = old.filtered_search(q[i], k, pred)
        same = a[0].tolist() == b[0].tolist() and a[1].view(np.uint32).tolist() == b[1].view(np.uint32).tolist()
        if not same:
            bad += 1
            print("MISMATCH", mod, i, a[0][:10], b[0][:10], a[1][:4], b[1][:4], flush=True)
if not big:
    print(f"parity: {bad} mismatching queries of 36; filter stats new {ix.filter_stats()} old {old.filter_stats()} pipe {ix.pipe_stats()}", flush=True)


class Res(C.Structure):
    _fields_ = [("seconds", C.c_double), ("queries", C.c_uint64), ("qps", C.c_double), ("latency_min_ns", C.c_int64), ("latency_max_ns", C.c_int64)] + \
               [(f"p{p:02d}_ns", C.c_int64) for p in (1, 10, 25, 50, 75, 90, 99)] + [("recall_avg", C.c_double), ("errors", C.c_uint64), ("launches", C.c_uint64), ("team_launches", C.c_uint64)]


L = C.CDLL(os.path.join("vector_store_amd", "libvs_callers.so"))
L.vs_callers_run_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint, C.c_double, C.POINTER(Res), C.POINTER(C.c_uint64)]
for name, index in ((("pipe" if os.environ.get("VS_HNSW_PIPE", "1") != "0" else "team"), ix),) + (() if big else (("team", old),)):
    for threads in threads_list:
        for mod in [int(m) for m in os.environ.get('PIPE_PROBE_MODS', '2,10,100').split(',')]:
            r, extra = Res(), (C.c_uint64 * 4)()
            L.vs_callers_run_filtered(index.h, q.ctypes.data, q.shape[0], dim, k, mod, threads, 0.3, C.byref(Res()), (C.c_uint64 * 4)())
            index.stats(reset=True)
            f0 = index.filter_stats()
            rc = L.vs_callers_run_filtered(index.h, q.ctypes.data, q.shape[0], dim, k, mod, threads, seconds, C.byref(r), extra)
            f1, st = index.filter_stats(), index.stats(reset=True)
            ps = index.pod_stats() if hasattr(index, "pod_stats") else {}
            nq = max(int(r.queries), 1)
            print(f"{name}: n {n} ef {ef} threads {threads} 1/{mod}: {r.qps:.1f} QPS, min {r.latency_min_ns/1e6:.2f} p50 {r.p50_ns/1e6:.2f} p99 {r.p99_ns/1e6:.2f} ms, "
                  f"pred calls/q {extra[0]/nq:.0f}, launches/q {(f1['lazy_rounds']-f0['lazy_rounds'])/nq:.1f}, "
                  f"evals/q {st['search_evals']/nq:.0f}, hops/q {st['search_hops']/nq:.0f}, errors {r.errors} rc {rc}; p90 {r.p90_ns/1e6:.1f} max {r.latency_max_ns/1e6:.0f} ms; "
                  f"so far: answered {ps.get('filtered_answered')} handed over {ps.get('filtered_handed_over')} rounds without a pod {ps.get('rounds_without_a_pod')} walked again {ps.get('rounds_walked_again')}", flush=True)
