"""Regression tests of the round-2 advisor findings: a walk that reports "outgrew its workspace" inside the
single-query dispatcher is answered by exhaustive ranking (never by a memcpy of 2^32 entries); the walk instance is chosen
by the slot domain of ITS OWN visited table; the inner-product certificate's norm bound follows rows that re-use a
removed slot."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

USEARCH_ORDER, GLOBAL_WALK, TINY_HEAP = 16, 32, 128
WALK_LDS_256, WALK_LDS_512, WALK_GLOBAL_512, WALK_LDS_320, WALK_LDS_256_DENSE = 1, 2, 3, 8, 9


def vs():
    import vector_store_amd as v
    return v


def _data(n, dim, seed):
    rng = np.random.default_rng(seed)
    r = min(16, dim)
    w = rng.standard_normal((r, dim)).astype(np.float32) / np.sqrt(r)
    return (rng.standard_normal((n, r)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


@pytest.mark.timeout(600)
def test_dispatcher_ranks_exhaustively_when_a_walk_outgrows_its_workspace():
    """Mass removes leave fewer live members than the beam: `top` never fills, the walk floods the whole graph, and with
    the 64-entry heap hook `next` overflows -> kWalkFailed.  vs_hnsw_search and vs_hnsw_search_async must then return the
    exact answer (the header's promise), through the dispatcher thread."""
    v = vs()
    n, dim, k = 20000, 16, 10
    base = _data(n, dim, 11)
    ix = v.HipUsearchIndex(dim, v.L2SQ, expansion_search=64, _stress=USEARCH_ORDER | GLOBAL_WALK | TINY_HEAP)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    keep = [17, 4321, 9999, 15000, 19999]
    for key in range(n):
        if key not in keep:
            assert ix.remove(key)
    q = _data(8, dim, 12)
    before = ix.walk_info()["ranked_fallbacks"]
    for i in range(len(q)):
        want = sorted(keep, key=lambda s: float(((q[i] - base[s]) ** 2).sum()))
        # vs_hnsw_search sees the global-walk route itself (search_host + rank_all) ...
        got_k, got_d = ix.search(q[i], k)
        assert got_k.tolist() == want, (i, got_k, want)
        # ... the non-blocking entry point goes through the dispatcher: that is where the sentinel used to be memcpy'd
        done = threading.Event()
        res = {}

        def on_done(keys, dist, status, res=res, done=done):
            res["keys"], res["dist"], res["status"] = keys.copy(), dist.copy(), status
            done.set()
        holder = ix.search_async(q[i], k, on_done)
        assert done.wait(120)
        del holder
        assert res["status"] == 0 and res["keys"].tolist() == want, (i, res)
        assert np.allclose(res["dist"], [float(((q[i] - base[s]) ** 2).sum()) for s in want], rtol=1e-5)
    assert ix.walk_info()["ranked_fallbacks"] >= before + len(q)


@pytest.mark.timeout(600)
def test_walk_instance_is_chosen_by_its_own_visited_domain(monkeypatch):
    """ef 288 takes the 320-entry instance, whose visited table is the 256 instance's (25 slot bits), not the 512
    instance's (26): an index of 2^25 < slots <= 2^26 must move to the 512 instance, beyond 2^26 to the global bitmap.
    VS_HNSW_WALK_DOMAIN_SLOTS makes a small index pretend; all three routes return the same ids."""
    v = vs()
    n, dim, k, ef = 30000, 64, 10, 288
    base = _data(n, dim, 21)
    q = _data(200, dim, 22)
    results, instances = [], []
    graph = None
    for pretend in (0, (1 << 25) + 1, (1 << 26) + 1):
        if pretend:
            monkeypatch.setenv("VS_HNSW_WALK_DOMAIN_SLOTS", str(pretend))
        else:
            monkeypatch.delenv("VS_HNSW_WALK_DOMAIN_SLOTS", raising=False)
        ix = v.HipUsearchIndex(dim, v.COS, expansion_search=ef, quantization=v.I8)
        if graph is None:
            ix.reserve(n)
            ix.add_batch(np.arange(n, dtype=np.uint64), base)
            graph = ix.export_graph()
        else:
            ix.import_graph(graph)
        keys, dist, found = ix.search_batch(q, k)
        results.append((keys, dist, found))
        instances.append(ix.walk_info()["last_instance"])
    assert instances == [WALK_LDS_320, WALK_LDS_512, WALK_GLOBAL_512], instances
    for keys, dist, found in results[1:]:
        assert np.array_equal(keys, results[0][0]) and np.array_equal(dist.view(np.uint32), results[0][1].view(np.uint32))


@pytest.mark.timeout(600)
def test_dense_visited_table_instance_returns_what_the_256_instance_returns(monkeypatch):
    """Beams of 129..256 below 2^24 slots take the dense instance (512 buckets x 12 tags, 600 entries of `next`: 8 walks per
    CU); beyond 2^24 slots its 24-bit domain does not tell slots apart and the 256 instance runs.  Same ids, same distances,
    bit for bit, and both equal the CPU restatement's walk."""
    v = vs()
    from oracle import OracleIndex
    n, dim, k, ef = 40000, 96, 10, 200
    base = _data(n, dim, 31)
    q = _data(600, dim, 32)
    results, instances = [], []
    graph = None
    for pretend in (0, (1 << 24) + 1):
        if pretend:
            monkeypatch.setenv("VS_HNSW_WALK_DOMAIN_SLOTS", str(pretend))
        else:
            monkeypatch.delenv("VS_HNSW_WALK_DOMAIN_SLOTS", raising=False)
        ix = v.HipUsearchIndex(dim, v.L2SQ, expansion_search=ef, quantization=v.I8)
        if graph is None:
            ix.reserve(n)
            ix.add_batch(np.arange(n, dtype=np.uint64), base)
            graph = ix.export_graph()
        else:
            ix.import_graph(graph)
        results.append(ix.search_batch(q, k))
        instances.append(ix.walk_info()["last_instance"])
    assert instances == [WALK_LDS_256_DENSE, WALK_LDS_256], instances
    assert np.array_equal(results[0][0], results[1][0])
    assert np.array_equal(results[0][1].view(np.uint32), results[1][1].view(np.uint32))
    o = OracleIndex(dim, v.L2SQ, expansion_search=ef, quantization=v.I8)
    o.import_graph(graph)
    ok, od, _ = o.search_batch(q[:200], k, threads=8)
    assert np.array_equal(ok, results[0][0][:200]) and np.array_equal(od, results[0][1][:200])


@pytest.mark.timeout(600)
def test_ip_certificate_follows_a_reused_slots_norm():
    """Inner product, >= 65,536 rows: the split-bf16 nomination certifies with eps ~ |q| * max|row|.  A vector 100x longer
    than anything before it is added into a REUSED slot (below the high-water mark the norm pass had covered): the exact
    search must still equal a float64 brute force."""
    v = vs()
    n, dim, k = 70000, 96, 10
    rng = np.random.default_rng(31)
    base = _data(n, dim, 31)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    ix = v.HipUsearchIndex(dim, v.IP)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    q = _data(64, dim, 32)
    ix.exact_search_batch(q, k)  # norm state now covers every slot
    assert ix.remove(12345)
    big = (100.0 * base[777] + 0.3 * rng.standard_normal(dim)).astype(np.float32)
    ix.add(1 << 40, big)
    keys, dist, found = ix.exact_search_batch(q, k)
    rows = base.copy().astype(np.float64)
    rows[12345] = big
    ids = np.arange(n, dtype=np.uint64)
    ids[12345] = 1 << 40
    d = 1.0 - q.astype(np.float64) @ rows.T
    for i in range(len(q)):
        order = np.argsort(d[i], kind="stable")[:k]
        assert keys[i].tolist() == ids[order].tolist(), (i, keys[i], ids[order])
        assert np.allclose(dist[i], d[i][order], rtol=1e-4, atol=1e-4)


@pytest.mark.timeout(900)
def test_crowds_of_callers_get_the_batch_paths_answers():
    """The reference's call patterns through the C ABI, with more callers than host cores: blocking single-query callers (the
    dispatcher's eight pipeline slots), the non-blocking entry point under load (two slots, large batches), and a crowd of
    blocking FILTERED callers (each its own rounds of walks; beyond the core count they sleep on the round's event).  Every
    answer equals the batch path's, no call fails, every filtered call returns k admitted members."""
    import ctypes as C
    v = vs()
    n, dim, k = 200000, 64, 10
    base = _data(n, dim, 51)
    q = np.ascontiguousarray(_data(2000, dim, 52))
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    truth, _, _ = ix.search_batch(q, k)
    truth = np.ascontiguousarray(truth, dtype=np.uint64)

    class Res(C.Structure):
        _fields_ = [("seconds", C.c_double), ("queries", C.c_uint64), ("qps", C.c_double), ("latency_min_ns", C.c_int64),
                    ("latency_max_ns", C.c_int64)] + [(f"p{p:02d}_ns", C.c_int64) for p in (1, 10, 25, 50, 75, 90, 99)] + [
                    ("recall_avg", C.c_double), ("errors", C.c_uint64), ("launches", C.c_uint64), ("team_launches", C.c_uint64)]

    L = C.CDLL(os.path.join(os.path.dirname(v.__file__), "libvs_callers.so"))
    L.vs_callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint,
                                 C.c_double, C.POINTER(Res)]
    L.vs_callers_run_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint, C.c_double,
                                          C.POINTER(Res), C.POINTER(C.c_uint64)]
    for threads, inflight in ((3, 1), (40, 1), (16, 128)):
        r = Res()
        rc = L.vs_callers_run(ix.h, q.ctypes.data, q.shape[0], dim, k, truth.ctypes.data, threads, inflight, 1.0, C.byref(r))
        assert rc == 0 and r.errors == 0 and r.queries > 0, (threads, inflight, rc, r.errors)
        assert r.recall_avg > 0.9999, (threads, inflight, r.recall_avg)  # agreement with the batch path's ids
    for threads, modulus in ((5, 10), (48, 10), (48, 50)):
        r, extra = Res(), (C.c_uint64 * 4)()
        rc = L.vs_callers_run_filtered(ix.h, q.ctypes.data, q.shape[0], dim, k, modulus, threads, 1.5, C.byref(r), extra)
        assert rc == 0 and r.errors == 0 and r.queries > 0, (threads, modulus, rc, r.errors)
        assert extra[1] == r.queries * k, (threads, modulus, extra[1], r.queries)
    # the same filtered answers whether one caller asks or a crowd does (24 threads: beyond the core count of the pool's boxes)
    pred = lambda key: key % 7 == 3
    want = [ix.filtered_search(q[i], k, pred) for i in range(24)]
    got = [None] * 24

    def ask(i):
        got[i] = ix.filtered_search(q[i], k, pred)
    th = [threading.Thread(target=ask, args=(i,)) for i in range(24)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(24):
        assert got[i] is not None and got[i][0].tolist() == want[i][0].tolist(), i
        assert np.array_equal(got[i][1].view(np.uint32), want[i][1].view(np.uint32)), i
