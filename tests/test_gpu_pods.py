"""Pods (vector_store_amd/csrc/pipe_pod.hpp): blocking callers -- the reference issues one query per FFI call (vs_index/usearch.rs:212,
:236) and gives every filtered query a blocking thread of its own (:937-948) -- post their query to a workgroup of a RESIDENT launch of
the pipelined walk instead of launching one.  A pod carries the index's view as its kernel arguments, so whatever changes the view closes
the index's pods first; an idle pod closes by itself.  Checked here: answers bit-identical to the team kernels' on the same graph while
adds and removes come between the searches, crowds of plain callers, more indexes than the device has pods, and that idle pods go."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NO_PIPE = 256  # vs_hnsw_options.reserved bit 8: never the pipelined walk (hence never a pod)


def _dataset(n, dim, seed, rank=16):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((rank, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, rank)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


def _same_bits(a, b, what):
    assert a[0].tolist() == b[0].tolist(), (what, a[0][:10], b[0][:10])
    assert a[1].view(np.uint32).tolist() == b[1].view(np.uint32).tolist(), (what, a[1][:6], b[1][:6])


def test_adds_and_removes_between_searches_close_the_pods_and_the_answers_follow_the_graph():
    import vector_store_amd as vs
    n, dim, k, step = 120_000, 96, 10, 20_000
    data = _dataset(n + 8, dim, 41)
    base, q = data[:n], data[n:]
    keys = np.arange(n, dtype=np.uint64) + 5
    ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=128)
    ix.reserve(n)
    if not ix.pod_stats()["pods_enabled"]:
        pytest.skip("VS_HNSW_PODS=0")
    pred = lambda key: key % 7 == 1
    opened = 0
    for lo in range(0, n, step):
        ix.add_batch(keys[lo:lo + step], base[lo:lo + step])
        if lo >= 2 * step:
            assert ix.remove(int(keys[lo - step + 3]))  # (the first remove turns the removed-members flag on: part of the view)
        if lo + step < 70_000:
            continue  # (filtered search below 65,536 slots asks the predicate about every member)
        old = vs.HipUsearchIndex(dim, vs.COS, expansion_search=128, _stress=NO_PIPE)
        old.import_graph(ix.export_graph())
        for i in range(len(q)):
            _same_bits(ix.search(q[i], k), old.search(q[i], k), ("plain", lo, i))
            _same_bits(ix.filtered_search(q[i], k, pred), old.filtered_search(q[i], k, pred), ("filtered", lo, i))
        st = ix.pod_stats()
        assert st["pods_opened"] > opened, st  # every step's searches ran in pods opened after the step's adds
        opened = st["pods_opened"]
        assert old.pod_stats()["pods_opened"] == 0
    assert ix.pod_stats()["pod_rounds"] >= 3 * len(q) * 2  # (a plain query and at least one posted round per filtered one: the very first has no history and takes rounds of its own)


def test_a_crowd_of_plain_callers_gets_the_batch_paths_answers():
    import vector_store_amd as vs
    n, dim, k, nq, callers = 150_000, 64, 10, 256, 48
    data = _dataset(n + nq, dim, 43)
    base, q = data[:n], data[n:]
    ix = vs.HipUsearchIndex(dim, vs.L2SQ, expansion_search=96)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    bk, bd, bf = ix.search_batch(q, k)
    got, errs = {}, []

    def caller(t):
        try:
            for i in range(t, nq, callers):
                got[i] = ix.search(q[i], k)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=caller, args=(t,)) for t in range(callers)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for i in range(nq):
        assert got[i][0].tolist() == bk[i][: bf[i]].tolist(), i
        assert got[i][1].view(np.uint32).tolist() == bd[i][: bf[i]].view(np.uint32).tolist(), i
    st = ix.pod_stats()
    if st["pods_enabled"]:
        assert st["pod_rounds"] >= nq // 2, st  # (Python's threads take turns: most of the queries still find a pod open)


def test_more_indexes_than_pods_and_idle_pods_close():
    import vector_store_amd as vs
    n, dim, k = 80_000, 48, 10
    pred = lambda key: key % 5 == 2
    handles = []
    for j in range(5):
        data = _dataset(n + 4, dim, 50 + j)
        ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=64)
        ix.reserve(n)
        ix.add_batch(np.arange(n, dtype=np.uint64), data[:n])
        old = vs.HipUsearchIndex(dim, vs.COS, expansion_search=64, _stress=NO_PIPE)
        old.import_graph(ix.export_graph())
        handles.append((ix, old, data[n:]))
    errs = []

    def caller(ix, old, q):
        try:
            for r in range(3):
                for i in range(len(q)):
                    _same_bits(ix.search(q[i], k), old.search(q[i], k), ("plain", i))
                    _same_bits(ix.filtered_search(q[i], k, pred), old.filtered_search(q[i], k, pred), ("filtered", i))
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=caller, args=h) for h in handles]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    if not handles[0][0].pod_stats()["pods_enabled"]:
        return
    assert sum(h[0].pod_stats()["pods_opened"] for h in handles) >= 1
    # nobody posts any more: the keeper closes the pods (their workgroups hold CUs while they poll)
    deadline = time.time() + 2.0
    while handles[0][0].pod_stats()["pods_open_on_device"] and time.time() < deadline:
        time.sleep(0.01)
    assert handles[0][0].pod_stats()["pods_open_on_device"] == 0


def test_a_growing_tiny_index_keeps_every_member_reachable():
    """The workspace a caller brings is a visited bitmap with the visited log behind it; when the index grows by a few slots the bitmap
    grows by a word OVER what was the log (slot numbers): the layout, not the byte size, decides when it is zeroed again."""
    import vector_store_amd as vs
    rng = np.random.default_rng(1)
    dim = 20
    ix = vs.HipUsearchIndex(dim, vs.L2SQ, expansion_search=64)
    ix.reserve(700)
    nxt = 0
    for phase in range(24):
        n = int(rng.integers(1, 30))
        keys = np.arange(nxt, nxt + n, dtype=np.uint64)
        vecs = rng.standard_normal((n, dim)).astype(np.float32)
        nxt += n
        if phase % 2:
            ix.add_batch(keys, vecs)
        else:
            for i in range(n):
                ix.add(int(keys[i]), vecs[i])
        if phase % 5 == 4:
            assert ix.remove(nxt - 3)
        k = min(ix.size() + 3, 250)
        for r in range(3):
            q = rng.standard_normal(dim).astype(np.float32)
            gk, gd = ix.search(q, k)
            bk, bd, bf = ix.search_batch(q[None, :], k)
            assert gk.tolist() == bk[0][: bf[0]].tolist(), (phase, r, len(gk), int(bf[0]))
            assert gd.view(np.uint32).tolist() == bd[0][: bf[0]].view(np.uint32).tolist()
