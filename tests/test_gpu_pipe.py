"""The pipelined walk (vector_store_amd/csrc/kernels_pipe.hip): lone queries -- one vector per call, as the reference issues
them (vs_index/usearch.rs:212 search, :236 filtered_search, every filtered query on its own blocking thread :937-948) -- are
walked by one wave on register-resident structures while the other waves of the workgroup measure candidates ahead of it.
Tie-free, its decisions are usearch's; where two equal distances could make the order matter it hands the query to the
usearch-order walk.  Checked here: bit-identical answers (ids AND distance bits) to the team kernels on the same graph, for
filtered and plain lone queries, storage types, row layouts, beams up to 512 and removed members; the hand-over on data made
of ties; and the oracle on the same graph."""
import numpy as np
import pytest

import oracle
from oracle import OracleIndex
from tests.parity_util import assert_same_results, lattice

pytestmark = pytest.mark.gpu
NO_PIPE = 256  # vs_hnsw_options.reserved bit 8


def _dataset(n, dim, seed, rank=16):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((rank, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, rank)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


def _pair(vs, dim, metric, kind, base, keys, ef, remove_every=0):
    """The same graph behind two handles: `new` serves lone queries with the pipelined walk, `old` never does."""
    new = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind], expansion_search=ef)
    new.reserve(len(base))
    new.add_batch(keys, base)
    if remove_every:
        for key in keys[::remove_every]:
            assert new.remove(int(key))
    old = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind], expansion_search=ef, _stress=NO_PIPE)
    old.import_graph(new.export_graph())
    return new, old


def _same_bits(a, b, what):
    assert a[0].tolist() == b[0].tolist(), (what, a[0][:10], b[0][:10])
    assert a[1].view(np.uint32).tolist() == b[1].view(np.uint32).tolist(), (what, a[1][:6], b[1][:6])


@pytest.mark.parametrize("metric,kind,dim,ef", [("cos", "f32", 768, 200), ("l2sq", "f32", 96, 64), ("ip", "f16", 256, 128),
                                                ("cos", "bf16", 128, 300), ("l2sq", "f32", 1536, 100), ("cos", "f32", 24, 64),
                                                ("l2sq", "f32", 2048, 64)])
def test_filtered_lone_queries_equal_the_team_walk_bit_for_bit(metric, kind, dim, ef):
    import vector_store_amd as vs
    n, nq, k = 120_000, 6, 10
    data = _dataset(n + nq, dim, 5)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    keys = np.arange(n, dtype=np.uint64) * 3 + 1
    new, old = _pair(vs, dim, metric, kind, base, keys, ef, remove_every=17)
    for modulo in (2, 10, 100):
        pred = lambda key, m=modulo: ((key - 1) // 3) % m == 0
        for i in range(nq):
            a, b = new.filtered_search(q[i], k, pred), old.filtered_search(q[i], k, pred)
            assert len(a[0]) == k and all(pred(int(x)) for x in a[0])
            _same_bits(a, b, (metric, kind, dim, ef, modulo, i))
    # round 6: an unnamed filter is ONE walk that asks while it runs (posted to a pod); a walk that hands over (an order-relevant tie) or
    # finds no pod goes through the rounds of rounds 3-5: an exploring and an exact one at least
    st = new.filter_ask_stats()
    assert st["queries"] + st["handed_over"] + st["no_pod"] == 3 * nq
    assert new.pipe_stats()["pipe_launches"] >= st["queries"] + 2 * (st["handed_over"] + st["no_pod"])
    assert old.pipe_stats()["pipe_launches"] == 0
    # the oracle on the same graph (the parity bar of tests/parity_util.py)
    o = OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[kind])
    o.import_graph(new.export_graph())
    o.set_expansion_search(ef)
    pred = lambda key: ((key - 1) // 3) % 10 == 0
    ties = 0
    for i in range(nq):
        fk, fd = new.filtered_search(q[i], k, pred)
        ek, ed = o.filtered_search(q[i], k, pred)
        ties += assert_same_results(fk, fd, ek, ed, lambda key, i=i: oracle.distance_as(oracle.METRICS[metric], oracle.SCALARS[kind], q[i], base[(key - 1) // 3]),
                                    what=(metric, kind, i))
    assert ties <= 3, ties


@pytest.mark.parametrize("metric,kind,dim,ef,k", [("cos", "f32", 768, 200, 10), ("l2sq", "f32", 64, 128, 10), ("cos", "f16", 384, 64, 5),
                                                  ("ip", "bf16", 128, 400, 100)])
def test_plain_lone_queries_equal_the_team_kernels_bit_for_bit(metric, kind, dim, ef, k):
    """vs_hnsw_search (one vector per call, through the dispatcher): pipelined walk against the team form of the fused-list kernel."""
    import vector_store_amd as vs
    n, nq = 150_000, 24
    data = _dataset(n + nq, dim, 9)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    keys = np.arange(n, dtype=np.uint64) + 7
    new, old = _pair(vs, dim, metric, kind, base, keys, ef, remove_every=11)
    for i in range(nq):
        _same_bits(new.search(q[i], k), old.search(q[i], k), (metric, kind, dim, ef, i))
    assert new.pipe_stats()["pipe_launches"] >= nq
    assert old.pipe_stats()["pipe_launches"] == 0
    # and the batch path (one wave per query) answers the same
    bk, bd, bf = new.search_batch(q[:8], k)
    for i in range(8):
        a = new.search(q[i], k)
        assert a[0].tolist() == bk[i][: bf[i]].tolist()
    # counters of a walk: evaluations and hops as the team kernels count them
    new.stats(reset=True), old.stats(reset=True)
    for i in range(8):
        new.search(q[i], k), old.search(q[i], k)
    sn, so = new.stats(), old.stats()
    # (the fused list counts a removed member's expansion differently from usearch's two structures: close, not equal)
    assert abs(sn["search_hops"] - so["search_hops"]) <= 0.15 * so["search_hops"], (sn, so)
    assert abs(sn["search_evals"] - so["search_evals"]) <= 0.15 * so["search_evals"], (sn, so)


def test_data_made_of_ties_is_handed_to_the_usearch_order_walk():
    """Lattice data: every distance is an integer, equal distances are the rule.  Where the order among them matters the
    pipelined walk gives the query up; either way the answer is the oracle's, bit for bit."""
    import vector_store_amd as vs
    n, dim = 90_000, 16
    data = lattice(n + 12, dim, 3, span=40)
    base, q = data[:n], data[n:]
    ix = vs.HipUsearchIndex(dim, vs.L2SQ, expansion_search=64)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.L2SQ)
    o.import_graph(ix.export_graph())
    o.set_expansion_search(64)
    for modulo in (3, 20):
        pred = lambda key, m=modulo: key % m == 1
        for i in range(12):
            fk, fd = ix.filtered_search(q[i], 10, pred)
            ek, ed = o.filtered_search(q[i], 10, pred)
            assert_same_results(fk, fd, ek, ed, exact=True, what=(modulo, i))
    assert ix.pipe_stats()["pipe_launches"] > 0


def test_heavily_duplicated_vectors_and_tiny_indexes():
    import vector_store_amd as vs
    rng = np.random.default_rng(2)
    dim = 48
    uniq = rng.standard_normal((400, dim)).astype(np.float32)
    base = np.repeat(uniq, 200, axis=0)  # 80,000 rows, every vector 200 times: every distance ties 200-fold
    n = len(base)
    new, old = _pair(vs, dim, "cos", "f32", base, np.arange(n, dtype=np.uint64), 64)
    q = uniq[:6] + 0.01 * rng.standard_normal((6, dim)).astype(np.float32)
    pred = lambda key: key % 7 == 0
    for i in range(6):
        a, b = new.filtered_search(q[i], 10, pred), old.filtered_search(q[i], 10, pred)
        assert len(a[0]) == len(b[0]) == 10 and all(pred(int(x)) for x in a[0])
        assert np.allclose(a[1], b[1], atol=1e-6), (i, a[1], b[1])  # (which of 200 copies is a matter of tie order; the distances are not)
        pa, pb = new.search(q[i], 10), old.search(q[i], 10)
        assert np.allclose(pa[1], pb[1], atol=1e-6)
    # fewer members than the beam; an empty index
    tiny = vs.HipUsearchIndex(dim, vs.COS, expansion_search=64)
    tiny.reserve(100)
    assert len(tiny.search(q[0], 5)[0]) == 0
    tiny.add_batch(np.arange(20, dtype=np.uint64), uniq[:20])
    fk, fd = tiny.search(q[0], 50)
    assert len(fk) == 20 and fk[0] == 0


def test_a_crowd_of_filtered_callers_shares_launches_and_gets_the_same_answers():
    """More filtered callers than the device has streams (the reference puts every filtered query on a blocking thread of its own,
    usearch.rs:937-948): their rounds are batched into shared launches.  Same answers, bit for bit, as one caller at a time."""
    import threading
    import vector_store_amd as vs
    n, dim, k, nq, callers = 200_000, 96, 10, 48, 40
    data = _dataset(n + nq, dim, 31)
    base, q = data[:n], data[n:]
    ix = vs.HipUsearchIndex(dim, vs.COS, expansion_search=128)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    pred = lambda key: key % 10 == 3
    want = [ix.filtered_search(q[i], k, pred) for i in range(nq)]   # one at a time (and the index learns the filter's appetite)
    assert ix.filter_batch_stats()["batched_rounds"] == 0
    got = [None] * (callers * 3)
    errs = []

    def caller(t):
        try:
            for r in range(3):
                i = (t * 3 + r) % nq
                got[t * 3 + r] = (i, ix.filtered_search(q[i], k, pred))
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=caller, args=(t,)) for t in range(callers)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for i, a in got:
        _same_bits(a, want[i], ("crowd", i))
    # their rounds were posted to resident workgroups (csrc/pipe_pod.hpp) or, when no pod could take one, shared launches
    st, pods = ix.filter_batch_stats(), ix.pod_stats()
    assert pods["pod_rounds"] + st["batched_rounds"] > 0 and st["batched_launches"] <= st["batched_rounds"], (st, pods)
    assert pods["pods_opened"] >= 1 or not pods["pods_enabled"], pods


@pytest.mark.parametrize("dim,span,ef,k,modulo", [(12, 6, 64, 10, 3), (8, 4, 128, 20, 10), (16, 10, 200, 10, 7), (6, 3, 48, 48, 2)])
def test_ties_everywhere_and_the_pipelined_walk_still_answers_most_rounds_itself(dim, span, ef, k, modulo):
    """Coarse lattices: a handful of distinct distances, so two equal ones wait in `next` at EVERY hop, with `top` full for most of the
    walk.  The window rule (pipe_device.hpp) lets the pipelined walk go on while the two orders cannot differ -- the radius stays at or
    above the window's distance, nothing meets the radius or an equal distance in `top` -- and hands over otherwise; either way ids,
    their order and the distance bits are the oracle's."""
    import vector_store_amd as vs
    n, nq = 80_000, 10
    data = lattice(n + nq, dim, 100 + dim, span=span)
    base, q = data[:n], data[n:] + 0.5  # (queries off the lattice: fewer exact ties with the radius, ties among candidates stay)
    ix = vs.HipUsearchIndex(dim, vs.L2SQ, expansion_search=ef)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.L2SQ)
    o.import_graph(ix.export_graph())
    o.set_expansion_search(ef)
    pred = lambda key: key % modulo == 1
    for rep in range(2):  # (the second pass runs with the index's filter history: exploring rounds that guess)
        for i in range(nq):
            fk, fd = ix.filtered_search(q[i], k, pred)
            ek, ed = o.filtered_search(q[i], k, pred)
            assert_same_results(fk, fd, ek, ed, exact=True, what=(dim, span, ef, modulo, rep, i))
    st = ix.pod_stats()
    assert ix.pipe_stats()["pipe_launches"] > 0
    print("rounds posted", st["pod_rounds"], "walked again", st["rounds_walked_again"], "handed over", st["filtered_handed_over"])
