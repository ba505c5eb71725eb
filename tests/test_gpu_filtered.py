"""filtered_search with the reference's semantics: usearch tests the predicate when
a node would enter `top` and keeps expanding rejected nodes (reference vs_index/usearch.rs:224-248, wrapper
filtered_ann :1107-1154; 19 integration tests tests/integration/vs_index.rs:718-1640).  The engine evaluates the host
predicate into an allow-bitmap over slots and tests it in the walk at admission, so on the same graph the result set
and its order equal the CPU restatement's -- compared here at 100,000 members for 50 % / 10 % / 1 % / 0.1 % selectivity."""
import numpy as np
import pytest

import oracle
from oracle import OracleIndex
from tests.parity_util import assert_same_results, lattice

pytestmark = pytest.mark.gpu


def _dataset(n, dim, seed):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


@pytest.mark.parametrize("metric,kind,dim", [("cos", "f32", 96), ("l2sq", "f32", 32), ("ip", "f16", 128), ("l2sq", "i8", 64),
                                             ("hamming", "b1", 128)])
def test_filtered_search_equals_the_cpu_algorithm_at_100k(metric, kind, dim):
    import vector_store_amd as vs
    n, nq = 100000, 8
    data = _dataset(n + nq, dim, 41)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind])
    ix.reserve(n)
    keys = np.arange(n, dtype=np.uint64) * 7 + 3
    ix.add_batch(keys, base)
    o = OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[kind])
    o.import_graph(ix.export_graph())
    exact = kind in ("i8", "b1")
    dist_of = lambda i: (lambda key: oracle.distance_as(oracle.METRICS[metric], oracle.SCALARS[kind], q[i], base[(key - 3) // 7]))
    ties = 0
    for modulo, k, ef in ((2, 10, 64), (10, 10, 64), (100, 10, 64), (100, 50, 128), (1000, 20, 64), (10, 600, 64)):
        pred = lambda key: ((key - 3) // 7) % modulo == 1
        ix.set_expansion_search(ef)
        o.set_expansion_search(ef)
        for i in range(nq):
            fk, fd = ix.filtered_search(q[i], k, pred)
            ek, ed = o.filtered_search(q[i], k, pred)
            assert all(pred(int(x)) for x in fk)
            assert len(fk) == len(ek) == k, (modulo, k, len(fk), len(ek))
            ties += assert_same_results(fk, fd, ek, ed, dist_of(i), exact=exact, what=(metric, kind, modulo, k, ef, i))
    assert ties <= 8, ties
    # 100,000 members > 65,536: the predicate was asked lazily -- a few rounds per query, far fewer calls than one per member
    fs = ix.filter_stats()
    queries = 6 * nq
    assert fs["lazy_rounds"] >= queries and fs["lazy_rounds"] <= 8 * queries, fs
    assert fs["lazy_predicate_calls"] < 0.5 * n * queries, fs
    # a predicate nothing passes, one a single member passes, and one everything passes (== plain search)
    fk, fd = ix.filtered_search(q[0], 10, lambda key: False)
    assert len(fk) == 0
    fk, fd = ix.filtered_search(q[0], 10, lambda key: key == int(keys[777]))
    assert fk.tolist() == [int(keys[777])]
    fk, fd = ix.filtered_search(q[0], 10, lambda key: True)
    pk, pd = ix.search(q[0], 10)
    assert fk.tolist() == pk.tolist() or not exact  # float indexes answer plain searches from the fused list


def test_filtered_search_with_removed_members_and_exact_data():
    """Lattice data (every distance exact in f32): bit-identical ids with removed members in the graph."""
    import vector_store_amd as vs
    n, dim = 20000, 24
    data = lattice(n + 16, dim, 77, span=200)
    base, q = data[:n], data[n:]
    ix = vs.HipUsearchIndex(dim, vs.L2SQ)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    for key in range(0, n, 5):
        assert ix.remove(key)
    o = OracleIndex(dim, oracle.L2SQ)
    o.import_graph(ix.export_graph())
    for modulo, k in ((3, 10), (50, 25)):
        pred = lambda key: key % modulo == 1
        for i in range(16):
            fk, fd = ix.filtered_search(q[i], k, pred)
            ek, ed = o.filtered_search(q[i], k, pred)
            assert_same_results(fk, fd, ek, ed, exact=True, what=(modulo, k, i))
            assert all(int(x) % 5 != 0 for x in fk)


def test_a_named_filter_remembers_verdicts_and_answers_like_the_unnamed_one():
    """vs_hnsw_filtered_search_keyed (round 5): queries that carry the same filter_key share the verdicts the predicate gave (two bits
    per slot on the device).  Answers are those of the plain call -- the oracle's, ids and distances --; the predicate is asked about a
    key at most once per filter; once the neighbourhoods are known a query is one walk without a single predicate call; a member that
    is removed and re-added under the same key is asked about again (its verdict may have changed); another key is another filter."""
    import threading

    import vector_store_amd as v
    n, dim, k = 100_000, 64, 10
    data = _dataset(n + 64, dim, 61)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(n + 64)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.COS, 16, 128, 96)
    o.import_graph(ix.export_graph())
    asked = {}
    flip = set()

    def pred(key):
        asked[key] = asked.get(key, 0) + 1
        return (key % 10 == 3) != (key in flip)

    FK = 0xF117E5
    for rnd in range(3):
        before = sum(asked.values())
        for i in range(len(q)):
            gk, gd = ix.filtered_search(q[i], k, pred, filter_key=FK)
            wk, wd = o.filtered_search(q[i], k, lambda key: (key % 10 == 3) != (key in flip))
            assert_same_results(gk, gd, wk, wd, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=("keyed", rnd, i))
        calls = sum(asked.values()) - before
        if rnd == 0:
            assert calls > 1000 * 8, calls          # the first pass asks
        else:
            assert calls == 0, (rnd, calls)         # the same queries again: every verdict is remembered
    assert max(asked.values()) == 1                 # nothing was asked twice
    st = ix.filter_memo_stats()
    assert st["queries"] == 3 * len(q) and st["memories_held"] == 1
    # the unnamed call still asks (and answers the same)
    before = sum(asked.values())
    gk, gd = ix.filtered_search(q[0], k, pred)
    assert sum(asked.values()) > before
    wk, wd = o.filtered_search(q[0], k, lambda key: key % 10 == 3)
    assert_same_results(gk, gd, wk, wd, lambda key: o.distance_to_slot(q[0], int(key)), what="unnamed")
    # a member that changes is asked about again: remove the best match of q[0], re-add it with a verdict that flipped
    best = int(ix.filtered_search(q[0], 1, pred, filter_key=FK)[0][0])
    asked.clear()
    assert ix.remove(best)
    ix.add(best, base[best])
    flip.add(best)                                   # the row's filterable column changed with it: now rejected
    gk, gd = ix.filtered_search(q[0], k, pred, filter_key=FK)
    assert best not in gk.tolist() and asked.get(best, 0) == 1, (best, gk, asked.get(best))
    # another name, another memory: the same key is asked about again
    asked.clear()
    ix.filtered_search(q[1], k, pred, filter_key=FK + 1)
    assert sum(asked.values()) > 100 and ix.filter_memo_stats()["memories_held"] == 2
    # concurrent callers of one filter: same answers
    want = [ix.filtered_search(q[i], k, lambda key: key % 7 == 1)[0].tolist() for i in range(16)]
    errors = []

    def caller(t):
        try:
            for i in range(t, 16, 4):
                assert ix.filtered_search(q[i], k, lambda key: key % 7 == 1, filter_key=77)[0].tolist() == want[i]
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=caller, args=(t,)) for t in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors[:2]
