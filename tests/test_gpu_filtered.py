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
