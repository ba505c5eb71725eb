"""Pins the CPU oracle (oracle/cpu_hnsw.cpp) against every known-answer test the reference
holds for the usearch-backed path (tests/golden/kat.json <- SURVEY.md Appendix B)."""
import threading

import numpy as np
import pytest

import oracle
from oracle import METRICS, OracleIndex
from tests import kat_runner as K


def factory(metric, dim, **kw):
    return OracleIndex(dim, METRICS[metric], **kw)


@pytest.mark.parametrize("name", ["B2_l2sq_3d_http", "B3_l2sq_1d_scores", "B4_empty", "B5_cos_winners",
                                  "B6_ip_winner", "B7_l2_winner", "B8_quant_f32", "B10_self_zero_f32"])
def test_simple_kats(name):
    keys, d = K.run_simple(factory, name)
    t = K.KAT[name]
    if "expect_similarity" in t:
        for x, want in zip(d, t["expect_similarity"]):
            assert abs(oracle.similarity(float(x), METRICS[t["metric"]], t["dim"]) - want) <= 1e-5
    for x in d:
        assert oracle.distance_valid(float(x), METRICS[t["metric"]], t["dim"])


def test_b1_add_remove_readd():
    K.run_b1(factory)


def test_b11_filtered_30():
    K.run_b11(factory)


def test_b12_fine_order_exact():
    first, _ = K.run_b12(factory)
    assert first == sorted(first)  # what the reference asserts
    assert first == list(range(100))  # f32 cosine order is exact here


def test_b13_zero_query():
    K.run_b13(factory)


def test_b14_b1_packing():
    for c in K.KAT["B14_b1_packing"]["cases"]:
        got = oracle.f32_to_b1x8(np.asarray(c["input"], dtype=np.float32))
        assert got.tolist() == c["expect"]


def test_b15_distance_ranges():
    t = K.KAT["B15_ranges"]
    for m in ("l2sq", "cos", "ip", "hamming"):
        dim = t[m].get("dim", 0)
        for v in t[m]["ok"]:
            assert oracle.distance_valid(K.special(v), METRICS[m], dim), (m, v)
        for v in t[m]["err"]:
            assert not oracle.distance_valid(K.special(v), METRICS[m], dim), (m, v)


def test_b16_similarity_scores():
    for c in K.KAT["B16_scores"]["cases"]:
        got = oracle.similarity(c["d"], METRICS[c["metric"]], c.get("dim", 0))
        assert got == pytest.approx(np.float32(c["score"]), abs=1e-6), c


def test_b10_self_zero_b1():
    # quantization.rs:292-358: B1 self-distance is exactly 0 at d=1536 and d=100
    for dim in (1536, 100):
        ix = OracleIndex(dim, oracle.HAMMING)
        ix.reserve(4)
        v = np.full(dim, 0.5, dtype=np.float32)
        ix.add(1, v)
        keys, d = ix.search(v, 1)
        assert keys.tolist() == [1] and d.tolist() == [0.0]


def test_b17_concurrent_add_and_search():
    t = K.KAT["B17_concurrency"]
    tasks, per = 16, t["adds_per_worker"]
    ix = OracleIndex(t["dim"], oracle.L2SQ)
    ix.reserve(tasks * per)
    z = np.zeros(t["dim"], dtype=np.float32)
    errs = []

    # The reference's actor never overlaps adds with searches (usearch.rs:590-612); adds || adds only.
    def adder(tid):
        try:
            keys = np.arange(tid * per, (tid + 1) * per, dtype=np.uint64)
            ix.L.orc_add_batch(ix.h, keys.ctypes.data, np.zeros((per, t["dim"]), np.float32).ctypes.data, per, 1)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ix.add(10 ** 9, z)
    ix.add_batch(np.arange(tasks * per - 1, dtype=np.uint64), np.zeros((tasks * per - 1, t["dim"]), np.float32),
                 threads=8)
    assert ix.size() == tasks * per
    res = []

    def searcher():
        res.append(ix.search_batch(np.zeros((per, t["dim"]), np.float32), t["search_k"], threads=2))

    th = [threading.Thread(target=searcher) for _ in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs
    for keys, d, found in res:
        assert (found == t["search_k"]).all() and (d == 0).all()


def test_errors_match_usearch_contract():
    ix = OracleIndex(3, oracle.L2SQ)
    with pytest.raises(oracle.OracleError, match="Reserve capacity"):
        ix.add(1, [1, 2, 3])
    ix.reserve(2)
    assert ix.capacity() == 2
    ix.add(1, [1, 2, 3])
    with pytest.raises(oracle.OracleError, match="Duplicate"):
        ix.add(1, [1, 2, 3])
    assert ix.remove(7) is False
    assert ix.remove(1) is True and ix.size() == 0
    keys, _ = ix.search([1, 2, 3], 5)
    assert len(keys) == 0  # removed entries are never returned


def test_level_stream_is_deterministic_and_geometric():
    lv = oracle.level_stream(16, 200000)
    assert (lv == oracle.level_stream(16, 200000)).all()
    frac = [(lv >= l).mean() for l in range(4)]
    for l, rel in ((1, 0.05), (2, 0.15), (3, 0.5)):
        assert frac[l] == pytest.approx(16.0 ** -l, rel=rel)


def test_recall_config1_10k_128_cosine():
    """BASELINE.json configs[0]: 10k random f32 vectors, dim=128, cosine, top-10 (CPU plumbing)."""
    rng = np.random.Generator(np.random.PCG64(1234))
    base = rng.standard_normal((10000, 128), dtype=np.float32)
    q = np.random.Generator(np.random.PCG64(4321)).standard_normal((200, 128), dtype=np.float32)
    ix = OracleIndex(128, oracle.COS, 16, 128, 64)
    ix.reserve(10000)
    ix.add_batch(np.arange(10000, dtype=np.uint64), base, threads=8)
    keys, d, found = ix.search_batch(q, 10, threads=8)
    bn = base / np.linalg.norm(base, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    truth = np.argsort(1.0 - qn @ bn.T, axis=1, kind="stable")[:, :10]

    def recall_of(kk):
        return np.mean([len(set(truth[i]) & set(kk[i].tolist())) / 10.0 for i in range(len(q))])

    assert (found == 10).all()
    # i.i.d. Gaussian in 128-d has no neighbourhood structure: HNSW recall is low at ef=64 and
    # must rise monotonically with the beam (measured: 0.61 / 0.82 / 0.95 at ef 64 / 128 / 256).
    r64 = recall_of(keys)
    ix.set_expansion_search(256)
    keys256, _, _ = ix.search_batch(q, 10, threads=8)
    r256 = recall_of(keys256)
    assert r64 >= 0.5 and r256 >= 0.9 and r256 > r64, (r64, r256)
    # distances agree with the exact formula
    for i in range(0, 200, 37):
        for j in range(10):
            want = 1.0 - float(qn[i] @ bn[int(keys[i, j])])
            assert abs(float(d[i, j]) - want) <= 1e-5
    # exact search agrees with numpy ground truth
    ek, ed = ix.exact_search(q[0], 10)
    assert set(ek.tolist()) == set(truth[0].tolist())


def test_export_import_roundtrip_same_results():
    rng = np.random.default_rng(7)
    base = rng.standard_normal((2000, 24)).astype(np.float32)
    q = rng.standard_normal((50, 24)).astype(np.float32)
    a = OracleIndex(24, oracle.L2SQ)
    a.reserve(2000)
    a.add_batch(np.arange(2000, dtype=np.uint64) + 5, base, threads=1)
    g = a.export_graph()
    assert g["adj0"].shape == (2000, 32) and (g["levels"] >= 0).all()
    b = OracleIndex(24, oracle.L2SQ)
    b.import_graph(g)
    for i in range(50):
        ka, da = a.search(q[i], 10)
        kb, db = b.search(q[i], 10)
        assert ka.tolist() == kb.tolist() and da.tolist() == db.tolist()


def test_filtered_timed_driver_equals_one_query_at_a_time():
    """oracle.filtered_search_timed (bench.py's CPU baseline beside boundary.filtered): threads x one query per call with the
    predicate key % m == 0 answers what filtered_search answers, and counts the predicate's calls."""
    rng = np.random.default_rng(11)
    base = rng.standard_normal((3000, 16)).astype(np.float32)
    q = rng.standard_normal((40, 16)).astype(np.float32)
    o = OracleIndex(16, oracle.L2SQ)
    o.reserve(3000)
    o.add_batch(np.arange(3000, dtype=np.uint64), base, threads=1)
    keys, d, found, answered, calls, wall = o.filtered_search_timed(q, 5, 7, threads=3, seconds=30.0)
    assert answered == 40 and wall > 0 and calls >= 40 * 5
    for i in range(40):
        ek, ed = o.filtered_search(q[i], 5, lambda key: key % 7 == 0)
        assert found[i] == len(ek) == 5
        assert keys[i].tolist() == ek.tolist() and d[i].tolist() == ed.tolist()
        assert all(int(x) % 7 == 0 for x in keys[i])
