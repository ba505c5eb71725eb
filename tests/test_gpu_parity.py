"""GPU parity tests: the HIP engine, called through the C ABI (libvs_hnsw.so), against the CPU
oracle on the same inputs and against the reference's known-answer tests.

Bar (integer / index work): identical ids.  Floating point: |d_gpu - d_cpu| <= 1e-5 * max(1, |d|)
(f32 re-association of a <=1536-term sum).  Where two candidates are
closer than that tolerance their order is unspecified (usearch leaves tie order unspecified too).
"""
import threading

import numpy as np
import pytest

import oracle
from oracle import OracleIndex
from tests import kat_runner as K
from tests.parity_util import TOL, assert_same_results, lattice

pytestmark = pytest.mark.gpu


def vs():
    import vector_store_amd as v
    return v


def gpu_factory(metric, dim, **kw):
    v = vs()
    return v.HipUsearchIndex(dim, v.METRICS[metric], **kw)


def close(a, b):
    return abs(float(a) - float(b)) <= TOL * max(1.0, abs(float(b)))


# ------------------------------------------------------------------ reference known-answer tests
@pytest.mark.parametrize("name", ["B2_l2sq_3d_http", "B3_l2sq_1d_scores", "B4_empty", "B5_cos_winners",
                                  "B6_ip_winner", "B7_l2_winner", "B8_quant_f32", "B10_self_zero_f32"])
def test_kat_simple(name):
    v = vs()
    keys, d = K.run_simple(gpu_factory, name)
    t = K.KAT[name]
    for x in d:
        assert v.distance_valid(float(x), v.METRICS[t["metric"]], t["dim"])
    if "expect_similarity" in t:
        for x, want in zip(d, t["expect_similarity"]):
            assert abs(v.similarity_score(float(x), v.METRICS[t["metric"]], t["dim"]) - want) <= 1e-5


def test_kat_b1_add_remove_readd():
    K.run_b1(gpu_factory)


def test_kat_b11_filtered_30():
    K.run_b11(gpu_factory)


def test_kat_b12_fine_order():
    first, _ = K.run_b12(gpu_factory)
    assert first == sorted(first)
    assert first == list(range(100))


def test_kat_b13_zero_query():
    K.run_b13(gpu_factory)


def test_kat_b17_concurrency():
    """usearch.rs:1526-1607: 2 x cores tasks x 50 adds and as many searches, no error, final count."""
    v = vs()
    t = K.KAT["B17_concurrency"]
    tasks, per = 16, t["adds_per_worker"]
    ix = v.HipUsearchIndex(t["dim"], v.L2SQ)
    ix.reserve(tasks * per)
    z = np.zeros(t["dim"], dtype=np.float32)
    errs = []

    def adder(tid):
        try:
            for i in range(per):
                ix.add(tid * per + i, z)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    def searcher():
        try:
            for _ in range(per):
                keys, d = ix.search(z, t["search_k"])
                assert len(keys) <= t["search_k"] and all(float(x) == 0.0 for x in d)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    # the reference's actor alternates families: adds || adds, then searches || searches (usearch.rs:590-612)
    th = [threading.Thread(target=adder, args=(i,)) for i in range(tasks)]
    [x.start() for x in th]
    [x.join() for x in th]
    th = [threading.Thread(target=searcher) for _ in range(tasks)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    assert ix.size() == tasks * per
    keys, d = ix.search(z, t["search_k"])
    assert len(keys) == t["search_k"]


# ------------------------------------------------------------------ same graph, same queries: ids identical
def _dataset(n, dim, seed, kind="lowrank"):
    rng = np.random.default_rng(seed)
    if kind == "gauss" or dim <= 8:
        return rng.standard_normal((n, dim)).astype(np.float32)
    r = min(16, dim)
    w = rng.standard_normal((r, dim)).astype(np.float32) / np.sqrt(r)
    return (rng.standard_normal((n, r)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


def _oracle_dist(metric, q, base, key_to_row, kind="f32"):
    return lambda key: oracle.distance_as(oracle.METRICS[metric], oracle.SCALARS[kind], q, base[key_to_row(key)])


@pytest.mark.parametrize("metric", ["cos", "l2sq", "ip"])
@pytest.mark.parametrize("dim,n", [(3, 500), (24, 3000), (100, 3000), (128, 4000), (768, 3000), (1536, 1500)])
def test_search_matches_oracle_on_same_graph(metric, dim, n):
    """Same graph, same queries => the same ids in the same order; the only admissible difference is an f32 near-tie,
    and every differing position is checked to be one (tests/parity_util.py)."""
    v = vs()
    m = oracle.METRICS[metric]
    data = _dataset(n + 64, dim, 11 + dim)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    o = OracleIndex(dim, m)
    o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64) * 3 + 7, base, threads=1)
    g = o.export_graph()
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.import_graph(g)
    assert ix.size() == n
    ties = rows = 0
    for ef, k in ((64, 10), (128, 10), (200, 100), (400, 100)):  # (the 64 queries of a batch take the team kernels, 512-entry one included)
        o.set_expansion_search(ef)
        ix.set_expansion_search(ef)
        gk, gd, gf = ix.search_batch(q, k)
        for i in range(len(q)):
            ok_, od_ = o.search(q[i], k)
            assert gf[i] == len(ok_)
            ties += assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_,
                                        _oracle_dist(metric, q[i], base, lambda key: (key - 7) // 3), what=(metric, dim, ef, i))
            rows += 1
        st = ix.stats(reset=True)
        assert st["visited_overflow"] == 0
    assert ties <= max(3, rows // 25), (metric, dim, ties)  # near-ties are rare events, not a loophole
    # single-query entry point == batch entry point
    k1, d1 = ix.search(q[0], 10)
    kb, db, _ = ix.search_batch(q[:1], 10)
    assert k1.tolist() == kb[0, : len(k1)].tolist()


@pytest.mark.parametrize("metric", ["l2sq", "ip"])
@pytest.mark.parametrize("dim,n", [(16, 4000), (128, 3000), (768, 2000), (1536, 1000)])
def test_usearch_order_walk_is_bit_identical_on_exactly_representable_data(metric, dim, n):
    """Integer lattice vectors: every distance is an integer below 2^24, exact in f32 in any summation order, so there is
    no rounding to excuse anything -- ids and distance bits must equal the oracle's, ties (plenty on a lattice)
    included.  options.reserved bit 4 selects the walk that keeps usearch's `top` and `next` apart for a float index
    (it is the default for i8 / b1); removed members and a wide beam are part of the case."""
    v = vs()
    data = lattice(n + 48, dim, 5 + dim, span=32 if dim > 768 else 64)
    base, q = data[:n], data[n:]
    o = OracleIndex(dim, oracle.METRICS[metric])
    o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64) + 1, base, threads=1)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric], _stress=16)
    ix.import_graph(o.export_graph())

    def compare():
        for ef, k in ((64, 10), (128, 50), (200, 100), (400, 400)):
            o.set_expansion_search(ef)
            ix.set_expansion_search(ef)
            gk, gd, gf = ix.search_batch(q, k)
            for i in range(len(q)):
                ok_, od_ = o.search(q[i], k)
                assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_, exact=True, what=(metric, dim, ef, i))
        k1, d1 = ix.search(q[0], 10)          # the single-query entry point takes the same walk
        o.set_expansion_search(64)
        ix.set_expansion_search(64)
        ok_, od_ = o.search(q[0], 10)
        assert_same_results(*ix.search(q[0], 10), ok_, od_, exact=True)

    compare()
    for key in range(1, n, 3):                # a third of the members removed: traversed, never returned
        assert ix.remove(key) and o.remove(key)
    compare()


def test_configs0_10k_x_128_cosine_top10():
    """BASELINE configs[0] (10k random f32 vectors, dim 128, cosine, top-10: the CPU-runnable plumbing case) through
    the HIP path: SURVEY.md section 8d's generator (numpy PCG64 standard_normal, seeds 1234 / 4321), M = 16,
    ef_add = 128, ef_search = 64; the oracle builds the graph, both search it."""
    v = vs()
    n, dim, nq, k = 10_000, 128, 1000, 10
    base = np.random.Generator(np.random.PCG64(1234)).standard_normal((n, dim), dtype=np.float32)
    q = np.random.Generator(np.random.PCG64(4321)).standard_normal((nq, dim), dtype=np.float32)
    o = OracleIndex(dim, oracle.COS, 16, 128, 64)
    o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64), base, threads=1)
    ix = v.HipUsearchIndex(dim, v.COS, 16, 128, 64)
    ix.import_graph(o.export_graph())
    gk, gd, gf = ix.search_batch(q, k)
    ok_, od_, of_ = o.search_batch(q, k, threads=8)
    ties = 0
    for i in range(nq):
        assert gf[i] == of_[i] == k
        ties += assert_same_results(gk[i], gd[i], ok_[i], od_[i], _oracle_dist("cos", q[i], base, lambda key: key), what=i)
    assert ties <= 10, ties
    # and built by the engine itself: recall against the exact search is what the CPU algorithm reaches on its own graph
    ix2 = v.HipUsearchIndex(dim, v.COS, 16, 128, 64)
    ix2.reserve(n)
    ix2.add_batch(np.arange(n, dtype=np.uint64), base)
    tk, _, _ = ix2.exact_search_batch(q, k)
    k2, _, _ = ix2.search_batch(q, k)
    rec_gpu = np.mean([len(set(tk[i].tolist()) & set(k2[i].tolist())) / k for i in range(nq)])
    rec_cpu = np.mean([len(set(tk[i].tolist()) & set(ok_[i].tolist())) / k for i in range(nq)])
    assert rec_gpu >= rec_cpu - 0.03, (rec_gpu, rec_cpu)


def test_wide_visited_tags_build_and_search_like_plain_ones():
    """options.reserved bit 6 runs the wide-tag instances (indexes above 2^25 / 2^26 slots) on a small index: the visited
    set is exact either way, so the graph and every answer must be bit-identical to the plain-tag build."""
    v = vs()
    n, dim = 40000, 64
    data = _dataset(n + 200, dim, 61)
    out = []
    for hook in (0, 64):
        ix = v.HipUsearchIndex(dim, v.COS, expansion_add=200, _stress=hook)   # expansion_add 200: the two-choice table
        ix.reserve(n)
        ix.add_batch(np.arange(n, dtype=np.uint64), data[:n])
        g = ix.export_graph()
        res = []
        for ef in (64, 200, 400):
            ix.set_expansion_search(ef)
            res.append(ix.search_batch(data[n:], 10))
        assert ix.stats()["visited_overflow"] == 0
        out.append((g, res))
    (g0, r0), (g1, r1) = out
    assert np.array_equal(g0["adj0"], g1["adj0"]) and np.array_equal(g0["upper"], g1["upper"])
    for a, b in zip(r0, r1):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_evals_and_hops_match_oracle_counters():
    """The engine counts E_q / H_q exactly as the CPU restatement does."""
    v = vs()
    n, dim = 5000, 64
    data = _dataset(n + 100, dim, 5)
    o = OracleIndex(dim, oracle.L2SQ)
    o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64), data[:n], threads=1)
    ix = v.HipUsearchIndex(dim, v.L2SQ)
    ix.import_graph(o.export_graph())
    o.stats(reset=True)
    ix.stats(reset=True)
    for i in range(100):
        o.search(data[n + i], 10)
    ix.search_batch(data[n:], 10)
    so, sg = o.stats(), ix.stats()
    assert sg["queries"] == 100
    assert abs(sg["search_evals"] - so["computed_distances"]) <= 0.01 * so["computed_distances"]
    assert abs(sg["search_hops"] - so["node_expansions"]) <= 0.01 * so["node_expansions"]


# ------------------------------------------------------------------ exact search == numpy brute force
@pytest.mark.parametrize("metric", ["cos", "l2sq", "ip"])
def test_exact_search_matches_numpy(metric):
    v = vs()
    n, dim, k = 70000, 96, 10  # crosses the 65536-row chunk boundary
    rng = np.random.default_rng(3)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((37, dim)).astype(np.float32)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64) + 100, base)
    keys, d, found = ix.exact_search_batch(q, k)
    b64, q64 = base.astype(np.float64), q.astype(np.float64)
    if metric == "l2sq":
        D = (q64 ** 2).sum(1)[:, None] + (b64 ** 2).sum(1)[None, :] - 2 * q64 @ b64.T
    elif metric == "ip":
        D = 1.0 - q64 @ b64.T
    else:
        D = 1.0 - (q64 / np.linalg.norm(q64, axis=1, keepdims=True)) @ (b64 / np.linalg.norm(b64, axis=1, keepdims=True)).T
    truth = np.argsort(D, axis=1, kind="stable")[:, :k]
    assert (found == k).all()
    for i in range(len(q)):
        assert set((keys[i] - 100).tolist()) == set(truth[i].tolist()), (metric, i)
        for j in range(k):
            assert close(d[i, j], D[i, int(keys[i, j]) - 100])
        assert all(d[i, j] <= d[i, j + 1] for j in range(k - 1))


# ------------------------------------------------------------------ GPU build
def _graph_invariants(g, M, M0):
    n = len(g["levels"])
    adj0 = g["adj0"]
    valid = adj0 != 0xFFFFFFFF
    assert adj0.shape == (n, M0)
    assert (adj0[valid] < n).all()
    for s in range(n):
        row = adj0[s][valid[s]]
        assert len(set(row.tolist())) == len(row), "duplicate link"
        assert s not in row, "self loop"
        # packed at the front
        assert valid[s][: len(row)].all()
    for s in np.nonzero(g["levels"] > 0)[0]:
        for l in range(1, g["levels"][s] + 1):
            blk = g["upper"][g["upper_off"][s] + l - 1]
            nb = blk[blk != 0xFFFFFFFF]
            assert (nb < n).all() and s not in nb and len(set(nb.tolist())) == len(nb)
            assert (g["levels"][nb] >= l).all(), "link to a node missing on that level"
    assert g["levels"][g["entry_slot"]] == g["max_level"] == g["levels"].max()


@pytest.mark.parametrize("metric,dim,n", [("cos", 128, 20000), ("l2sq", 24, 20000), ("ip", 768, 6000)])
def test_gpu_build_recall_and_invariants(metric, dim, n):
    v = vs()
    data = _dataset(n + 200, dim, 21)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    keys = np.arange(n, dtype=np.uint64) + (1 << 48)  # epoch bits set, like the reference's PrimaryId
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.reserve(n)
    ix.add_batch(keys[: n // 2], base[: n // 2])
    ix.add_batch(keys[n // 2:], base[n // 2:])
    assert ix.size() == n
    g = ix.export_graph()
    _graph_invariants(g, 16, 32)
    assert np.array_equal(g["vectors"], base)
    assert (g["levels"] == oracle.level_stream(16, n)).all(), "level stream == single-thread usearch context"
    st = ix.stats(reset=True)
    assert st["added"] == n - 1 and st["visited_overflow"] == 0
    tk, td, _ = ix.exact_search_batch(q, 10)
    ix.set_expansion_search(128)
    gk, gd, gf = ix.search_batch(q, 10)
    rec_gpu = np.mean([len(set(tk[i].tolist()) & set(gk[i].tolist())) / 10.0 for i in range(len(q))])
    # the CPU restatement on the same data, same parameters
    o = OracleIndex(dim, oracle.METRICS[metric])
    o.reserve(n)
    o.add_batch(keys, base, threads=8)
    o.set_expansion_search(128)
    ok_, _, _ = o.search_batch(q, 10, threads=8)
    rec_cpu = np.mean([len(set(tk[i].tolist()) & set(ok_[i].tolist())) / 10.0 for i in range(len(q))])
    assert rec_gpu >= rec_cpu - 0.03, (rec_gpu, rec_cpu)
    assert rec_gpu >= 0.9, rec_gpu
    # the GPU-built graph searched by the CPU algorithm gives the same answers as the GPU search
    o2 = OracleIndex(dim, oracle.METRICS[metric])
    o2.import_graph(g)
    o2.set_expansion_search(128)
    ties = 0
    for i in range(len(q)):
        k2, d2 = o2.search(q[i], 10)
        ties += assert_same_results(gk[i], gd[i], k2, d2, _oracle_dist(metric, q[i], base, lambda key: key - (1 << 48)),
                                    what=(metric, i))
    assert ties <= 4, ties


@pytest.mark.timeout(900)
@pytest.mark.parametrize("generator", ["gaussian_pcg64", "clustered_pcg64"])
def test_gpu_built_graph_on_the_surveys_generators_at_768d(generator):
    """SURVEY.md section 8(d) names the synthetic inputs: numpy PCG64 standard_normal (seeds 1234 base / 4321 queries) -- the
    hardest case for any graph index -- and its clustered variant (256 centres of seed 99, sigma 0.2).  At 768-d, 100,000 rows:
    the batched GPU build (sub-batches against a frozen graph) must be as good a graph as the CPU restatement's insertion of the
    same rows, measured as recall@10 against exact search at beams of 128 and 512 -- whatever that recall is (on i.i.d.
    Gaussian data it is low for every HNSW: that is the data, and the point of the comparison)."""
    v = vs()
    n, nq, dim, k = 100_000, 300, 768, 10
    if generator == "gaussian_pcg64":
        base = np.random.Generator(np.random.PCG64(1234)).standard_normal((n, dim), dtype=np.float32)
        q = np.random.Generator(np.random.PCG64(4321)).standard_normal((nq, dim), dtype=np.float32)
    else:
        centres = np.random.Generator(np.random.PCG64(99)).standard_normal((256, dim), dtype=np.float32)
        g1, g2 = np.random.Generator(np.random.PCG64(1234)), np.random.Generator(np.random.PCG64(4321))
        base = centres[g1.integers(0, 256, n)] + 0.2 * g1.standard_normal((n, dim), dtype=np.float32)
        q = centres[g2.integers(0, 256, nq)] + 0.2 * g2.standard_normal((nq, dim), dtype=np.float32)
    keys = np.arange(n, dtype=np.uint64)
    ix = v.HipUsearchIndex(dim, v.COS)
    ix.reserve(n)
    ix.add_batch(keys, base)
    tk, _, _ = ix.exact_search_batch(q, k)
    o = OracleIndex(dim, oracle.COS)
    o.reserve(n)
    o.add_batch(keys, base, threads=8)
    got = {}
    for ef in (128, 512):
        ix.set_expansion_search(ef)
        o.set_expansion_search(ef)
        gk, _, _ = ix.search_batch(q, k)
        ck, _, _ = o.search_batch(q, k, threads=8)
        rec_gpu = float(np.mean([len(set(tk[i].tolist()) & set(gk[i].tolist())) / k for i in range(nq)]))
        rec_cpu = float(np.mean([len(set(tk[i].tolist()) & set(ck[i].tolist())) / k for i in range(nq)]))
        got[ef] = (rec_gpu, rec_cpu)
        assert rec_gpu >= rec_cpu - 0.03, (generator, ef, rec_gpu, rec_cpu)
    print(f"[{generator} 100k x 768] recall@10 GPU-built / CPU-built: ef 128 {got[128][0]:.3f} / {got[128][1]:.3f}, ef 512 {got[512][0]:.3f} / {got[512][1]:.3f}")
    assert got[512][0] >= got[128][0]
    if generator == "clustered_pcg64":
        assert got[512][0] >= 0.9, got


@pytest.mark.parametrize("metric,dim,span", [("l2sq", 8, 500), ("ip", 16, 200)])
def test_sequential_adds_build_the_oracle_graph(metric, dim, span):
    """One add at a time (a barrier after each: sub-batch of 1) is the sequential usearch algorithm.  On exactly
    representable data (integer coordinates, every distance an integer below 2^24) nothing depends on rounding, so
    the graph must equal the single-threaded CPU restatement's: levels, entry point, every adjacency row."""
    v = vs()
    n = 1500
    base = lattice(n, dim, 9, span=span)
    o = OracleIndex(dim, oracle.METRICS[metric])
    o.reserve(n)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.reserve(n)
    for i in range(n):
        o.add(i, base[i])
        ix.add(i, base[i])
        assert ix.size() == i + 1  # size() is a barrier: the staged vector is inserted now, alone
    go, gg = o.export_graph(), ix.export_graph()
    assert (go["levels"] == gg["levels"]).all()
    assert go["entry_slot"] == gg["entry_slot"] and go["max_level"] == gg["max_level"]
    differ = [s_ for s_ in range(n) if set(go["adj0"][s_].tolist()) != set(gg["adj0"][s_].tolist())]
    # wide coordinates make equal distances rare but not impossible; a row may differ only if an exact tie is involved
    for s_ in differ[:20]:
        d = ((base - base[s_]) ** 2).sum(1) if metric == "l2sq" else 1.0 - base @ base[s_]
        cand = set(go["adj0"][s_].tolist()) ^ set(gg["adj0"][s_].tolist())
        cand.discard(0xFFFFFFFF)
        assert any(np.sum(d == d[c]) > 1 for c in cand), (s_, "rows differ without an exact tie")
    assert len(differ) <= n // 100, len(differ)
    for s_ in np.nonzero(go["levels"] > 0)[0]:
        for l in range(1, go["levels"][s_] + 1):
            a, b = go["upper"][go["upper_off"][s_] + l - 1], gg["upper"][gg["upper_off"][s_] + l - 1]
            if s_ not in differ:
                assert set(a.tolist()) == set(b.tolist())


def test_remove_update_and_errors():
    v = vs()
    dim = 8
    rng = np.random.default_rng(1)
    base = rng.standard_normal((300, dim)).astype(np.float32)
    ix = v.HipUsearchIndex(dim, v.L2SQ)
    with pytest.raises(v.VsError, match="Reserve capacity"):
        ix.add(1, base[0])
    ix.reserve(300)
    assert ix.capacity() == 300
    ix.add_batch(np.arange(300, dtype=np.uint64), base)
    with pytest.raises(v.VsError, match="Duplicate"):
        ix.add(5, base[5])
    with pytest.raises(v.VsError) as e:
        ix.add(1000, base[0][:4])
    assert e.value.code == -2  # VS_ERR_DIMENSION -> HTTP 400 in the reference (validator.rs:12-26)
    with pytest.raises(v.VsError) as e:
        ix.search(base[0][:4], 3)
    assert e.value.code == -2
    with pytest.raises(v.VsError, match="Reserve capacity"):
        ix.add(1000, base[0])
    # removed members are never returned, the slot is reused by the next add (update path)
    assert ix.remove(17) and not ix.remove(17)
    assert ix.size() == 299
    keys, d = ix.search(base[17], 5)
    assert 17 not in keys.tolist()
    newv = base[17] + 0.001
    ix.add((3 << 48) | 17, newv)  # same row index, next epoch: how the reference re-keys an update
    assert ix.size() == 300
    keys, d = ix.search(newv, 1)
    assert keys.tolist() == [(3 << 48) | 17] and float(d[0]) == 0.0
    keys, _ = ix.search(base[17], 300)
    assert len(keys) == 300 and 17 not in keys.tolist()
    # growing keeps everything
    ix.reserve(1000)
    assert ix.capacity() == 1000
    keys, d = ix.search(base[3], 1)
    assert keys.tolist() == [3] and float(d[0]) == 0.0
    ix.add(2000, base[0] * 2)
    assert ix.size() == 301


@pytest.mark.parametrize("metric,dim,n", [("l2sq", 12, 1200), ("cos", 96, 30000), ("ip", 768, 8000)])
def test_limits_beyond_512_take_a_wide_walk(metric, dim, n):
    """The reference passes any `limit` through (httproutes.rs:842-847; benchmark CLI up to 10,000): 513..10,240 is a
    walk with a wide `top` (global visited bitmap, heap spilling to global memory), not an exhaustive ranking --
    same ids as the CPU algorithm with the same expansion on the same graph."""
    v = vs()
    data = _dataset(n + 16, dim, 2)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.METRICS[metric])
    o.import_graph(ix.export_graph())
    for ef, k in ((64, 1000), (2000, 100), (2000, 2000), (64, 6000)):
        k = min(k, n - 100)
        ix.set_expansion_search(ef)
        o.set_expansion_search(ef)
        ix.stats(reset=True)
        gk, gd, gf = ix.search_batch(q[:8], k)
        assert ix.stats()["queries"] == 8          # a walk ran (the exhaustive path counts no queries)
        for i in range(8):
            ok_, od_ = o.search(q[i], k)
            assert gf[i] == len(ok_)
            assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_, _oracle_dist(metric, q[i], base, lambda key: key),
                                what=(metric, ef, k, i))
        k1, d1 = ix.search(q[0], k)                # single-query entry point
        assert k1.tolist() == gk[0, : gf[0]].tolist()
    if n <= 2000:
        keys, d = ix.search(base[0], 1000)
        want = np.sort(((base - base[0]) ** 2).sum(1))[:1000]
        assert len(keys) == 1000 and keys[0] == 0 and np.allclose(d, want, rtol=1e-5, atol=1e-5)


def test_topk_merge_matches_numpy():
    import torch
    v = vs()
    parts, nq, k = 8, 300, 10
    rng = np.random.default_rng(4)
    d = np.sort(rng.random((parts, nq, k)).astype(np.float32), axis=2)
    keys = rng.integers(0, 1 << 60, size=(parts, nq, k), dtype=np.uint64)
    d[3, :, 7:] = np.inf
    keys[3, :, 7:] = 0xFFFFFFFFFFFFFFFF
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    dd = torch.from_numpy(d).cuda()
    ok_ = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    of = torch.empty((nq,), dtype=torch.int32, device="cuda")
    v.topk_merge_device(dk.data_ptr(), dd.data_ptr(), parts, nq, k, ok_.data_ptr(), od.data_ptr(), of.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got_k = ok_.cpu().numpy().view(np.uint64)
    got_d = od.cpu().numpy()
    for qi in range(nq):
        flat_d = d[:, qi, :].reshape(-1)
        flat_k = keys[:, qi, :].reshape(-1)
        order = np.argsort(flat_d, kind="stable")[:k]
        assert np.array_equal(got_d[qi], flat_d[order])
        assert np.array_equal(got_k[qi], flat_k[order])


def test_device_entry_points_with_torch_buffers():
    import torch
    v = vs()
    n, dim, k = 8000, 768, 10
    data = _dataset(n + 256, dim, 31)
    ix = v.HipUsearchIndex(dim, v.COS)
    ix.reserve(n)
    dv = torch.from_numpy(data[:n]).cuda()
    ix.add_batch_device(np.arange(n, dtype=np.uint64), dv.data_ptr(), n, dim)
    dq = torch.from_numpy(data[n:]).cuda()
    ok_ = torch.empty((256, k), dtype=torch.int64, device="cuda")
    od = torch.empty((256, k), dtype=torch.float32, device="cuda")
    of = torch.empty((256,), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ix.search_batch_device(dq.data_ptr(), 256, k, ok_.data_ptr(), od.data_ptr(), of.data_ptr(), s)
    torch.cuda.synchronize()
    hk, hd, hf = ix.search_batch(data[n:], k)
    assert np.array_equal(ok_.cpu().numpy().view(np.uint64), hk)
    assert np.array_equal(od.cpu().numpy(), hd)
    ek = torch.empty((256, k), dtype=torch.int64, device="cuda")
    ed = torch.empty((256, k), dtype=torch.float32, device="cuda")
    ix.exact_search_batch_device(dq.data_ptr(), 256, k, ek.data_ptr(), ed.data_ptr(), of.data_ptr(), s)
    torch.cuda.synchronize()
    truth = ek.cpu().numpy().view(np.uint64)
    rec = np.mean([len(set(truth[i].tolist()) & set(hk[i].tolist())) / k for i in range(256)])
    assert rec >= 0.9, rec


def test_visited_table_overflow_is_graceful():
    """When the LDS visited table overflows, only evaluations are repeated: results stay
    duplicate-free and identical to the CPU algorithm on the same graph.  The test hook
    (options.reserved bit 0) swaps in a 256-bucket table so that every query overflows."""
    v = vs()
    n, dim, k = 60000, 32, 100  # 32 f32 = 8 chunks = 8 lanes x 1: the hook exists for 1-chunk-per-lane layouts
    data = _dataset(n + 64, dim, 77)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.L2SQ, _stress=1)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    ix.set_expansion_search(128)
    ix.stats(reset=True)
    gk, gd, gf = ix.search_batch(q, k)
    st = ix.stats()
    assert st["visited_overflow"] > 0, "test no longer reaches the overflow path"
    assert (gf == k).all()
    for i in range(len(q)):
        assert len(set(gk[i].tolist())) == k, "duplicate result"
        assert all(gd[i, j] <= gd[i, j + 1] for j in range(k - 1))
    o = OracleIndex(dim, oracle.L2SQ)
    o.import_graph(ix.export_graph())
    o.set_expansion_search(128)
    ties = 0
    for i in range(len(q)):
        ok_, od_ = o.search(q[i], k)
        ties += assert_same_results(gk[i], gd[i], ok_, od_, _oracle_dist("l2sq", q[i], base, lambda key: key), what=i)
    assert ties <= 4, ties


@pytest.mark.parametrize("frac", [0.05, 0.4])
def test_removed_members_are_traversed_but_never_returned(frac):
    """usearch keeps removed nodes linked; `top` holds ef LIVE members while removed ones still steer
    the walk.  Same graph + same removals => same ids as the CPU restatement."""
    v = vs()
    n, dim, k = 6000, 48, 10
    data = _dataset(n + 64, dim, 123)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    rng = np.random.default_rng(8)
    gone = rng.choice(n, size=int(frac * n), replace=False)
    for key in gone.tolist():
        assert ix.remove(key)
    assert ix.size() == n - len(gone)
    o = OracleIndex(dim, oracle.COS)
    o.import_graph(ix.export_graph())  # keys of removed slots are the free key in the export
    assert o.size() == n - len(gone)
    gone_set = set(gone.tolist())
    for ef in (64, 200):
        ix.set_expansion_search(ef)
        o.set_expansion_search(ef)
        gk, gd, gf = ix.search_batch(q, k)
        ties = 0
        for i in range(len(q)):
            ok_, od_ = o.search(q[i], k)
            assert gf[i] == len(ok_) == k
            assert not (set(gk[i].tolist()) & gone_set)
            ties += assert_same_results(gk[i], gd[i], ok_, od_, _oracle_dist("cos", q[i], base, lambda key: key), what=(frac, ef, i))
        assert ties <= 3, (frac, ef, ties)


def test_async_search_matches_blocking_search():
    v = vs()
    n, dim, k = 5000, 32, 10
    data = _dataset(n + 200, dim, 55)
    ix = v.HipUsearchIndex(dim, v.L2SQ)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), data[:n])
    want = [ix.search(data[n + i], k) for i in range(200)]
    got, done, lock = {}, threading.Event(), threading.Lock()
    holders = []

    def make(i):
        def on_done(keys, d, status):
            with lock:
                got[i] = (keys.copy(), d.copy(), status)
                if len(got) == 200:
                    done.set()
        return on_done

    for i in range(200):
        holders.append(ix.search_async(data[n + i], k, make(i)))
    assert done.wait(30)
    for i in range(200):
        assert got[i][2] == 0
        assert got[i][0].tolist() == want[i][0].tolist() and got[i][1].tolist() == want[i][1].tolist()


@pytest.mark.parametrize("metric,quant", [("cos", "f32"), ("ip", "f32"), ("cos", "f16"), ("ip", "i8")])
def test_exact_mfma_block_distances_equal_valu_kernel(metric, quant):
    """K8: the MFMA block-distance kernel (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain) returns what the
    VALU tile kernel returns -- same keys, distances to the last bit or within 1 ulp-scale tolerance."""
    v = vs()
    n, dim, k = 30000, 768, 10
    data = _dataset(n + 256, dim, 91)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    out = {}
    for hook in (0, 2):  # options.reserved bit 1: dot-product family on the VALU kernel
        ix = v.HipUsearchIndex(dim, v.METRICS[metric], quantization=v.SCALARS[quant], _stress=hook)
        ix.reserve(n)
        ix.add_batch(np.arange(n, dtype=np.uint64), base)
        out[hook] = ix.exact_search_batch(q, k)  # q = 256: the batched configuration C5 of BASELINE.json
    (mk, md, mf), (vk, vd, vf) = out[0], out[2]
    assert (mf == k).all() and (vf == k).all()
    assert np.allclose(md, vd, rtol=1e-6, atol=1e-6)
    assert np.mean([mk[i].tolist() == vk[i].tolist() for i in range(len(q))]) >= 0.99
    if quant == "f32":
        b64, q64 = base.astype(np.float64), q.astype(np.float64)
        if metric == "ip":
            D = 1.0 - q64 @ b64.T
        else:
            D = 1.0 - (q64 / np.linalg.norm(q64, axis=1, keepdims=True)) @ (b64 / np.linalg.norm(b64, axis=1, keepdims=True)).T
        truth = np.argsort(D, axis=1, kind="stable")[:, :k]
        assert np.mean([set(mk[i].tolist()) == set(truth[i].tolist()) for i in range(len(q))]) >= 0.99
        assert all(close(md[i, j], D[i, int(mk[i, j])]) for i in range(0, 256, 17) for j in range(k))


@pytest.mark.parametrize("metric,dim", [("cos", 768), ("l2sq", 100)])
def test_wide_beam_up_to_512(metric, dim):
    """k = 500 (CQL LIMIT 100 x oversampling 5, validator quantization_and_rescoring.rs:109) stays on the LDS
    beam kernel (ef <= 512) and matches the CPU algorithm on the same graph."""
    v = vs()
    n = 6000
    data = _dataset(n + 32, dim, 333)
    o = OracleIndex(dim, oracle.METRICS[metric])
    o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64), data[:n], threads=1)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.import_graph(o.export_graph())
    for ef, k in ((300, 100), (64, 500)):
        o.set_expansion_search(ef)
        ix.set_expansion_search(ef)
        ix.stats(reset=True)
        gk, gd, gf = ix.search_batch(data[n:], k)
        assert ix.stats()["queries"] == 32  # the beam kernel ran (the exhaustive path does not count queries)
        ties = 0
        for i in range(32):
            ok_, od_ = o.search(data[n + i], k)
            assert gf[i] == len(ok_) == k
            ties += assert_same_results(gk[i], gd[i], ok_, od_, _oracle_dist(metric, data[n + i], data, lambda key: key), what=(ef, k, i))
        assert ties <= 6, (ef, k, ties)
    k1, d1 = ix.search(data[n], 500)  # single-query entry point too
    assert len(k1) == 500


@pytest.mark.parametrize("metric", ["ip", "cos", "l2sq"])
def test_exhaustive_ranking_on_the_device_matches_numpy(metric):
    """k beyond the LDS beam and selective filters rank every member on the device (distances, 64-bit radix sort of
    (distance, slot), members in ascending order, fetched in chunks): against a float64 numpy ranking, with removed
    members, negative distances (ip) and a predicate that only a few members pass."""
    v = vs()
    n, dim, k = 30000, 48, 12000  # beyond the widest walk (10,240): every member is ranked
    rng = np.random.default_rng(17)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal(dim).astype(np.float32)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    removed = set(range(0, n, 7))
    for key in removed:
        assert ix.remove(key)
    b64, q64 = base.astype(np.float64), q.astype(np.float64)
    if metric == "ip":
        ref = 1.0 - b64 @ q64
    elif metric == "cos":
        ref = 1.0 - (b64 @ q64) / (np.linalg.norm(b64, axis=1) * np.linalg.norm(q64))
    else:
        ref = ((b64 - q64) ** 2).sum(axis=1)
    live = np.array([i for i in range(n) if i not in removed])
    order = live[np.argsort(ref[live], kind="stable")]
    keys, dist = ix.search(q, k)                       # k = 12000 > 10240: exhaustive
    assert len(keys) == k and (np.diff(dist) >= 0).all() and not (set(keys.tolist()) & removed)
    assert len(set(keys.tolist()) ^ set(order[:k].tolist())) <= 8          # f32 near-ties at the cut only
    assert np.allclose(dist, ref[keys.astype(np.int64)], rtol=1e-4, atol=1e-4)
    if metric == "ip":
        assert dist[0] < 0                               # negative distances keep their order
    ix.set_expansion_search(20000)                     # a filter with a beam beyond every walk: ranked exhaustively too
    fk, fd = ix.filtered_search(q, 50, lambda key: key % 101 == 5)         # ~1 % pass
    want = [int(x) for x in order if x % 101 == 5][:50]
    assert len(fk) == 50 and len(set(fk.tolist()) ^ set(want)) <= 2
    ix.set_expansion_search(64)
    allk, alld = ix.search(q, 40000)                   # more than the members: everything live, once, in order
    assert len(allk) == len(live) and len(set(allk.tolist())) == len(live) and (np.diff(alld) >= 0).all()


@pytest.mark.timeout(120)
def test_non_finite_inputs_do_not_hang_or_poison_the_index():
    """NaN / Inf in a query or in a stored vector: the reference would surface an out-of-range distance as an error
    (distance.rs:58-105); the engine must at least stay bounded (every walk ends) and keep answering clean queries."""
    v = vs()
    n, dim = 5000, 64
    base = _dataset(n + 8, dim, 23)
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=64)
    ix.reserve(n + 16)
    ix.add_batch(np.arange(n, dtype=np.uint64), base[:n])
    good_k, good_d = ix.search(base[n], 10)
    for bad in (np.nan, np.inf, -np.inf):
        q = base[n + 1].copy()
        q[3] = bad
        keys, dist = ix.search(q, 10)                 # must return (whatever it returns) without hanging
        assert len(keys) <= 10
        kb, db, fb = ix.search_batch(np.stack([q, base[n]]), 10)
        assert np.array_equal(kb[1], good_k) and np.array_equal(db[1], good_d)   # the clean neighbour query is untouched
    poisoned = base[n + 2].copy()
    poisoned[0] = np.nan
    ix.add(n + 1, poisoned)                           # a stored NaN vector
    ix.add_batch(np.arange(n + 2, n + 6, dtype=np.uint64), base[n + 2:n + 6])
    assert ix.size() == n + 5
    k2, d2 = ix.search(base[n], 10)
    assert len(k2) == 10 and np.isfinite(d2[:9]).all()
    assert len(set(k2.tolist()) & set(good_k.tolist())) >= 8
    tk, td, _ = ix.exact_search_batch(base[n:n + 1], 10)
    assert len(set(tk[0].tolist()) & set(good_k.tolist())) >= 8


@pytest.mark.timeout(300)
@pytest.mark.parametrize("metric", ["cos", "ip", "l2sq"])
def test_nan_distances_rank_last_and_never_corrupt_a_walk(metric):
    """A NaN distance used to break the (distance, slot) order of the LDS lists: ranks collided, holes kept stale slot ids
    and the walk followed them out of bounds.  NaN now ranks as +inf.  Many non-finite queries, small (team kernel) and
    large batches, large k, exact search: nothing faults, clean queries keep their answers."""
    v = vs()
    n, dim = 20000, 24
    base = _dataset(n + 600, dim, 29)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric], expansion_search=200)
    ix.reserve(n + 64)
    poisoned = base[:n].copy()
    poisoned[::97, 5] = np.nan                        # ~200 stored vectors carry a NaN
    poisoned[50::97, 7] = np.inf
    ix.add_batch(np.arange(n, dtype=np.uint64), poisoned)
    q = base[n:n + 600].copy()
    q[::3, 1] = np.nan
    q[1::3, 2] = np.inf
    clean = np.arange(2, 600, 3)
    ref_k, ref_d, _ = ix.search_batch(q[clean], 50)
    for lo, hi in ((0, 600), (0, 7), (7, 64), (64, 300)):
        k, d, f = ix.search_batch(q[lo:hi], 50)
        assert (f <= 50).all() and all(k[i, :f[i]].max(initial=0) < n for i in range(hi - lo))
        sel = [i for i in range(lo, hi) if i % 3 == 2]
        pos = [clean.tolist().index(i) for i in sel]
        assert np.array_equal(k[[i - lo for i in sel]], ref_k[pos])
    ek, ed, ef_ = ix.exact_search_batch(q[:64], 50)
    assert (ef_ <= 50).all() and all(ek[i, :ef_[i]].max(initial=0) < n for i in range(64))
    bk, bd = ix.search(q[0], 2000)                     # a wide walk with NaNs in the table: bounded, no duplicates (the
    assert len(bk) <= 2000 and len(set(bk.tolist())) == len(bk)   # poisoned rows form a clique of their own, so it may end early)
    bk, bd = ix.search(q[0], 11000)                    # beyond the widest walk: exhaustive ranking with NaNs in the table
    assert len(bk) == 11000 and len(set(bk.tolist())) == 11000
    for i in range(0, 30):
        ix.search(q[i], 10)                            # one-query entry point (team kernel, zero-copy results)



def test_identical_vectors_stay_reachable():
    """12,000 copies of one vector: every distance ties.  With ties ordered by slot in the link kernel the same 32
    lowest slots stayed in every list and only 34 members were reachable; ties are now ordered pseudo-randomly per
    target, and a top-100 query finds 100 distinct members at distance 0."""
    v = vs()
    dim, n = 40, 12000
    same = np.tile(np.random.default_rng(3).standard_normal(dim).astype(np.float32), (n, 1))
    for metric in ("cos", "l2sq"):
        ix = v.HipUsearchIndex(dim, v.METRICS[metric], expansion_search=128)
        ix.reserve(n)
        ix.add_batch(np.arange(n, dtype=np.uint64), same)
        k, d, f = ix.search_batch(same[:200], 100)
        assert (f == 100).all() and all(len(set(r.tolist())) == 100 for r in k)
        assert np.abs(d).max() <= 1e-6


def test_heavily_duplicated_data_builds_as_well_as_the_cpu_algorithm():
    """100,000 vectors, every vector present 50 times: HNSW's heuristic fills a node's list with copies of its own vector, so
    the algorithm itself degrades (the CPU restatement reaches ~0.67 of the tied top-10).  What is left depends on the order
    among EQUAL distances in the build; with usearch's order (a new entry precedes the equal ones already listed; in a
    re-selected row later members first, the new link last) the batched GPU build matches the sequential CPU build
    (round 1, pseudo-random order: 0.46)."""
    v = vs()
    dim, copies = 64, 50
    rng = np.random.default_rng(5)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    uniq = 100000 // copies
    u = (rng.standard_normal((uniq, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((uniq, dim)).astype(np.float32))
    base = np.repeat(u, copies, axis=0)[rng.permutation(uniq * copies)]
    q = (rng.standard_normal((500, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((500, dim)).astype(np.float32))
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=128)
    ix.reserve(len(base))
    ix.add_batch(np.arange(len(base), dtype=np.uint64), base)
    k, d, f = ix.search_batch(q, 10)
    tk, td, tf = ix.exact_search_batch(q, 10)
    tied = lambda dd, ff: float(np.mean([(dd[i, :ff[i]] <= td[i, 9] * (1 + 1e-5) + 1e-7).sum() / 10 for i in range(len(q))]))
    gpu = tied(d, f)
    o = OracleIndex(dim, oracle.COS, 16, 128, 128)
    o.reserve(len(base))
    o.add_batch(np.arange(len(base), dtype=np.uint64), base, threads=1)
    ko, do, fo = o.search_batch(q, 10, threads=8)
    cpu = tied(do, fo)
    assert (f == 10).all()
    assert gpu >= cpu - 0.03 and gpu >= 0.6, (gpu, cpu)


@pytest.mark.parametrize("path", ["plane", "bf16x3"])
@pytest.mark.parametrize("metric,quant", [("cos", "f32"), ("ip", "f32"), ("cos", "f16")])
def test_exact_search_split_bf16_nomination_is_certified_or_falls_back(metric, quant, path):
    """Exact search on large float indexes nominates with bf16 MFMA tiles -- round 3: ONE product per score over a bf16 plane of
    the rows ("plane", the default); round 2: three products of split-bf16 pairs (VS_HNSW_EXACT=bf16x3) --, re-scores the
    nominees in f32 and certifies the answer; a query whose certificate fails (a crowd of equal scores at the cut) sends the
    batch on: plane -> split bf16 -> f32-input MFMA path.  Same ids as the f32 path either way; every case is provoked."""
    import os
    v = vs()
    fast_env = None if path == "plane" else "bf16x3"
    batches, fallbacks = ("plane_batches", "plane_fallbacks") if path == "plane" else ("block_batches", "block_fallbacks")
    n, dim, k = 80000, 96, 10
    data = _dataset(n + 64, dim, 13)
    base, q = data[:n].copy(), data[n:]
    if metric == "ip":
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    def build(env):
        old = os.environ.pop("VS_HNSW_EXACT", None)
        if env:
            os.environ["VS_HNSW_EXACT"] = env
        try:
            ix = v.HipUsearchIndex(dim, v.METRICS[metric], quantization=v.SCALARS[quant])
        finally:
            os.environ.pop("VS_HNSW_EXACT", None)
            if old:
                os.environ["VS_HNSW_EXACT"] = old
        ix.reserve(n)
        ix.add_batch(np.arange(n, dtype=np.uint64), base)
        return ix
    fast, ref = build(fast_env), build("f32")
    fk, fd, ff = fast.exact_search_batch(q, k)
    rk, rd, rf = ref.exact_search_batch(q, k)
    st, rst = fast.exact_stats(), ref.exact_stats()
    assert st[batches] >= 1 and st[fallbacks] == 0 and rst["block_batches"] == 0 and rst["plane_batches"] == 0
    if path == "plane":
        assert st["block_batches"] == 0  # nothing was handed on
    assert (ff == k).all() and np.allclose(fd, rd, rtol=1e-5, atol=2e-5)
    assert np.mean([fk[i].tolist() == rk[i].tolist() for i in range(len(q))]) >= 0.97      # f32 near-ties may swap
    assert all(set(fk[i].tolist()) == set(rk[i].tolist()) or np.isclose(fd[i, -1], rd[i, -1], rtol=1e-5, atol=2e-5) for i in range(len(q)))
    # 300 copies of one vector right at the query: the 64 nominees all tie with 236 rows outside -> no certificate -> f32 path
    # (round 6: the 8-bit plane's lists hold 512 nominees and would certify 300 copies -- 700 of them for the plane stages)
    copies = 700 if path == "plane" else 300
    base[1000:1000 + copies] = q[0] / (np.linalg.norm(q[0]) if metric == "ip" else 1.0)
    fast2, ref2 = build(fast_env), build("f32")
    fk, fd, ff = fast2.exact_search_batch(q[:4], k)
    rk, rd, rf = ref2.exact_search_batch(q[:4], k)
    assert fast2.exact_stats()[fallbacks] == 1 and fast2.exact_stats()["block_fallbacks"] == 1  # (the planes' 512 / 256 nominees tie with rows outside, too)
    if path == "plane":
        assert fast2.exact_stats()["plane8_batches"] == 1 and fast2.exact_stats()["plane8_fallbacks"] == 1  # 8-bit plane -> bf16 plane -> split bf16 -> f32
    assert np.array_equal(fk, rk) and np.array_equal(fd, rd)          # the very same kernels answered
    assert set(fk[0].tolist()) <= set(range(1000, 1000 + copies))
    # Row blocks after the first pass on only the scores at or below each query's threshold.  (a) members of the second block
    # removed: never results; (b) rows stored farthest-first from one query: its second block beats everything seen before,
    # the per-query buffer overflows and the batch is answered by the f32 path -- same ids either way.
    base = data[:n].copy()
    if metric == "ip":
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    fast3, ref3 = build(fast_env), build("f32")
    gone = list(range(65536, n, 3))
    for key in gone:
        assert fast3.remove(key) and ref3.remove(key)
    fk, fd, ff = fast3.exact_search_batch(q, k)
    rk, rd, rf = ref3.exact_search_batch(q, k)
    assert fast3.exact_stats()[fallbacks] == 0 and not (set(fk.ravel().tolist()) & set(gone))
    assert np.allclose(fd, rd, rtol=1e-5, atol=2e-5)
    assert all(set(fk[i].tolist()) == set(rk[i].tolist()) or np.isclose(fd[i, -1], rd[i, -1], rtol=1e-5, atol=2e-5) for i in range(len(q)))
    qn = q[0] / np.linalg.norm(q[0])
    order = np.argsort(-(1.0 - (base @ qn) / (np.linalg.norm(base, axis=1) if metric == "cos" else 1.0)), kind="stable")
    base = base[order]
    fast4, ref4 = build(fast_env), build("f32")
    fk, fd, ff = fast4.exact_search_batch(q[:8], k)
    rk, rd, rf = ref4.exact_search_batch(q[:8], k)
    assert fast4.exact_stats()[fallbacks] == 1
    assert np.array_equal(fk, rk) and np.array_equal(fd, rd)
    assert fk[0].min() >= n - 64                                       # the nearest rows are the last ones stored


@pytest.mark.timeout(900)
@pytest.mark.parametrize("metric,quant,dim,k", [("cos", "f32", 128, 10), ("ip", "f32", 200, 64), ("cos", "bf16", 96, 10), ("ip", "f16", 768, 10)])
def test_one_product_plane_search_equals_the_f32_path(metric, quant, dim, k):
    """The bf16-plane pass at a size where its persistent tile kernel runs several chunks (300k rows: 64k through the score
    block, then 64k..300k thresholded), more than 256 queries (two query blocks), adds after the first search (the plane is
    extended), a removed member and a re-used slot (the plane is rebuilt): every batch certified, ids == the f32 path's."""
    import os
    v = vs()
    n, nq = 300_000, 300
    data = _dataset(n + nq + 2000, dim, 29)
    base, q, extra = data[:n].copy(), data[n:n + nq], data[n + nq:]
    if metric == "ip":
        base /= np.linalg.norm(base, axis=1, keepdims=True)
        extra = extra / np.linalg.norm(extra, axis=1, keepdims=True)

    def build(env):
        old = os.environ.pop("VS_HNSW_EXACT", None)
        if env:
            os.environ["VS_HNSW_EXACT"] = env
        try:
            ix = v.HipUsearchIndex(dim, v.METRICS[metric], quantization=v.SCALARS[quant])
        finally:
            os.environ.pop("VS_HNSW_EXACT", None)
            if old:
                os.environ["VS_HNSW_EXACT"] = old
        ix.reserve(n + len(extra))
        ix.add_batch(np.arange(n, dtype=np.uint64), base)
        return ix

    fast, ref = build(None), build("f32")

    def same(queries):
        fk, fd, ff = fast.exact_search_batch(queries, k)
        rk, rd, rf = ref.exact_search_batch(queries, k)
        assert (ff == rf).all() and np.allclose(fd, rd, rtol=1e-5, atol=2e-5)
        assert np.mean([fk[i].tolist() == rk[i].tolist() for i in range(len(queries))]) >= 0.97  # f32 near-ties may swap
        assert all(set(fk[i].tolist()) == set(rk[i].tolist()) or np.isclose(fd[i, -1], rd[i, -1], rtol=1e-5, atol=2e-5) for i in range(len(queries)))
        return fk

    same(q)
    st = fast.exact_stats()
    assert st["plane_batches"] == 1 and st["plane_fallbacks"] == 0 and st["block_batches"] == 0
    assert fast.memory_info()["bytes"] > ref.memory_info()["bytes"]  # the plane is accounted for
    for ix in (fast, ref):   # rows added after the plane was built, one of them the nearest neighbour of query 0
        e = extra.copy()
        e[0] = q[0] / (np.linalg.norm(q[0]) if metric == "ip" else 1.0)
        ix.add_batch(np.arange(n, n + len(e), dtype=np.uint64), e)
    fk = same(q)
    assert fk[0][0] == n
    for ix in (fast, ref):   # a removed member is never a result; its slot is re-used by a row that must be found
        assert ix.remove(n)
        ix.add(1 << 40, q[1] / (np.linalg.norm(q[1]) if metric == "ip" else 1.0))
    fk = same(q)
    assert n not in fk and fk[1][0] == 1 << 40
    st = fast.exact_stats()
    assert st["plane_batches"] == 3 and st["plane_fallbacks"] == 0
