"""The reference passes `Dimensions`, `ExpansionAdd` (and `Connectivity`) straight from the CQL index options with no upper
bound (reference crates/vector-store/src/lib.rs:378,392,412, db.rs:921-930).  Round 3 lifts two of the engine's limits:
stored vectors up to 16 KiB (3072-d f32 = text-embedding-3-large, 4096-d f32) and construction beams up to 512.
Same parity bar as everywhere: ids identical to the oracle's on the same graph, near-ties asserted one by one."""
import numpy as np
import pytest

import oracle
from oracle import OracleIndex
from tests.parity_util import assert_same_results, lattice

pytestmark = pytest.mark.gpu


def vs():
    import vector_store_amd as v
    return v


def _dataset(n, dim, seed):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("metric,kind,dim", [("cos", "f32", 3072), ("l2sq", "f32", 4096), ("ip", "f32", 2500), ("cos", "f16", 6144),
                                             ("cos", "i8", 12288)])
def test_rows_of_up_to_16_kib(metric, kind, dim):
    """8 KiB < row <= 16 KiB: 12 or 16 wave-loads per row (I = 12 / 16).  GPU build, then search parity with the oracle on the
    SAME graph, the exact search against float64 numpy, and one vector per call through the single-query entry points."""
    v = vs()
    n, k = 2500, 10
    data = _dataset(n + 48, dim, 3 + dim)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    ix = v.HipUsearchIndex(dim, v.METRICS[metric], expansion_search=96, quantization=v.SCALARS[kind])
    assert ix.bytes_per_vector() > 8192 or dim == 2500
    ix.reserve(n)
    ix.add_batch(np.arange(n - 100, dtype=np.uint64), base[: n - 100])
    for i in range(n - 100, n):  # the last hundred one per FFI call
        ix.add(i, base[i])
    assert ix.size() == n
    o = OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[kind])
    o.import_graph(ix.export_graph())
    exact = kind == "i8"
    ties = 0
    for ef in (96, 300):
        o.set_expansion_search(ef)
        ix.set_expansion_search(ef)
        gk, gd, gf = ix.search_batch(q, k)
        for i in range(len(q)):
            ok_, od_ = o.search(q[i], k)
            ties += assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_, lambda key, i=i: o.distance_to_slot(q[i], int(key)),
                                        exact=exact, what=(metric, kind, dim, ef, i))
    assert ties <= 6, ties
    assert ix.stats()["visited_overflow"] == 0
    k1, d1 = ix.search(q[0], k)
    assert k1.tolist() == gk[0].tolist()
    if kind == "f32":
        tk, td, _ = ix.exact_search_batch(q[:16], k)
        b64, q64 = base.astype(np.float64), q[:16].astype(np.float64)
        if metric == "l2sq":
            d = ((q64[:, None, :] - b64[None, :, :]) ** 2).sum(-1)
        elif metric == "ip":
            d = 1.0 - q64 @ b64.T
        else:
            d = 1.0 - (q64 @ b64.T) / (np.linalg.norm(q64, axis=1)[:, None] * np.linalg.norm(b64, axis=1)[None, :])
        want = np.argsort(d, axis=1, kind="stable")[:, :k]
        agree = np.mean([len(set(want[i].tolist()) & set(tk[i].tolist())) / k for i in range(16)])
        assert agree >= 0.99, agree
        recall = np.mean([len(set(want[i].tolist()) & set(gk[i].tolist())) / k for i in range(16)])
        assert recall >= 0.9, recall


def test_rows_above_16_kib_are_refused_with_a_status():
    v = vs()
    with pytest.raises(v.VsError) as e:
        v.HipUsearchIndex(4097, v.COS)
    assert e.value.code == -7 and "16 KiB" in e.value.msg
    v.HipUsearchIndex(4096, v.COS)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("ef_add", [400, 512])
def test_construction_beams_up_to_512(ef_add):
    """expansion_add of 257..512 (`construction_beam_width`): the GPU-built graph is at least as good as the CPU
    restatement's at the same setting, and searches on it equal the oracle's on the same graph."""
    v = vs()
    n, dim, k = 20000, 48, 10
    data = _dataset(n + 200, dim, 17)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, 16, ef_add, 64)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    st = ix.stats()
    assert st["visited_overflow"] == 0 and st["added"] == n - 1 and ix.size() == n  # (the first member becomes the entry point without a walk)
    tk, _, _ = ix.exact_search_batch(q, k)
    gk, gd, gf = ix.search_batch(q, k)
    recall_gpu = np.mean([len(set(tk[i].tolist()) & set(gk[i].tolist())) / k for i in range(len(q))])
    o_built = OracleIndex(dim, oracle.COS, 16, ef_add, 64)
    o_built.reserve(n)
    o_built.add_batch(np.arange(n, dtype=np.uint64), base, threads=1)
    recall_cpu = np.mean([len(set(tk[i].tolist()) & set(o_built.search(q[i], k)[0].tolist())) / k for i in range(len(q))])
    assert recall_gpu >= recall_cpu - 0.03, (recall_gpu, recall_cpu)
    # a wider construction beam must not be worse than the default one
    dflt = v.HipUsearchIndex(dim, v.COS, 16, 128, 64)
    dflt.reserve(n)
    dflt.add_batch(np.arange(n, dtype=np.uint64), base)
    dk, _, _ = dflt.search_batch(q, k)
    recall_128 = np.mean([len(set(tk[i].tolist()) & set(dk[i].tolist())) / k for i in range(len(q))])
    assert recall_gpu >= recall_128 - 0.01, (recall_gpu, recall_128)
    o = OracleIndex(dim, oracle.COS, 16, ef_add, 64)
    o.import_graph(ix.export_graph())
    ties = 0
    for i in range(len(q)):
        ok_, od_ = o.search(q[i], k)
        ties += assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=(ef_add, i))
    assert ties <= 4, ties


@pytest.mark.timeout(900)
def test_sequential_adds_with_a_construction_beam_of_400_build_the_oracle_graph():
    """One add per call = the sequential usearch algorithm; on exactly representable data the graph built with
    expansion_add = 400 equals the CPU restatement's row for row (rows that involve an exact tie excepted)."""
    v = vs()
    n, dim = 1200, 8
    base = lattice(n, dim, 9, span=500)
    o = OracleIndex(dim, oracle.L2SQ, 16, 400, 64)
    o.reserve(n)
    ix = v.HipUsearchIndex(dim, v.L2SQ, 16, 400, 64)
    ix.reserve(n)
    for i in range(n):
        o.add(i, base[i])
        ix.add(i, base[i])
        assert ix.size() == i + 1
    go, gg = o.export_graph(), ix.export_graph()
    assert (go["levels"] == gg["levels"]).all() and go["entry_slot"] == gg["entry_slot"]
    differ = [s_ for s_ in range(n) if set(go["adj0"][s_].tolist()) != set(gg["adj0"][s_].tolist())]
    for s_ in differ[:20]:
        d = ((base - base[s_]) ** 2).sum(1)
        cand = set(go["adj0"][s_].tolist()) ^ set(gg["adj0"][s_].tolist())
        cand.discard(0xFFFFFFFF)
        assert any(np.sum(d == d[c]) > 1 for c in cand), (s_, "rows differ without an exact tie")
    assert len(differ) <= n // 100, len(differ)


def test_construction_beam_above_512_is_refused_with_a_status():
    v = vs()
    with pytest.raises(v.VsError) as e:
        v.HipUsearchIndex(16, v.COS, 16, 513, 64)
    assert e.value.code == -7


@pytest.mark.timeout(900)
@pytest.mark.parametrize("M", [48, 64])
def test_connectivity_up_to_64(M):
    """`maximum_node_connections` above 32: level-0 rows of up to 128 ids are taken two per lane.  (1) a graph built by the
    oracle at that connectivity, imported: the fused-list search, the usearch-order walk and an i8 index return the oracle's ids;
    (2) one add per call on exactly representable data builds the oracle's graph row for row; (3) the batched GPU build is as good
    as the CPU restatement's."""
    v = vs()
    dim, n, k = 32, 6000, 10
    data = _dataset(n + 100, dim, 5)
    base, q = data[:n], data[n:]
    o = OracleIndex(dim, oracle.COS, M, 128, 64)
    o.reserve(n)
    o.add_batch(np.arange(n, dtype=np.uint64), base, threads=1)
    g = o.export_graph()
    assert g["adj0"].shape[1] == 2 * M and (g["adj0"][:, 64:] != 0xFFFFFFFF).any()  # rows really use the second half
    for stress in (0, 16):  # fused list; usearch-order walk
        ix = v.HipUsearchIndex(dim, v.COS, M, 128, 64, _stress=stress)
        ix.import_graph(g)
        ties = 0
        for ef in (64, 200):
            o.set_expansion_search(ef)
            ix.set_expansion_search(ef)
            gk, gd, gf = ix.search_batch(q, k)
            for i in range(len(q)):
                ok_, od_ = o.search(q[i], k)
                ties += assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=(M, stress, ef, i))
        assert ties <= 6, ties
    # i8 storage (always the walk): bit-identical
    o8 = OracleIndex(dim, oracle.COS, M, 128, 100, quantization=oracle.I8)
    o8.reserve(n)
    o8.add_batch(np.arange(n, dtype=np.uint64), base, threads=1)
    i8 = v.HipUsearchIndex(dim, v.COS, M, 128, 100, quantization=v.I8)
    i8.import_graph(o8.export_graph())
    gk, gd, gf = i8.search_batch(q, k)
    for i in range(len(q)):
        ok_, od_ = o8.search(q[i], k)
        assert_same_results(gk[i, : gf[i]], gd[i, : gf[i]], ok_, od_, exact=True, what=("i8", M, i))
    # (2) sequential adds on a lattice: the graph itself
    m2 = 900
    lat = lattice(m2, 8, 9, span=500)
    os_ = OracleIndex(8, oracle.L2SQ, M, 128, 64)
    os_.reserve(m2)
    gs = v.HipUsearchIndex(8, v.L2SQ, M, 128, 64)
    gs.reserve(m2)
    for i in range(m2):
        os_.add(i, lat[i])
        gs.add(i, lat[i])
        assert gs.size() == i + 1
    go, gg = os_.export_graph(), gs.export_graph()
    assert (go["levels"] == gg["levels"]).all() and go["entry_slot"] == gg["entry_slot"]
    differ = [s_ for s_ in range(m2) if set(go["adj0"][s_].tolist()) != set(gg["adj0"][s_].tolist())]
    for s_ in differ[:20]:
        d = ((lat - lat[s_]) ** 2).sum(1)
        cand = set(go["adj0"][s_].tolist()) ^ set(gg["adj0"][s_].tolist())
        cand.discard(0xFFFFFFFF)
        assert any(np.sum(d == d[c]) > 1 for c in cand), (s_, "rows differ without an exact tie")
    assert len(differ) <= m2 // 50, len(differ)
    # (3) batched GPU build
    gb = v.HipUsearchIndex(dim, v.COS, M, 128, 64)
    gb.reserve(n)
    gb.add_batch(np.arange(n, dtype=np.uint64), base)
    tk, _, _ = gb.exact_search_batch(q, k)
    bk, _, _ = gb.search_batch(q, k)
    o.set_expansion_search(64)
    r_gpu = np.mean([len(set(tk[i].tolist()) & set(bk[i].tolist())) / k for i in range(len(q))])
    r_cpu = np.mean([len(set(tk[i].tolist()) & set(o.search(q[i], k)[0].tolist())) / k for i in range(len(q))])
    assert r_gpu >= r_cpu - 0.03, (r_gpu, r_cpu)
    info = gb.graph_info()
    assert info["connectivity"] == M and info["connectivity_base"] == 2 * M


def test_connectivity_above_64_is_refused_with_a_status():
    v = vs()
    with pytest.raises(v.VsError) as e:
        v.HipUsearchIndex(16, v.COS, 65, 128, 64)
    assert e.value.code == -7
