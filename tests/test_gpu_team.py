"""Team search kernel (TEAM wavefronts per query, used for batches too small to fill the chip): it must return
exactly what the one-wavefront-per-query kernel returns -- same keys, same distances bit for bit, same evaluation
and hop counts -- because wave 0 makes every decision and the helpers only evaluate distances with the same code.
Both are compared with the oracle (the usearch restatement) on the same graph as well."""
import numpy as np
import pytest

from oracle import OracleIndex

pytestmark = pytest.mark.gpu

TEAM_ALWAYS, TEAM_NEVER = 4, 8


def vs():
    import vector_store_amd as v
    return v


def _data(n, dim, seed):
    rng = np.random.default_rng(seed)
    r = min(16, dim)
    w = rng.standard_normal((r, dim)).astype(np.float32) / np.sqrt(r)
    return (rng.standard_normal((n, r)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


def _pair(v, dim, metric, kind, base, ef):
    """The same graph behind both kernels: built once, imported into the second handle."""
    a = v.HipUsearchIndex(dim, metric, expansion_search=ef, quantization=v.SCALARS[kind], _stress=TEAM_NEVER)
    a.reserve(len(base))
    a.add_batch(np.arange(len(base), dtype=np.uint64), base)
    b = v.HipUsearchIndex(dim, metric, expansion_search=ef, quantization=v.SCALARS[kind], _stress=TEAM_ALWAYS)
    b.import_graph(a.export_graph())
    return a, b


@pytest.mark.parametrize("kind,metric,dim", [
    ("f32", "cos", 768), ("f32", "l2sq", 96), ("f32", "ip", 33), ("f32", "cos", 1536), ("f32", "l2sq", 2048),
    ("f16", "cos", 768), ("bf16", "l2sq", 200), ("i8", "cos", 768), ("i8", "l2sq", 64), ("b1", "hamming", 1024),
])
def test_team_kernel_equals_single_wave_kernel(kind, metric, dim):
    v = vs()
    m = v.METRICS[metric]
    n = 6000
    base = _data(n, dim, 5)
    q = _data(300, dim, 6)
    for ef in (64, 200, 300):  # 300: the 512-entry fused-list team kernel; i8 / b1: the team forms of the usearch-order walk
        a, b = _pair(v, dim, m, kind, base, ef)
        a.stats(reset=True), b.stats(reset=True)
        ka, da, fa = a.search_batch(q, 10)
        kb, db, fb = b.search_batch(q, 10)
        assert np.array_equal(fa, fb)
        assert np.array_equal(ka, kb)
        assert np.array_equal(da.view(np.uint32), db.view(np.uint32))  # bit for bit
        sa, sb = a.stats(), b.stats()
        assert sa["search_evals"] == sb["search_evals"] and sa["search_hops"] == sb["search_hops"]
        assert sb["visited_overflow"] == 0


def test_team_kernel_equals_oracle_on_same_graph():
    v = vs()
    n, dim = 5000, 128
    base, q = _data(n, dim, 11), _data(200, dim, 12)
    a, b = _pair(v, dim, v.COS, "f32", base, 128)
    o = OracleIndex(dim, 0, 16, 128, 128)
    o.import_graph(a.export_graph())
    kb, db, _ = b.search_batch(q, 10)
    same = 0
    for i in range(len(q)):
        ko, do = o.search(q[i], 10)
        same += int(np.array_equal(ko, kb[i][: len(ko)]))
        assert np.allclose(do, db[i][: len(do)], rtol=1e-5, atol=1e-6)
    assert same >= len(q) - 2  # last-bit distance differences may swap near-ties (same bound as the single-wave test)


def test_team_kernel_with_removed_members_and_tiny_batches():
    v = vs()
    n, dim = 4000, 64
    base, q = _data(n, dim, 21), _data(64, dim, 22)
    a, b = _pair(v, dim, v.L2SQ, "f32", base, 96)
    for ix in (a, b):
        for key in range(0, n, 3):
            assert ix.remove(key)
    for nq in (1, 2, 7, 64):
        ka, da, fa = a.search_batch(q[:nq], 10)
        kb, db, fb = b.search_batch(q[:nq], 10)
        assert np.array_equal(ka, kb) and np.array_equal(da.view(np.uint32), db.view(np.uint32)) and np.array_equal(fa, fb)
        assert not np.any(kb % 3 == 0)
    # single-vector entry point (SearchService) goes through the same kernel choice
    k1, d1 = b.search(q[0], 10)
    k2, d2 = a.search(q[0], 10)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)


def test_default_policy_small_batches_take_the_team_kernel_and_match():
    v = vs()
    n, dim = 5000, 768
    base, q = _data(n, dim, 31), _data(1024, dim, 32)
    a = v.HipUsearchIndex(dim, v.COS, expansion_search=128)  # default: team for nq <= 256
    a.reserve(n)
    a.add_batch(np.arange(n, dtype=np.uint64), base)
    big_k, big_d, _ = a.search_batch(q, 10)            # 1024 queries: one wave per query
    for lo in range(0, 1024, 256):                     # 4 x 256 queries: team kernel
        k, d, _ = a.search_batch(q[lo:lo + 256], 10)
        assert np.array_equal(k, big_k[lo:lo + 256]) and np.array_equal(d, big_d[lo:lo + 256])


def test_team_insert_builds_the_same_graph():
    """Sub-batches of at most 256 nodes take the team insert kernel (always, with the test hook): same links, same
    levels, same entry point as the one-wave-per-node kernel -- wave 0 decides, the helpers only measure."""
    v = vs()
    for kind, metric, dim in (("f32", "cos", 768), ("f32", "l2sq", 48), ("f16", "ip", 200), ("i8", "cos", 256), ("b1", "hamming", 512)):
        n = 3000
        base = _data(n, dim, 41)
        graphs = []
        for stress in (TEAM_NEVER, TEAM_ALWAYS):
            ix = v.HipUsearchIndex(dim, v.METRICS[metric], quantization=v.SCALARS[kind], _stress=stress)
            ix.reserve(n)
            ix.add_batch(np.arange(n, dtype=np.uint64), base)
            graphs.append(ix.export_graph())
        a, b = graphs
        assert a["max_level"] == b["max_level"] and a["entry_slot"] == b["entry_slot"]
        assert np.array_equal(a["levels"], b["levels"])
        assert np.array_equal(a["adj0"], b["adj0"]), (kind, metric, dim)
        assert np.array_equal(a["upper"], b["upper"])


def test_link_kernel_with_accepted_rows_in_lds_builds_the_same_graph(monkeypatch):
    """The link kernel re-selects a full neighbour list with the accepted rows cached in LDS (one HBM read per candidate);
    VS_HNSW_LINK_CACHE=0 is the plain refine_ from HBM, =3 forces the mixed path (3 rows cached, the rest from HBM)."""
    v = vs()
    for kind, metric, dim in (("f32", "cos", 768), ("f32", "l2sq", 100), ("f32", "ip", 1536), ("f16", "cos", 384), ("i8", "l2sq", 128),
                              ("b1", "hamming", 256)):
        n = 6000
        base = _data(n, dim, 51)
        graphs = []
        for cache in ("0", None, "3"):
            if cache is None:
                monkeypatch.delenv("VS_HNSW_LINK_CACHE", raising=False)
            else:
                monkeypatch.setenv("VS_HNSW_LINK_CACHE", cache)
            ix = v.HipUsearchIndex(dim, v.METRICS[metric], quantization=v.SCALARS[kind])
            ix.reserve(n)
            ix.add_batch(np.arange(n, dtype=np.uint64), base)
            graphs.append((ix.export_graph(), ix.stats()))
        (a, sa), (b, sb), (c, sc) = graphs
        assert sa["link_evals"] > 0 and sa["link_evals"] == sb["link_evals"] == sc["link_evals"]
        for g in (b, c):
            assert np.array_equal(a["adj0"], g["adj0"]), (kind, metric, dim)
            assert np.array_equal(a["upper"], g["upper"])
            assert a["entry_slot"] == g["entry_slot"] and a["max_level"] == g["max_level"]


def test_four_wave_team_tier_equals_single_wave_kernel(monkeypatch):
    """257..768 queries on the device take teams of 4 waves (two per CU); VS_HNSW_TEAM=mid forces that kernel."""
    v = vs()
    n, dim = 6000, 768
    base, q = _data(n, dim, 71), _data(512, dim, 72)
    monkeypatch.setenv("VS_HNSW_TEAM", "never")
    a = v.HipUsearchIndex(dim, v.COS, expansion_search=150)
    a.reserve(n)
    a.add_batch(np.arange(n, dtype=np.uint64), base)
    monkeypatch.setenv("VS_HNSW_TEAM", "mid")
    b = v.HipUsearchIndex(dim, v.COS, expansion_search=150)
    b.import_graph(a.export_graph())
    monkeypatch.delenv("VS_HNSW_TEAM")
    c = v.HipUsearchIndex(dim, v.COS, expansion_search=150)   # default policy: 400 queries -> the 4-wave tier
    c.import_graph(a.export_graph())
    for ef in (64, 150):
        for ix in (a, b, c):
            ix.set_expansion_search(ef)
        ka, da, fa = a.search_batch(q[:400], 10)
        for other in (b, c):
            ko, do, fo = other.search_batch(q[:400], 10)
            assert np.array_equal(ka, ko) and np.array_equal(da.view(np.uint32), do.view(np.uint32)) and np.array_equal(fa, fo)
