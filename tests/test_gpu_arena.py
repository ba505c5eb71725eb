"""HBM arenas that grow in place (virtual range + mapped chunks): `reserve` -- the reference grows by +1,000,000
whenever free slots run low (usearch.rs:442, :908-921) -- must keep every vector and link, must not need a second
copy of the index in HBM, and must hand memory back when the index is dropped."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def vs():
    import vector_store_amd as v
    return v


def _data(n, dim, seed):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


def test_growth_in_steps_keeps_the_index_and_copies_almost_nothing():
    v = vs()
    n, dim, step = 60_000, 768, 30_000
    base, q = _data(n, dim, 1), _data(200, dim, 2)
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(step)                                   # 92 MB of vectors: above the in-place threshold
    m0 = ix.memory_info()
    assert m0["in_place_bytes"] >= step * dim * 4, m0
    ix.add_batch(np.arange(step, dtype=np.uint64), base[:step])
    before = ix.search_batch(q, 10)
    ix.reserve(2 * step)                               # grow: one more chunk mapped behind the same range
    after = ix.search_batch(q, 10)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    ix.add_batch(np.arange(step, n, dtype=np.uint64), base[step:])
    for cap in (3 * step, 4 * step, 10 * step):        # 10 x: the virtual range itself is outgrown and remapped
        ix.reserve(cap)
        assert ix.capacity() == cap
    m1 = ix.memory_info()
    assert m1["chunks"] >= 4 and m1["in_place_bytes"] >= 10 * step * dim * 4
    # only the small arrays (keys, links, levels: 148 B per slot, below the threshold) were ever copied; a copying
    # regrow would also have moved (30 + 60 + 90 + 120)k x 3 KiB = 921 MB of vectors
    assert m1["copied_bytes"] - m0["copied_bytes"] < 0.1 * 300_000 * dim * 4, (m0, m1)
    k, d, f = ix.search_batch(base[:500], 1)
    assert np.array_equal(k[:, 0], np.arange(500, dtype=np.uint64))
    # same graph as an index that was reserved once
    ref = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ref.reserve(n)
    ref.add_batch(np.arange(step, dtype=np.uint64), base[:step])
    ref.add_batch(np.arange(step, n, dtype=np.uint64), base[step:])
    a, b = ix.search_batch(q, 10), ref.search_batch(q, 10)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # shrink to the members: whole chunks go back
    ix.reserve(n)
    assert ix.capacity() == n and ix.memory_info()["bytes"] < m1["bytes"]
    a2 = ix.search_batch(q, 10)
    assert np.array_equal(a[0], a2[0])


def test_index_larger_than_half_of_hbm_can_still_grow():
    """12M x 2048 f32 = 98 GB of vectors, then 24M = 197 GB: a copy-based regrow would need both at once (295 GB)."""
    import torch
    v = vs()
    free, total = torch.cuda.mem_get_info()
    if free < 240 * 2**30:
        pytest.skip(f"needs 240 GiB of free HBM, have {free / 2**30:.0f}")
    dim = 2048
    ix = v.HipUsearchIndex(dim, v.L2SQ)
    ix.reserve(12_000_000)
    x = _data(2000, dim, 3)
    ix.add_batch(np.arange(2000, dtype=np.uint64), x)
    ix.reserve(24_000_000)
    assert ix.capacity() == 24_000_000
    mi = ix.memory_info()
    assert mi["in_place_bytes"] > 190 * 10**9
    k, d, _ = ix.search_batch(x[:100], 1)
    assert np.array_equal(k[:, 0], np.arange(100, dtype=np.uint64)) and np.all(d[:, 0] == 0)
    with pytest.raises(v.VsError):                     # 40M x 8 KiB = 328 GB: refused, capacity unchanged
        ix.reserve(40_000_000)
    assert ix.capacity() == 24_000_000
    del ix
    import gc
    gc.collect()
    free2, _ = torch.cuda.mem_get_info()
    assert free2 > free - 2**30                        # everything was handed back
