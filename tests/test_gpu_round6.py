"""Round 6: edge cases of the 8-bit plane of the exact block search (configs[4]) -- the paths around it are
covered by tests/test_gpu_parity.py (every hand-over provoked) and tests/test_gpu_full_size.py (the float64 check at 10M)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dataset(n, dim, seed):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


def _build(v, dim, metric, base, env=None):
    old = os.environ.pop("VS_HNSW_EXACT", None)
    if env:
        os.environ["VS_HNSW_EXACT"] = env
    try:
        ix = v.HipUsearchIndex(dim, v.METRICS[metric])
    finally:
        os.environ.pop("VS_HNSW_EXACT", None)
        if old:
            os.environ["VS_HNSW_EXACT"] = old
    ix.reserve(len(base))
    ix.add_batch(np.arange(len(base), dtype=np.uint64), base)
    return ix


def _same(fast, ref, q, k):
    fk, fd, ff = fast.exact_search_batch(q, k)
    rk, rd, rf = ref.exact_search_batch(q, k)
    assert (ff == rf).all() and np.allclose(fd, rd, rtol=1e-5, atol=2e-5)
    assert all(set(fk[i].tolist()) == set(rk[i].tolist()) or np.isclose(fd[i, -1], rd[i, -1], rtol=1e-5, atol=2e-5) for i in range(len(q)))


@pytest.mark.parametrize("metric", ["cos", "ip"])
def test_the_8bit_plane_reports_its_band_and_serves_ordinary_rows(metric):
    import vector_store_amd as v
    n, dim, k = 70_000, 200, 10   # (dim 200: the plane's rows are padded to 256 bytes, two 128-element K steps)
    data = _dataset(n + 40, dim, 3)
    base, q = data[:n].copy(), data[n:]
    if metric == "ip":
        base /= np.linalg.norm(base, axis=1, keepdims=True)
    fast, ref = _build(v, dim, metric, base), _build(v, dim, metric, base, "f32")
    _same(fast, ref, q, k)
    st = fast.exact_stats()
    assert st["plane8_batches"] == 1 and st["plane8_fallbacks"] == 0 and st["plane8_rows"] == n
    assert 0.002 < st["plane8_rho"] < 0.05, st["plane8_rho"]   # int8 with a per-row step: ~1 % of a row's norm
    assert st["plane_batches"] == 1 and st["plane_fallbacks"] == 0 and st["block_batches"] == 0  # (the plane stages, counted once)
    # VS_HNSW_EXACT=bf16: the round-5 order of the stages (A/B)
    bf = _build(v, dim, metric, base, "bf16")
    _same(bf, ref, q, k)
    st = bf.exact_stats()
    assert st["plane8_batches"] == 0 and st["plane_batches"] == 1 and st["plane_fallbacks"] == 0


def test_the_8bit_plane_with_zero_rows_a_zero_query_and_a_wild_row():
    """Zero rows quantise exactly (a zero-norm cosine QUERY has no meaningful approximate scores: its batch goes on to the other paths,
    as on the bf16 plane); a row with one component 10^12 times the others is its own int8 image up to 10^-12 of its norm.  Answers
    equal the f32 path's in every case."""
    import vector_store_amd as v
    n, dim, k = 70_000, 96, 10
    data = _dataset(n + 16, dim, 5)
    base, q = data[:n].copy(), data[n:].copy()
    base[7] = 0.0
    base[1234] = 0.0
    fast, ref = _build(v, dim, "cos", base), _build(v, dim, "cos", base, "f32")
    _same(fast, ref, q, k)
    st = fast.exact_stats()
    assert st["plane8_batches"] == 1 and st["plane8_fallbacks"] == 0
    q0 = q.copy()
    q0[3] = 0.0
    _same(fast, ref, q0, k)
    st = fast.exact_stats()
    assert st["plane8_batches"] == 2 and st["plane8_fallbacks"] == 1   # the zero query's batch was handed on
    wild = base.copy()
    wild[500, 17] = 1e12
    fast, ref = _build(v, dim, "cos", wild), _build(v, dim, "cos", wild, "f32")
    _same(fast, ref, q, k)
    st = fast.exact_stats()
    assert st["plane8_batches"] == 1 and st["plane8_fallbacks"] == 0 and st["plane8_rho"] < 0.05, st
