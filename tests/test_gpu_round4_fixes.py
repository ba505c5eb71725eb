"""Regression tests for the advisor's round-3 findings that a test can provoke (the others are structural: see DESIGN.md, status block)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_the_bf16_plane_survives_a_copying_resize_with_a_partial_last_tile():
    """engine.hip ensure_plane(): the plane is tile-major (a 256-row tile is k-steps blocks of 256 x 128 B), so "the bytes of the
    first r rows" only exist for whole tiles.  A reserve() that grows the plane by COPY (below the 64 MiB in-place threshold) with
    slots % 256 != 0 and no new rows used to leave the later k-steps of the last partial tile uninitialised: approximate scores of up
    to 255 rows were garbage and the exact search could drop a true top-k row without its certificate noticing.  Exact search before
    and after such a reserve, against the f32 VALU path on the same rows."""
    import vector_store_amd as vs
    n, dim, k = 70_001, 128, 10          # 70,001 % 256 = 113 rows in the last tile; plane 70k x 128 x 2 B = 18 MB: a copying arena
    rng = np.random.default_rng(8)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    base[-113:] *= 3.0                   # the partial tile holds the rows an inner-product query likes best
    q = rng.standard_normal((64, dim)).astype(np.float32)
    keys = np.arange(n, dtype=np.uint64)
    ix = vs.HipUsearchIndex(dim, vs.IP)
    ix.reserve(n)
    ix.add_batch(keys, base)
    ref = vs.HipUsearchIndex(dim, vs.IP, _stress=2)  # exact search on the f32 VALU tile kernel
    ref.reserve(n)
    ref.add_batch(keys, base)
    rk, rd, rf = ref.exact_search_batch(q, k)
    k1, d1, f1 = ix.exact_search_batch(q, k)         # builds the plane
    assert ix.exact_stats()["plane_batches"] >= 1
    assert np.array_equal(k1, rk), "before the reserve"
    ix.reserve(n + 40_000)                            # capacity changes, no new rows: the plane is resized by copy
    k2, d2, f2 = ix.exact_search_batch(q, k)
    assert np.array_equal(k2, rk), "after the reserve"
    assert np.allclose(d2, rd, rtol=1e-5, atol=1e-5)
    assert ix.exact_stats()["plane_fallbacks"] == 0
