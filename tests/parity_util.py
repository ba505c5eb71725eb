"""The parity bar of the GPU tests, in one place.

Integer / index work: ids identical to the oracle's, position by position.  Floating point: distances within
TOL * max(1, |d|); an id may differ from the oracle's ONLY at a position where the two candidates are an f32 near-tie,
i.e. the ORACLE's own distance of the engine's id is within TOL of the oracle's distance at that position (the engine
sums a row across lanes, the oracle in 8-wide AVX partial sums: re-association flips such pairs).  Every differing
position is checked for exactly that; nothing is accepted on a percentage.
"""
import numpy as np

TOL = 1e-5


def close(a, b, tol=TOL):
    return abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b)))


def assert_same_results(got_keys, got_dist, want_keys, want_dist, oracle_distance_of=None, exact=False, what=""):
    """got_*: the engine's answer for one query (already cut to `found`); want_*: the oracle's.
    exact: no exception at all (integer metrics, exactly representable data): ids AND distance bits identical.
    oracle_distance_of(key) -> the oracle's distance from the query to that member.
    Returns the number of near-tie positions (0 for an identical row)."""
    gk, wk = [int(x) for x in got_keys], [int(x) for x in want_keys]
    assert len(gk) == len(wk), (what, len(gk), len(wk))
    gd, wd = np.asarray(got_dist, dtype=np.float32), np.asarray(want_dist, dtype=np.float32)
    if exact:
        assert gd.tolist() == wd.tolist(), (what, "distances", gd[:8], wd[:8])
        assert gk == wk, (what, "ids", gk[:12], wk[:12])
        return 0
    for j in range(len(wk)):
        assert close(gd[j], wd[j]), (what, j, float(gd[j]), float(wd[j]))
    assert all(gd[j] <= gd[j + 1] for j in range(len(gd) - 1)), (what, "ascending")
    assert len(set(gk)) == len(gk), (what, "duplicate id")
    ties = 0
    for j in range(len(wk)):
        if gk[j] != wk[j]:
            assert oracle_distance_of is not None, (what, j, gk[j], wk[j])
            d_other = oracle_distance_of(gk[j])
            assert close(d_other, wd[j]), (what, "ids differ and it is not a near-tie", j, gk[j], wk[j], d_other, float(wd[j]))
            ties += 1
    return ties


def lattice(n, dim, seed, span=64):
    """Vectors with small integer coordinates: every l2sq / inner product of two of them is an integer far below 2^24,
    hence exact in f32 whatever the summation order -- the engine and the oracle then compute bit-identical distances
    and every decision (admission, ties, heuristic) must come out the same."""
    rng = np.random.default_rng(seed)
    return rng.integers(-span, span + 1, size=(n, dim)).astype(np.float32)
