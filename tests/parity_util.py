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


def count_parity(got_keys, got_dist, want_keys, want_dist, want_found, oracle_distance_of=None, exact=False, got_found=None):
    """The same bar as assert_same_results over a whole batch, counted instead of asserted (bench.py's cpu_baseline leg,
    the full-size tests).  got_* / want_*: nq x k arrays (engine / oracle), want_found: oracle's result count per query;
    oracle_distance_of(query_index, key) -> the oracle's distance from that query to that member.
    A VIOLATION is a position where the two disagree and the parity bar does not allow it: a different result count,
    a distance beyond TOL, or ids that differ although the oracle's own distances of the two are not within TOL
    (exact=True: any difference in ids or distance bits).  Returns {rows, identical_rows, near_tie_positions,
    near_tie_rows, violations, violation_rows, first_violations}."""
    gk = np.asarray(got_keys).view(np.uint64) if np.asarray(got_keys).dtype == np.int64 else np.asarray(got_keys, dtype=np.uint64)
    wk = np.asarray(want_keys, dtype=np.uint64)
    gd, wd = np.asarray(got_dist, dtype=np.float32), np.asarray(want_dist, dtype=np.float32)
    nq, k = wk.shape
    found = np.asarray(want_found, dtype=np.int64)
    pos = np.arange(k)[None, :] < found[:, None]          # positions the oracle filled
    ids_differ = (gk != wk) & pos
    if exact:
        dist_bad = (gd.view(np.uint32) != wd.view(np.uint32)) & pos
    else:
        with np.errstate(invalid="ignore"):  # padding: inf - inf
            dist_bad = (np.abs(gd.astype(np.float64) - wd.astype(np.float64)) > TOL * np.maximum(1.0, np.abs(wd.astype(np.float64)))) & pos
    count_bad = np.zeros(nq, dtype=bool)
    if got_found is not None:
        count_bad = np.asarray(got_found, dtype=np.int64) != found
    else:  # the engine pads with the free key
        gfound = (gk != np.uint64(0xFFFFFFFFFFFFFFFF)).sum(axis=1)
        count_bad = gfound != found
    violations, ties, first = 0, 0, []
    tie_rows, bad_rows = set(), set()
    for qi in np.nonzero(count_bad)[0]:
        violations += 1
        bad_rows.add(int(qi))
        if len(first) < 5:
            first.append({"query": int(qi), "what": "result count", "oracle": int(found[qi])})
    for qi, j in zip(*np.nonzero(dist_bad & ~ids_differ)):
        violations += 1
        bad_rows.add(int(qi))
        if len(first) < 5:
            first.append({"query": int(qi), "position": int(j), "what": "distance", "engine": float(gd[qi, j]), "oracle": float(wd[qi, j])})
    for qi, j in zip(*np.nonzero(ids_differ)):
        ok = False
        if not exact and oracle_distance_of is not None and not dist_bad[qi, j]:
            ok = close(oracle_distance_of(int(qi), int(gk[qi, j])), wd[qi, j])
        if ok:
            ties += 1
            tie_rows.add(int(qi))
        else:
            violations += 1
            bad_rows.add(int(qi))
            if len(first) < 5:
                first.append({"query": int(qi), "position": int(j), "what": "id", "engine": int(gk[qi, j]), "oracle": int(wk[qi, j]),
                              "engine_distance": float(gd[qi, j]), "oracle_distance": float(wd[qi, j])})
    differing = set(np.nonzero((ids_differ | dist_bad).any(axis=1) | count_bad)[0].tolist())
    return {"rows": int(nq), "identical_rows": int(nq - len(differing)), "near_tie_positions": int(ties), "near_tie_rows": len(tie_rows - bad_rows),
            "violations": int(violations), "violation_rows": len(bad_rows), "first_violations": first,
            "bar": "exact: ids and distance bits" if exact else f"ids position by position; a differing id only where the oracle's own two distances are within {TOL} (f32 near-tie); distances within {TOL}*max(1,|d|)"}


def lattice(n, dim, seed, span=64):
    """Vectors with small integer coordinates: every l2sq / inner product of two of them is an integer far below 2^24,
    hence exact in f32 whatever the summation order -- the engine and the oracle then compute bit-identical distances
    and every decision (admission, ties, heuristic) must come out the same."""
    rng = np.random.default_rng(seed)
    return rng.integers(-span, span + 1, size=(n, dim)).astype(np.float32)
