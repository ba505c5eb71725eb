"""Runs the reference's known-answer tests (tests/golden/kat.json)
against any index object exposing the reference's `trait UsearchIndex` surface
(crates/vector-store/src/vs_index/usearch.rs:142-160):

    reserve(capacity), capacity(), size(), add(key, vector), remove(key) -> bool,
    search(vector, k) -> (keys, distances), filtered_search(vector, k, predicate)

`factory(metric_name, dim, **opts)` builds an empty index.  Used with the CPU oracle
(not gpu) and with the HIP engine through the C ABI (gpu).
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat.json")

with open(GOLDEN) as f:
    KAT = json.load(f)


def _fill(ix, base, dim):
    ix.reserve(max(len(base), 1) + 8)
    for row in base:
        v = row["vector"] if "vector" in row else [row["vector_fill"]] * dim
        ix.add(row["key"], np.asarray(v, dtype=np.float32))


def run_simple(factory, name):
    """B2, B3, B4, B5, B6, B7, B8, B10."""
    t = KAT[name]
    dim = t["dim"]
    ix = factory(t["metric"], dim)
    _fill(ix, t["base"], dim)
    assert ix.size() == len(t["base"])
    q = t["query"] if "query" in t else [t["query_fill"]] * dim
    keys, d = ix.search(np.asarray(q, dtype=np.float32), t["k"])
    keys = [int(k) for k in keys]
    tol = t.get("tolerance", 1e-6)
    if "expect_keys" in t:
        assert keys == t["expect_keys"], (name, keys)
    if "expect_key_set" in t:
        assert sorted(keys) == sorted(t["expect_key_set"]), (name, keys)
    if "expect_distances" in t:
        assert len(d) == len(t["expect_distances"])
        for got, want in zip(d, t["expect_distances"]):
            assert abs(float(got) - want) <= tol * max(1.0, abs(want)), (name, got, want)
    if "expect_distances_exact" in t:
        assert [float(x) for x in d] == t["expect_distances_exact"], (name, d)
    if "expect_distance_below" in t:
        assert all(float(x) < t["expect_distance_below"] for x in d)
    assert all(float(d[i]) <= float(d[i + 1]) for i in range(len(d) - 1)), "ascending order"
    if "all_distances" in t:  # every stored key's distance, via a full-length search
        keys_all, d_all = ix.search(np.asarray(q, dtype=np.float32), len(t["base"]))
        got = {str(int(k)): float(x) for k, x in zip(keys_all, d_all)}
        assert got.keys() == t["all_distances"].keys()
        for k, want in t["all_distances"].items():
            assert abs(got[k] - want) <= 1e-6, (name, k, got[k], want)
    return keys, d


def run_b1(factory):
    t = KAT["B1_l2sq_3d_basic"]
    ix = factory(t["metric"], t["dim"])
    ix.reserve(16)
    for s in t["steps"]:
        if s["op"] == "add":
            ix.add(s["key"], np.asarray(s["vector"], dtype=np.float32))
        elif s["op"] == "remove":
            assert ix.remove(s["key"]) == s["expect"]
        elif s["op"] == "count":
            assert ix.size() == s["expect"], s
        elif s["op"] == "search":
            keys, d = ix.search(np.asarray(s["query"], dtype=np.float32), s["k"])
            assert [int(k) for k in keys] == s["expect_keys"], (s, keys)
            assert len(d) == len(keys)


def run_b11(factory, case=None):
    t = KAT["B11_filter_30"]
    ix = factory(t["metric"], t["dim"])
    _fill(ix, t["base"], t["dim"])
    rows = {r["key"]: r for r in t["base"]}
    q = np.asarray(t["query"], dtype=np.float32)
    for c in t["cases"]:
        if case and c["name"] != case:
            continue
        allowed = set(c["expect"])
        keys, d = ix.filtered_search(q, t["k"], lambda key: int(key) in allowed)
        got = sorted(int(k) for k in keys)
        assert got == c["expect"], (c["name"], got)
        assert len(got) == t["expect_counts"][c["name"]]
        # ascending by exact squared-L2 distance to [1,2,3]
        want_d = sorted(float(np.sum((np.asarray(rows[k]["vector"], dtype=np.float32) - q) ** 2)) for k in got)
        assert [float(x) for x in d] == want_d, (c["name"], d)


def b12_rows():
    t = KAT["B12_fine_order"]
    q = np.asarray(t["query"], dtype=np.float32)
    dirv = np.asarray(t["direction"], dtype=np.float32)
    step = np.float32(t["step"])
    rows = np.stack([q + step * np.float32(i) * dirv for i in range(t["rows"])]).astype(np.float32)
    return t, q, rows


def run_b12(factory):
    t, q, rows = b12_rows()
    ix = factory(t["metric"], t["dim"])
    ix.reserve(t["rows"] + 8)
    for i in range(t["rows"]):
        ix.add(i, rows[i])
    keys, d = ix.search(q, t["k"])
    assert len(keys) == t["rows"]
    first = [int(k) for k in keys[: t["returned"]]]
    assert all(float(d[i]) <= float(d[i + 1]) for i in range(len(d) - 1))
    assert all(0.0 <= float(x) <= 2.0 for x in d)
    return first, d


def run_b13(factory):
    t = KAT["B13_zero_query"]
    ix = factory(t["metric"], t["dim"])
    ix.reserve(t["rows"] + 8)
    for i in range(t["rows"]):
        v = [0.0, 0.0, 0.0] if i < t["zero_rows"] else [float(i % 3), float(i % 5), float(i % 7)]
        ix.add(i, np.asarray(v, dtype=np.float32))
    keys, d = ix.search(np.asarray(t["query"], dtype=np.float32), t["k"])
    assert 0 < len(keys) <= t["k"]
    assert all(0 <= int(k) < t["rows"] for k in keys)
    assert len(set(int(k) for k in keys)) == len(keys)
    for k, x in zip(keys, d):
        x = float(x)
        assert 0.0 <= x <= 2.0
        i = int(k)
        is_zero = i < t["zero_rows"] or (i % 3 == 0 and i % 5 == 0 and i % 7 == 0)
        assert x == (0.0 if is_zero else 1.0), (i, x)
    assert all(float(d[i]) <= float(d[i + 1]) for i in range(len(d) - 1))


def special(v):
    if isinstance(v, str):
        return {"max": float(np.finfo(np.float32).max), "inf": math.inf, "-inf": -math.inf, "nan": math.nan}[v]
    return float(v)
