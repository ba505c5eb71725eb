#!/usr/bin/env python3
"""Writes tests/golden/kat.json: the known-answer vectors the reference's own tests hold
for the usearch-backed index/search path.

Every entry is DATA (inputs + expected outputs) transcribed from the cited reference test,
not reference source.  The reference cannot be executed in this image (no rustc / cargo,
usearch 2.22.0 is not vendored), so expectations are exactly what those tests assert.
Run:  python tests/golden/make_golden.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def rows30():
    # tests/integration/vs_index.rs:1018-1059: i -> pk=i/10, ck=i%10, f=i, vector [i,i,i]
    return [{"key": i, "pk": i // 10, "ck": i % 10, "f": i, "vector": [float(i)] * 3} for i in range(30)]


def sel(pred):
    return sorted(r["key"] for r in rows30() if pred(r))


kat = {
    "_doc": "Known-answer tests transcribed from scylladb/vector-store tests; see make_golden.py",
    # crates/vector-store/src/vs_index/usearch.rs:1298-1458 (add_or_replace_size_ann)
    "B1_l2sq_3d_basic": {
        "source": "crates/vector-store/src/vs_index/usearch.rs:1298-1458",
        "metric": "l2sq", "dim": 3,
        "steps": [
            {"op": "add", "key": 1, "vector": [1.0, 1.0, 1.0]},
            {"op": "add", "key": 2, "vector": [2.0, -2.0, 2.0]},
            {"op": "add", "key": 3, "vector": [3.0, 3.0, 3.0]},
            {"op": "count", "expect": 3},
            {"op": "search", "query": [2.2, -2.2, 2.2], "k": 1, "expect_keys": [2]},
            {"op": "remove", "key": 3, "expect": True},
            {"op": "count", "expect": 2},
            {"op": "add", "key": 3, "vector": [2.1, -2.1, 2.1]},
            {"op": "count", "expect": 3},
            {"op": "search", "query": [2.2, -2.2, 2.2], "k": 1, "expect_keys": [3]},
            {"op": "remove", "key": 3, "expect": True},
            {"op": "count", "expect": 2},
            {"op": "search", "query": [2.2, -2.2, 2.2], "k": 1, "expect_keys": [2]},
        ],
    },
    # crates/vector-store/tests/integration/vs_index.rs:242-301
    "B2_l2sq_3d_http": {
        "source": "crates/vector-store/tests/integration/vs_index.rs:242-301",
        "metric": "l2sq", "dim": 3,
        "base": [{"key": 1, "vector": [1.0, 1.0, 1.0]}, {"key": 2, "vector": [2.0, -2.0, 2.0]},
                 {"key": 3, "vector": [3.0, 3.0, 3.0]}],
        "query": [2.1, -2.0, 2.0], "k": 1, "expect_keys": [2],
    },
    # crates/vector-store/tests/integration/vs_index.rs:1795-1886
    "B3_l2sq_1d_scores": {
        "source": "crates/vector-store/tests/integration/vs_index.rs:1746-1887",
        "metric": "l2sq", "dim": 1,
        "base": [{"key": 0, "vector": [0.0]}, {"key": 1, "vector": [1.0]}, {"key": 2, "vector": [3.0]}],
        "query": [0.0], "k": 3,
        "expect_keys": [0, 1, 2], "expect_distances": [0.0, 1.0, 9.0],
        "expect_similarity": [1.0, 0.5, 0.1], "tolerance": 1e-5,
    },
    # crates/vector-store/tests/integration/vs_index.rs:1889-1951
    "B4_empty": {
        "source": "crates/vector-store/tests/integration/vs_index.rs:1889-1951",
        "metric": "l2sq", "dim": 3, "base": [], "query": [1.0, 2.0, 3.0], "k": 10, "expect_keys": [],
    },
    # crates/validator/src/similarity_functions.rs:113-194
    "B5_cos_winners": {
        "source": "crates/validator/src/similarity_functions.rs:130-144,164-178",
        "metric": "cos", "dim": 3,
        "base": [{"key": 1, "vector": [1.0, 0.0, 0.0]}, {"key": 2, "vector": [0.0, 1.0, 0.0]},
                 {"key": 3, "vector": [0.0, 0.0, 1.0]}, {"key": 4, "vector": [2.0, 0.0, 0.0]}],
        "query": [1.0, 0.0, 0.0], "k": 2, "expect_key_set": [1, 4], "expect_distances": [0.0, 0.0],
        "all_distances": {"1": 0.0, "2": 1.0, "3": 1.0, "4": 0.0},
    },
    "B6_ip_winner": {
        "source": "crates/validator/src/similarity_functions.rs:147-161; crates/vector-store/src/similarity.rs:94-100",
        "metric": "ip", "dim": 3,
        "base": [{"key": 1, "vector": [1.0, 0.0, 0.0]}, {"key": 2, "vector": [0.0, 1.0, 0.0]},
                 {"key": 3, "vector": [0.0, 0.0, 1.0]}, {"key": 4, "vector": [2.0, 0.0, 0.0]}],
        "query": [1.0, 0.0, 0.0], "k": 1, "expect_keys": [4], "expect_distances": [-1.0],
        "expect_similarity": [1.5],
        "all_distances": {"1": 0.0, "2": 1.0, "3": 1.0, "4": -1.0},
    },
    "B7_l2_winner": {
        "source": "crates/validator/src/similarity_functions.rs:114-127",
        "metric": "l2sq", "dim": 3,
        "base": [{"key": 1, "vector": [1.0, 0.0, 0.0]}, {"key": 2, "vector": [0.0, 1.0, 0.0]},
                 {"key": 3, "vector": [0.0, 0.0, 1.0]}, {"key": 4, "vector": [1.0, 1.0, 1.0]}],
        "query": [1.0, 0.0, 0.0], "k": 1, "expect_keys": [1], "expect_distances": [0.0],
        "all_distances": {"1": 0.0, "2": 2.0, "3": 2.0, "4": 2.0},
    },
    # crates/vector-store/tests/integration/quantization.rs:95-117
    "B8_quant_f32": {
        "source": "crates/vector-store/tests/integration/quantization.rs:95-117",
        "metric": "l2sq", "dim": 3, "base": [{"key": 1, "vector": [0.9, 0.1, 0.1]}],
        "query": [1.0, 0.0, 0.0], "k": 1, "expect_keys": [1], "expect_distance_below": 0.1,
        "expect_distances": [0.03], "tolerance": 1e-6,
    },
    # crates/vector-store/tests/integration/quantization.rs:175-259 (F32 leg)
    "B10_self_zero_f32": {
        "source": "crates/vector-store/tests/integration/quantization.rs:175-259",
        "metric": "l2sq", "dim": 1536, "base": [{"key": 1, "vector_fill": 0.5}],
        "query_fill": 0.5, "k": 1, "expect_keys": [1], "expect_distances_exact": [0.0],
    },
    # crates/vector-store/tests/integration/vs_index.rs:718-1640 (30-row filter fixture)
    "B11_filter_30": {
        "source": "crates/vector-store/tests/integration/vs_index.rs:718-1640",
        "metric": "l2sq", "dim": 3, "base": rows30(), "query": [1.0, 2.0, 3.0], "k": 100,
        "cases": [
            {"name": "pk_eq_1", "src": ":721-775", "expect": sel(lambda r: r["pk"] == 1)},
            {"name": "ck_eq_1", "src": ":779-833", "expect": sel(lambda r: r["ck"] == 1)},
            {"name": "pk_in_1_2", "src": ":837-895", "expect": sel(lambda r: r["pk"] in (1, 2))},
            {"name": "ck_in_1_3", "src": ":899-957", "expect": sel(lambda r: r["ck"] in (1, 3))},
            {"name": "pkck_eq_1_5", "src": ":961-985", "expect": sel(lambda r: (r["pk"], r["ck"]) == (1, 5))},
            {"name": "pkck_in_07_15", "src": ":989-1012",
             "expect": sel(lambda r: (r["pk"], r["ck"]) in ((0, 7), (1, 5)))},
            {"name": "ck_lt_3", "src": ":1122-1158", "expect": sel(lambda r: r["ck"] < 3)},
            {"name": "ck_lte_2", "src": ":1163-1199", "expect": sel(lambda r: r["ck"] <= 2)},
            {"name": "ck_gt_6", "src": ":1204-1240", "expect": sel(lambda r: r["ck"] > 6)},
            {"name": "ck_gte_7", "src": ":1245-1281", "expect": sel(lambda r: r["ck"] >= 7)},
            {"name": "ck_range_3_6", "src": ":1286-1328", "expect": sel(lambda r: 3 <= r["ck"] < 6)},
            {"name": "pkck_lt_1_5", "src": ":1333-1378", "expect": sel(lambda r: (r["pk"], r["ck"]) < (1, 5))},
            {"name": "pkck_lte_1_5", "src": ":1382-1428", "expect": sel(lambda r: (r["pk"], r["ck"]) <= (1, 5))},
            {"name": "pkck_gt_1_5", "src": ":1432-1476", "expect": sel(lambda r: (r["pk"], r["ck"]) > (1, 5))},
            {"name": "pkck_gte_1_5", "src": ":1480-1525", "expect": sel(lambda r: (r["pk"], r["ck"]) >= (1, 5))},
            {"name": "f_eq_1", "src": ":1616-1640", "expect": sel(lambda r: r["f"] == 1)},
        ],
        "expect_counts": {"pk_eq_1": 10, "ck_eq_1": 3, "pk_in_1_2": 20, "ck_in_1_3": 6, "pkck_eq_1_5": 1,
                          "pkck_in_07_15": 2, "ck_lt_3": 9, "ck_lte_2": 9, "ck_gt_6": 9, "ck_gte_7": 9,
                          "ck_range_3_6": 9, "pkck_lt_1_5": 15, "pkck_lte_1_5": 16, "pkck_gt_1_5": 14,
                          "pkck_gte_1_5": 15, "f_eq_1": 1},
    },
    # crates/validator/src/quantization_and_rescoring.rs:21-37,98-146
    "B12_fine_order": {
        "source": "crates/validator/src/quantization_and_rescoring.rs:21-37,98-146",
        "metric": "cos", "dim": 3, "rows": 500, "query": [0.5, 0.3, 0.7], "step": 0.001,
        "direction": [2.0, 4.0, 8.0], "k": 500, "returned": 100,
        "doc": "row i (key i) = query + step*i*direction in f32; the first `returned` keys of the "
               "k=500 result must be non-decreasing (== 0..99 when the order is exact)",
    },
    # crates/validator/src/ann.rs:34-103
    "B13_zero_query": {
        "source": "crates/validator/src/ann.rs:34-103",
        "metric": "cos", "dim": 3, "rows": 1000, "zero_rows": 100, "query": [0.0, 0.0, 0.0], "k": 100,
        "doc": "row i<100 = [0,0,0], else [i%3, i%5, i%7]; result has <=100 rows, every key exists, "
               "every distance lies in [0,2] (zero-vs-zero 0, zero-vs-nonzero 1)",
    },
    # crates/vector-store/src/vs_index/usearch.rs:1622-1664
    "B14_b1_packing": {
        "source": "crates/vector-store/src/vs_index/usearch.rs:1622-1664",
        "cases": [
            {"input": [], "expect": []},
            {"input": [1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0], "expect": [0x0F]},
            {"input": [1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, -1.0, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, 1.0],
             "expect": [0x55, 0xF0]},
            {"input": [1.0] * 64, "expect": [0xFF] * 8},
            {"input": [1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, -1.0, 1.0], "expect": [0x55, 0x05]},
        ],
    },
    # crates/vector-store/src/distance.rs:123-195
    "B15_ranges": {
        "source": "crates/vector-store/src/distance.rs:118-196",
        "l2sq": {"ok": [0.0, 0.123, 1.0, 2.0, 5.0, 100.5, "max", "inf"], "err": [-0.1, -1.0, "-inf", "nan"]},
        "cos": {"ok": [0.0, 0.123, 1.0, 2.0], "err": [5.0, 100.5, "max", -0.1, -1.0, "inf", "-inf", "nan"]},
        "ip": {"ok": [0.0, 0.123, 1.0, 2.0, 5.0, 100.5, "max", -0.1, -1.0, "inf", "-inf"], "err": ["nan"]},
        "hamming": {"dim": 3, "ok": [0.0, 1.0, 2.0],
                    "err": [0.123, 5.0, 100.5, "max", -0.1, -1.0, "inf", "-inf", "nan"]},
    },
    # crates/vector-store/src/similarity.rs:47-132
    "B16_scores": {
        "source": "crates/vector-store/src/similarity.rs:40-133",
        "cases": [
            {"metric": "l2sq", "d": 0.0, "score": 1.0}, {"metric": "l2sq", "d": 1.0, "score": 0.5},
            {"metric": "l2sq", "d": 99.0, "score": 0.01},
            {"metric": "cos", "d": 0.0, "score": 1.0}, {"metric": "cos", "d": 1.0, "score": 0.5},
            {"metric": "cos", "d": 2.0, "score": 0.0},
            {"metric": "ip", "d": 0.0, "score": 1.0}, {"metric": "ip", "d": 1.0, "score": 0.5},
            {"metric": "ip", "d": 2.0, "score": 0.0}, {"metric": "ip", "d": 6.7, "score": -2.35},
            {"metric": "ip", "d": -1.8, "score": 1.9},
            {"metric": "hamming", "dim": 128, "d": 0.0, "score": 1.0},
            {"metric": "hamming", "dim": 128, "d": 64.0, "score": 0.5},
            {"metric": "hamming", "dim": 128, "d": 128.0, "score": 0.0},
            {"metric": "hamming", "dim": 50, "d": 35.0, "score": 0.3},
            {"metric": "hamming", "dim": 50, "d": 50.0, "score": 0.0},
        ],
    },
    # crates/vector-store/src/vs_index/usearch.rs:1526-1607
    "B17_concurrency": {
        "source": "crates/vector-store/src/vs_index/usearch.rs:1526-1607",
        "metric": "l2sq", "dim": 1024, "adds_per_worker": 50, "search_k": 5,
        "doc": "2 x cores tasks x 50 adds of all-zero vectors with unique keys, and as many searches "
               "(k=5), issued concurrently: no error, final count = tasks*50",
    },
}

with open(os.path.join(HERE, "kat.json"), "w") as f:
    json.dump(kat, f, indent=1, sort_keys=True)
print("wrote", os.path.join(HERE, "kat.json"))
