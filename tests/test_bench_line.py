"""bench.py's stdout line stays a line the driver can parse (round-5 review, item 1): contract fields only, under 8 KB, whatever the
side legs produced -- the full record goes to a file.  CPU only: feeds `bench_line.contract_line` a worst-case record and every full
bench record kept under profiles/."""
import glob
import json
import os

import pytest

import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _latency(n=1.0):
    return {"per_s": 12345.678901 * n, "count": 123456, "latency_min_ms": 0.123, "latency_max_ms": 811.384, "p50_ms": 1.0, "p90_ms": 1.0, "p99_ms": 1.0,
            "callers": 16, "predicate": "key % 10 == 0", "predicate_calls_per_query": 10833.259072580646, "filter_named": False}


def _leg(n=1.0):
    return {"threads": 128, "in_flight_per_thread": 256, "queries_per_s": 361854.76161886106 * n, "seconds": 3.012299728, "latency_min_ms": 2.577,
            "p50_ms": 10.801, "p90_ms": 15.424, "p99_ms": 18.276, "recall_at_10": 0.9511, "errors": 0, "status": 0, "kernel_launches": 629,
            "queries_or_rounds_posted_to_pods": 107149, "pods_opened": 3, "predicate": "key % 10 == 0", "walk_launches_per_query": 2.15,
            "id_parity": {"rows": 100000, "identical_rows": 99975, "near_tie_positions": 37, "violations": 0, "first_violations": ["x" * 400] * 5},
            "cpu_queries_per_s": 790.1446580905963, "vs_cpu": 1.3654001607855706}


def _mixed_leg():
    return {"seconds": 1.5, "items": 524, "items_per_s": 348.33982148492225, "adds_applied": 524, "removes_applied": 0, "errors": 0, "producers": 16,
            "item": _latency(), "plain": _latency(), "filtered": _latency(), "vs_cpu": {"items": 1.9075, "plain": 1.96168, "filtered": 0.85004}}


def worst_case():
    mixed_legs = ("cdc_insert", "cdc_update", "cdc_delete", "search_while_updating:16+0", "search_while_updating", "search:0+16@named",
                  "search_while_updating@named", "search_while_inserting", "search_while_deleting", "search_while_inserting@named",
                  "search_while_deleting@named", "search_while_updating:64+64")
    group = {name: _mixed_leg() for name in mixed_legs}
    group["actor_counters"] = {"adds": 2136, "searches": 10917, "removes": 102391, "mode_switches": 2971}
    sel = ("selectivity_10pct", "selectivity_10pct_64_callers", "selectivity_10pct_128_callers", "selectivity_1pct", "selectivity_50pct", "selectivity_0_1pct")
    boundary = {"cores": 16, "pods_enabled": True, "note": "n" * 600,
                "blocking_callers": _leg(), "blocking_callers_64": _leg(), "blocking_callers_256": _leg(), "async_in_flight": _leg(),
                "pods_beside_async": {"blocking_callers": _leg(), "async_in_flight": _leg(), "pods_opened": 2},
                "filtered": {s: _leg() for s in sel}, "filtered_named": {s: _leg() for s in sel}, "filtered_resumed": {s: _leg() for s in sel},
                "mixed": {"producers_1": group, "producers_16": dict(group), "producers_64": dict(group), "engine": {"flushes": 2764},
                          "two_indexes": {"updates_per_s": 77721.5, "update_item": _latency(), "note": "n" * 300}}}
    side = {"config": "configs[4]", "workload": "w" * 200, "queries_per_s": 53883.2, "ms_per_batch": 4.75, "recall_at_10": 0.95, "build_vectors_per_s": 477384.8,
            "plane_fallback_batches": 0, "roofline": {"bound": "hbm", "frac": 0.404, "kernel": "k" * 300}, "blocking_callers": _leg(), "blocking_callers_64": _leg(),
            "hnsw_walk_ef200": {"ms_per_batch": 1.24, "queries_per_s": 205508.1}}
    return {
        "metric": "QPS at recall@10>=0.95 (HNSW search, inputs resident in HBM)", "value": 457801.4204413891, "unit": "queries/s", "n_gpus": 8, "steps": 20,
        "warmup": 3, "ms_per_step": 21.843532050115755, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "10000000x768 cos top-10 per GPU, 10000 queries/step, M=16 ef_add=128 ef_search=200 bf16", "distribution": "lowrank24", "mode": "replica",
                   "index_vectors_total": 80000000, "query_batches_rotated": 4},
        "recall_at_10": 0.9511, "ef_search": 200, "ef_sweep": [{"ef": e, "recall": 0.9} for e in range(64, 512, 8)],
        "roofline": {"bound": "hbm", "achieved": 6678.6308, "peak": 8000.0, "unit": "GB/s", "frac": 0.83482885, "traffic": 150922041386.6667, "kernel": "hnsw_search_kernel",
                     "kernel_ms": 21.836331939697267, "bytes_per_query": 14583679.9056, "evals_per_query": 4737.08675,
                     "traffic_source": {"file": "profiles/r06_h_traffic.json", "stale": False}, "dram": {"note": "n" * 900, "counters_per_launch": {"a": 1.0}}},
        "build": {"vectors_per_s": 469159.8, "seconds": 21.3},
        "boundary": boundary,
        "cpu_baseline": {"value": 6100.77, "unit": "queries/s", "cores": 16, "kind": "port", "sample": "s" * 500,
                         "id_parity": {"rows": 10000, "identical_rows": 9997, "near_tie_positions": 5, "violations": 0, "first_violations": ["v" * 300] * 5, "bar": "b" * 300},
                         "filtered": {s: {"queries_per_s": 790.1, "queries": 2048} for s in sel}, "build_vectors_per_s": 7292.16, "build_sample": "b" * 200,
                         "mixed": {"producers_1": group, "producers_16": dict(group)}, "recall_at_10": 0.9511},
        "configs": [dict(side, config=f"configs[{i}]") for i in range(6)] + [{"config": "configs[2]", "error": "e" * 2000}],
        "generators_at_1m": [{"data": "g" * 30, "points": [{"ef": 128, "recall_at_10": 0.01}] * 8}] * 6,
        "sharded": {"weak": {"queries_per_s": 3.2e6, "recall_at_10": 0.95}, "fixed_total": {"error": "e" * 1000}, "collective": "c" * 200},
        "rccl_ranks": 8, "comm_ranks": 8, "exchange": "rccl", "sharded_weak_queries_per_s": 3.2e6, "sharded_weak_recall_at_10": 0.95,
        "error": "e" * 2000,
    }


def check(line, full):
    assert "\n" not in line and len(line) < bench_line.MAX_LINE_BYTES
    rec = json.loads(line)
    for key in CONTRACT:
        assert rec[key] == pytest.approx(full[key], rel=1e-4) if isinstance(full[key], float) else rec[key] is not None or full[key] is None, key
    assert rec["config"]["workload"] == full["config"]["workload"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rec["roofline"]
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    return rec


def test_worst_case_record_stays_under_the_cap_and_keeps_the_contract():
    full = worst_case()
    assert len(json.dumps(full)) > 60_000  # twice what round 5's line had grown to
    rec = check(bench_line.contract_line(full, "gpurun_out/bench_full.json"), full)
    assert rec["n_gpus"] == 8 and rec["rccl_ranks"] == 8
    for key in ("value", "unit", "cores", "kind", "sample", "id_parity"):
        assert key in rec["cpu_baseline"]
    assert rec["cpu_baseline"]["id_parity"] == {"rows": 10000, "identical_rows": 9997, "violations": 0}
    assert rec["full_record"] == "gpurun_out/bench_full.json"
    assert len(rec["error"]) <= 300


def test_a_record_no_shedding_can_save_is_refused():
    full = worst_case()
    full["config"]["workload"] = "w" * 9000
    with pytest.raises(ValueError):
        bench_line.contract_line(full, "x.json")


def test_error_legs_do_not_break_the_line():
    full = worst_case()
    full["boundary"] = {"error": "RuntimeError('boom')"}
    full["cpu_baseline"] = {"error": "RuntimeError('boom')"}
    rec = check(bench_line.contract_line(full, None), full)
    assert "error" in rec["boundary"] and "error" in rec["cpu_baseline"] and "full_record" not in rec
    full = worst_case()
    full["boundary"]["mixed"] = {"error": "x"}
    full["boundary"]["filtered"]["selectivity_1pct"] = {"error": "y"}
    rec = check(bench_line.contract_line(full, None), full)
    assert rec["boundary"]["queries_per_s"]["filtered.selectivity_1pct"] is None


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*bench*.json"))))
def test_every_recorded_full_line_compacts(path):
    try:
        full = json.load(open(path))
    except ValueError:
        pytest.skip("not a single JSON document")
    if not isinstance(full, dict) or "roofline" not in full or "metric" not in full:
        pytest.skip("not a bench record")
    check(bench_line.contract_line(full, "gpurun_out/bench_full.json"), full)
