"""The ONE stdout line of bench.py: the driver's contract fields and nothing else.

Round 5's line had grown to 34 KB (every boundary leg with its latency table, the mixed legs, the side configs) and the driver, which
reads the tail of stdout, could no longer parse it.  The full record now goes to a FILE (`gpurun_out/bench_full.json`, named in the
line as `full_record`); the line keeps the contract -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / dtype / config /
recall / roofline / cpu_baseline -- plus ONE number per boundary leg and per side config.  `contract_line` refuses to return more than
`MAX_LINE_BYTES`; `tests/test_bench_line.py` feeds it a worst-case record.
"""
import json

MAX_LINE_BYTES = 8192

_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "recall_at_10", "ef_search", "rccl_ranks", "comm_ranks", "exchange", "same_device", "sharded_weak_queries_per_s",
        "sharded_weak_recall_at_10", "sharded_fixed_total_queries_per_s", "sharded_fixed_total_recall_at_10", "error")
_ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "bytes_per_query")
_CONFIG = ("workload", "distribution", "mode", "index_vectors_total", "query_batches_rotated")


def _num(v, digits=5):
    """Numbers at five significant digits: a line of a hundred of them stays short, and nothing here is known more finely."""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    return v


def _clip(s, n=160):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def _leg_rate(rec):
    """One number for a boundary leg: what its callers got per second (None where the leg failed)."""
    if not isinstance(rec, dict):
        return None
    for key in ("queries_per_s", "items_per_s", "updates_per_s"):
        if isinstance(rec.get(key), (int, float)):
            return _num(rec[key])
    return None


def _parity(par):
    if not isinstance(par, dict):
        return None
    return {k: par.get(k) for k in ("rows", "identical_rows", "violations")}


def _boundary(b):
    """{leg: rate} -- nested groups flattened with '.', `vs_cpu` where the leg has one, the parity violations summed."""
    if not isinstance(b, dict):
        return None
    if "error" in b and len(b) == 1:
        return {"error": _clip(b["error"])}
    legs, vs_cpu, rows, identical, violations = {}, {}, 0, 0, 0

    def visit(prefix, rec):
        nonlocal rows, identical, violations
        if not isinstance(rec, dict):
            return
        if "error" in rec and _leg_rate(rec) is None and not any(isinstance(v, dict) for v in rec.values()):
            legs[prefix] = None
            return
        rate = _leg_rate(rec)
        if rate is not None:
            legs[prefix] = rate
            for kind in ("plain", "filtered"):  # the searches beside a mixed leg's producers
                if isinstance(rec.get(kind), dict) and isinstance(rec[kind].get("per_s"), (int, float)):
                    legs[f"{prefix}/{kind}"] = _num(rec[kind]["per_s"])
            v = rec.get("vs_cpu")
            if isinstance(v, (int, float)):
                vs_cpu[prefix] = _num(v, 3)
            elif isinstance(v, dict):
                for kk, vv in v.items():
                    if isinstance(vv, (int, float)):
                        vs_cpu[prefix if kk in ("items", "updates") else f"{prefix}/{kk}"] = _num(vv, 3)
            par = rec.get("id_parity")
            if isinstance(par, dict):
                rows += par.get("rows", 0)
                identical += par.get("identical_rows", 0)
                violations += par.get("violations", 0)
            return
        for name, sub in rec.items():
            if name in ("actor_counters", "engine", "id_parity", "vs_cpu", "mixed"):
                continue
            visit(f"{prefix}.{name}" if prefix else name, sub)

    visit("", b)
    out = {"queries_per_s": legs}
    mixed = b.get("mixed")
    if isinstance(mixed, dict):  # per leg: the producers' items/s; the searches beside them and the CPU ratios only for search_while_updating
        if "error" in mixed and len(mixed) == 1:
            out["mixed"] = {"error": _clip(mixed["error"])}
        else:
            items, beside = {}, {}
            for pk, group in mixed.items():
                short = pk.replace("producers_", "p")
                if pk == "two_indexes" and isinstance(group, dict):
                    items["two_indexes"] = _leg_rate(group)
                    continue
                if not pk.startswith("producers_") or not isinstance(group, dict):
                    continue
                for leg, rec in group.items():
                    if leg == "actor_counters" or not isinstance(rec, dict):
                        continue
                    items[f"{short}.{leg}"] = _leg_rate(rec)
                    if leg.startswith("search_while_updating"):
                        v = rec.get("vs_cpu") if isinstance(rec.get("vs_cpu"), dict) else {}
                        beside[f"{short}.{leg}"] = {"plain_per_s": _num((rec.get("plain") or {}).get("per_s")),
                                                    "filtered_per_s": _num((rec.get("filtered") or {}).get("per_s")),
                                                    "vs_cpu": [_num(v.get(kk), 3) for kk in ("items", "plain", "filtered")]}
            out["mixed"] = {"items_per_s": items, "search_while_updating": beside}
    if vs_cpu:
        out["vs_cpu"] = vs_cpu
    out["id_parity"] = {"rows": rows, "identical_rows": identical, "violations": violations}
    for name in ("p50_ms", "p99_ms"):  # the lone caller's latency: the figure every caller-facing rate is a multiple of
        v = (b.get("blocking_callers") or {}).get(name) if isinstance(b.get("blocking_callers"), dict) else None
        if v is not None:
            out[f"blocking_callers_{name}"] = v
    return out


def _cpu_baseline(c):
    if not isinstance(c, dict):
        return None
    if "error" in c and "value" not in c:
        return {"error": _clip(c["error"])}
    out = {k: _num(c.get(k)) for k in ("value", "unit", "cores", "kind")}
    out["sample"] = _clip(c.get("sample", ""), 200)
    out["id_parity"] = _parity(c.get("id_parity"))
    for k in ("build_vectors_per_s", "recall_at_10"):
        if k in c:
            out[k] = _num(c[k])
    filt = c.get("filtered")
    if isinstance(filt, dict):
        out["filtered_queries_per_s"] = {name: _leg_rate(rec) for name, rec in filt.items()}
    return out


def _side_config(c):
    if not isinstance(c, dict):
        return None
    out = {"config": c.get("config")}
    for k in ("error", "skipped"):
        if k in c:
            out[k] = _clip(c[k], 100)
    for k in ("queries_per_s", "ms_per_batch", "recall_at_10", "build_vectors_per_s", "plane_fallback_batches", "plane8_batches", "plane8_fallback_batches"):
        if k in c:
            out[k] = _num(c[k])
    r = c.get("roofline")
    if isinstance(r, dict):
        out["roofline_frac"] = _num(r.get("frac"), 3)
        if "frac_of_bf16_plane_floor" in r:
            out["frac_of_bf16_plane_floor"] = _num(r.get("frac_of_bf16_plane_floor"), 3)
        out["kernel"] = _clip(r.get("kernel"), 48)
    for leg in ("blocking_callers", "blocking_callers_64"):
        if isinstance(c.get(leg), dict):
            out[f"{leg}_queries_per_s"] = _leg_rate(c[leg])
    return out


def contract_line(out, full_record=None):
    """The compact line for the full record `out`.  Raises ValueError when it would exceed MAX_LINE_BYTES."""
    line = {k: _num(out[k]) for k in _TOP if k in out}
    if "error" in line:
        line["error"] = _clip(line["error"], 300)
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in _CONFIG if k in cfg}
    r = out.get("roofline")
    if isinstance(r, dict):
        line["roofline"] = {k: _num(r.get(k), 6) for k in _ROOFLINE}
        src = r.get("traffic_source")
        if isinstance(src, dict):  # which PMC record `traffic` is (null when the kernel sources changed since it was taken)
            line["roofline"]["traffic_source"] = src.get("file")
            line["roofline"]["traffic_stale"] = src.get("stale")
    line["cpu_baseline"] = _cpu_baseline(out.get("cpu_baseline"))
    b = out.get("build")
    if isinstance(b, dict):
        line["build_vectors_per_s"] = _num(b.get("vectors_per_s"))
    if "boundary" in out:
        line["boundary"] = _boundary(out["boundary"])
    if isinstance(out.get("configs"), list):
        line["configs"] = [_side_config(c) for c in out["configs"]]
    if isinstance(out.get("sharded"), dict):
        line["sharded"] = {leg: (_leg_rate(rec) if "error" not in rec else {"error": _clip(rec["error"], 120)})
                           for leg, rec in out["sharded"].items() if isinstance(rec, dict)}
    if full_record:
        line["full_record"] = full_record
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= MAX_LINE_BYTES:  # shed the optional blocks in order of weight before giving up
        for victim in ("configs", "boundary"):
            if victim in line:
                line[victim] = {"omitted": f"see {full_record}"}
                text = json.dumps(line, separators=(",", ":"))
                if len(text) < MAX_LINE_BYTES:
                    break
    if len(text) >= MAX_LINE_BYTES:
        raise ValueError(f"bench line of {len(text)} bytes: the contract line must stay under {MAX_LINE_BYTES}")
    return text
