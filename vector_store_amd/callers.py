"""ctypes binding of libvs_callers.so (include/vs_callers.h): the reference's load loops as library calls -- the search loop of
crates/benchmark (main.rs:435-525) over the C ABI, and the mixed add / search workloads of crates/vector-store/benches/pipeline.rs
through a dispatch actor."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import actor as _actor

_HERE = os.path.dirname(os.path.abspath(__file__))


class CallersResult(C.Structure):
    _fields_ = [("seconds", C.c_double), ("queries", C.c_uint64), ("qps", C.c_double), ("latency_min_ns", C.c_int64),
                ("latency_max_ns", C.c_int64)] + [(f"p{p:02d}_ns", C.c_int64) for p in (1, 10, 25, 50, 75, 90, 99)] + [
                ("recall_avg", C.c_double), ("errors", C.c_uint64), ("launches", C.c_uint64), ("team_launches", C.c_uint64)]

    def as_dict(self) -> dict:
        ms = lambda ns: None if ns >= 2 ** 62 else round(ns / 1e6, 3)  # noqa: E731
        return {"per_s": self.qps, "count": int(self.queries), "latency_min_ms": round(self.latency_min_ns / 1e6, 3),
                "latency_max_ms": round(self.latency_max_ns / 1e6, 3), "p50_ms": ms(self.p50_ns), "p90_ms": ms(self.p90_ns), "p99_ms": ms(self.p99_ns)}


class MixedOptions(C.Structure):
    _fields_ = [("plain_callers", C.c_uint), ("filtered_callers", C.c_uint), ("producers", C.c_uint), ("modify", C.c_int),
                ("modulus", C.c_uint64), ("partition", C.c_uint64), ("first_new_key", C.c_uint64), ("existing_keys", C.c_uint64),
                ("delete_from", C.c_uint64), ("max_items", C.c_uint64), ("k", C.c_size_t), ("seconds", C.c_double), ("filter_key", C.c_uint64)]


class MixedResult(C.Structure):
    _fields_ = [("seconds", C.c_double), ("items", C.c_uint64), ("adds_applied", C.c_uint64), ("removes_applied", C.c_uint64),
                ("predicate_calls", C.c_uint64), ("filtered_results", C.c_uint64), ("errors", C.c_uint64),
                ("item", CallersResult), ("plain", CallersResult), ("filtered", CallersResult)]


class CallersRecord(C.Structure):
    _fields_ = [("query", C.c_void_p), ("found", C.c_void_p), ("keys", C.c_void_p), ("distances", C.c_void_p), ("cap", C.c_size_t), ("n", C.c_size_t)]


NONE, INSERT, UPDATE, DELETE = 0, 1, 2, 3
_lib = None


def lib():
    global _lib
    if _lib is None:
        _actor.lib()  # libvs_hnsw.so, libvs_actor.so first
        L = C.CDLL(os.path.join(os.environ.get("VS_LIB_DIR") or _HERE, "libvs_callers.so"))
        L.vs_mixed_run.argtypes = [C.c_void_p, C.POINTER(MixedOptions), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                   C.POINTER(MixedResult)]
        L.vs_callers_run_recorded.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint,
                                              C.c_double, C.POINTER(CallersResult), C.POINTER(CallersRecord)]
        L.vs_callers_run_filtered_keyed.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint, C.c_double,
                                                    C.POINTER(CallersResult), C.POINTER(C.c_uint64), C.POINTER(CallersRecord)]
        _lib = L
    return _lib


def _record(cap: int, k: int):
    arrays = {"query": np.zeros(cap, dtype=np.uint32), "found": np.zeros(cap, dtype=np.uint32),
              "keys": np.zeros((cap, k), dtype=np.uint64), "distances": np.zeros((cap, k), dtype=np.float32)}
    rec = CallersRecord(arrays["query"].ctypes.data, arrays["found"].ctypes.data, arrays["keys"].ctypes.data, arrays["distances"].ctypes.data, cap, 0)
    return rec, arrays


def _cut(rec, arrays):
    n = int(rec.n)
    return {name: a[:n] for name, a in arrays.items()}


def run(index, queries, k, truth=None, threads=17, inflight=1, seconds=3.0, record=0):
    """The reference's search loop (crates/benchmark/src/main.rs:435-525) over vs_hnsw_search / vs_hnsw_search_async on `index`
    (vector_store_amd.HipUsearchIndex).  record > 0: also what the first `record` completed calls RECEIVED ({"query", "found",
    "keys", "distances"}), for id parity against the oracle.  Returns (CallersResult, record or None, status)."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    t = None if truth is None else np.ascontiguousarray(truth, dtype=np.uint64)
    r = CallersResult()
    rec, arrays = _record(record, k) if record else (None, None)
    rc = lib().vs_callers_run_recorded(index.h, q.ctypes.data, q.shape[0], q.shape[1], k, None if t is None else t.ctypes.data, threads, inflight,
                                       seconds, C.byref(r), None if rec is None else C.byref(rec))
    return r, (None if rec is None else _cut(rec, arrays)), rc


def run_filtered(index, queries, k, modulus, threads=17, seconds=3.0, record=0, filter_key=0):
    """The same loop over vs_hnsw_filtered_search with the predicate key % modulus == 0 (usearch.rs:937-948: every filtered query on a
    blocking thread of its own).  Returns (CallersResult, [predicate calls, results returned], record or None, status)."""
    q = np.ascontiguousarray(queries, dtype=np.float32)
    r = CallersResult()
    extra = (C.c_uint64 * 4)()
    rec, arrays = _record(record, k) if record else (None, None)
    rc = lib().vs_callers_run_filtered_keyed(index.h, q.ctypes.data, q.shape[0], q.shape[1], k, modulus, filter_key, threads, seconds, C.byref(r), extra,
                                             None if rec is None else C.byref(rec))
    return r, [int(extra[0]), int(extra[1])], (None if rec is None else _cut(rec, arrays)), rc


def mixed_run(actor, queries, vectors, *, modify=NONE, plain_callers=0, filtered_callers=0, producers=1, modulus=10, partition=0,
              first_new_key=0, existing_keys=0, delete_from=0, max_items=0, k=10, seconds=2.0, filter_key=0) -> dict:
    """One leg of the reference's pipeline benches (include/vs_callers.h: vs_mixed_run) through `actor` (vector_store_amd.actor.IndexActor)."""
    q = None if queries is None else np.ascontiguousarray(queries, dtype=np.float32)
    v = None if vectors is None else np.ascontiguousarray(vectors, dtype=np.float32)
    o = MixedOptions(plain_callers, filtered_callers, producers, modify, modulus, partition, first_new_key, existing_keys, delete_from,
                     max_items, k, seconds, filter_key)
    r = MixedResult()
    rc = lib().vs_mixed_run(actor.h, C.byref(o), None if q is None else q.ctypes.data, 0 if q is None else q.shape[0],
                            None if v is None else v.ctypes.data, 0 if v is None else v.shape[0], actor.dim, C.byref(r))
    if rc != 0:
        raise RuntimeError(f"vs_mixed_run: status {rc}")
    out = {"seconds": r.seconds, "items": int(r.items), "items_per_s": r.items / r.seconds if r.seconds > 0 else 0.0,
           "adds_applied": int(r.adds_applied), "removes_applied": int(r.removes_applied), "errors": int(r.errors),
           "producers": producers if modify != NONE else 0}
    if modify != NONE:
        out["item"] = r.item.as_dict()
    if plain_callers:
        out["plain"] = dict(r.plain.as_dict(), callers=plain_callers)
    if filtered_callers:
        out["filtered"] = dict(r.filtered.as_dict(), callers=filtered_callers, predicate=f"key % {modulus} == 0",
                               predicate_calls_per_query=r.predicate_calls / max(int(r.filtered.queries), 1), filter_named=bool(filter_key))
    return out


PIPELINE_LEGS = ("cdc_insert", "cdc_update", "cdc_delete", "search_while_inserting", "search_while_updating", "search_while_deleting")


def pipeline_legs(actor, queries, vectors, existing_keys, legs=PIPELINE_LEGS, *, seconds=2.0, producers=1, plain_callers=16, filtered_callers=16,
                  modulus=10, state=None, log=None, filter_key=0) -> dict:
    """The reference's pipeline benches (crates/vector-store/benches/pipeline.rs:1407-1418) as legs of vs_mixed_run through `actor`:
    cdc_insert / cdc_update / cdc_delete alone, and search_while_{inserting,updating,deleting} with `plain_callers` + `filtered_callers`
    blocking searchers on the same partition.  Keys 0 .. existing_keys - 1 exist: updates draw from the lower half, deletes eat the upper
    half upwards from state["delete_from"]; inserts use fresh keys from state["next_key"].  A leg name may carry a caller mix:
    "search_while_updating:16+0" = 16 plain, no filtered callers; "...@named": the filtered callers name their filter."""
    state = state if state is not None else {}
    state.setdefault("next_key", 1 << 40)
    state.setdefault("delete_from", existing_keys // 2)
    out = {}
    for leg in legs:
        name, _, mix = leg.partition(":")
        mix, _, named = mix.partition("@")
        if "@" in name:
            name, _, named = name.partition("@")
        plain, filtered = (int(x) for x in mix.split("+")) if mix else (plain_callers, filtered_callers)
        # "...@named": the filtered callers name their filter (vs_actor_filtered_ann_keyed: verdicts remembered across queries)
        kw = dict(k=10, seconds=seconds, producers=producers, modulus=modulus, filter_key=(filter_key or 0xF117E5) if named else filter_key)
        if name.startswith("search"):
            kw.update(plain_callers=plain, filtered_callers=filtered)
        what = name.replace("search_while_", "cdc_")
        if what in ("cdc_insert", "cdc_inserting"):
            kw.update(modify=INSERT, first_new_key=state["next_key"])
        elif what in ("cdc_update", "cdc_updating"):
            kw.update(modify=UPDATE, existing_keys=existing_keys // 2)
        elif what in ("cdc_delete", "cdc_deleting"):
            kw.update(modify=DELETE, delete_from=state["delete_from"])
        r = mixed_run(actor, queries, vectors, **kw)
        if kw.get("modify") == INSERT:
            state["next_key"] += r["items"] + producers + 1
        if kw.get("modify") == DELETE:
            state["delete_from"] += r["items"] + producers + 1
        out[leg] = r
        if log:
            log(leg, r)
    return out
