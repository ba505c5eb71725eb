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
                ("delete_from", C.c_uint64), ("max_items", C.c_uint64), ("k", C.c_size_t), ("seconds", C.c_double)]


class MixedResult(C.Structure):
    _fields_ = [("seconds", C.c_double), ("items", C.c_uint64), ("adds_applied", C.c_uint64), ("removes_applied", C.c_uint64),
                ("predicate_calls", C.c_uint64), ("filtered_results", C.c_uint64), ("errors", C.c_uint64),
                ("item", CallersResult), ("plain", CallersResult), ("filtered", CallersResult)]


NONE, INSERT, UPDATE, DELETE = 0, 1, 2, 3
_lib = None


def lib():
    global _lib
    if _lib is None:
        _actor.lib()  # libvs_hnsw.so, libvs_actor.so first
        L = C.CDLL(os.path.join(_HERE, "libvs_callers.so"))
        L.vs_mixed_run.argtypes = [C.c_void_p, C.POINTER(MixedOptions), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t,
                                   C.POINTER(MixedResult)]
        _lib = L
    return _lib


def mixed_run(actor, queries, vectors, *, modify=NONE, plain_callers=0, filtered_callers=0, producers=1, modulus=10, partition=0,
              first_new_key=0, existing_keys=0, delete_from=0, max_items=0, k=10, seconds=2.0) -> dict:
    """One leg of the reference's pipeline benches (include/vs_callers.h: vs_mixed_run) through `actor` (vector_store_amd.actor.IndexActor)."""
    q = None if queries is None else np.ascontiguousarray(queries, dtype=np.float32)
    v = None if vectors is None else np.ascontiguousarray(vectors, dtype=np.float32)
    o = MixedOptions(plain_callers, filtered_callers, producers, modify, modulus, partition, first_new_key, existing_keys, delete_from,
                     max_items, k, seconds)
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
                               predicate_calls_per_query=r.predicate_calls / max(int(r.filtered.queries), 1))
    return out
