"""ctypes binding of libvs_actor.so (include/vs_actor.h): the reference's per-index dispatch actor
(crates/vector-store/src/vs_index/usearch.rs:688-1177) over the HIP engine.  Method names follow the
reference's sender helpers (vs_index/actor.rs:63-175): add_vector / remove_vector / remove_partition /
ann / filtered_ann / count."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import index as _ix

_HERE = os.path.dirname(os.path.abspath(__file__))


class _ActorOptions(C.Structure):
    _fields_ = [("index", _ix._Options), ("workers", C.c_size_t), ("local", C.c_int), ("reserve_increment", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _ix.lib()  # libvs_hnsw.so first (RTLD_GLOBAL), then the actor on top of it
        L = C.CDLL(os.path.join(os.environ.get("VS_LIB_DIR") or _HERE, "libvs_actor.so"))
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.vs_actor_create.argtypes = [C.POINTER(_ActorOptions), C.POINTER(vp)]
        L.vs_actor_create_with.argtypes = [C.POINTER(_ActorOptions), vp, C.POINTER(vp)]
        L.vs_actor_adopt_partition.argtypes = [vp, u64, vp, sz]
        L.vs_actor_add_vector_wait.argtypes = [vp, u64, u64, vp, sz, C.POINTER(C.c_int)]
        L.vs_actor_remove_vector_wait.argtypes = [vp, u64, u64, C.POINTER(C.c_int)]
        L.vs_actor_stop.argtypes = [vp]
        L.vs_actor_add_vector.argtypes = [vp, u64, u64, vp, sz]
        L.vs_actor_remove_vector.argtypes = [vp, u64, u64]
        L.vs_actor_remove_partition.argtypes = [vp, u64]
        L.vs_actor_ann.argtypes = [vp, u64, vp, sz, sz, vp, vp, C.POINTER(sz)]
        L.vs_actor_filtered_ann.argtypes = [vp, u64, vp, sz, sz, _ix.PRED, vp, vp, vp, C.POINTER(sz)]
        L.vs_actor_filtered_ann_keyed.argtypes = [vp, u64, vp, sz, sz, _ix.PRED, vp, u64, vp, vp, C.POINTER(sz)]
        L.vs_actor_count.restype = sz
        L.vs_actor_count.argtypes = [vp]
        L.vs_actor_set_allocate.argtypes = [vp, C.c_int]
        L.vs_actor_partition_capacity.restype = sz
        L.vs_actor_partition_capacity.argtypes = [vp, u64]
        L.vs_actor_partitions.restype = sz
        L.vs_actor_partitions.argtypes = [vp]
        L.vs_actor_counters.argtypes = [vp, vp]
        L.vs_actor_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class IndexActor:
    """index_vtable: nine function pointers in the order of include/vs_actor.h's vs_actor_index_vtable (a ctypes array of
    c_void_p) -- the actor over another implementation of `trait UsearchIndex` (tests and bench.py's cpu_baseline bind the CPU
    oracle: oracle.trait_vtable()); None: the HIP engine."""

    def __init__(self, dimensions: int, metric: int = _ix.COS, connectivity: int = 16, expansion_add: int = 128,
                 expansion_search: int = 64, workers: int = 0, local: bool = False, reserve_increment: int = 0,
                 quantization: int = 0, reserved: int = 0, index_vtable=None):
        self.L = lib()
        self.dim = dimensions
        o = _ActorOptions(_ix._Options(dimensions, connectivity, expansion_add, expansion_search, metric, quantization, -1, reserved),
                          workers, int(local), reserve_increment)
        h = C.c_void_p()
        self._vtable = index_vtable  # (kept alive: the actor copies it, the functions must stay loaded)
        if index_vtable is None:
            rc = self.L.vs_actor_create(C.byref(o), C.byref(h))
        else:
            rc = self.L.vs_actor_create_with(C.byref(o), C.cast(index_vtable, C.c_void_p), C.byref(h))
        if rc != 0:
            raise _ix.VsError(rc, self.L.vs_actor_last_error().decode())
        self.h = h

    def adopt_partition(self, partition: int, index_handle, size: int):
        """An index that exists already (bulk-built) becomes partition `partition`; the actor does not own it."""
        rc = self.L.vs_actor_adopt_partition(self.h, partition, index_handle, size)
        if rc != 0:
            raise _ix.VsError(rc, self.L.vs_actor_last_error().decode())

    def add_vector_wait(self, partition: int, primary_id: int, vector) -> bool:
        v = np.ascontiguousarray(vector, dtype=np.float32)
        applied = C.c_int(0)
        self.L.vs_actor_add_vector_wait(self.h, partition, primary_id, v.ctypes.data, v.size, C.byref(applied))
        return bool(applied.value)

    def remove_vector_wait(self, partition: int, primary_id: int) -> bool:
        applied = C.c_int(0)
        self.L.vs_actor_remove_vector_wait(self.h, partition, primary_id, C.byref(applied))
        return bool(applied.value)

    def stop(self):
        if getattr(self, "h", None):
            self.L.vs_actor_stop(self.h)
            self.h = None

    __del__ = stop

    def add_vector(self, partition: int, primary_id: int, vector):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        self.L.vs_actor_add_vector(self.h, partition, primary_id, v.ctypes.data, v.size)

    def remove_vector(self, partition: int, primary_id: int):
        self.L.vs_actor_remove_vector(self.h, partition, primary_id)

    def remove_partition(self, partition: int):
        self.L.vs_actor_remove_partition(self.h, partition)

    def _search(self, fn, partition, vector, limit, *extra):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        keys = np.zeros(limit, dtype=np.uint64)
        d = np.zeros(limit, dtype=np.float32)
        found = C.c_size_t(0)
        rc = fn(self.h, partition, v.ctypes.data, v.size, limit, *extra, keys.ctypes.data, d.ctypes.data, C.byref(found))
        if rc != 0:
            msg = "wrong embedding dimension" if rc == -2 else self.L.vs_actor_last_error().decode()
            raise _ix.VsError(rc, msg)
        return keys[: found.value], d[: found.value]

    def ann(self, partition: int, vector, limit: int):
        return self._search(self.L.vs_actor_ann, partition, vector, limit)

    def filtered_ann(self, partition: int, vector, limit: int, predicate, filter_key: int = 0):
        cb = _ix.PRED(lambda key, _ctx: 1 if predicate(key) else 0)
        if filter_key:
            return self._search(self.L.vs_actor_filtered_ann_keyed, partition, vector, limit, cb, None, filter_key)
        return self._search(self.L.vs_actor_filtered_ann, partition, vector, limit, cb, None)

    def count(self) -> int:
        return self.L.vs_actor_count(self.h)

    def set_allocate(self, can: bool):
        self.L.vs_actor_set_allocate(self.h, int(can))

    def partition_capacity(self, partition: int) -> int:
        return self.L.vs_actor_partition_capacity(self.h, partition)

    def partitions(self) -> int:
        return self.L.vs_actor_partitions(self.h)

    def counters(self) -> dict:
        out = np.zeros(8, dtype=np.uint64)
        self.L.vs_actor_counters(self.h, out.ctypes.data)
        names = ["adds", "adds_dropped", "reserves", "searches", "removes", "mode_switches", "max_in_flight", "errors"]
        return {n: int(v) for n, v in zip(names, out)}
