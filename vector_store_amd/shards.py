"""ctypes binding of libvs_shards.so (include/vs_shards.h): one `UsearchIndex`-shaped handle over several GPUs in
one process (each an independent HNSW graph; keys dealt in 4096-row ranges; search = per-shard top-k + merge)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import index as _ix

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _ix.lib()
        L = C.CDLL(os.path.join(_HERE, "libvs_shards.so"))
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.vs_shards_create.argtypes = [C.POINTER(_ix._Options), vp, sz, C.POINTER(vp)]
        L.vs_shards_free.argtypes = [vp]
        for f in ("vs_shards_count", "vs_shards_capacity", "vs_shards_size"):
            getattr(L, f).restype = sz
            getattr(L, f).argtypes = [vp]
        L.vs_shards_owner.restype = sz
        L.vs_shards_owner.argtypes = [vp, u64]
        L.vs_shards_reserve.argtypes = [vp, sz, sz]
        L.vs_shards_add.argtypes = [vp, u64, vp, sz]
        L.vs_shards_add_batch.argtypes = [vp, vp, vp, sz, sz]
        L.vs_shards_remove.argtypes = [vp, u64, C.POINTER(C.c_int)]
        L.vs_shards_search.argtypes = [vp, vp, sz, sz, vp, vp, C.POINTER(sz)]
        L.vs_shards_filtered_search.argtypes = [vp, vp, sz, sz, _ix.PRED, vp, vp, vp, C.POINTER(sz)]
        L.vs_shards_search_batch.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp]
        L.vs_shards_set_expansion_search.argtypes = [vp, sz]
        L.vs_shards_stats.argtypes = [vp, vp, C.c_int]
        L.vs_shards_last_error.restype = C.c_char_p
        _lib = L
    return _lib


class ShardedIndex:
    def __init__(self, dimensions: int, metric: int = _ix.COS, devices=(0,), connectivity: int = 16, expansion_add: int = 128,
                 expansion_search: int = 64, quantization: int = _ix.F32):
        self.L = lib()
        self.dim = dimensions
        o = _ix._Options(dimensions, connectivity, expansion_add, expansion_search, metric, quantization, -1, 0)
        dev = np.asarray(devices, dtype=np.int32)
        h = C.c_void_p()
        self._check(self.L.vs_shards_create(C.byref(o), dev.ctypes.data, dev.size, C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise _ix.VsError(rc, self.L.vs_shards_last_error().decode())

    def stop(self):
        if getattr(self, "h", None):
            self.L.vs_shards_free(self.h)
            self.h = None

    __del__ = stop

    def shards(self) -> int:
        return self.L.vs_shards_count(self.h)

    def owner(self, key: int) -> int:
        return self.L.vs_shards_owner(self.h, key)

    def reserve(self, size: int, threads: int = 0):
        self._check(self.L.vs_shards_reserve(self.h, size, threads))

    def capacity(self) -> int:
        return self.L.vs_shards_capacity(self.h)

    def size(self) -> int:
        return self.L.vs_shards_size(self.h)

    def add(self, primary_id: int, vector):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        self._check(self.L.vs_shards_add(self.h, primary_id, v.ctypes.data, v.size))

    def add_batch(self, keys, vectors):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        self._check(self.L.vs_shards_add_batch(self.h, keys.ctypes.data, vectors.ctypes.data, keys.size, vectors.shape[-1]))

    def remove(self, primary_id: int) -> bool:
        r = C.c_int(0)
        self._check(self.L.vs_shards_remove(self.h, primary_id, C.byref(r)))
        return bool(r.value)

    def _one(self, fn, vector, limit, *extra):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        keys = np.zeros(limit, dtype=np.uint64)
        d = np.zeros(limit, dtype=np.float32)
        found = C.c_size_t(0)
        self._check(fn(self.h, v.ctypes.data, v.size, limit, *extra, keys.ctypes.data, d.ctypes.data, C.byref(found)))
        return keys[: found.value], d[: found.value]

    def search(self, vector, limit: int):
        return self._one(self.L.vs_shards_search, vector, limit)

    def filtered_search(self, vector, limit: int, predicate):
        cb = _ix.PRED(lambda key, _ctx: 1 if predicate(key) else 0)
        return self._one(self.L.vs_shards_filtered_search, vector, limit, cb, None)

    def search_batch(self, queries, k: int):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        keys = np.zeros((nq, k), dtype=np.uint64)
        d = np.zeros((nq, k), dtype=np.float32)
        found = np.zeros(nq, dtype=np.uint64)
        self._check(self.L.vs_shards_search_batch(self.h, q.ctypes.data, nq, q.shape[1], k, keys.ctypes.data, d.ctypes.data,
                                                  found.ctypes.data))
        return keys, d, found.astype(np.int64)

    def set_expansion_search(self, ef: int):
        self._check(self.L.vs_shards_set_expansion_search(self.h, ef))

    def stats(self, reset: bool = False) -> dict:
        out = np.zeros(8, dtype=np.uint64)
        self._check(self.L.vs_shards_stats(self.h, out.ctypes.data, int(reset)))
        names = ["search_evals", "search_hops", "queries", "add_evals", "add_hops", "added", "visited_overflow", "link_evals"]
        return {n: int(v) for n, v in zip(names, out)}
