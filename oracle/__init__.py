"""CPU oracle for the usearch-backed index/search path -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/cpu_hnsw.cpp (a C++ restatement of the usearch 2.22.0
algorithm the reference calls at crates/vector-store/src/vs_index/usearch.rs:172-236).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product package (vector_store_amd) never does.

Parity status: pinned by the reference's known-answer tests (tests/golden/),
unpinned for large-graph traversal order -- see the header of cpu_hnsw.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("VS_ORACLE_LIB") or os.path.join(_HERE, "liboracle_hnsw.so")  # VS_ORACLE_LIB: sanitizer build

COS, L2SQ, IP, HAMMING = 0, 1, 2, 3
F32, F16, BF16, I8, B1 = 0, 1, 2, 3, 4
SCALARS = {"f32": F32, "f16": F16, "bf16": BF16, "i8": I8, "b1": B1}
METRICS = {"cos": COS, "l2sq": L2SQ, "ip": IP, "hamming": HAMMING}
FREE_KEY = 0xFFFFFFFFFFFFFFFF
INVALID_SLOT = 0xFFFFFFFF

PRED = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (g++ only)."""
    src = os.path.join(_HERE, "cpu_hnsw.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, sz, u64p, f32p, u32p, i32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        L.orc_last_error.restype = C.c_char_p
        L.orc_create.restype = vp
        L.orc_create.argtypes = [sz, C.c_int, sz, sz, sz]
        L.orc_create_ex.restype = vp
        L.orc_create_ex.argtypes = [sz, C.c_int, C.c_int, sz, sz, sz]
        L.orc_bytes_per_vector.restype = sz
        L.orc_bytes_per_vector.argtypes = [vp]
        L.orc_distance_as.restype = C.c_float
        L.orc_distance_as.argtypes = [C.c_int, C.c_int, vp, vp, sz]
        L.orc_free.argtypes = [vp]
        L.orc_reserve.argtypes = [vp, sz]
        for f in ("orc_capacity", "orc_size", "orc_slots", "orc_upper_blocks"):
            getattr(L, f).restype = sz
            getattr(L, f).argtypes = [vp]
        L.orc_max_level.argtypes = [vp]
        L.orc_entry_slot.restype = C.c_uint32
        L.orc_entry_slot.argtypes = [vp]
        L.orc_set_expansion_search.argtypes = [vp, sz]
        L.orc_add.argtypes = [vp, C.c_uint64, vp, sz]
        L.orc_add_with_level.argtypes = [vp, C.c_uint64, vp, C.c_int]
        L.orc_remove.argtypes = [vp, C.c_uint64]
        L.orc_search.argtypes = [vp, vp, sz, u64p, f32p, C.POINTER(sz)]
        L.orc_search_slots.argtypes = [vp, vp, sz, u64p, f32p, u32p, C.POINTER(sz)]
        L.orc_filtered_search.argtypes = [vp, vp, sz, PRED, vp, u64p, f32p, C.POINTER(sz)]
        L.orc_add_batch.argtypes = [vp, vp, vp, sz, sz]
        L.orc_search_batch.argtypes = [vp, vp, sz, sz, vp, vp, vp, sz]
        L.orc_filtered_search_timed.argtypes = [vp, vp, sz, sz, C.c_uint64, vp, vp, vp, sz, C.c_double, vp]
        L.orc_stats.argtypes = [vp, u64p, C.c_int]
        L.orc_exact_search.argtypes = [vp, vp, sz, u64p, f32p, C.POINTER(sz)]
        L.orc_distance_to_slot.restype = C.c_float
        L.orc_distance_to_slot.argtypes = [vp, vp, C.c_uint32]
        L.orc_export_graph.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_import_graph.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_uint32]
        L.orc_vectors.restype = vp
        L.orc_vectors.argtypes = [vp]
        L.orc_level_stream.argtypes = [sz, sz, vp]
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [C.c_int, vp, vp, sz]
        L.orc_f32_to_b1x8.argtypes = [vp, sz, vp]
        L.orc_distance_valid.argtypes = [C.c_float, C.c_int, sz]
        L.orc_similarity.restype = C.c_float
        L.orc_similarity.argtypes = [C.c_float, C.c_int, sz]
        L.orc_trait_vtable.argtypes = [vp]
        _lib = L
    return _lib


def trait_vtable():
    """The oracle as an implementation of `trait UsearchIndex` for libvs_actor (include/vs_actor.h: vs_actor_index_vtable):
    nine function pointers and the optional tenth (filtered_search_keyed) left NULL -- a CPU usearch has no verdict memory.  Only
    tests and bench.py's cpu_baseline leg pass it to vs_actor_create_with."""
    t = (C.c_void_p * 10)()
    lib().orc_trait_vtable(t)
    return t


class OracleError(RuntimeError):
    pass


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def f32_to_b1x8(v) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros((v.size + 7) // 8, dtype=np.uint8)
    lib().orc_f32_to_b1x8(_ptr(v), v.size, _ptr(out))
    return out


def distance_as(metric: int, scalar: int, a, b) -> float:
    """Distance between two f32 vectors as an index with the given storage type computes it."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_distance_as(metric, scalar, _ptr(a), _ptr(b), a.size))


def distance(metric: int, a, b) -> float:
    if metric == HAMMING:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return float(lib().orc_distance(metric, _ptr(a), _ptr(b), a.size * 8))
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().orc_distance(metric, _ptr(a), _ptr(b), a.size))


def distance_valid(v: float, metric: int, dim: int = 0) -> bool:
    return bool(lib().orc_distance_valid(v, metric, dim))


def similarity(d: float, metric: int, dim: int = 0) -> float:
    return float(lib().orc_similarity(d, metric, dim))


def level_stream(connectivity: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.int32)
    lib().orc_level_stream(connectivity, n, _ptr(out))
    return out


class OracleIndex:
    """Mirrors the reference's private `trait UsearchIndex` (usearch.rs:142-160)."""

    def __init__(self, dim: int, metric: int = COS, connectivity: int = 16, expansion_add: int = 128,
                 expansion_search: int = 64, quantization: int = F32):
        self.L = lib()
        if metric == HAMMING:
            quantization = B1
        if quantization == B1:
            metric = HAMMING
        self.dim, self.metric, self.scalar = dim, metric, quantization
        self.M = connectivity or 16
        self.M0 = 2 * self.M
        self.h = self.L.orc_create_ex(dim, metric, quantization, connectivity, expansion_add, expansion_search)
        if not self.h:
            raise OracleError(self.L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_free(self.h)
            self.h = None

    def _vec(self, v) -> np.ndarray:
        v = np.ascontiguousarray(v, dtype=np.float32)
        if v.shape[-1] != self.dim:
            raise OracleError("wrong embedding dimension")
        return v

    def _check(self, rc):
        if rc != 0:
            raise OracleError(self.L.orc_last_error().decode())

    def reserve(self, capacity: int):
        self._check(self.L.orc_reserve(self.h, capacity))

    def capacity(self) -> int:
        return self.L.orc_capacity(self.h)

    def size(self) -> int:
        return self.L.orc_size(self.h)

    def slots(self) -> int:
        return self.L.orc_slots(self.h)

    def set_expansion_search(self, ef: int):
        self.L.orc_set_expansion_search(self.h, ef)

    def add(self, key: int, vector, level: int | None = None, thread: int = 0):
        """thread: the usearch thread slot (context) the call runs on -- every context has its own level generator."""
        v = self._vec(vector)
        if level is None:
            self._check(self.L.orc_add(self.h, key, _ptr(v), thread))
        else:
            self._check(self.L.orc_add_with_level(self.h, key, _ptr(v), level))

    def add_batch(self, keys, vectors, threads: int = 1):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        assert vectors.shape == (keys.size, self.dim)
        self._check(self.L.orc_add_batch(self.h, _ptr(keys), _ptr(vectors), keys.size, threads))

    def remove(self, key: int) -> bool:
        return self.L.orc_remove(self.h, key) != 0

    def search(self, vector, k: int, return_slots: bool = False):
        v = self._vec(vector)
        keys = np.zeros(k, dtype=np.uint64)
        d = np.zeros(k, dtype=np.float32)
        slots = np.zeros(k, dtype=np.uint32)
        found = C.c_size_t(0)
        self._check(self.L.orc_search_slots(self.h, _ptr(v), k, keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            d.ctypes.data_as(C.POINTER(C.c_float)),
                                            slots.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(found)))
        n = found.value
        if return_slots:
            return keys[:n], d[:n], slots[:n]
        return keys[:n], d[:n]

    def filtered_search(self, vector, k: int, predicate):
        v = self._vec(vector)
        keys = np.zeros(k, dtype=np.uint64)
        d = np.zeros(k, dtype=np.float32)
        found = C.c_size_t(0)
        cb = PRED(lambda key, _ctx: 1 if predicate(key) else 0)
        self._check(self.L.orc_filtered_search(self.h, _ptr(v), k, cb, None,
                                               keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               d.ctypes.data_as(C.POINTER(C.c_float)), C.byref(found)))
        return keys[:found.value], d[:found.value]

    def search_batch(self, queries, k: int, threads: int = 1):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        keys = np.zeros((nq, k), dtype=np.uint64)
        d = np.zeros((nq, k), dtype=np.float32)
        found = np.zeros(nq, dtype=np.uint64)
        self._check(self.L.orc_search_batch(self.h, _ptr(q), nq, k, _ptr(keys), _ptr(d), _ptr(found), threads))
        return keys, d, found.astype(np.int64)

    def filtered_search_timed(self, queries, k: int, modulus: int, threads: int = 1, seconds: float = 5.0):
        """filtered_search with the predicate `key % modulus == 0` from `threads` threads, one query per call, for at most
        `seconds`: (keys, distances, found, queries answered, predicate calls, wall seconds)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        keys = np.zeros((nq, k), dtype=np.uint64)
        d = np.zeros((nq, k), dtype=np.float32)
        found = np.zeros(nq, dtype=np.uint64)
        out = np.zeros(3, dtype=np.uint64)
        self._check(self.L.orc_filtered_search_timed(self.h, _ptr(q), nq, k, modulus, _ptr(keys), _ptr(d), _ptr(found), threads,
                                                     float(seconds), _ptr(out)))
        return keys, d, found.astype(np.int64), int(out[0]), int(out[1]), out[2] / 1e9

    def exact_search(self, vector, k: int):
        v = self._vec(vector)
        keys = np.zeros(k, dtype=np.uint64)
        d = np.zeros(k, dtype=np.float32)
        found = C.c_size_t(0)
        self._check(self.L.orc_exact_search(self.h, _ptr(v), k, keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            d.ctypes.data_as(C.POINTER(C.c_float)), C.byref(found)))
        return keys[:found.value], d[:found.value]

    def distance_to_slot(self, query, slot: int) -> float:
        """The index's own distance from an f32 query (cast as search casts it) to the stored row of `slot`."""
        q = np.ascontiguousarray(query, dtype=np.float32)
        return float(self.L.orc_distance_to_slot(self.h, _ptr(q), int(slot)))

    def stats(self, reset: bool = False):
        out = (C.c_uint64 * 2)()
        self.L.orc_stats(self.h, out, int(reset))
        return {"computed_distances": int(out[0]), "node_expansions": int(out[1])}

    def export_graph(self) -> dict:
        n = self.slots()
        blocks = self.L.orc_upper_blocks(self.h)
        g = {
            "levels": np.zeros(n, dtype=np.int32),
            "keys": np.zeros(n, dtype=np.uint64),
            "adj0": np.zeros((n, self.M0), dtype=np.uint32),
            "upper_off": np.zeros(n, dtype=np.uint32),
            "upper": np.zeros((max(blocks, 1), self.M), dtype=np.uint32),
        }
        self.L.orc_export_graph(self.h, _ptr(g["levels"]), _ptr(g["keys"]), _ptr(g["adj0"]), _ptr(g["upper_off"]),
                                _ptr(g["upper"]))
        g["upper"] = g["upper"][:blocks]
        g["max_level"] = self.L.orc_max_level(self.h)
        g["entry_slot"] = self.L.orc_entry_slot(self.h)
        bpv = self.L.orc_bytes_per_vector(self.h)
        raw = (C.c_uint8 * (n * bpv)).from_address(self.L.orc_vectors(self.h)) if n else b""
        arr = np.frombuffer(raw, dtype=np.float32 if self.scalar == F32 else np.uint8).copy()  # storage format
        g["vectors"] = arr.reshape(n, -1) if n else arr.reshape(0, self.dim if self.scalar == F32 else bpv)
        return g

    def vector_arena(self, n: int) -> np.ndarray:
        """Writable view of the first n rows of the index's own vector arena (storage format), after reserving
        n slots: export a graph's vectors straight into it, then import_graph() skips the copy."""
        if n > self.capacity():
            self.reserve(n)
        bpv = self.L.orc_bytes_per_vector(self.h)
        raw = (C.c_uint8 * (n * bpv)).from_address(self.L.orc_vectors(self.h))
        return np.frombuffer(raw, dtype=np.float32 if self.scalar == F32 else np.uint8).reshape(n, -1)

    def import_graph(self, g: dict):
        n = len(g["levels"])
        vec = np.ascontiguousarray(g["vectors"])
        upper = np.ascontiguousarray(g["upper"], dtype=np.uint32)
        if upper.size == 0:
            upper = np.zeros((1, self.M), dtype=np.uint32)
        self._check(self.L.orc_import_graph(
            self.h, n, _ptr(vec), _ptr(np.ascontiguousarray(g["levels"], dtype=np.int32)),
            _ptr(np.ascontiguousarray(g["keys"], dtype=np.uint64)),
            _ptr(np.ascontiguousarray(g["adj0"], dtype=np.uint32)),
            _ptr(np.ascontiguousarray(g["upper_off"], dtype=np.uint32)), _ptr(upper),
            int(g["max_level"]), int(g["entry_slot"])))
