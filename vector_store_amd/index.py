"""ctypes binding of libvs_hnsw.so and the host-side mirror of the reference's `UsearchIndex` trait.

Reference interface mirrored (crates/vector-store/src/vs_index/usearch.rs:142-160, 180-251):
    reserve(size) / capacity() / add(primary_id, vector) / remove(primary_id) -> bool /
    search(vector, limit) / filtered_search(vector, limit, filter) / stop()
Errors surface as VsError (the reference wraps every usearch call in anyhow::Result).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# concurrent filtered searches are one small kernel launch per caller: see HwQueuesDefault in csrc/engine.hip (only effective
# while the HIP runtime has not initialised yet -- importing torch does not initialise it, the first CUDA call does)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
COS, L2SQ, IP, HAMMING = 0, 1, 2, 3
METRICS = {"cos": COS, "l2sq": L2SQ, "ip": IP, "hamming": HAMMING}
F32, F16, BF16, I8, B1 = 0, 1, 2, 3, 4
SCALARS = {"f32": F32, "f16": F16, "bf16": BF16, "i8": I8, "b1": B1}
FREE_KEY = 0xFFFFFFFFFFFFFFFF

PRED = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_void_p)
DONE = C.CFUNCTYPE(None, C.c_void_p, C.c_int)


class VsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


class _Options(C.Structure):
    _fields_ = [("dimensions", C.c_size_t), ("connectivity", C.c_size_t), ("expansion_add", C.c_size_t),
                ("expansion_search", C.c_size_t), ("metric", C.c_int), ("quantization", C.c_int),
                ("device", C.c_int), ("reserved", C.c_int)]


class _GraphInfo(C.Structure):
    _fields_ = [("slots", C.c_size_t), ("upper_blocks", C.c_size_t), ("max_level", C.c_int32),
                ("entry_slot", C.c_uint32), ("connectivity", C.c_size_t), ("connectivity_base", C.c_size_t)]


def lib_path() -> str:
    # VS_HNSW_LIB: an instrumented build of the engine alone; VS_LIB_DIR: a directory with instrumented builds of every library (development)
    return os.environ.get("VS_HNSW_LIB") or os.path.join(os.environ.get("VS_LIB_DIR") or _HERE, "libvs_hnsw.so")


_lib = None


def lib():
    """Loads libvs_hnsw.so.  torch (if it is going to be used in this process) must be imported
    first: it bundles its own libamdhip64.so.7 and two HIP runtimes in one process do not mix."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise VsError(-6, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    if "torch" not in sys.modules and os.environ.get("VS_HNSW_NO_TORCH") != "1":
        try:  # keep a single HIP runtime in the process (see docstring)
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp, sz, u64, f32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_float
    L.vs_hnsw_version.restype = C.c_char_p
    L.vs_hnsw_last_error.restype = C.c_char_p
    L.vs_hnsw_create.argtypes = [C.POINTER(_Options), C.POINTER(vp)]
    L.vs_hnsw_free.argtypes = [vp]
    L.vs_hnsw_reserve.argtypes = [vp, sz, sz]
    L.vs_hnsw_capacity.restype = sz
    L.vs_hnsw_capacity.argtypes = [vp]
    L.vs_hnsw_size.restype = sz
    L.vs_hnsw_size.argtypes = [vp]
    L.vs_hnsw_bytes_per_vector.restype = sz
    L.vs_hnsw_bytes_per_vector.argtypes = [vp]
    L.vs_hnsw_add.argtypes = [vp, u64, vp, sz]
    L.vs_hnsw_add_batch.argtypes = [vp, vp, vp, sz, sz]
    L.vs_hnsw_add_batch_device.argtypes = [vp, vp, vp, sz, sz]
    L.vs_hnsw_remove.argtypes = [vp, u64, C.POINTER(C.c_int)]
    L.vs_hnsw_search.argtypes = [vp, vp, sz, sz, vp, vp, C.POINTER(sz)]
    L.vs_hnsw_search_async.argtypes = [vp, vp, sz, sz, vp, vp, C.POINTER(sz), DONE, vp]
    L.vs_hnsw_filtered_search.argtypes = [vp, vp, sz, sz, PRED, vp, vp, vp, C.POINTER(sz)]
    L.vs_hnsw_search_batch.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp]
    L.vs_hnsw_exact_search_batch.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp]
    L.vs_hnsw_search_batch_device.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp]
    L.vs_hnsw_exact_search_batch_device.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp]
    L.vs_hnsw_set_expansion_search.argtypes = [vp, sz]
    L.vs_hnsw_stats.argtypes = [vp, vp, C.c_int]
    L.vs_hnsw_memory_info.argtypes = [vp, vp]
    L.vs_hnsw_filter_stats.argtypes = [vp, vp]
    # (VS_HNSW_LIB may name an older build for A/B measurements: symbols younger than round 3 are bound only where they exist,
    # and their accessors below return zeros without them)
    if hasattr(L, "vs_hnsw_filtered_search_keyed"):
        L.vs_hnsw_filtered_search_keyed.argtypes = [vp, vp, sz, sz, PRED, vp, u64, vp, vp, C.POINTER(sz)]
    if hasattr(L, "vs_hnsw_filter_forget"):
        L.vs_hnsw_filter_forget.argtypes = [vp, u64, C.POINTER(sz)]
        L.vs_hnsw_filter_forget_keys.argtypes = [vp, vp, sz]
    for young in ("vs_hnsw_filter_ask_stats", "vs_hnsw_pipe_stats", "vs_hnsw_filter_batch_stats", "vs_hnsw_pod_stats", "vs_hnsw_modify_stats", "vs_hnsw_filter_memo_stats", "vs_hnsw_call_stats"):
        if hasattr(L, young):
            getattr(L, young).argtypes = [vp, vp]
    if hasattr(L, "vs_hnsw_streams_created"):  # (VS_HNSW_LIB may name an older build: A/B measurements)
        L.vs_hnsw_streams_created.argtypes = []
        L.vs_hnsw_streams_created.restype = C.c_uint64
    L.vs_hnsw_exact_stats.argtypes = [vp, vp]
    L.vs_hnsw_walk_info.argtypes = [vp, vp]
    L.vs_hnsw_exact_stats2.argtypes = [vp, vp]
    if hasattr(L, "vs_hnsw_exact_stats3"):
        L.vs_hnsw_exact_stats3.argtypes = [vp, vp]
    L.vs_hnsw_graph_info_get.argtypes = [vp, C.POINTER(_GraphInfo)]
    L.vs_hnsw_export_graph.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.vs_hnsw_import_graph.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, sz, C.c_int32, C.c_uint32]
    L.vs_topk_merge_device.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp]
    L.vs_topk_merge_packed_device.argtypes = [vp, sz, sz, sz, sz, vp, vp, vp, vp]
    L.vs_f32_to_b1x8.argtypes = [vp, sz, vp]
    L.vs_distance_valid.argtypes = [f32, C.c_int, sz]
    L.vs_similarity_score.restype = f32
    L.vs_similarity_score.argtypes = [f32, C.c_int, sz]
    _lib = L
    return L


def streams_created() -> int:
    """HIP streams the engine has created in this process (a fixed set per device, never one per index)."""
    return int(lib().vs_hnsw_streams_created()) if hasattr(lib(), "vs_hnsw_streams_created") else 0


def version() -> str:
    return lib().vs_hnsw_version().decode()


def _check(rc: int):
    if rc != 0:
        raise VsError(rc, lib().vs_hnsw_last_error().decode())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def f32_to_b1x8(v) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros((v.size + 7) // 8, dtype=np.uint8)
    lib().vs_f32_to_b1x8(_p(v), v.size, _p(out))
    return out


def distance_valid(value: float, metric: int, dim: int = 0) -> bool:
    return bool(lib().vs_distance_valid(value, metric, dim))


def similarity_score(distance: float, metric: int, dim: int = 0) -> float:
    return float(lib().vs_similarity_score(distance, metric, dim))


def topk_merge_device(part_keys_ptr: int, part_dist_ptr: int, parts: int, nq: int, k: int, keys_ptr: int,
                      dist_ptr: int, found_ptr: int, stream: int = 0):
    _check(lib().vs_topk_merge_device(part_keys_ptr, part_dist_ptr, parts, nq, k, keys_ptr, dist_ptr, found_ptr,
                                      stream))


class HipUsearchIndex:
    """`impl UsearchIndex` over the HIP engine (what ThreadedUsearchIndex is over usearch)."""

    def __init__(self, dimensions: int, metric: int = COS, connectivity: int = 16, expansion_add: int = 128,
                 expansion_search: int = 64, quantization: int = F32, device: int = -1, _stress: int = 0):
        self.L = lib()
        if quantization == B1:  # reference metric_kind(): B1 => Hamming (usearch.rs:450-457)
            metric = HAMMING
        self.dim, self.metric, self.scalar = int(dimensions), int(metric), int(quantization)
        self.M = connectivity or 16
        self.M0 = 2 * self.M
        o = _Options(dimensions, connectivity, expansion_add, expansion_search, metric, quantization, device, _stress)
        h = C.c_void_p()
        _check(self.L.vs_hnsw_create(C.byref(o), C.byref(h)))
        self.h = h

    def __del__(self):
        self.stop()

    # --- trait UsearchIndex -------------------------------------------------------------
    def stop(self):
        if getattr(self, "h", None):
            self.L.vs_hnsw_free(self.h)
            self.h = None

    def reserve(self, size: int, threads: int = 0):
        _check(self.L.vs_hnsw_reserve(self.h, size, threads))

    def capacity(self) -> int:
        return self.L.vs_hnsw_capacity(self.h)

    def size(self) -> int:
        return self.L.vs_hnsw_size(self.h)

    def bytes_per_vector(self) -> int:
        return self.L.vs_hnsw_bytes_per_vector(self.h)

    def add(self, primary_id: int, vector):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        _check(self.L.vs_hnsw_add(self.h, primary_id, _p(v), v.size))

    def remove(self, primary_id: int) -> bool:
        r = C.c_int(0)
        _check(self.L.vs_hnsw_remove(self.h, primary_id, C.byref(r)))
        return bool(r.value)

    def search(self, vector, limit: int):
        v = np.ascontiguousarray(vector, dtype=np.float32)
        keys = np.zeros(limit, dtype=np.uint64)
        d = np.zeros(limit, dtype=np.float32)
        found = C.c_size_t(0)
        _check(self.L.vs_hnsw_search(self.h, _p(v), v.size, limit, _p(keys), _p(d), C.byref(found)))
        return keys[: found.value], d[: found.value]

    def search_async(self, vector, limit: int, on_done):
        """Non-blocking search: `on_done(keys, distances, status)` runs on the engine's dispatcher thread."""
        v = np.ascontiguousarray(vector, dtype=np.float32)
        keys = np.zeros(limit, dtype=np.uint64)
        d = np.zeros(limit, dtype=np.float32)
        found = C.c_size_t(0)
        holder = {}

        def _done(_ctx, status):
            cb = holder.pop("cb", None)  # keeps the ctypes thunk alive until it has run
            on_done(keys[: found.value], d[: found.value], status)
            del cb

        holder["cb"] = DONE(_done)
        holder["bufs"] = (keys, d, found)
        _check(self.L.vs_hnsw_search_async(self.h, _p(v), v.size, limit, _p(keys), _p(d), C.byref(found), holder["cb"],
                                           None))
        return holder

    def filtered_search(self, vector, limit: int, predicate, filter_key: int = 0):
        """filter_key != 0: the filter has a name (a fingerprint of its restrictions) -- the engine remembers verdicts across the
        queries that carry it (include/vs_hnsw.h: vs_hnsw_filtered_search_keyed)."""
        v = np.ascontiguousarray(vector, dtype=np.float32)
        keys = np.zeros(limit, dtype=np.uint64)
        d = np.zeros(limit, dtype=np.float32)
        found = C.c_size_t(0)
        cb = PRED(lambda key, _ctx: 1 if predicate(key) else 0)
        if filter_key:
            _check(self.L.vs_hnsw_filtered_search_keyed(self.h, _p(v), v.size, limit, cb, None, filter_key, _p(keys), _p(d), C.byref(found)))
        else:
            _check(self.L.vs_hnsw_filtered_search(self.h, _p(v), v.size, limit, cb, None, _p(keys), _p(d),
                                                  C.byref(found)))
        return keys[: found.value], d[: found.value]

    def call_stats(self) -> dict:
        """Where single-query calls spend their time (include/vs_hnsw.h: vs_hnsw_call_stats), totals in ms."""
        out = np.zeros(8, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_call_stats"):
            _check(self.L.vs_hnsw_call_stats(self.h, _p(out)))
        return {"searches": int(out[0]), "search_ms": int(out[1]) / 1e6, "filtered": int(out[2]), "filtered_ms": int(out[3]) / 1e6,
                "filtered_device_wait_ms": int(out[4]) / 1e6, "filtered_predicate_ms": int(out[5]) / 1e6, "flush_wait_ms": int(out[6]) / 1e6}

    def filter_forget(self, filter_key: int = 0) -> int:
        """The named filter starts over (0: every named filter of the index); returns the memories dropped."""
        dropped = C.c_size_t(0)
        _check(self.L.vs_hnsw_filter_forget(self.h, filter_key, C.byref(dropped)))
        return dropped.value

    def filter_forget_keys(self, keys) -> None:
        """Every named filter forgets its verdicts for these members: what the host calls after rewriting their filtering columns
        (the reference's `update_columns`, table/mod.rs:676-695)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        _check(self.L.vs_hnsw_filter_forget_keys(self.h, _p(keys), keys.size))

    def filter_ask_stats(self) -> dict:
        """Unnamed filters: the walk that asks while it runs (include/vs_hnsw_debug.h: vs_hnsw_filter_ask_stats)."""
        out = np.zeros(8, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_filter_ask_stats"):
            _check(self.L.vs_hnsw_filter_ask_stats(self.h, _p(out)))
        return {"queries": int(out[0]), "handed_over": int(out[1]), "no_pod": int(out[2]), "predicate_calls": int(out[3]),
                "device_waits": int(out[4]), "device_wait_ms": int(out[5]) / 1e5, "hops": int(out[6]), "device_walk_ms": int(out[7]) / 1e5}

    def filter_memo_stats(self) -> dict:
        out = np.zeros(6, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_filter_memo_stats"):
            _check(self.L.vs_hnsw_filter_memo_stats(self.h, _p(out)))
        return {"queries": int(out[0]), "verdicts_asked": int(out[1]), "memories_created": int(out[2]), "memories_held": int(out[3]),
                "forget_calls": int(out[4]), "members_forgotten": int(out[5])}

    # --- bulk / device paths used by the benchmark driver --------------------------------
    def add_batch(self, keys, vectors):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        n = keys.size
        dim = vectors.shape[-1] if vectors.ndim > 1 else (vectors.size // max(n, 1))
        _check(self.L.vs_hnsw_add_batch(self.h, _p(keys), _p(vectors), n, dim))

    def add_batch_device(self, keys, d_vectors_ptr: int, n: int, dim: int):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        _check(self.L.vs_hnsw_add_batch_device(self.h, _p(keys), d_vectors_ptr, n, dim))

    def _batch(self, fn, queries, k):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        nq = q.shape[0]
        keys = np.zeros((nq, k), dtype=np.uint64)
        d = np.zeros((nq, k), dtype=np.float32)
        found = np.zeros(nq, dtype=np.uint64)
        _check(fn(self.h, _p(q), nq, q.shape[1], k, _p(keys), _p(d), _p(found)))
        return keys, d, found.astype(np.int64)

    def search_batch(self, queries, k: int):
        return self._batch(self.L.vs_hnsw_search_batch, queries, k)

    def exact_search_batch(self, queries, k: int):
        return self._batch(self.L.vs_hnsw_exact_search_batch, queries, k)

    def search_batch_device(self, d_queries: int, nq: int, k: int, d_keys: int, d_dist: int, d_found: int,
                            stream: int = 0):
        _check(self.L.vs_hnsw_search_batch_device(self.h, d_queries, nq, self.dim, k, d_keys, d_dist, d_found,
                                                  stream))

    def exact_search_batch_device(self, d_queries: int, nq: int, k: int, d_keys: int, d_dist: int, d_found: int,
                                  stream: int = 0):
        _check(self.L.vs_hnsw_exact_search_batch_device(self.h, d_queries, nq, self.dim, k, d_keys, d_dist, d_found,
                                                        stream))

    def set_expansion_search(self, ef: int):
        _check(self.L.vs_hnsw_set_expansion_search(self.h, ef))

    def stats(self, reset: bool = False) -> dict:
        out = np.zeros(8, dtype=np.uint64)
        _check(self.L.vs_hnsw_stats(self.h, _p(out), int(reset)))
        names = ["search_evals", "search_hops", "queries", "add_evals", "add_hops", "added", "visited_overflow",
                 "link_evals"]
        return {n: int(v) for n, v in zip(names, out)}

    def memory_info(self) -> dict:
        out = np.zeros(4, dtype=np.uint64)
        _check(self.L.vs_hnsw_memory_info(self.h, _p(out)))
        return {n: int(v) for n, v in zip(["bytes", "in_place_bytes", "chunks", "copied_bytes"], out)}

    def filter_stats(self) -> dict:
        out = np.zeros(2, dtype=np.uint64)
        _check(self.L.vs_hnsw_filter_stats(self.h, _p(out)))
        return {"lazy_rounds": int(out[0]), "lazy_predicate_calls": int(out[1])}

    def filter_batch_stats(self) -> dict:
        out = np.zeros(2, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_filter_batch_stats"):
            _check(self.L.vs_hnsw_filter_batch_stats(self.h, _p(out)))
        return {"batched_launches": int(out[0]), "batched_rounds": int(out[1])}

    def pod_stats(self) -> dict:
        """Resident launches of the pipelined walk (csrc/pipe_pod.hpp) that blocking callers post their queries to."""
        out = np.zeros(12, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_pod_stats"):
            _check(self.L.vs_hnsw_pod_stats(self.h, _p(out)))
        return {"pods_opened": int(out[0]), "pod_rounds": int(out[1]), "pods_open_on_device": int(out[2]), "pods_enabled": bool(out[3]),
                "plain_queries": int(out[4]), "plain_ns": int(out[5]), "plain_wait_ns": int(out[6]), "plain_device_ns": int(out[7]),
                "filtered_answered": int(out[8]), "filtered_handed_over": int(out[9]), "rounds_without_a_pod": int(out[10]), "rounds_walked_again": int(out[11])}

    def pipe_stats(self) -> dict:
        out = np.zeros(2, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_pipe_stats"):
            _check(self.L.vs_hnsw_pipe_stats(self.h, _p(out)))
        return {"pipe_launches": int(out[0]), "lone_queries_handed_over": int(out[1])}

    def modify_stats(self) -> dict:
        """Where modifications spend their time (include/vs_hnsw.h: vs_hnsw_modify_stats)."""
        out = np.zeros(8, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_modify_stats"):
            _check(self.L.vs_hnsw_modify_stats(self.h, _p(out)))
        return {"flushes": int(out[0]), "vectors_flushed": int(out[1]), "flush_ms": int(out[2]) / 1e6, "pod_closings": int(out[3]),
                "pod_closing_ms": int(out[4]) / 1e6, "removes": int(out[5]), "remove_ms": int(out[6]) / 1e6, "pods_opened": int(out[7])}

    def exact_stats(self) -> dict:
        out = np.zeros(2, dtype=np.uint64)
        _check(self.L.vs_hnsw_exact_stats(self.h, _p(out)))
        out4 = np.zeros(4, dtype=np.uint64)
        _check(self.L.vs_hnsw_exact_stats2(self.h, _p(out4)))
        out8 = np.zeros(4, dtype=np.uint64)
        if hasattr(self.L, "vs_hnsw_exact_stats3"):
            _check(self.L.vs_hnsw_exact_stats3(self.h, _p(out8)))
        return {"block_batches": int(out[0]), "block_fallbacks": int(out[1]), "plane_batches": int(out4[0]), "plane_fallbacks": int(out4[1]),
                "plane8_batches": int(out8[0]), "plane8_fallbacks": int(out8[1]), "plane8_rho": float(np.array([int(out8[2])], dtype=np.uint32).view(np.float32)[0]),
                "plane8_rows": int(out8[3])}

    def walk_info(self) -> dict:
        out = np.zeros(2, dtype=np.uint64)
        _check(self.L.vs_hnsw_walk_info(self.h, _p(out)))
        return {"last_instance": None if int(out[0]) == 0xFFFFFFFFFFFFFFFF else int(out[0]), "ranked_fallbacks": int(out[1])}

    def graph_info(self) -> dict:
        gi = _GraphInfo()
        _check(self.L.vs_hnsw_graph_info_get(self.h, C.byref(gi)))
        return {f[0]: getattr(gi, f[0]) for f in _GraphInfo._fields_}

    def export_graph(self, vectors_out=None) -> dict:
        gi = self.graph_info()
        n, blocks = gi["slots"], gi["upper_blocks"]
        if vectors_out is not None:
            assert vectors_out.flags.c_contiguous and vectors_out.nbytes == n * self.bytes_per_vector()
        g = {
            # storage format: f32 rows for F32, raw bytes (bytes_per_vector per row) otherwise
            "vectors": vectors_out if vectors_out is not None
            else np.zeros((n, self.dim), dtype=np.float32) if self.scalar == F32
            else np.zeros((n, self.bytes_per_vector()), dtype=np.uint8),
            "levels": np.zeros(n, dtype=np.int32),
            "keys": np.zeros(n, dtype=np.uint64),
            "adj0": np.zeros((n, self.M0), dtype=np.uint32),
            "upper_off": np.zeros(n, dtype=np.uint32),
            "upper": np.zeros((max(blocks, 1), self.M), dtype=np.uint32),
        }
        _check(self.L.vs_hnsw_export_graph(self.h, _p(g["vectors"]), _p(g["levels"]), _p(g["keys"]), _p(g["adj0"]),
                                           _p(g["upper_off"]), _p(g["upper"])))
        g["upper"] = g["upper"][:blocks]
        g["max_level"], g["entry_slot"] = gi["max_level"], gi["entry_slot"]
        return g

    def import_graph(self, g: dict):
        n = len(g["levels"])
        upper = np.ascontiguousarray(g["upper"], dtype=np.uint32)
        blocks = upper.shape[0] if upper.size else 0
        if blocks == 0:
            upper = np.zeros((1, self.M), dtype=np.uint32)
        _check(self.L.vs_hnsw_import_graph(
            self.h, n, _p(np.ascontiguousarray(g["vectors"], dtype=np.float32 if self.scalar == F32 else np.uint8)),
            _p(np.ascontiguousarray(g["levels"], dtype=np.int32)), _p(np.ascontiguousarray(g["keys"], dtype=np.uint64)),
            _p(np.ascontiguousarray(g["adj0"], dtype=np.uint32)),
            _p(np.ascontiguousarray(g["upper_off"], dtype=np.uint32)), _p(upper), blocks, int(g["max_level"]),
            int(g["entry_slot"])))
