"""ctypes binding of libvs_ranks.so (include/vs_ranks.h): key-range shards, one process per GPU, per-shard top-k
exchanged by ONE RCCL ncclAllGather per batch on the library's own stream and merged on every rank.

`RankedSearcher` is the step object bench.py times in `--mode shard`; the communicator id travels over
torch.distributed (any backend: it is 128 bytes), everything on the data path is native."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import index as _ix

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
ID_BYTES = 128


def lib():
    global _lib
    if _lib is None:
        _ix.lib()
        path = os.path.join(_HERE, "libvs_ranks.so")
        if not os.path.exists(path):
            raise _ix.VsError(-6, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.vs_ranks_unique_id.argtypes = [vp]
        L.vs_ranks_create.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.POINTER(vp)]
        L.vs_ranks_create_ex.argtypes = [vp, C.c_int, C.c_int, vp, u64, C.c_int, C.POINTER(vp)]
        L.vs_ranks_exchange_kind.argtypes = [vp]
        L.vs_ranks_rccl_ranks.argtypes = [vp, vp]
        L.vs_ranks_free.argtypes = [vp]
        L.vs_ranks_world.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vs_ranks_unanswered.argtypes = [vp, C.POINTER(u64)]
        L.vs_ranks_owner.argtypes = [vp, u64]
        L.vs_ranks_range.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.vs_ranks_add_batch.argtypes = [vp, vp, vp, sz, sz, C.POINTER(sz)]
        L.vs_ranks_search_batch_device.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp]
        L.vs_ranks_exact_search_batch_device.argtypes = [vp, vp, sz, sz, sz, vp, vp, vp, vp]
        L.vs_ranks_search_submit.argtypes = [vp, C.c_int, vp, sz, sz, sz, vp, vp, vp, vp]
        L.vs_ranks_wait.argtypes = [vp, C.c_int, vp]
        L.vs_ranks_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def unique_id() -> np.ndarray:
    out = np.zeros(ID_BYTES, dtype=np.uint8)
    rc = lib().vs_ranks_unique_id(out.ctypes.data)
    if rc != 0:
        raise _ix.VsError(rc, lib().vs_ranks_last_error().decode())
    return out


class Ranks:
    """This rank's member of the sharded index: `shard` (a HipUsearchIndex) + the RCCL communicator."""

    def __init__(self, shard, rank: int, world: int, comm_id: np.ndarray | None, total_rows: int):
        self.L = lib()
        self.shard, self.rank, self.world = shard, rank, world
        h = C.c_void_p()
        idp = np.ascontiguousarray(comm_id, dtype=np.uint8).ctypes.data if comm_id is not None else None
        self._check(self.L.vs_ranks_create(shard.h, rank, world, idp, total_rows, C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise _ix.VsError(rc, self.L.vs_ranks_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.vs_ranks_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def comm_info(self) -> dict:
        """rank / world the handle was created with; comm_ranks: ranks that joined its exchange; rccl_ranks: ncclCommCount of its RCCL
        communicator (0: the host exchange serves it)."""
        r, w, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self.L.vs_ranks_world(self.h, C.byref(r), C.byref(w), C.byref(c)))
        n = C.c_int(0)
        self._check(self.L.vs_ranks_rccl_ranks(self.h, C.byref(n)))
        return {"rank": r.value, "world": w.value, "comm_ranks": c.value, "rccl_ranks": n.value,
                "exchange": {0: "rccl", 1: "hostshm"}.get(self.L.vs_ranks_exchange_kind(self.h), "?")}

    def unanswered(self) -> int:
        v = C.c_uint64(0)
        self._check(self.L.vs_ranks_unanswered(self.h, C.byref(v)))
        return v.value

    def owner(self, key: int) -> int:
        return self.L.vs_ranks_owner(self.h, key)

    def key_range(self):
        lo, hi = C.c_uint64(0), C.c_uint64(0)
        self.L.vs_ranks_range(self.h, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def add_batch(self, keys, vectors) -> int:
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        vectors = np.ascontiguousarray(vectors, dtype=np.float32)
        added = C.c_size_t(0)
        self._check(self.L.vs_ranks_add_batch(self.h, keys.ctypes.data, vectors.ctypes.data, keys.size, vectors.shape[-1],
                                              C.byref(added)))
        return added.value

    def search_batch_device(self, d_q, nq, dim, k, d_keys, d_dist, d_found, stream=0, exact=False):
        fn = self.L.vs_ranks_exact_search_batch_device if exact else self.L.vs_ranks_search_batch_device
        self._check(fn(self.h, d_q, nq, dim, k, d_keys, d_dist, d_found, stream))

    def submit(self, slot, d_q, nq, dim, k, d_keys, d_dist, d_found, stream=0):
        self._check(self.L.vs_ranks_search_submit(self.h, slot, d_q, nq, dim, k, d_keys, d_dist, d_found, stream))

    def wait(self, slot, stream=0):
        self._check(self.L.vs_ranks_wait(self.h, slot, stream))


class RankedSearcher:
    """search(batch) over all shards through libvs_ranks: walk -> ncclAllGather -> merge, two batches in flight.
    `dist` (torch.distributed or None) only carries the 128-byte communicator id."""

    def __init__(self, ix, queries, k: int, dist, total_rows: int):
        import torch
        self.torch, self.ix, self.q, self.k = torch, ix, queries, k
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        comm_id = None
        if world > 1:
            t = torch.from_numpy(unique_id() if rank == 0 else np.zeros(ID_BYTES, dtype=np.uint8))
            t = t.to(queries.device) if dist.get_backend() == "nccl" else t
            dist.broadcast(t, src=0)
            comm_id = t.cpu().numpy()
        self.ranks = Ranks(ix, rank, world, comm_id, total_rows)
        nq, dev = queries.shape[0], queries.device
        self.out = [(torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), dtype=torch.float32, device=dev),
                     torch.empty((nq,), dtype=torch.int32, device=dev)) for _ in range(2)]
        self.keys, self.dists, self.found = self.out[0]
        self.turn = 0
        self.pending = None

    def _stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def step(self):
        """Pipelined: submits this batch, then makes the stream wait for the PREVIOUS batch's merge, whose results are
        then in self.keys / self.dists -- the all-gather and merge of a batch overlap the next batch's walk."""
        s = self._stream()
        k_, d_, f_ = self.out[self.turn]
        self.ranks.submit(self.turn, self.q.data_ptr(), self.q.shape[0], self.q.shape[1], self.k, k_.data_ptr(), d_.data_ptr(),
                          f_.data_ptr(), s)
        if self.pending is not None:
            self.ranks.wait(self.pending, s)
            self.keys, self.dists, self.found = self.out[self.pending]
        self.pending = self.turn
        self.turn ^= 1

    def flush(self):
        """Results of the last submitted batch."""
        if self.pending is not None:
            self.ranks.wait(self.pending, self._stream())
            self.keys, self.dists, self.found = self.out[self.pending]
            self.pending = None
        self.torch.cuda.synchronize()

    def step_sync(self):
        s = self._stream()
        k_, d_, f_ = self.out[0]
        self.ranks.search_batch_device(self.q.data_ptr(), self.q.shape[0], self.q.shape[1], self.k, k_.data_ptr(), d_.data_ptr(),
                                       f_.data_ptr(), s)
        self.keys, self.dists, self.found = self.out[0]

    def exact(self) -> np.ndarray:
        s = self._stream()
        k_, d_, f_ = self.out[0]
        self.ranks.search_batch_device(self.q.data_ptr(), self.q.shape[0], self.q.shape[1], self.k, k_.data_ptr(), d_.data_ptr(),
                                       f_.data_ptr(), s, exact=True)
        self.torch.cuda.synchronize()
        return k_.cpu().numpy().copy()
