"""Key-range sharded search across the GPUs of one node (one process per GPU).

Each rank owns an independent HNSW graph over its key range.  A query batch is searched on every
shard; the per-shard top-k lists (k x (u64 key, f32 distance) per query) are exchanged with ONE
all-gather per batch of packed blocks [nq x k keys u64 | nq x k distances f32] (torch.distributed; backend "nccl" is
RCCL) and merged on every rank by the HIP kernel behind vs_topk_merge_packed_device (include/vs_hnsw.h).  This module is
the torch.distributed TWIN of the native path (csrc/ranks.cpp, libvs_ranks.so: the same block layout and the same single
collective, issued by the library itself) -- what the product binds is the native library; the twin serves the CPU tests.  The payload is tiny (nq*k*12 B per
rank), so the collective is latency-bound: one collective per batch, never per query
(SURVEY.md section 5 "Distributed communication backend").

The reference has no analogue (a query touches exactly one partition index, reference
crates/vector-store/src/vs_index/usearch.rs:787-803); this is the C4 configuration of BASELINE.json.
"""
from __future__ import annotations

import numpy as np
import torch


def key_range(rank: int, world: int, total: int) -> tuple[int, int]:
    """[lo, hi) of the row indices (low 48 bits of the PrimaryId) owned by `rank`."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def owner_of(key: int, world: int, total: int) -> int:
    per = (total + world - 1) // world
    return min(int(key & ((1 << 48) - 1)) // per, world - 1)


def merge_topk_reference(part_keys: np.ndarray, part_dist: np.ndarray, k: int):
    """numpy statement of vs_topk_merge_device: parts x nq x k -> nq x k, ascending (distance, part, pos)."""
    parts, nq, _ = part_keys.shape
    keys = np.full((nq, k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    dist = np.full((nq, k), np.inf, dtype=np.float32)
    for q in range(nq):
        fk = part_keys[:, q, :].reshape(-1)
        fd = part_dist[:, q, :].reshape(-1)
        valid = fk != np.uint64(0xFFFFFFFFFFFFFFFF)
        order = np.argsort(np.where(valid, fd, np.inf), kind="stable")[:k]
        order = order[valid[order]]
        keys[q, : len(order)] = fk[order]
        dist[q, : len(order)] = fd[order]
    return keys, dist


def block_bytes(nq: int, k: int) -> int:
    """Size of one rank's packed block in the gather buffer of csrc/ranks.cpp: [nq x k keys u64 | nq x k distances f32],
    rounded up to 16 bytes."""
    return (nq * k * 12 + 15) // 16 * 16


def pack_block(keys: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """numpy statement of what a rank's walk writes into its block (ranks.cpp `mine`)."""
    nq, k = keys.shape
    out = np.zeros(block_bytes(nq, k), dtype=np.uint8)
    out[: nq * k * 8] = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1).view(np.uint8)
    out[nq * k * 8: nq * k * 12] = np.ascontiguousarray(dist, dtype=np.float32).reshape(-1).view(np.uint8)
    return out


def merge_packed_reference(blocks: np.ndarray, parts: int, nq: int, k: int):
    """numpy statement of vs_topk_merge_packed_device over the receive buffer of the one all-gather: `parts` packed blocks."""
    bb = block_bytes(nq, k)
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(parts, bb)
    pk = np.stack([blocks[p, : nq * k * 8].view(np.uint64).reshape(nq, k) for p in range(parts)])
    pd = np.stack([blocks[p, nq * k * 8: nq * k * 12].view(np.float32).reshape(nq, k) for p in range(parts)])
    return merge_topk_reference(pk, pd, k)


class ShardedSearcher:
    """search(batch) over all shards = local search -> ONE all-gather of packed blocks -> merge: the torch.distributed twin
    of csrc/ranks.cpp (same block layout, same single collective per batch), kept for the gloo tests and for backends
    where the native library cannot run.

    `local_search(exact) -> (keys int64 [nq,k], dist f32 [nq,k])` and `merge(blocks u8 [world, block_bytes], out_k, out_d)`
    default to the HIP engine; tests on CPU (gloo) inject numpy stand-ins (merge_packed_reference) to exercise the
    collective plumbing only.
    """

    def __init__(self, ix, queries: torch.Tensor, k: int, dist, vs=None, local_search=None, merge=None):
        self.ix, self.q, self.k, self.dist, self.vs = ix, queries, k, dist, vs
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        nq, dev = queries.shape[0], queries.device
        self.bb = block_bytes(nq, k)
        self.gathered = torch.zeros((self.world, self.bb), dtype=torch.uint8, device=dev)  # the all-gather's receive buffer
        mine = self.gathered[self.rank]
        # this rank's block, viewed as the two arrays the walk writes (in place, as ranks.cpp does)
        self.lkeys = mine[: nq * k * 8].view(torch.int64).view(nq, k)
        self.ldist = mine[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
        self.lfound = torch.empty((nq,), dtype=torch.int32, device=dev)
        self.keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
        self.dists = torch.empty((nq, k), dtype=torch.float32, device=dev)
        self.found = torch.empty((nq,), dtype=torch.int32, device=dev)
        self._local = local_search or self._hip_local
        self._merge = merge or self._hip_merge

    def _hip_local(self, exact: bool):
        s = torch.cuda.current_stream().cuda_stream
        fn = self.ix.exact_search_batch_device if exact else self.ix.search_batch_device
        fn(self.q.data_ptr(), self.q.shape[0], self.k, self.lkeys.data_ptr(), self.ldist.data_ptr(),
           self.lfound.data_ptr(), s)
        return self.lkeys, self.ldist

    def _hip_merge(self, blocks, out_k, out_d):
        s = torch.cuda.current_stream().cuda_stream
        rc = self.vs.lib().vs_topk_merge_packed_device(blocks.data_ptr(), self.world, self.bb, self.q.shape[0], self.k, out_k.data_ptr(),
                                                       out_d.data_ptr(), self.found.data_ptr(), s)
        if rc != 0:
            raise RuntimeError(self.vs.lib().vs_hnsw_last_error().decode())

    def _gather(self):
        if self.dist is None or self.world == 1:
            return
        mine = self.gathered[self.rank]
        if self.dist.get_backend() == "nccl":
            self.dist.all_gather_into_tensor(self.gathered, mine)  # in place: the send block aliases its slot of the receive buffer
        else:  # gloo (CPU tests)
            self.dist.all_gather(list(self.gathered.unbind(0)), mine.clone())

    def _run(self, exact: bool):
        lk, ld = self._local(exact)
        if lk.data_ptr() != self.lkeys.data_ptr():  # an injected local search returned its own arrays
            self.lkeys.copy_(lk)
            self.ldist.copy_(ld)
        self._gather()
        self._merge(self.gathered, self.keys, self.dists)

    def step(self):
        self._run(False)

    def exact(self) -> np.ndarray:
        """Global exact top-k keys (ground truth for recall of the sharded index)."""
        self._run(True)
        if self.q.is_cuda:
            torch.cuda.synchronize()
        return self.keys.cpu().numpy().copy()
