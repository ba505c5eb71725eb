"""Key-range sharded search across the GPUs of one node (one process per GPU).

Each rank owns an independent HNSW graph over its key range.  A query batch is searched on every
shard; the per-shard top-k lists (k x (u64 key, f32 distance) per query) are exchanged with ONE
RCCL all-gather per batch (torch.distributed backend "nccl") and merged on every rank by the
HIP kernel behind vs_topk_merge_device (include/vs_hnsw.h).  The payload is tiny (nq*k*12 B per
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


class ShardedSearcher:
    """search(batch) over all shards = local search -> all-gather -> merge.

    `local_search(exact) -> (keys int64 [nq,k], dist f32 [nq,k])` and `merge(gk, gd, out_k, out_d)`
    default to the HIP engine; tests on CPU (gloo) inject numpy stand-ins to exercise the
    collective plumbing only.
    """

    def __init__(self, ix, queries: torch.Tensor, k: int, dist, vs=None, local_search=None, merge=None):
        self.ix, self.q, self.k, self.dist, self.vs = ix, queries, k, dist, vs
        self.world = dist.get_world_size() if dist is not None else 1
        nq, dev = queries.shape[0], queries.device
        self.lkeys = torch.empty((nq, k), dtype=torch.int64, device=dev)
        self.ldist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        self.lfound = torch.empty((nq,), dtype=torch.int32, device=dev)
        self.gkeys = torch.empty((self.world, nq, k), dtype=torch.int64, device=dev)
        self.gdist = torch.empty((self.world, nq, k), dtype=torch.float32, device=dev)
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

    def _hip_merge(self, gk, gd, out_k, out_d):
        s = torch.cuda.current_stream().cuda_stream
        self.vs.topk_merge_device(gk.data_ptr(), gd.data_ptr(), self.world, self.q.shape[0], self.k, out_k.data_ptr(),
                                  out_d.data_ptr(), self.found.data_ptr(), s)

    def _gather(self, lk, ld):
        if self.dist is None or self.world == 1:
            self.gkeys[0].copy_(lk)
            self.gdist[0].copy_(ld)
            return
        if self.dist.get_backend() == "nccl":
            self.dist.all_gather_into_tensor(self.gkeys, lk)
            self.dist.all_gather_into_tensor(self.gdist, ld)
        else:  # gloo (CPU tests)
            self.dist.all_gather(list(self.gkeys.unbind(0)), lk)
            self.dist.all_gather(list(self.gdist.unbind(0)), ld)

    def _run(self, exact: bool):
        lk, ld = self._local(exact)
        self._gather(lk, ld)
        self._merge(self.gkeys, self.gdist, self.keys, self.dists)

    def step(self):
        self._run(False)

    def exact(self) -> np.ndarray:
        """Global exact top-k keys (ground truth for recall of the sharded index)."""
        self._run(True)
        if self.q.is_cuda:
            torch.cuda.synchronize()
        return self.keys.cpu().numpy().copy()
