"""Worker of tests/test_gpu_ranks.py::test_two_ranks_share_one_device_over_the_host_exchange -- started twice by
torch.distributed.run (gloo), both ranks on GPU 0, libvs_ranks with VS_RANKS_EXCHANGE=hostshm.  Each rank builds the shard
of its key range, then the NATIVE sharded path runs with world = 2: in-place packed blocks, exchange, packed merge, the
two-slot pipeline.  Checked on every rank: the merged exact search equals a float64 brute force over ALL rows; the merged
HNSW search equals the numpy merge (sharded.merge_topk_reference) of the two shards' own local answers; both ranks hold
the same answer; pipelined == synchronous."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import vector_store_amd as vs
    from vector_store_amd import ranks, sharded
    total, dim, nq, k = 60000, 64, 300, 10
    rng = np.random.default_rng(77)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    base = (rng.standard_normal((total, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((total, dim))).astype(np.float32)
    qh = (rng.standard_normal((nq, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((nq, dim))).astype(np.float32)
    lo, hi = sharded.key_range(rank, world, total)
    ix = vs.HipUsearchIndex(dim, vs.L2SQ, expansion_search=96)
    ix.reserve(hi - lo)
    ix.add_batch(np.arange(lo, hi, dtype=np.uint64), base[lo:hi])
    q = torch.from_numpy(qh).cuda()
    rs = ranks.RankedSearcher(ix, q, k, dist, total)
    info = rs.ranks.comm_info()
    assert info == {"rank": rank, "world": world, "comm_ranks": world, "rccl_ranks": 0, "exchange": "hostshm"}, info
    assert rs.ranks.key_range() == (lo, hi)
    # exact: merged answer == brute force over all rows
    truth = rs.exact().view(np.uint64)
    d_all = ((qh[:, None, :].astype(np.float64) - base[None, :, :].astype(np.float64)) ** 2).sum(-1)
    want = np.argsort(d_all, axis=1, kind="stable")[:, :k]
    assert np.array_equal(truth, want.astype(np.uint64)), "merged exact search differs from the brute force"
    # HNSW: merged == numpy merge of the shards' own local answers (gathered over gloo)
    lk, ld, _ = ix.search_batch(qh, k)
    gk = [torch.empty((nq, k), dtype=torch.int64) for _ in range(world)]
    gd = [torch.empty((nq, k), dtype=torch.float32) for _ in range(world)]
    dist.all_gather(gk, torch.from_numpy(lk.view(np.int64)))
    dist.all_gather(gd, torch.from_numpy(ld))
    mk, md = sharded.merge_topk_reference(np.stack([t.numpy().view(np.uint64) for t in gk]), np.stack([t.numpy() for t in gd]), k)
    rs.step_sync()
    torch.cuda.synchronize()
    sk, sd = rs.keys.cpu().numpy().view(np.uint64).copy(), rs.dists.cpu().numpy().copy()
    assert np.array_equal(sk, mk) and np.array_equal(sd, md), "merged walk differs from the numpy merge of the local answers"
    for _ in range(7):  # pipelined: two batches in flight, the exchange of batch i overlaps the walk of batch i + 1
        rs.step()
    rs.flush()
    assert np.array_equal(rs.keys.cpu().numpy().view(np.uint64), mk) and np.array_equal(rs.dists.cpu().numpy(), md)
    # every rank holds the same merged answer
    t = torch.from_numpy(sk.view(np.int64)).clone()
    dist.broadcast(t, src=0)
    assert np.array_equal(t.numpy().view(np.uint64), sk)
    assert rs.ranks.unanswered() == 0
    recall = float(np.mean([len(set(want[i].tolist()) & set(sk[i].tolist())) / k for i in range(nq)]))
    dist.barrier()
    rs.ranks.close()
    if rank == 0:
        print(json.dumps({"ok": True, "world": world, "exchange": info["exchange"], "recall_at_10": recall}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
