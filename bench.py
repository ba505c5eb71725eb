#!/usr/bin/env python3
"""bench.py -- QPS at recall@10 >= 0.95 of the HNSW search hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` is authoritative.  Started WITHOUT a torch.distributed environment and N > 1, this process -- before it touches
a GPU -- starts the N ranks itself (a child `python -m torch.distributed.run` on 127.0.0.1, one rank per GPU), relays
rank 0's JSON line and exits with the children's status; it refuses (exit 2) when the box shows fewer than N GPUs.
Started by a launcher, every rank checks WORLD_SIZE == N and exits 2 otherwise.  `--dry-run` only brings the ranks up,
has them agree on who they are over the chosen backend and prints that (CPU test of the launcher: gloo).

stdout's LAST line is the driver's contract line and nothing else (`bench_line.py`: metric / value / ... / roofline / cpu_baseline plus
one number per boundary leg and side config, under 8 KB, asserted); the FULL record -- every leg's latency table, the mixed legs, the
side configs, the generators -- is written to `gpurun_out/bench_full.json` (`--full-record`), which the line names as `full_record`.

A "step" is one pass of the hot path (hnsw_search, one wavefront per query) over one batch of `--nq` synthetic queries
that are already resident in HBM; `--query-batches` (4) distinct batches rotate through the steps, so no step replays the
previous one's rows.  Workload at N=1 = the configuration BASELINE.json's metric is quoted on: 10M x 768 cosine, top-10,
one GPU (31 GB of vectors + 1.4 GB of graph in HBM); ef_search = the smallest beam width (coarse sweep 64,96,...,256,
320,...,512, then bisection in steps of 8) reaching recall@10 >= 0.95 against the exact brute-force ground truth computed
on the GPU.  `--vectors 1000000` is configs[1] (ef_search 128).

The N=1 default run also times, as `configs` side records with their own roofline blocks, BASELINE.json's configs[1]
(1M x 768 cos, ef_search 128), configs[4] (q = 256 batches over 10M x 768 inner product: the MFMA block-distance path)
and -- when the device has >= 180 GiB free -- configs[2] (10M x 1536 L2); `--configs` picks them.  The cpu_baseline
leg times the CPU restatement on the SAME graph and compares its ids with the GPU's for every query of the batch
(`cpu_baseline.id_parity`); a violation (ids that differ where the oracle sees no f32 near-tie) makes the run exit 3.

Multi-GPU (`--mode replica`, default): the reference scales by replication -- every vector-store process holds the whole
index -- so each rank builds a full replica and serves its own query stream; no data-path
collective; scaling "weak"; value = all ranks' queries per second.  With N>1 two sharded legs (SURVEY.md section 8e:
key-range shards, per-shard top-k all-gathered by ONE ncclAllGather per batch issued by libvs_ranks over RCCL, merged by
topk_merge_kernel) are also run and reported under "sharded": `weak` (--vectors per GPU, index = N x that) and
`fixed_total` (--vectors / N per GPU: the same index as N = 1, cut into N key ranges).  `--mode shard` makes the weak
sharded form the timed path.  `rccl_ranks` is ncclCommCount of the library's communicator (a run over N > 1 GPUs whose value is not N exits 4),
`comm_ranks` the ranks that joined its exchange, `sharded_{weak,fixed_total}_queries_per_s` the sharded legs beside the replica `value`.

Synthetic data: `--dist lowrank` (default) = 24-d Gaussian latent mapped by a fixed random 24 x dim matrix plus 0.05
isotropic noise: embedding-like local intrinsic dimension, on which HNSW reaches the recall target at the reference's
beam widths.  `--dist gaussian` is the i.i.d. generator of SURVEY.md section 8d, on which no graph index reaches useful
recall at dim 768.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")  # before the HIP runtime initialises: concurrent filtered searches (see csrc/engine.hip HwQueuesDefault)
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from bench_common import (ADJ_BYTES, BF16_PEAK_TFLOPS, HBM_ACHIEVABLE_GBS, HBM_PEAK_GBS, Searcher, build_index, effective_cores, hbm_roofline,  # noqa: E402,F401
                          make_data, recall_at_k, timed_steps)
from bench_side import (boundary_record, config_c3, config_c5, i8_callers_record, mixed_record, side_records, two_index_record)  # noqa: E402

def cpu_baseline(ix, queries_host, k, ef, seconds, gpu_keys, gpu_dist, extra_host=None, filtered_seconds=0.0, boundary_answers=None,
                 mixed_seconds=0.0, mixed_fresh=None, headroom=0, other_index=None):
    """The CPU restatement of the usearch algorithm (oracle/, kind "port") on the host cores of this box, on the
    SAME graph: searches one query per call from T threads (reference usearch.rs:212), then -- the build half of the
    metric -- inserts `extra_host` further vectors into that full-size index from T threads (usearch.rs:194-196).
    id_parity: the oracle's (keys, distances) for EVERY query of the batch against the GPU's, position by position, with
    the bar of tests/parity_util.py (counted here instead of asserted)."""
    import oracle
    from tests.parity_util import count_parity
    o = oracle.OracleIndex(ix.dim, ix.metric, 16, 128, ef, quantization=ix.scalar)
    slots = ix.graph_info()["slots"]
    extra = 0 if extra_host is None else len(extra_host)
    # (headroom: the mixed legs insert through the actor, whose growth step -- capacity + 1,000,000, usearch.rs:442 -- is a realloc of
    # the whole 30 GB arena in a CPU usearch: it would be the only thing a 2-second leg measures)
    o.reserve(slots + extra + headroom)
    g = ix.export_graph(vectors_out=o.vector_arena(slots))  # straight into the oracle's arena
    o.import_graph(g)
    slot_of_key = None if np.array_equal(g["keys"], np.arange(slots, dtype=np.uint64)) else {int(kk): i for i, kk in enumerate(g["keys"])}
    del g
    o.set_expansion_search(ef)
    threads = effective_cores()
    nq = queries_host.shape[0]
    o.search_batch(queries_host[: min(nq, 4 * threads)], k, threads=threads)  # page in, create contexts
    done, t0 = 0, time.perf_counter()
    while True:
        keys, dists, found = o.search_batch(queries_host, k, threads=threads)
        done += nq
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    out = {"value": done / el, "unit": "queries/s", "cores": threads, "kind": "port",
           "sample": f"{done} queries ({done // nq} passes over the bench batch) in {el:.1f}s on the GPU-built graph, "
                     f"ef_search={ef}, usearch-algorithm CPU restatement (not the usearch binary)"}

    def oracle_distance(qi, key):  # the oracle's own distance from query qi to the member the ENGINE returned
        s = int(key) if slot_of_key is None else slot_of_key.get(int(key), -1)
        return float("nan") if s < 0 or s >= slots else o.distance_to_slot(queries_host[qi], s)

    exact = ix.scalar in (oracle.I8, oracle.B1)
    out["id_parity"] = count_parity(gpu_keys, gpu_dist, keys, dists, found, oracle_distance, exact=exact)
    brief = ("rows", "identical_rows", "near_tie_positions", "violations")

    def parity_of_answers(rec, want_k, want_d, want_f, answered):
        """What a crowd of callers RECEIVED through the C ABI (boundary_record's recorded answers) against the oracle's answers for
        the same queries: every recorded row whose query the oracle answered."""
        if rec is None or len(rec["query"]) == 0:
            return None
        use = rec["query"] < answered
        qi = rec["query"][use].astype(np.int64)
        if len(qi) == 0:
            return None
        par = count_parity(rec["keys"][use], rec["distances"][use], want_k[qi], want_d[qi], want_f[qi],
                           lambda row, key: oracle_distance(int(qi[row]), key), exact=exact, got_found=rec["found"][use])
        return dict({kk: par[kk] for kk in brief}, first_violations=par["first_violations"][:2], distinct_queries=int(len(np.unique(qi))))

    out["boundary_id_parity"] = {}
    for leg, rec in (boundary_answers or {}).items():
        if not leg.startswith("filtered"):
            par = parity_of_answers(rec, keys, dists, found, nq)
            if par is not None:
                out["boundary_id_parity"][leg] = par
    # filtered_search on the same graph, same predicate and call pattern as boundary.filtered (one query per call from T threads,
    # reference filtered_ann usearch.rs:1107-1154): the CPU number beside the GPU's, and id parity of the engine's answers
    if filtered_seconds > 0:
        out["filtered"] = {}
        for name, modulus, nq_used in (("selectivity_10pct", 10, 2048), ("selectivity_1pct", 100, 256)):
            # (the first nq_used queries of the batch, to the end if the seconds allow: the set boundary.filtered draws from)
            fk, fd, ff, answered, calls, wall = o.filtered_search_timed(queries_host[:nq_used], k, modulus, threads=threads, seconds=filtered_seconds)
            rec = {"queries_per_s": answered / wall, "queries": answered, "seconds": wall, "threads": threads,
                   "predicate": f"key % {modulus} == 0", "predicate_calls_per_query": calls / max(answered, 1)}
            # the engine's answers for the first queries of the batch, through the C ABI with the same predicate, against the oracle's
            m = min(answered, 24)
            if m and not np.any(ff[:m] == 0):
                gk = np.zeros((m, k), dtype=np.uint64)
                gd = np.full((m, k), np.inf, dtype=np.float32)
                for i in range(m):
                    a_k, a_d = ix.filtered_search(queries_host[i], k, lambda key, mod=modulus: key % mod == 0)
                    gk[i, : len(a_k)] = a_k
                    gd[i, : len(a_d)] = a_d
                par = count_parity(gk, gd, fk[:m], fd[:m], ff[:m], oracle_distance, exact=exact)
                rec["id_parity"] = {kk: par[kk] for kk in ("rows", "identical_rows", "near_tie_positions", "violations")}
            for leg, brec in (boundary_answers or {}).items():
                if leg.startswith("filtered." + name) or leg.startswith("filtered_named." + name):
                    par = parity_of_answers(brec, fk, fd, ff, answered)
                    if par is not None:
                        out["boundary_id_parity"][leg] = par
            out["filtered"][name] = rec
    if extra:
        t0 = time.perf_counter()
        o.add_batch(np.arange(slots, slots + extra, dtype=np.uint64) + np.uint64(1 << 40), extra_host, threads=threads)
        el = time.perf_counter() - t0
        out["build_vectors_per_s"] = extra / el
        out["build_sample"] = f"{extra} further vectors inserted into the {slots}-vector index in {el:.1f}s, {threads} threads"
    # the reference's mixed workloads through the SAME dispatch actor (libvs_actor bound to the oracle: vs_actor_create_with) and the
    # same driver as boundary.mixed -- last: they modify the index
    if mixed_seconds > 0 and mixed_fresh is not None:
        from vector_store_amd import actor as _actor

        def actor_of():
            act = _actor.IndexActor(ix.dim, ix.metric, 16, 128, ef, workers=threads, quantization=ix.scalar, index_vtable=oracle.trait_vtable())
            act.adopt_partition(0, o.h, o.size())
            return act
        try:
            out["mixed"] = mixed_record(actor_of, queries_host, mixed_fresh, slots, mixed_seconds, named=False)  # (a CPU usearch has no verdict memory)
            if other_index is not None:  # the reference's own shape: searches on ANOTHER index (its own actor) beside the updates
                o_b = oracle.OracleIndex(ix.dim, ix.metric, 16, 128, ef, quantization=ix.scalar)
                o_b.import_graph(other_index.export_graph())
                act_b = _actor.IndexActor(ix.dim, ix.metric, 16, 128, ef, workers=threads, quantization=ix.scalar, index_vtable=oracle.trait_vtable())
                act_b.adopt_partition(0, o_b.h, o_b.size())
                act_a = actor_of()
                try:
                    out["mixed"]["two_indexes"] = two_index_record(act_a, act_b, queries_host, mixed_fresh, slots, mixed_seconds)
                finally:
                    act_a.stop()
                    act_b.stop()
        except Exception as e:  # noqa: BLE001
            out["mixed"] = dict(out.get("mixed") or {}, error=repr(e))
    return out, keys


def make_sharded_searcher(ix, queries, k, dist, vs, ranks, sharded, total_rows, backend):
    """The native path (libvs_ranks: ncclAllGather issued by the library) when RCCL is the backend -- or, with
    VS_RANKS_EXCHANGE=hostshm, the same library over its host-shared-memory exchange (several ranks on ONE device, where
    RCCL refuses to pair them: test boxes); the torch.distributed twin for gloo smoke tests without it, with
    VS_RANKS=torch, or when the native communicator cannot be created (every rank agrees on the choice through an
    all-reduce, so no rank is left waiting in a collective the others never enter)."""
    native = (backend == "nccl" or os.environ.get("VS_RANKS_EXCHANGE") == "hostshm") and os.environ.get("VS_RANKS", "native") != "torch"
    gs = None
    if dist is None:
        return ranks.RankedSearcher(ix, queries, k, None, total_rows)
    if native:
        # RankedSearcher's constructor is itself collective (ncclCommInitRank, or the host exchange's attach wait): a rank that cannot even
        # load the library must say so BEFORE the others enter it, or they wait in the bootstrap for a rank that never comes
        can_load = 1
        try:
            ranks.lib()
        except Exception as e:  # noqa: BLE001
            can_load = 0
            print(f"[bench] libvs_ranks does not load on rank {dist.get_rank()}: {e!r}", file=sys.stderr)
        agree = torch.tensor([can_load], device=queries.device if backend == "nccl" else "cpu")
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        native = int(agree.item()) == 1
    if native:
        try:
            gs = ranks.RankedSearcher(ix, queries, k, dist, total_rows)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] libvs_ranks unavailable on rank {dist.get_rank()}: {e!r}", file=sys.stderr)
        ok = torch.tensor([1 if gs is not None else 0], device=queries.device if backend == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            gs = None
    return gs if gs is not None else sharded.ShardedSearcher(ix, queries, k, dist, vs)


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a) -> int:
    """--gpus N > 1 without a torch.distributed environment: start the N ranks (no GPU call has happened in this process;
    torch.cuda.device_count() does not initialise the runtime on this image), relay their output, return their status."""
    have = torch.cuda.device_count()
    if not (a.same_device or a.dry_run) and have < a.gpus:
        print(json.dumps({"error": f"bench.py --gpus {a.gpus}: this box shows {have} GPU(s); refusing to report a {a.gpus}-GPU number from fewer devices",
                          "n_gpus_requested": a.gpus, "n_gpus_visible": have}), file=sys.stderr)
        return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(a, world, rank, local):
    """Brings the process group up on the chosen backend and has every rank report who it is: the launcher's CPU test."""
    import torch.distributed as dist
    me = {"rank": rank, "local_rank": local, "world_size": world, "pid": os.getpid(), "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}
    everyone = [me]
    if world > 1:
        dist.init_process_group(a.backend if a.backend != "nccl" or torch.cuda.is_available() else "gloo")
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": a.backend, "ranks": everyone}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vectors", "--n", dest="n", type=int, default=10_000_000,
                    help="vectors per GPU; default = the headline workload BASELINE.json's metric is quoted on (10M x 768 cosine); 1000000 = configs[1]")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=10_000, help="queries per step per GPU")
    ap.add_argument("--query-batches", type=int, default=4, help="distinct query batches rotated through the steps (recall / parity are measured on the first)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cos", choices=["cos", "l2sq", "ip"])
    ap.add_argument("--dist", default="lowrank", choices=["lowrank", "gaussian", "clustered"])
    ap.add_argument("--rank", type=int, default=24, help="latent dimension of the lowrank generator")
    ap.add_argument("--ef", type=int, default=0, help="expansion_search; 0 = smallest multiple of 8 (coarse sweep + bisection) with recall >= target")
    ap.add_argument("--target-recall", type=float, default=0.95)
    ap.add_argument("--quantization", default="f32", choices=["f32", "f16", "bf16", "i8", "b1"], help="storage type (usearch ScalarKind)")
    ap.add_argument("--mode", default="replica", choices=["replica", "shard"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall seconds of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--cpu-build-vectors", type=int, default=20_000, help="vectors the cpu_baseline leg inserts into the full-size index (0 = skip)")
    ap.add_argument("--boundary-seconds", type=float, default=3.0, help="seconds per leg of the through-the-C-ABI record (0 = skip)")
    ap.add_argument("--mixed-seconds", type=float, default=1.5, help="seconds per leg of boundary.mixed / cpu_baseline.mixed: the reference's pipeline benches through the dispatch actor (0 = skip)")
    ap.add_argument("--no-side-records", action="store_true", help="skip the generator side records (gaussian / clustered / rank sweep at 1M) and the configs side records")
    ap.add_argument("--configs", default="auto", help="BASELINE configs timed as side records at N=1: comma list of c2,c5,c3, 'none', or 'auto' (c2, c5, and c3 when >= 180 GiB of HBM is free)")
    ap.add_argument("--no-sharded-leg", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="testing aid: every rank uses GPU 0")
    ap.add_argument("--full-record", default="", help="where the FULL record (every leg's latency table, the mixed legs, side configs) is written, relative to the "
                    "repo root; default gpurun_out/bench_full.json.  stdout carries only the contract line (bench_line.py)")
    ap.add_argument("--dry-run", action="store_true", help="bring the ranks up, report their environment, measure nothing")
    a = ap.parse_args()

    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(json.dumps({"error": f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}: the two must agree", "n_gpus_requested": a.gpus,
                              "world_size": world}), file=sys.stderr)
        sys.exit(2)
    if a.dry_run:
        dry_run(a, world, rank, local)
        return
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    import vector_store_amd as vs  # after torch: one HIP runtime per process
    from vector_store_amd import ranks, sharded

    def barrier():
        if dist is not None:
            dist.barrier()

    def reduce_scalar(v, op):
        t = torch.tensor([v], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return float(t.item())

    n, dim, nq, k = a.n, a.dim, a.nq, a.k
    nb = max(1, a.query_batches)
    shard_mode = a.mode == "shard"  # (at N = 1: the same code path over a world of one -- no collective, same pack / merge / pipeline)
    # replica: same base on every rank, own queries; shard: own base (key range r*n..), same queries
    base = make_data(n, dim, a.dist, 1234 + (rank if shard_mode else 0), dev, a.rank)
    qseed = 4321 + (0 if shard_mode else rank)
    batches = [make_data(nq, dim, a.dist, qseed + 1000 * b, dev, a.rank) for b in range(nb)]
    queries = batches[0]
    keys = np.arange(n, dtype=np.uint64) + (np.uint64(rank * n) if shard_mode else np.uint64(0))
    ix, build_s = build_index(vs, base, keys, a.metric, quantization=a.quantization)
    del base
    torch.cuda.empty_cache()
    st = ix.stats(reset=True)
    build_info = {"vectors_per_s": n / build_s, "seconds": build_s, "vectors": n,
                  "evals_per_add": st["add_evals"] / max(st["added"], 1),
                  "insert_evals_per_add": (st["add_evals"] - st["link_evals"]) / max(st["added"], 1),  # hnsw_insert_kernel's share
                  "hops_per_add": st["add_hops"] / max(st["added"], 1)}
    se = Searcher(ix, batches, k)
    finish = lambda: None  # pipelined steppers: completes the batch still in flight
    rccl_ranks, comm_ranks, gs, exchange_kind = None, None, None, None
    rccl_failed = False
    if shard_mode:
        # native path (libvs_ranks: one ncclAllGather per batch on its own stream, overlapped with the next walk);
        # the torch.distributed twin only where RCCL is not the backend (gloo smoke tests)
        gs = make_sharded_searcher(ix, queries, k, dist, vs, ranks, sharded, n * world, a.backend)
        if hasattr(gs, "ranks"):
            ci = gs.ranks.comm_info()
            rccl_ranks, comm_ranks, exchange_kind = ci["rccl_ranks"], ci["comm_ranks"], ci["exchange"]
        truth = gs.exact()
        turn = [0]

        def step(batch=None):
            gs.q = batches[turn[0] % nb] if batch is None else batches[batch]
            if batch is None:
                turn[0] += 1
            gs.step()
        finish = getattr(gs, "flush", finish)
        result_keys = lambda: gs.keys.cpu().numpy()
        result_dist = lambda: gs.dists.cpu().numpy()
    else:
        truth, _ = se.exact()
        step = se.step
        result_keys = lambda: se.keys.cpu().numpy()
        result_dist = lambda: se.dist.cpu().numpy()

    # ---- beam width: smallest of {64,96,...,256,320,...,512} reaching the recall target
    sweep = []
    chosen = None

    def probe(ef):
        ix.set_expansion_search(ef)
        step(0)
        finish()
        torch.cuda.synchronize()
        r = recall_at_k(truth, result_keys())
        if dist is not None:
            r = reduce_scalar(r, dist.ReduceOp.MIN)
        sweep.append({"ef": ef, "recall": round(r, 4)})
        return r

    prev = None
    for ef in ([a.ef] if a.ef else [64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 512]):
        r = probe(ef)
        if r >= a.target_recall:
            chosen = (ef, r)
            break
        prev = ef
    if chosen is not None and prev is not None and not a.ef:  # refine between the last miss and the first hit, in steps of 8
        lo, hi = prev, chosen[0]
        while hi - lo > 8:
            mid = (lo + hi) // 16 * 8
            r = probe(mid)
            if r >= a.target_recall:
                hi, chosen = mid, (mid, r)
            else:
                lo = mid
    if chosen is None:
        chosen = (sweep[-1]["ef"], sweep[-1]["recall"])
    ef, recall = chosen
    ix.set_expansion_search(ef)

    # ---- timed region: W warmup steps, then exactly K steps between barrier + synchronize
    for _ in range(a.warmup):
        step()
    finish()
    torch.cuda.synchronize()
    ix.stats(reset=True)
    barrier()
    kernel_ms, elapsed = timed_steps(step, finish, a.steps)  # HIP events on the launch stream
    barrier()
    if dist is not None:
        elapsed = reduce_scalar(elapsed, dist.ReduceOp.MAX)
    st = ix.stats(reset=True)
    total_q = nq * a.steps * (1 if shard_mode else world)
    value = total_q / elapsed

    # i8 / b1 indexes (and VS_HNSW_ORDER=usearch) are served by the usearch-order walk, everything else by the fused-list kernel
    order = os.environ.get("VS_HNSW_ORDER", "")
    search_kernel_name = "hnsw_walk_kernel" if (order == "usearch" or (a.quantization in ("i8", "b1") and order != "fused")) else "hnsw_search_kernel"
    out = {
        "metric": "QPS at recall@10>=0.95 (HNSW search, inputs resident in HBM)",
        "value": value, "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.quantization, "data": "synthetic",
        "config": {"workload": f"{n}x{dim} {a.metric} top-{k} per GPU, {nq} queries/step, M=16 ef_add=128 ef_search={ef}" + ("" if a.quantization == "f32" else f" {a.quantization}"),
                   "distribution": a.dist + (f"{a.rank}" if a.dist == "lowrank" else ""), "mode": a.mode if (world > 1 or shard_mode) else "single",
                   "index_vectors_total": n * (world if shard_mode else 1), "query_batches_rotated": nb},
        "recall_at_10": round(recall, 4), "ef_search": ef, "ef_sweep": sweep,
        "roofline": hbm_roofline(ix, st, nq, dim, kernel_ms, search_kernel_name),
        "build": build_info,
    }
    # PMC-measured HBM bytes per launch for this exact workload (profiles/*traffic.json, see profiles/README.md).  A record
    # is only used while the kernel sources it was measured on are the ones in the tree; otherwise traffic stays null and
    # traffic_source says which record went stale.
    import glob
    from scripts.summarise_profiles import kernel_sources_sha16
    sha_now = kernel_sources_sha16()
    out["roofline"]["traffic_source"] = None
    for tr in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            rec = json.load(open(tr))
            if rec.get("workload") == out["config"]["workload"]:
                fresh = rec.get("kernel_sources_sha16") == sha_now
                out["roofline"]["traffic_source"] = {"file": os.path.relpath(tr, ROOT), "kernel_sources_sha16": rec.get("kernel_sources_sha16"),
                                                     "current_kernel_sources_sha16": sha_now, "stale": not fresh}
                out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"] if fresh else None
                if fresh and "dram" in rec:
                    out["roofline"]["dram"] = rec["dram"]  # how much of the traffic came from HBM itself (TCC EA counters), how much from the Infinity Cache
        except Exception:
            pass

    # ---- sharded legs at N>1 (RCCL all-gather + merge), reported beside the replica number: weak (n per GPU) and fixed total (n / N per GPU)
    if world > 1 and not shard_mode and not a.no_sharded_leg:
        out["sharded"] = {}
        for leg, per in (("weak", n), ("fixed_total", max(n // world, 1))):
            # every rank builds its shard first and says whether it could: a rank that failed must not leave the others waiting in the
            # collectives below (communicator creation, barriers, the all-gathers)
            six = sq = None
            setup_err = None
            try:
                sbase = make_data(per, dim, a.dist, 777 + rank, dev, a.rank)
                skeys = np.arange(per, dtype=np.uint64) + np.uint64(rank * per)
                six, sbuild = build_index(vs, sbase, skeys, a.metric, quantization=a.quantization)
                del sbase
                six.set_expansion_search(ef)
                sq = [make_data(nq, dim, a.dist, 4321 + 1000 * b, dev, a.rank) for b in range(nb)]
            except Exception as e:  # noqa: BLE001
                setup_err = e
            if reduce_scalar(0.0 if setup_err is not None else 1.0, dist.ReduceOp.MIN) < 0.5:
                out["sharded"][leg] = {"error": repr(setup_err) if setup_err is not None else "another rank could not build its shard"}
                del six, sq
                torch.cuda.empty_cache()
                continue
            try:
                gs = make_sharded_searcher(six, sq[0], k, dist, vs, ranks, sharded, per * world, a.backend)
                if hasattr(gs, "ranks"):
                    ci = gs.ranks.comm_info()
                    rccl_ranks, comm_ranks, exchange_kind = ci["rccl_ranks"], ci["comm_ranks"], ci["exchange"]
                sfinish = getattr(gs, "flush", lambda: None)
                struth = gs.exact()
                gs.step()
                sfinish()
                torch.cuda.synchronize()
                srec = recall_at_k(struth, gs.keys.cpu().numpy())
                srec = reduce_scalar(srec, dist.ReduceOp.MIN)
                sturn = [0]

                def sstep():
                    gs.q = sq[sturn[0] % nb]
                    sturn[0] += 1
                    gs.step()
                ssteps = max(a.steps // 2, 1)
                barrier()
                _, sel = timed_steps(sstep, sfinish, ssteps)
                barrier()
                sel = reduce_scalar(sel, dist.ReduceOp.MAX)
                out["sharded"][leg] = {"path": type(gs).__name__, "vectors_per_gpu": per, "index_vectors_total": per * world, "queries_per_s": nq * ssteps / sel,
                                       "recall_at_10": round(srec, 4), "ef_search": ef, "steps": ssteps, "build_vectors_per_s_per_gpu": per / sbuild,
                                       "unanswered_queries": gs.ranks.unanswered() if hasattr(gs, "ranks") else None}
                del six, gs, sq
                torch.cuda.empty_cache()
            except Exception as e:  # the replica number stands on its own
                out["sharded"][leg] = {"error": repr(e)}
        out["sharded"]["collective"] = "one ncclAllGather (RCCL, libvs_ranks) of packed per-shard top-k per batch + topk_merge_kernel, overlapped with the next walk"
    if world > 1 or shard_mode:
        out["rccl_ranks"] = rccl_ranks  # ncclCommCount of libvs_ranks' communicator (0: its host exchange served; None: the torch twin did)
        out["comm_ranks"] = comm_ranks  # ranks that joined the library's exchange, whichever it is
        out["exchange"] = exchange_kind
        # the sharded legs beside the replica `value`: the north_star's own path (key ranges + one all-gather per batch)
        for leg in ("weak", "fixed_total"):
            rec = out.get("sharded", {}).get(leg)
            if isinstance(rec, dict) and "queries_per_s" in rec:
                out[f"sharded_{leg}_queries_per_s"] = rec["queries_per_s"]
                out[f"sharded_{leg}_recall_at_10"] = rec["recall_at_10"]
        # An N-GPU line is only worth reporting when RCCL itself carried the exchange between N ranks (--same-device: a testing aid on
        # one GPU, marked as such, never a scaling number).
        if world > 1 and not a.same_device and (exchange_kind != "rccl" or rccl_ranks != world):
            out["error"] = f"the sharded path did not run over RCCL with {world} ranks (exchange {exchange_kind}, rccl_ranks {rccl_ranks})"
            rccl_failed = True
    if a.same_device and world > 1:
        out["same_device"] = True  # testing aid: every rank on GPU 0 -- exercises the N > 1 code, is NOT a scaling measurement

    # ---- through the boundary: what a drop-in caller of the trait gets on this index (N=1 only)
    boundary_answers = None
    if world == 1 and a.boundary_seconds > 0:
        try:
            out["boundary"], boundary_answers = boundary_record(ix, queries.cpu().numpy(), truth, k, a.boundary_seconds)
        except Exception as e:
            out["boundary"] = {"error": repr(e)}

    # ---- CPU baseline + id parity at full size: rank 0, N=1 only, bounded
    violations = 0
    other_ix = None
    if world == 1 and a.cpu_seconds > 0:
        try:
            step(0)
            finish()
            torch.cuda.synchronize()
            gk, gd = result_keys().view(np.uint64).copy(), result_dist().copy()
            extra = make_data(a.cpu_build_vectors, dim, a.dist, 97531, dev, a.rank).cpu().numpy() if a.cpu_build_vectors else None
            want_filtered = a.quantization == "f32" and isinstance(out.get("boundary"), dict) and "filtered" in out["boundary"]
            mixed_fresh = make_data(8192, dim, a.dist, 24680, dev, a.rank).cpu().numpy() if a.mixed_seconds > 0 else None
            if a.mixed_seconds > 0 and a.quantization == "f32" and other_ix is None:
                try:  # the index the background searches of boundary.mixed.two_indexes go to (1M x dim, same generator)
                    ob = make_data(1_000_000, dim, a.dist, 777, dev, a.rank)
                    other_ix, _ = build_index(vs, ob, np.arange(1_000_000, dtype=np.uint64), a.metric)
                    other_ix.set_expansion_search(ef)
                    del ob
                except Exception:  # noqa: BLE001
                    other_ix = None
            cb, ckeys = cpu_baseline(ix, queries.cpu().numpy(), k, ef, a.cpu_seconds, gk, gd, extra,
                                     filtered_seconds=min(6.0, a.cpu_seconds) if want_filtered else 0.0, boundary_answers=boundary_answers,
                                     mixed_seconds=a.mixed_seconds if a.quantization == "f32" else 0.0, mixed_fresh=mixed_fresh, headroom=300_000,
                                     other_index=other_ix)
            boundary_answers = None
            cb["recall_at_10"] = round(recall_at_k(truth, ckeys), 4)
            out["cpu_baseline"] = cb
            violations = cb["id_parity"]["violations"]
            # what the crowds of the boundary legs received, against the oracle: the parity record moves next to each leg
            for leg, par in cb.pop("boundary_id_parity", {}).items():
                violations += par["violations"]
                group, _, short = leg.partition(".")
                where = out["boundary"].get(group, {}) if short else out["boundary"]
                if isinstance(where.get(short or leg), dict):
                    where[short or leg]["id_parity"] = par
            # the CPU's filtered rate beside each GPU record of the same predicate (same graph, same call pattern, `cores` threads)
            for name, crec in cb.get("filtered", {}).items():
                violations += crec.get("id_parity", {}).get("violations", 0)
                for group in ("filtered", "filtered_named"):
                    for gname, grec in out["boundary"].get(group, {}).items():
                        if gname.startswith(name) and isinstance(grec, dict) and "queries_per_s" in grec and crec["queries_per_s"] > 0:
                            grec["cpu_queries_per_s"] = crec["queries_per_s"]
                            grec["vs_cpu"] = grec["queries_per_s"] / crec["queries_per_s"]
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}

    # ---- the reference's mixed add / search workloads through the dispatch actor on this index (N=1 only; last: they modify it)
    if world == 1 and a.mixed_seconds > 0 and a.quantization == "f32":
        try:
            from vector_store_amd import actor as _actor
            mixed_fresh = make_data(8192, dim, a.dist, 24680, dev, a.rank).cpu().numpy()

            def actor_of():
                act = _actor.IndexActor(dim, vs.METRICS[a.metric], 16, 128, ef, workers=effective_cores())
                act.adopt_partition(0, ix.h, ix.size())
                return act
            m0 = ix.modify_stats()
            mixed = mixed_record(actor_of, queries.cpu().numpy(), mixed_fresh, n, a.mixed_seconds)
            m1 = ix.modify_stats()
            mixed["engine"] = {kk: m1[kk] - m0[kk] for kk in m1}
            if other_ix is None:
                ob = make_data(1_000_000, dim, a.dist, 777, dev, a.rank)
                other_ix, _ = build_index(vs, ob, np.arange(1_000_000, dtype=np.uint64), a.metric)
                other_ix.set_expansion_search(ef)
                del ob
            act_a = actor_of()
            act_b = _actor.IndexActor(dim, vs.METRICS[a.metric], 16, 128, ef, workers=effective_cores())
            act_b.adopt_partition(0, other_ix.h, other_ix.size())
            try:
                mixed["two_indexes"] = two_index_record(act_a, act_b, queries.cpu().numpy(), mixed_fresh, n, a.mixed_seconds)
                mixed["two_indexes"]["note"] = "updates on the 10M index through its actor, 16 plain + 16 filtered searchers on a 1M index through its own actor"
                ctwo = (out.get("cpu_baseline", {}).get("mixed", {}) or {}).get("two_indexes") if isinstance(out.get("cpu_baseline"), dict) else None
                if isinstance(ctwo, dict) and ctwo.get("updates_per_s"):
                    mixed["two_indexes"]["vs_cpu"] = {"updates": mixed["two_indexes"]["updates_per_s"] / ctwo["updates_per_s"]}
            finally:
                act_a.stop()
                act_b.stop()
            cm = out.get("cpu_baseline", {}).get("mixed", {}) if isinstance(out.get("cpu_baseline"), dict) else {}
            for pk, legs in mixed.items():  # the CPU's rate beside each leg (same actor, same driver, the oracle behind it)
                if not pk.startswith("producers_") or not isinstance(cm.get(pk), dict):
                    continue
                for leg, rec in legs.items():
                    crec = cm[pk].get(leg.replace("@named", ""))
                    if isinstance(rec, dict) and isinstance(crec, dict) and "items_per_s" in rec:
                        rec["vs_cpu"] = {"items": rec["items_per_s"] / crec["items_per_s"] if crec.get("items_per_s") else None}
                        for kind in ("plain", "filtered"):
                            if kind in rec and kind in crec and crec[kind]["per_s"] > 0:
                                rec["vs_cpu"][kind] = rec[kind]["per_s"] / crec[kind]["per_s"]
            if isinstance(out.get("boundary"), dict):
                out["boundary"]["mixed"] = mixed
            else:
                out["boundary"] = {"mixed": mixed}
        except Exception as e:
            out.setdefault("boundary", {})["mixed"] = {"error": repr(e)}

    # ---- the other BASELINE configs and the survey's own generators as side records (N=1 only); the headline index is released first
    if world == 1 and not a.no_side_records and a.quantization == "f32":
        want = a.configs.split(",") if a.configs not in ("auto", "none") else []
        import gc
        se = ix = step = finish = probe = result_keys = result_dist = gs = other_ix = None  # every reference to the headline index
        gc.collect()
        torch.cuda.empty_cache()
        free_b, _ = torch.cuda.mem_get_info()
        if a.configs == "auto":
            want = ["c2", "c5"] + (["c3"] if free_b >= 180 * 2 ** 30 else [])
        out["configs"] = []
        try:
            gens, c2 = side_records(vs, dev, dim, a.metric, k)
            out["generators_at_1m"] = gens
            if c2 is not None and "c2" in want:
                out["configs"].append(c2)
        except Exception as e:
            out["generators_at_1m"] = {"error": repr(e)}
        if "c5" in want:
            try:
                out["configs"].append(config_c5(vs, dev, 10_000_000, 768, k, a.dist, a.rank))
            except Exception as e:
                out["configs"].append({"config": "configs[4]", "error": repr(e)})
        if "c5" in want:  # (with the other 10M side records: the i8 index behind the reference's call pattern)
            try:
                out["configs"].append(i8_callers_record(vs, dev, 10_000_000, 768, k, 200, a.dist, a.rank, 1.5))
            except Exception as e:
                out["configs"].append({"config": "i8_blocking_callers", "error": repr(e)})
            try:  # (round 6: b1 storage, lone queries through the walk pods)
                out["configs"].append(i8_callers_record(vs, dev, 10_000_000, 768, k, 200, a.dist, a.rank, 1.5, kind="b1"))
            except Exception as e:
                out["configs"].append({"config": "b1_blocking_callers", "error": repr(e)})
        if "c3" in want:
            try:
                out["configs"].append(config_c3(vs, dev, 10_000_000, k, a.dist, a.rank, a.target_recall))
            except Exception as e:
                out["configs"].append({"config": "configs[2]", "error": repr(e)})
        elif a.configs == "auto":
            out["configs"].append({"config": "configs[2]", "skipped": f"{free_b / 2 ** 30:.0f} GiB of HBM free, 180 needed"})

    if rank == 0:
        # The full record goes to a file; stdout's LAST line is the contract line alone (bench_line.py: < 8 KB, asserted).
        from bench_line import contract_line
        full_path = a.full_record or os.path.join("gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(os.path.join(ROOT, full_path)) or ".", exist_ok=True)
            with open(os.path.join(ROOT, full_path), "w") as fh:
                json.dump(out, fh)
                fh.write("\n")
        except OSError as e:
            print(f"[bench] could not write the full record to {full_path}: {e}", file=sys.stderr)
            full_path = None
        sys.stderr.flush()
        print(contract_line(out, full_path))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if violations:
        print(f"[bench] id parity against the CPU restatement: {violations} violation(s)", file=sys.stderr)
        sys.exit(3)
    if rccl_failed:
        print(f"[bench] {out['error']}", file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
