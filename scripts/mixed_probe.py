#!/usr/bin/env python3
"""The reference's mixed add / search workloads (crates/vector-store/benches/pipeline.rs:508-1292: cdc_insert, cdc_update,
cdc_delete, search_while_{inserting,updating,deleting}) through the dispatch actor (libvs_actor) on a bulk-built index, on
the GPU engine and -- the same actor, the same driver (libvs_callers: vs_mixed_run) -- on the CPU oracle holding a copy of
the same graph.  Prints one JSON object; bench.py's `boundary.mixed` / `cpu_baseline.mixed` run the same legs.

    python scripts/mixed_probe.py [--vectors 2000000] [--dim 768] [--seconds 2] [--cpu] [--legs a,b,...] [--producers 1,16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch  # noqa: E402

import bench  # noqa: E402  (make_data, build_index, effective_cores)

LEGS = ("cdc_insert", "cdc_update", "cdc_delete", "search", "search@named", "search:16+0", "search_while_inserting", "search_while_updating", "search_while_updating@named",
        "search_while_updating:16+0", "search_while_deleting")


def run_legs(actor, callers, queries, vectors, n, legs, seconds, producers, plain, filtered, modulus, state, tag, ix=None):
    out = {}
    for leg in legs:  # one at a time: the engine's own account of each leg beside it
        before = (ix.call_stats(), ix.modify_stats(), ix.filter_stats(), ix.pod_stats()) if ix is not None else None
        r = callers.pipeline_legs(actor, queries, vectors, n, (leg,), seconds=seconds, producers=producers, plain_callers=plain, filtered_callers=filtered,
                                  modulus=modulus, state=state)[leg]
        if ix is not None:
            after = (ix.call_stats(), ix.modify_stats(), ix.filter_stats(), ix.pod_stats())
            d = [{k: (a[k] - b[k]) for k in a if isinstance(a[k], (int, float)) and not isinstance(a[k], bool)} for a, b in zip(after, before)]
            c = d[0]
            r["engine"] = {"ms_per_search": c["search_ms"] / max(c["searches"], 1), "ms_per_filtered": c["filtered_ms"] / max(c["filtered"], 1),
                           "filtered_device_wait_ms": c["filtered_device_wait_ms"] / max(c["filtered"], 1),
                           "filtered_predicate_ms": c["filtered_predicate_ms"] / max(c["filtered"], 1),
                           "flush_wait_ms_per_call": c["flush_wait_ms"] / max(c["searches"] + c["filtered"], 1),
                           "flushes": d[1]["flushes"], "flush_ms": d[1]["flush_ms"], "rounds_per_filtered": d[2]["lazy_rounds"] / max(c["filtered"], 1),
                           "pods_opened": d[3]["pods_opened"], "rounds_without_a_pod": d[3]["rounds_without_a_pod"]}
        out[leg] = r
        print(f"[{tag}] {leg}: " + json.dumps(r), file=sys.stderr, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vectors", type=int, default=2_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--ef", type=int, default=200)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--cpu", action="store_true", help="the same legs over the CPU oracle holding the same graph")
    ap.add_argument("--no-gpu-legs", action="store_true")
    ap.add_argument("--legs", default=",".join(LEGS))
    ap.add_argument("--producers", default="1", help="comma list: concurrent CDC producers (BENCHES_CONCURRENCY)")
    ap.add_argument("--plain", type=int, default=16)
    ap.add_argument("--filtered", type=int, default=16)
    ap.add_argument("--modulus", type=int, default=10)
    ap.add_argument("--workers", type=int, default=0, help="actor workers (0 = usable cores)")
    a = ap.parse_args()
    import vector_store_amd as vs
    from vector_store_amd import actor, callers
    dev = torch.device("cuda", 0)
    n, dim = a.vectors, a.dim
    legs = [x for x in a.legs.split(",") if x]
    workers = a.workers or bench.effective_cores()
    base = bench.make_data(n, dim, "lowrank", 1234, dev, 24)
    queries = bench.make_data(4096, dim, "lowrank", 4321, dev, 24).cpu().numpy()
    fresh = bench.make_data(8192, dim, "lowrank", 97531, dev, 24).cpu().numpy()
    ix = vs.HipUsearchIndex(dim, vs.COS, 16, 128, a.ef)
    ix.reserve(n + 400_000)  # both sides grow nowhere during the legs (the CPU's reserve is a 30 GB realloc at 10M)
    t0 = time.perf_counter()
    ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
    torch.cuda.synchronize()
    out = {"workload": f"{n} x {dim} cos, ef_search {a.ef}, through libvs_actor with {workers} workers; {a.plain} plain + {a.filtered} filtered callers "
                       f"(key % {a.modulus} == 0) beside the producers", "build_s": time.perf_counter() - t0, "cores": bench.effective_cores()}
    del base
    torch.cuda.empty_cache()
    o = None
    if a.cpu:
        import oracle
        o = oracle.OracleIndex(dim, oracle.COS, 16, 128, a.ef)
        o.reserve(n + 400_000)
        g = ix.export_graph(vectors_out=o.vector_arena(n))
        o.import_graph(g)
        del g
    for p in [int(x) for x in a.producers.split(",")]:
        if not a.no_gpu_legs:
            act = actor.IndexActor(dim, vs.COS, 16, 128, a.ef, workers=workers)
            act.adopt_partition(0, ix.h, ix.size())
            state = {"next_key": (1 << 40) + p * (1 << 32), "delete_from": n // 2 + (p % 7) * 50_000}
            m0 = ix.modify_stats()
            rec = run_legs(act, callers, queries, fresh, n, legs, a.seconds, p, a.plain, a.filtered, a.modulus, state, f"gpu p={p}", ix=ix)
            m1 = ix.modify_stats()
            rec["engine"] = {k: m1[k] - m0[k] for k in m1}
            rec["actor_counters"] = act.counters()
            act.stop()
            out[f"gpu_producers_{p}"] = rec
        if o is not None:
            import oracle
            act = actor.IndexActor(dim, vs.COS, 16, 128, a.ef, workers=workers, index_vtable=oracle.trait_vtable())
            act.adopt_partition(0, o.h, o.size())
            state = {"next_key": (1 << 40) + p * (1 << 32), "delete_from": n // 2 + (p % 7) * 50_000}
            rec = run_legs(act, callers, queries, fresh, n, legs, a.seconds, p, a.plain, a.filtered, a.modulus, state, f"cpu p={p}")
            rec["actor_counters"] = act.counters()
            act.stop()
            out[f"cpu_producers_{p}"] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
