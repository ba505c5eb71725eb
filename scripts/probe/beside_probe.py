#!/usr/bin/env python3
"""Plain and filtered blocking callers on ONE index at the same time, through the C ABI (no actor): each kind alone, then both
together -- what the reference produces (plain Ann inline on its workers + every filtered query on a blocking thread, usearch.rs:928-948).
    python scripts/probe/beside_probe.py [vectors=10000000] [seconds=2]
"""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import torch  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
import vector_store_amd as vs  # noqa: E402
from vector_store_amd import callers  # noqa: E402

dev = torch.device("cuda", 0)
base = bench.make_data(n, 768, "lowrank", 1234, dev, 24)
q = bench.make_data(2048, 768, "lowrank", 4321, dev, 24).cpu().numpy()
ix = vs.HipUsearchIndex(768, vs.COS, 16, 128, 200)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, 768)
torch.cuda.synchronize()
del base
out = {}


def leg(name, plain, filtered, fkey=0):
    res = {}

    def p():
        res["plain"] = callers.run(ix, q, 10, None, plain, 1, secs)[0]

    def f():
        res["filtered"] = callers.run_filtered(ix, q, 10, 10, filtered, secs, filter_key=fkey)

    f0, p0 = ix.filter_stats(), ix.pod_stats()
    th = ([threading.Thread(target=p)] if plain else []) + ([threading.Thread(target=f)] if filtered else [])
    [x.start() for x in th]
    [x.join() for x in th]
    f1, p1 = ix.filter_stats(), ix.pod_stats()
    rec = {}
    if plain:
        r = res["plain"]
        rec["plain"] = {"qps": r.qps, "p50_ms": r.p50_ns / 1e6, "p99_ms": r.p99_ns / 1e6, "min_ms": r.latency_min_ns / 1e6}
    if filtered:
        r, extra = res["filtered"][0], res["filtered"][1]
        nq = max(int(r.queries), 1)
        rec["filtered"] = {"qps": r.qps, "p50_ms": r.p50_ns / 1e6, "p99_ms": r.p99_ns / 1e6, "min_ms": r.latency_min_ns / 1e6, "calls_per_query": extra[0] / nq,
                           "rounds_per_query": (f1["lazy_rounds"] - f0["lazy_rounds"]) / nq}
    rec["pods"] = {k: p1[k] - p0[k] for k in ("pods_opened", "pod_rounds", "rounds_without_a_pod", "rounds_walked_again", "filtered_handed_over")}
    out[name] = rec
    print(name, json.dumps(rec), file=sys.stderr, flush=True)


leg("warm_filtered", 0, 16)
leg("plain_16", 16, 0)
leg("filtered_16", 0, 16)
leg("both_16_16", 16, 16)
leg("both_4_16", 4, 16)
leg("both_16_4", 16, 4)
leg("filtered_16_named_warm", 0, 16, 0xBEEF)
leg("filtered_16_named", 0, 16, 0xBEEF)
leg("both_16_16_named", 16, 16, 0xBEEF)
print(json.dumps(out))
