"""The walk that asks while it runs (round 6) against the rounds of rounds 3-5 and the named filter's exact walk: lone-caller latency,
device time, waits, and 16 / 17 callers, on the bench's own index (default 10M x 768 cos, ef 200, key % modulus == 0)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import vector_store_amd as vs  # noqa: E402
from vector_store_amd import callers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--threads", default="1,16,17")
ap.add_argument("--named", type=int, default=0, help="filter_key: the named filter's exact walk instead (warmed by a first pass)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
base = bench.make_data(a.n, 768, "lowrank", 1234, dev)
ix, bs = bench.build_index(vs, base, np.arange(a.n, dtype=np.uint64), "cos")
del base
ix.set_expansion_search(200)
q = bench.make_data(2048, 768, "lowrank", 4321, dev).cpu().numpy()
out = {"build_s": bs}
for modulus, nq in ((10, 2048), (100, 256)):
    for threads in [int(t) for t in a.threads.split(",")]:
        callers.run_filtered(ix, q[:nq], 10, modulus, threads, 0.5 if not a.named else 3.0, filter_key=a.named * modulus)
        s0, f0 = ix.filter_ask_stats(), ix.filter_stats()
        r, extra, _, rc = callers.run_filtered(ix, q[:nq], 10, modulus, threads, a.seconds, filter_key=a.named * modulus)
        s1, f1 = ix.filter_ask_stats(), ix.filter_stats()
        nqd = max(int(r.queries), 1)
        d = {k: s1[k] - s0[k] for k in s1}
        aq = max(d["queries"], 1)
        out[f"mod{modulus}_t{threads}"] = {
            "qps": round(r.queries / r.seconds, 1), "p50_ms": r.as_dict().get("p50_ms"), "min_ms": r.as_dict().get("latency_min_ms"), "rc": rc,
            "calls_per_q": round(extra[0] / nqd), "walks_per_q": round((f1["lazy_rounds"] - f0["lazy_rounds"]) / nqd, 2),
            "ask_queries": d["queries"], "handed_over": d["handed_over"], "no_pod": d["no_pod"],
            "device_walk_ms_per_q": round(d["device_walk_ms"] / aq, 3), "device_wait_ms_per_q": round(d["device_wait_ms"] / aq, 3),
            "waits_per_q": round(d["device_waits"] / aq, 1), "hops_per_q": round(d["hops"] / aq)}
        print(json.dumps({f"mod{modulus}_t{threads}": out[f"mod{modulus}_t{threads}"]}), flush=True)
print(json.dumps(out))
