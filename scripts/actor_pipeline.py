#!/usr/bin/env python3
"""The reference's pipeline scenarios (crates/vector-store/benches/pipeline.rs: fullscan-insert, search, cdc-update,
cdc-delete, search-while-updating) through the dispatch actor (libvs_actor: search-first channels, Operation permits,
+1,000,000 growth) over the GPU engine.  The reference runs them against its Simulator to price the actor hops; here
the real index is behind the actor.  Prints one JSON line.

    python scripts/actor_pipeline.py [vectors=200000] [dim=768] [threads=16] [seconds=4]
"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime in the process)
from vector_store_amd import actor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
rng = np.random.default_rng(1)
w = rng.standard_normal((24, dim)).astype(np.float32) / np.sqrt(24)
base = (rng.standard_normal((n, 24)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim)).astype(np.float32))
queries = (rng.standard_normal((4096, 24)).astype(np.float32) @ w + 0.05 * rng.standard_normal((4096, dim)).astype(np.float32))
a = actor.IndexActor(dim, expansion_search=128)
out = {"workload": f"{n} x {dim} cos through the dispatch actor, {T} caller threads"}


def run_threads(fn, count):
    th = [threading.Thread(target=fn, args=(t,)) for t in range(count)]
    [x.start() for x in th]
    [x.join() for x in th]


# fullscan-insert: T producers, fire-and-forget adds (usearch.rs:1019-1034), done when every vector is counted
t0 = time.perf_counter()
run_threads(lambda t: [a.add_vector(0, i, base[i]) for i in range(t, n, T)], T)
while a.count() < n:
    time.sleep(0.002)
out["fullscan_insert_vectors_per_s"] = n / (time.perf_counter() - t0)

stop = threading.Event()
done = [0] * T


def searcher(t):
    i = t
    while not stop.is_set():
        a.ann(0, queries[i % len(queries)], 10)
        i += T
        done[t] += 1


def timed_search(background=None):
    global done
    done = [0] * T
    stop.clear()
    bg = threading.Thread(target=background) if background else None
    th = [threading.Thread(target=searcher, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    if bg:
        bg.start()
    time.sleep(secs)
    stop.set()
    [x.join() for x in th]
    el = time.perf_counter() - t0
    if bg:
        bg.join()
    return sum(done) / el


out["search_queries_per_s"] = timed_search()
upd = [0]


def updater():  # cdc-update: a key gets a new vector = RemoveVector then AddVector (usearch.rs:1036-1049, :1019-1034)
    i = 0
    while not stop.is_set():
        a.remove_vector(0, i % n)
        a.add_vector(0, i % n, base[(i * 7 + 1) % n])
        i += 1
    upd[0] = i


out["search_while_updating_queries_per_s"] = timed_search(updater)
out["updates_per_s_meanwhile"] = upd[0] / secs
t0 = time.perf_counter()
m = min(n, 50_000)
for i in range(m):  # cdc-delete
    a.remove_vector(0, i)
while a.count() > n - m:
    time.sleep(0.002)
out["cdc_delete_vectors_per_s"] = m / (time.perf_counter() - t0)
out["actor_counters"] = a.counters()
a.stop()
print(json.dumps(out))
