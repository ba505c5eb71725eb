"""Lone inserts: one staged vector per flush (the reference's add is one vector per call, usearch.rs:191-197, and a mixed workload
flushes between two families of searches).  Times `add` + the barrier (`size()`), and prints the engine's own flush clock."""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench_common as bc  # noqa: E402
import vector_store_amd as v  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--vectors", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--n", type=int, default=300)
a = ap.parse_args()
import torch  # noqa: E402
g = torch.Generator(device="cuda")
g.manual_seed(1)
lat = torch.randn((a.vectors + a.n, 24), generator=g, device="cuda")
mp = torch.randn((24, a.dim), generator=g, device="cuda")
base = lat @ mp + 0.05 * torch.randn((a.vectors + a.n, a.dim), generator=g, device="cuda")
ix = v.HipUsearchIndex(a.dim, v.COS)
ix.reserve(a.vectors + a.n)
t0 = time.time()
ix.add_batch_device(np.arange(a.vectors, dtype=np.uint64), base.data_ptr(), a.vectors, a.dim)
print("built", a.vectors, "in", round(time.time() - t0, 2), "s", flush=True)
extra = base[a.vectors:].cpu().numpy()
ts = []
for i in range(a.n):
    t = time.perf_counter()
    ix.add(a.vectors + i, extra[i])
    ix.size()
    ts.append(time.perf_counter() - t)
ts = np.array(ts[20:]) * 1e3
print("lone insert ms: p50 %.3f mean %.3f p99 %.3f" % (np.percentile(ts, 50), ts.mean(), np.percentile(ts, 99)))
print(ix.modify_stats())
