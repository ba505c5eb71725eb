#!/usr/bin/env python3
"""Exact search: split-bf16 MFMA nomination + f32 re-score + certificate (default) against the f32-input MFMA path
(VS_HNSW_EXACT=f32), same index, same queries: identical ids, distances within f32 rounding, and the time per batch.
    python scripts/probe/block_search_probe.py [vectors=1000000] [metric=ip]"""
import json, os, subprocess, sys, time
import numpy as np

if len(sys.argv) > 3 and sys.argv[3] == "child":
    import torch
    sys.path.insert(0, os.getcwd())
    import vector_store_amd as vs
    from bench import make_data
    n, metric = int(sys.argv[1]), sys.argv[2]
    dim, k = 768, 10
    dev = torch.device("cuda:0")
    base = make_data(n, dim, "lowrank", 1234, dev, 24)
    if metric == "ip":
        base /= base.norm(dim=1, keepdim=True)
    q = make_data(1024, dim, "lowrank", 4321, dev, 24)
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric])
    ix.reserve(n)
    ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, dim)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for nq in (256, 1024):
        keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
        dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        found = torch.empty((nq,), dtype=torch.int32, device=dev)
        ix.exact_search_batch_device(q.data_ptr(), nq, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            ix.exact_search_batch_device(q.data_ptr(), nq, k, keys.data_ptr(), dist.data_ptr(), found.data_ptr(), st)
        torch.cuda.synchronize()
        out[str(nq)] = {"ms_per_batch": (time.time() - t0) / 3 * 1e3, "keys": keys.cpu().numpy().tolist(), "dist": dist.cpu().numpy().tolist()}
    out["stats"] = ix.exact_stats()
    print("RESULT" + json.dumps(out))
    sys.exit(0)

n = sys.argv[1] if len(sys.argv) > 1 else "1000000"
metric = sys.argv[2] if len(sys.argv) > 2 else "ip"
res = {}
for mode in ("block", "f32"):
    env = dict(os.environ)
    if mode == "f32":
        env["VS_HNSW_EXACT"] = "f32"
    o = subprocess.run([sys.executable, __file__, n, metric, "child"], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(mode, "FAILED", o.stderr[-2000:])
        sys.exit(1)
    res[mode] = json.loads(line[0][6:])
for nq in ("256", "1024"):
    a, b = res["block"][nq], res["f32"][nq]
    ka, kb = np.array(a["keys"]), np.array(b["keys"])
    da, db = np.array(a["dist"]), np.array(b["dist"])
    same = float(np.mean([ka[i].tolist() == kb[i].tolist() for i in range(len(ka))]))
    sets = float(np.mean([set(ka[i].tolist()) == set(kb[i].tolist()) for i in range(len(ka))]))
    flops = 2.0 * int(nq) * int(n) * 768
    print(f"n {n} {metric} q={nq}: block {a['ms_per_batch']:.2f} ms ({flops / a['ms_per_batch'] / 1e9:.0f} TFLOP/s equivalent), f32 {b['ms_per_batch']:.2f} ms "
          f"({flops / b['ms_per_batch'] / 1e9:.0f} TFLOP/s); rows with identical id lists {same:.4f}, identical id sets {sets:.4f}, max |d diff| {np.abs(da - db).max():.2e}")
print("block path stats:", res["block"]["stats"], " f32 path stats:", res["f32"]["stats"])
