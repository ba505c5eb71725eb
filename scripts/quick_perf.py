#!/usr/bin/env python3
"""Quick GPU perf probe: build + search timings at a given size (development aid, not the bench contract)."""
import argparse
import time

import numpy as np
import torch

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vector_store_amd as vs


def make_data(n, dim, kind, seed, device, rank=32):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if kind == "gaussian":
        return torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    gw = torch.Generator(device=device)
    gw.manual_seed(99)
    w = torch.randn((rank, dim), generator=gw, device=device, dtype=torch.float32) / rank ** 0.5
    z = torch.randn((n, rank), generator=g, device=device, dtype=torch.float32)
    out = z @ w
    out += 0.05 * torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="cos")
    ap.add_argument("--dist", default="lowrank")
    ap.add_argument("--efs", default="64,128,256")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rank", type=int, default=32)
    ap.add_argument("--quant", default="f32")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    base = make_data(a.n, a.dim, a.dist, 1234, dev, a.rank)
    q = make_data(a.nq, a.dim, a.dist, 4321, dev, a.rank)
    torch.cuda.synchronize()
    ix = vs.HipUsearchIndex(a.dim, vs.METRICS[a.metric], quantization=vs.SCALARS[a.quant])
    ix.reserve(a.n)
    keys = np.arange(a.n, dtype=np.uint64)
    t = time.time()
    ix.add_batch_device(keys, base.data_ptr(), a.n, a.dim)
    tb = time.time() - t
    st = ix.stats(reset=True)
    print(f"build n={a.n} dim={a.dim}: {tb:.2f}s = {a.n / tb:.0f} vec/s; evals/add={st['add_evals'] / max(st['added'], 1):.0f} (link {st['link_evals'] / max(st['added'], 1):.0f}) "
          f"hops/add={st['add_hops'] / max(st['added'], 1):.0f} overflow={st['visited_overflow']}", flush=True)
    k = a.k
    ok = torch.empty((a.nq, k), dtype=torch.int64, device=dev)
    od = torch.empty((a.nq, k), dtype=torch.float32, device=dev)
    of = torch.empty((a.nq,), dtype=torch.int32, device=dev)
    tk = torch.empty((a.nq, k), dtype=torch.int64, device=dev)
    td = torch.empty((a.nq, k), dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    t = time.time()
    ix.exact_search_batch_device(q.data_ptr(), a.nq, k, tk.data_ptr(), td.data_ptr(), of.data_ptr(), s)
    torch.cuda.synchronize()
    te = time.time() - t
    print(f"exact: {te:.2f}s ({2.0 * a.nq * a.n * a.dim / te / 1e12:.1f} TFLOP/s)", flush=True)
    truth = tk.cpu().numpy()
    for ef in [int(x) for x in a.efs.split(",")]:
        ix.set_expansion_search(ef)
        ix.search_batch_device(q.data_ptr(), a.nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
        torch.cuda.synchronize()
        ix.stats(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ix.search_batch_device(q.data_ptr(), a.nq, k, ok.data_ptr(), od.data_ptr(), of.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        st = ix.stats(reset=True)
        eq, hq = st["search_evals"] / st["queries"], st["search_hops"] / st["queries"]
        got = ok.cpu().numpy()
        rec = np.mean([len(set(truth[i].tolist()) & set(got[i].tolist())) / k for i in range(0, a.nq, 5)])
        bq = eq * ix.bytes_per_vector() + hq * 132 + a.dim * 4
        print(f"ef={ef}: {ms:.2f} ms/batch, {a.nq / ms * 1e3:.0f} QPS, recall@{k}={rec:.4f}, E_q={eq:.0f} H_q={hq:.0f} "
              f"B_q={bq / 1e6:.2f} MB -> {bq * a.nq / ms / 1e6:.0f} GB/s ({bq * a.nq / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s) "
              f"overflow={st['visited_overflow']}", flush=True)


if __name__ == "__main__":
    main()
