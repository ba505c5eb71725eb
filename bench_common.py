"""What bench.py, its side legs (bench_side.py) and the probe scripts share: the synthetic data generators, the index build, the timed
loop with HIP events on the launch stream, the roofline arithmetic.  (Split out of bench.py in round 6: the contract path stays there.)"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")  # before the HIP runtime initialises: concurrent filtered searches (see csrc/engine.hip HwQueuesDefault)
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
HBM_ACHIEVABLE_GBS = 6300.0  # ... and what a streaming copy measures on it (same guide)
BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 2:1-sparsity headline figure is not used)
ADJ_BYTES = 132             # level-0 adjacency record, 4 + 32*4


def make_data(n, dim, kind, seed, device, rank=24):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if kind == "gaussian":
        return torch.randn((n, dim), generator=g, device=device, dtype=torch.float32)
    if kind == "clustered":  # 256 Gaussian centres (seed 99), sigma = 0.2 around them
        gc = torch.Generator(device=device)
        gc.manual_seed(99)
        centres = torch.randn((256, dim), generator=gc, device=device, dtype=torch.float32)
        which = torch.randint(0, 256, (n,), generator=g, device=device)
        out = centres[which]
        chunk = 1 << 18
        for i in range(0, n, chunk):
            m = min(chunk, n - i)
            out[i:i + m] += 0.2 * torch.randn((m, dim), generator=g, device=device, dtype=torch.float32)
        return out
    gw = torch.Generator(device=device)
    gw.manual_seed(99)
    w = torch.randn((rank, dim), generator=gw, device=device, dtype=torch.float32) / rank ** 0.5
    out = torch.randn((n, rank), generator=g, device=device, dtype=torch.float32) @ w
    chunk = 1 << 18
    for i in range(0, n, chunk):  # noise in slices: keeps the generator's transient footprint small
        m = min(chunk, n - i)
        out[i:i + m] += 0.05 * torch.randn((m, dim), generator=g, device=device, dtype=torch.float32)
    return out


def recall_at_k(truth: np.ndarray, got: np.ndarray) -> float:
    # crates/benchmark/src/db.rs:308: |neighbors ∩ found| / |neighbors|
    k = truth.shape[1]
    return float(np.mean([len(set(truth[i].tolist()) & set(got[i].tolist())) / k for i in range(truth.shape[0])]))


class Searcher:
    """Device-resident query batches + output buffers; one call = one step of the hot path.  `batches` rotate through
    the steps (step i searches batch i mod B); batch 0 is the one recall and parity are measured on."""

    def __init__(self, ix, batches, k):
        self.ix, self.k = ix, k
        self.batches = batches if isinstance(batches, (list, tuple)) else [batches]
        self.q = self.batches[0]
        nq, dev = self.q.shape[0], self.q.device
        self.keys = torch.empty((nq, k), dtype=torch.int64, device=dev)
        self.dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
        self.found = torch.empty((nq,), dtype=torch.int32, device=dev)
        self.turn = 0

    def step(self, batch=None):
        q = self.batches[self.turn % len(self.batches)] if batch is None else self.batches[batch]
        if batch is None:
            self.turn += 1
        s = torch.cuda.current_stream().cuda_stream
        self.ix.search_batch_device(q.data_ptr(), q.shape[0], self.k, self.keys.data_ptr(), self.dist.data_ptr(), self.found.data_ptr(), s)

    def exact(self):
        s = torch.cuda.current_stream().cuda_stream
        self.ix.exact_search_batch_device(self.q.data_ptr(), self.q.shape[0], self.k, self.keys.data_ptr(),
                                          self.dist.data_ptr(), self.found.data_ptr(), s)
        torch.cuda.synchronize()
        return self.keys.cpu().numpy().copy(), self.dist.cpu().numpy().copy()


def build_index(vs, base, keys, metric, ef_add=128, quantization="f32"):
    n, dim = base.shape
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], 16, ef_add, 64, quantization=vs.SCALARS[quantization])
    ix.reserve(n)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ix.add_batch_device(keys, base.data_ptr(), n, dim)
    torch.cuda.synchronize()
    return ix, time.perf_counter() - t


def effective_cores() -> int:
    """Host cores this process may actually use: affinity mask and cgroup CPU quota included."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def timed_steps(step, finish, steps):
    """`steps` steps between two HIP events each, on torch's current stream = the stream the kernels are launched on;
    returns (mean ms per launch by the events, wall seconds of the whole region)."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    finish()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])), wall


def hbm_roofline(ix, st, nq, dim, kernel_ms, kernel_name):
    """B_q = E_q * row_bytes + H_q * 132 + dim * 4 (E_q, H_q counted in-kernel), one launch = nq queries."""
    e_q = st["search_evals"] / max(st["queries"], 1)
    h_q = st["search_hops"] / max(st["queries"], 1)
    b_q = e_q * ix.bytes_per_vector() + h_q * ADJ_BYTES + dim * 4
    achieved = b_q * nq / (kernel_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            # against what a streaming copy reaches on this chip (MI355X_MICROARCH.md: 6.29 TB/s; once-read random 2,304-byte rows
            # 5.7-5.8).  A value near or above 1 does NOT mean the DRAM interface is saturated: part of the bytes counted are served by
            # the 256 MB Infinity Cache (hub rows, upper levels, adjacency), which no counter of rocprofv3 separates from HBM reads.
            "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS, "achievable": HBM_ACHIEVABLE_GBS,
            "kernel": kernel_name, "kernel_ms": kernel_ms, "bytes_per_query": b_q, "evals_per_query": e_q, "hops_per_query": h_q,
            "visited_overflow": st["visited_overflow"]}


