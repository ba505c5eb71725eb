"""Phase clocks of lone usearch-order walks on an integer-storage index (a -DVS_WALK_PROFILE build, VS_HNSW_WALK_DEBUG=1, VS_HNSW_B1_PODS=0)."""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import vector_store_amd as vs
from bench_common import make_data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 200
quant = {"i8": vs.I8, "b1": vs.B1, "f32": vs.F32}[sys.argv[3] if len(sys.argv) > 3 else "b1"]
dev = torch.device("cuda:0")
base = make_data(n, 768, "lowrank", 1234, dev)
q = make_data(64, 768, "lowrank", 4321, dev).cpu().numpy()
ix = vs.HipUsearchIndex(768, vs.COS, expansion_search=ef, quantization=quant)
ix.reserve(n)
ix.add_batch_device(np.arange(n, dtype=np.uint64), base.data_ptr(), n, 768)
for i in range(24):
    t = time.perf_counter()
    ix.search(q[i], 10)
    print("query", i, "ms", round((time.perf_counter() - t) * 1e3, 3), flush=True)
