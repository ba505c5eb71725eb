"""vector_store_amd -- MI355X-native HNSW engine behind the usearch boundary of scylladb/vector-store.

The product is `libvs_hnsw.so` (HIP/gfx950, C ABI in include/vs_hnsw.h).  This package is the
thin host-side binding used by tests and bench.py; it mirrors the reference's private
`trait UsearchIndex` (crates/vector-store/src/vs_index/usearch.rs:142-160).  It never falls
back to a CPU path: if the library or a GPU is missing, it raises.
"""
from .index import (  # noqa: F401
    B1, BF16, COS, F16, F32, HAMMING, I8, IP, L2SQ, METRICS, SCALARS, HipUsearchIndex, VsError, distance_valid, f32_to_b1x8, lib, lib_path,
    similarity_score, streams_created, topk_merge_device, version,
)
