/* vs_hnsw.h -- C ABI of the MI355X-native HNSW engine (libvs_hnsw.so).
 *
 * Drop-in boundary for the usearch-backed index/search path of scylladb/vector-store:
 * every entry point replaces one call the reference makes into `usearch::Index` from
 * `ThreadedUsearchIndex` (reference crates/vector-store/src/vs_index/usearch.rs:162-251),
 * i.e. what an `impl UsearchIndex for HipIndex` (trait at usearch.rs:142-160) binds over
 * FFI.  INTEGRATION.md shows that Rust binding.  Plain pointers and sizes only.
 *
 * Conventions
 *  - Every function returning `int` returns VS_OK (0) or a negative vs_status; the
 *    message is available from vs_hnsw_last_error() on the calling thread.  No C++
 *    exception or abort crosses this boundary (reference: every usearch call returns
 *    Result, usearch.rs:182-200).
 *  - Host-pointer entry points borrow their inputs for the duration of the call and
 *    write results into caller-allocated arrays (reference: `vector.as_slice()`,
 *    usearch.rs:196,212).  `_device` variants take HBM-resident buffers + a hipStream_t.
 *  - Thread safety: concurrent add||add and search||search are supported; the caller
 *    guarantees (as the reference's Operation permits do, usearch.rs:545-612) that
 *    reserve/remove run alone and adds never overlap searches.
 *  - Keys are u64 PrimaryIds (epoch16 || idx48, reference table/primary_id.rs:35-45);
 *    the value ~0 is reserved (usearch free key).
 */
#ifndef VS_HNSW_H
#define VS_HNSW_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define VS_API __attribute__((visibility("default")))
#else
#define VS_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_hnsw vs_hnsw;

typedef enum vs_status {
    VS_OK = 0,
    VS_ERR_INVALID_ARGUMENT = -1,
    VS_ERR_DIMENSION = -2,     /* wrong embedding dimension (reference validator.rs:12-26 -> HTTP 400) */
    VS_ERR_CAPACITY = -3,      /* "Reserve capacity ahead of insertions!" */
    VS_ERR_DUPLICATE_KEY = -4, /* usearch multi=false: duplicate keys are an error */
    VS_ERR_OUT_OF_MEMORY = -5, /* HBM budget exceeded in reserve (reference memory.rs analogue) */
    VS_ERR_DEVICE = -6,        /* HIP runtime error / no GPU */
    VS_ERR_UNSUPPORTED = -7
} vs_status;

/* usearch::MetricKind as mapped by the reference (usearch.rs:450-501). */
typedef enum vs_metric_kind { VS_METRIC_COS = 0, VS_METRIC_L2SQ = 1, VS_METRIC_IP = 2, VS_METRIC_HAMMING = 3 } vs_metric_kind;
/* usearch::ScalarKind as mapped by the reference (usearch.rs:503-513). */
typedef enum vs_scalar_kind { VS_SCALAR_F32 = 0, VS_SCALAR_F16 = 1, VS_SCALAR_BF16 = 2, VS_SCALAR_I8 = 3, VS_SCALAR_B1 = 4 } vs_scalar_kind;

/* usearch::IndexOptions as filled by the reference (usearch.rs:74-82); 0 => usearch default
 * (connectivity 16, expansion_add 128, expansion_search 64). */
typedef struct vs_hnsw_options {
    size_t dimensions;
    size_t connectivity;
    size_t expansion_add;
    size_t expansion_search;
    int metric;       /* vs_metric_kind */
    int quantization; /* vs_scalar_kind */
    int device;       /* HIP device ordinal; -1 = current device */
    int reserved;     /* 0; test hooks: bit 0 = tiny visited table in search (forces the overflow path),
                         bit 1 = exact search on the VALU tile kernel instead of MFMA,
                         bit 2 = always serve a query with a team of wavefronts, bit 3 = never (default: batches
                         of at most one team per CU),
                         bit 4 = usearch-order walk (two structures, exact tie order) for every search of this index (default:
                         i8 / b1 storage only), bit 5 = always the global-bitmap instance of that walk,
                         bit 6 = wide visited tags (the instances for indexes above 2^25 / 2^26 slots) on a small index,
                         bit 7 = 64-entry global heap for the global-bitmap walk (a flooding walk then reports "outgrew its
                         workspace", which the host entry points answer by ranking exhaustively),
                         bit 8 = lone queries never take the pipelined walk (kernels_pipe.hip): the team kernels serve them, as
                         before round 4 (A/B in tests) */
} vs_hnsw_options;

/* -- lifecycle: usearch::Index::new (usearch.rs:172), drop ------------------------------- */
VS_API int vs_hnsw_create(const vs_hnsw_options* options, vs_hnsw** out);
VS_API void vs_hnsw_free(vs_hnsw* index);

/* -- usearch::Index::reserve_capacity_and_threads / capacity (usearch.rs:181-189) ------- */
VS_API int vs_hnsw_reserve(vs_hnsw* index, size_t capacity, size_t threads);
VS_API size_t vs_hnsw_capacity(const vs_hnsw* index);
VS_API size_t vs_hnsw_size(const vs_hnsw* index); /* live members; the reference counts outside (usearch.rs:1031) */
/* bytes of one stored vector: dim*4 (f32), dim*2 (f16, bf16), dim (i8), ceil(dim/8) (b1) */
VS_API size_t vs_hnsw_bytes_per_vector(const vs_hnsw* index);

/* -- usearch::Index::add(key, &[f32]) (usearch.rs:191-197) ------------------------------
 * Validates (reserved / duplicate key, capacity, dimension) and copies the vector before returning.  The
 * graph insertion itself is deferred and done in bulk: every later call on the index (search, remove,
 * reserve, size, stats, export ...) first inserts what is staged, so callers observe exactly the
 * sequential semantics -- adds are fire-and-forget in the reference too (usearch.rs:1028-1034). */
VS_API int vs_hnsw_add(vs_hnsw* index, uint64_t key, const float* vector, size_t dim);
/* n vectors, row-major n x dim.  The benchmark driver's bulk path (crates/benchmark build-index). */
VS_API int vs_hnsw_add_batch(vs_hnsw* index, const uint64_t* keys, const float* vectors, size_t n, size_t dim);
/* keys on the host (they feed the host key->slot map), vectors already resident in HBM. */
VS_API int vs_hnsw_add_batch_device(vs_hnsw* index, const uint64_t* keys, const float* d_vectors, size_t n, size_t dim);

/* -- usearch::Index::remove(key) (usearch.rs:199-201): *removed = 1 when the key existed -- */
VS_API int vs_hnsw_remove(vs_hnsw* index, uint64_t key, int* removed);

/* -- usearch::Index::search(&[f32], k) (usearch.rs:203-222) ------------------------------
 * keys/distances: caller arrays of length k, ascending by distance; *found <= k. */
VS_API int vs_hnsw_search(vs_hnsw* index, const float* query, size_t dim, size_t k, uint64_t* keys, float* distances,
                   size_t* found);
/* Non-blocking form of vs_hnsw_search for async runtimes (the reference runs searches inline on tokio
 * workers, usearch.rs:928-935; blocking them caps the queries in flight at the worker count).
 * `query` is copied before the call returns; keys/distances/found must stay valid until `done(ctx,
 * status)` runs on the engine's dispatcher thread (it must not block; e.g. complete a oneshot). */
typedef void (*vs_hnsw_completion)(void* ctx, int status);
VS_API int vs_hnsw_search_async(vs_hnsw* index, const float* query, size_t dim, size_t k, uint64_t* keys,
                         float* distances, size_t* found, vs_hnsw_completion done, void* ctx);
/* -- usearch::Index::filtered_search(&[f32], k, |key| bool) (usearch.rs:224-248) --------
 * predicate(key, ctx) != 0 admits the key into the result set. */
typedef int (*vs_hnsw_predicate)(uint64_t key, void* ctx);
VS_API int vs_hnsw_filtered_search(vs_hnsw* index, const float* query, size_t dim, size_t k, vs_hnsw_predicate predicate,
                            void* ctx, uint64_t* keys, float* distances, size_t* found);
/* The same call for a filter that has a NAME: `filter_key` != 0 is a fingerprint of the restrictions the predicate evaluates (the
 * reference's predicate is a table read-lock + restriction evaluation per candidate, usearch.rs:1118-1124: the same function of the key
 * for every query with the same `Filter::restrictions`).  The engine then remembers verdicts across the queries of that filter (two
 * bits per slot in HBM, up to 4 filters per index, least recently used first out): a key is asked about at most once, and once a
 * filter's neighbourhoods are known a query is one exact walk with no predicate call.  Results are those of vs_hnsw_filtered_search
 * as long as predicate(key) depends on nothing but (filter_key, key); a member that is removed or re-added is asked about again.
 * filter_key == 0: no memory (= vs_hnsw_filtered_search). */
VS_API int vs_hnsw_filtered_search_keyed(vs_hnsw* index, const float* query, size_t dim, size_t k, vs_hnsw_predicate predicate,
                                         void* ctx, uint64_t filter_key, uint64_t* keys, float* distances, size_t* found);
/* [0] queries answered with a filter memory, [1] verdicts they still asked the host for, [2] memories created, [3] memories held now */
VS_API int vs_hnsw_filter_memo_stats(vs_hnsw* index, uint64_t out[4]);
/* nq queries, row-major nq x dim; keys/distances are nq x k, found is nq. */
VS_API int vs_hnsw_search_batch(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, uint64_t* keys,
                         float* distances, size_t* found);
/* Device form: enqueued on hip_stream, nothing is synchronised.  k (or expansion_search) up to 10,240: beams beyond 512 take
 * the wide walk, whose per-workgroup workspace is bounded -- a query that outgrows it (not observed; the host entry points
 * then rank exhaustively) is flagged with d_found[i] = 0xFFFFFFFF instead of an answer. */
VS_API int vs_hnsw_search_batch_device(vs_hnsw* index, const float* d_queries, size_t nq, size_t dim, size_t k,
                                uint64_t* d_keys, float* d_distances, uint32_t* d_found, void* hip_stream);
/* Exact brute-force top-k (usearch `exact` search; ground truth for recall). Device buffers. */
VS_API int vs_hnsw_exact_search_batch_device(vs_hnsw* index, const float* d_queries, size_t nq, size_t dim, size_t k,
                                      uint64_t* d_keys, float* d_distances, uint32_t* d_found, void* hip_stream);
VS_API int vs_hnsw_exact_search_batch(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, uint64_t* keys,
                               float* distances, size_t* found);

/* expansion_search can be changed between searches (usearch change_expansion_search). */
VS_API int vs_hnsw_set_expansion_search(vs_hnsw* index, size_t expansion_search);

/* -- counters for the roofline figure: cumulative since reset -----
 * [0] distance evaluations in search, [1] node expansions in search, [2] queries,
 * [3] distance evaluations in add, [4] node expansions in add, [5] vectors added,
 * [6] visited-table overflows (must stay 0), [7] the part of [3] spent re-selecting neighbours' links */
VS_API int vs_hnsw_stats(vs_hnsw* index, uint64_t out[8], int reset);
/* HBM held by the index: [0] bytes in all arenas, [1] of which grow in place (virtual range + mapped chunks),
 * [2] physical chunks mapped, [3] bytes copied device-to-device by arena growth so far (process-wide). */
VS_API int vs_hnsw_memory_info(vs_hnsw* index, uint64_t out[4]);
/* Single-query dispatcher (vs_hnsw_search / _async), process-wide: [0] kernel launches, [1] queries,
 * [2] launches and [3] queries that took the team kernel (8 wavefronts per query, lightly loaded device). */
VS_API int vs_search_service_stats(uint64_t out[4]);

/* The usearch-order walk: [0] the instance (kernels.hpp WALK_*) the index's last search launch took, ~0 if none yet;
 * [1] queries (process-wide) the single-query dispatcher answered by exhaustive ranking because their walk outgrew its workspace. */
VS_API int vs_hnsw_walk_info(vs_hnsw* index, uint64_t out[2]);

/* Filtered search on indexes above 65,536 slots asks the predicate lazily (only for members a walk needs a verdict for, in
 * rounds): [0] walk launches and [1] predicate calls spent that way so far. */
VS_API int vs_hnsw_filter_stats(vs_hnsw* index, uint64_t out[2]);

/* A crowd of lazily filtered queries (more callers than the device has streams; the reference runs every filtered query on a blocking
 * thread of its own, usearch.rs:937-948) shares launches: [0] launches that served rounds of several callers at once, [1] rounds served
 * that way. */
VS_API int vs_hnsw_filter_batch_stats(vs_hnsw* index, uint64_t out[2]);

/* Lone queries (one vector per call, usearch.rs:212 / :236) on float indexes take the pipelined walk (kernels_pipe.hip):
 * [0] walks of it so far for this index (launched, or posted to a pod), [1] lone plain queries (process-wide) it handed over to the team kernels because two
 * equal distances met where their order matters. */
VS_API int vs_hnsw_pipe_stats(vs_hnsw* index, uint64_t out[2]);

/* HIP streams the engine has created in this process so far, over all devices and indexes: a fixed set per device (16 unless
 * VS_HNSW_STREAMS says otherwise) shared by every index handle -- thousands of per-partition handles (usearch.rs:704-705,
 * 766-778) own none. */
VS_API uint64_t vs_hnsw_streams_created(void);

/* Pods (vector_store_amd/csrc/pipe_pod.hpp): blocking callers of vs_hnsw_search / vs_hnsw_filtered_search on float indexes -- one
 * query per call (usearch.rs:212, :236), every filtered query on a thread of its own (:937-948) -- post their query to a workgroup of
 * a resident launch of the pipelined walk instead of launching one: as many walks in flight as callers, no launch per query.
 * [0] pods opened for this index so far, [1] queries / filter rounds its pods have served, [2] pods open on the index's device now,
 * [3] 1 unless VS_HNSW_PODS=0; where the time of the plain queries posted on the device went: [4] their number, [5] ns inside the
 * library, [6] of them waiting for the answer, [7] ns the workgroups spent on them by the device's clock; filtered queries of this
 * index: [8] answered through posted (or batched) rounds, [9] handed over to rounds of their own after such a round met two equal
 * distances where their order matters, [10] rounds no pod could take (launched in a batch instead), [11] rounds walked again in usearch's order on the caller's own stream
 * for the same reason. */
VS_API int vs_hnsw_pod_stats(vs_hnsw* index, uint64_t out[12]);

/* Where modifications spend their time (the reference's mixed add / search workloads, benches/pipeline.rs:508-1292):
 * [0] flushes of staged single-vector adds (vs_hnsw_add defers the insertion to the next call that observes the index),
 * [1] vectors they inserted, [2] ns they took; [3] times this index's pods were closed for a modification, [4] ns spent waiting
 * for their workgroups to leave; [5] vs_hnsw_remove calls, [6] ns inside them; [7] pods opened for this index. */
VS_API int vs_hnsw_modify_stats(vs_hnsw* index, uint64_t out[8]);

/* Where a single-query call spends its time: [0] vs_hnsw_search calls, [1] ns inside them; [2] vs_hnsw_filtered_search[_keyed] calls,
 * [3] ns inside them, of which [4] waiting for the device's rounds and [5] asking the predicate; [6] ns callers of either waited for
 * staged modifications to be applied first. */
VS_API int vs_hnsw_call_stats(vs_hnsw* index, uint64_t out[8]);

/* Exact search on float storage (cos / ip, k <= 64, >= 65,536 slots) nominates with split-bf16 MFMA tiles, re-scores the nominees
 * in f32 and certifies the answer: [0] batches that took that path, [1] of them re-run on the f32-input MFMA path because a query's
 * certificate failed. */
VS_API int vs_hnsw_exact_stats(vs_hnsw* index, uint64_t out[2]);
/* Round 3: the first stage of the exact search is a ONE-product bf16 pass over a bf16 plane of the rows (built lazily, +2 bytes per
 * element of HBM): [0] batches that took it, [1] of them handed on to the split-bf16 pass (uncertified), [2] / [3] as exact_stats. */
VS_API int vs_hnsw_exact_stats2(vs_hnsw* index, uint64_t out[4]);

/* -- graph export / import (flat layout; see oracle/cpu_hnsw.cpp orc_export_graph) ------- */
typedef struct vs_hnsw_graph_info {
    size_t slots;        /* nodes ever allocated (including removed) */
    size_t upper_blocks; /* total upper-level adjacency blocks */
    int32_t max_level;
    uint32_t entry_slot;
    size_t connectivity;      /* M  */
    size_t connectivity_base; /* M0 */
} vs_hnsw_graph_info;
VS_API int vs_hnsw_graph_info_get(vs_hnsw* index, vs_hnsw_graph_info* info);
VS_API int vs_hnsw_export_graph(vs_hnsw* index, void* vectors /* slots x bytes_per_vector, storage format */, int32_t* levels, uint64_t* keys,
                         uint32_t* adj0 /* slots x M0 */, uint32_t* upper_off, uint32_t* upper /* blocks x M */);
VS_API int vs_hnsw_import_graph(vs_hnsw* index, size_t slots, const void* vectors /* storage format */, const int32_t* levels,
                         const uint64_t* keys, const uint32_t* adj0, const uint32_t* upper_off, const uint32_t* upper,
                         size_t upper_blocks, int32_t max_level, uint32_t entry_slot);

/* -- multi-GPU: merge `parts` per-shard top-k lists (each nq x k, ascending, padded with
 * key ~0 / +inf) into one nq x k list.  Device buffers; parts are contiguous:
 * d_part_keys[p][q][k].  Follows the RCCL all-gather in the sharded search. ------------- */
VS_API int vs_topk_merge_device(const uint64_t* d_part_keys, const float* d_part_dists, size_t parts, size_t nq, size_t k,
                         uint64_t* d_keys, float* d_dists, uint32_t* d_found, void* hip_stream);

/* The same merge over `parts` packed blocks of block_bytes each (a multiple of 16), block p = [nq x k keys u64 | nq x k
 * distances f32 | padding]: the receive buffer of the ONE ncclAllGather per batch of include/vs_ranks.h. */
VS_API int vs_topk_merge_packed_device(const void* d_blocks, size_t parts, size_t block_bytes, size_t nq, size_t k,
                                uint64_t* d_keys, float* d_dists, uint32_t* d_found, void* hip_stream);

/* -- small host helpers that the reference keeps next to the wrapper ---------------------- */
/* f32_to_b1x8 (usearch.rs:1179-1205): out has ceil(n/8) bytes. */
VS_API void vs_f32_to_b1x8(const float* v, size_t n, uint8_t* out);
/* Distance::try_from (distance.rs:58-105): 1 when the value is acceptable for the metric. */
VS_API int vs_distance_valid(float value, int metric, size_t dimensions);
/* SimilarityScore::from (similarity.rs:28-35). */
VS_API float vs_similarity_score(float distance, int metric, size_t dimensions);

/* usearch::version() analogue; the service reports "usearch-<ver>" (usearch.rs:109-114). */
VS_API const char* vs_hnsw_version(void);
VS_API const char* vs_hnsw_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* VS_HNSW_H */
