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
    int reserved;     /* 0 (test hooks: include/vs_hnsw_debug.h, VS_DEBUG_*) */
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
/* The same call for a filter that has a NAME: `filter_key` != 0 identifies the restrictions the predicate evaluates.  The reference's
 * predicate is a table read-lock + restriction evaluation per candidate (usearch.rs:1118-1124), the same function of the key for every
 * query with the same `Filter::restrictions` -- UNTIL the table changes: restrictions cover the index's filtering columns, which
 * `Table::upsert` rewrites in place under a fixed PrimaryId (`update_columns`, table/mod.rs:676-695, 1053-1061; no remove / add follows
 * when the vector's timestamp is not newer, :905-910).  The engine remembers verdicts across the queries of a named filter (two bits per
 * slot in HBM, up to 4 filters per index, least recently used first out): a key is asked about at most once, and once a filter's
 * neighbourhoods are known a query is one exact walk with no predicate call.
 * CONTRACT: results are those of vs_hnsw_filtered_search provided the caller tells the engine whenever predicate(key) may have changed
 * for a filter it named: vs_hnsw_filter_forget_keys for the rows whose filtering columns were rewritten, or vs_hnsw_filter_forget for the
 * whole filter (or a new filter_key: e.g. one drawn from a counter per (restrictions, table generation)).  A query that overlaps such a
 * call may see either verdict -- as a usearch query that overlaps the upsert's `table.write()` may.  The engine itself forgets a member that
 * is removed or whose slot is re-used.  `filter_key` must be unique per restrictions (a registry id, not a bare hash of caller-supplied
 * values: INTEGRATION.md section 5c).  filter_key == 0: no memory (= vs_hnsw_filtered_search); this is what the trait's signature binds. */
VS_API int vs_hnsw_filtered_search_keyed(vs_hnsw* index, const float* query, size_t dim, size_t k, vs_hnsw_predicate predicate,
                                         void* ctx, uint64_t filter_key, uint64_t* keys, float* distances, size_t* found);
/* The named filter `filter_key` starts over (0: every named filter of the index); *dropped (may be NULL) = memories dropped.  Queries
 * that start after the call returns ask the predicate afresh. */
VS_API int vs_hnsw_filter_forget(vs_hnsw* index, uint64_t filter_key, size_t* dropped);
/* Every named filter forgets what it knows about these members (unknown keys are ignored): what the host calls after it rewrote
 * their filtering columns (table/mod.rs:1053-1061).  Queries that start after the call returns ask about them again. */
VS_API int vs_hnsw_filter_forget_keys(vs_hnsw* index, const uint64_t* keys, size_t n);
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

/* Counters, timers and test hooks: include/vs_hnsw_debug.h (same library; nothing a binding of the trait needs). */

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
