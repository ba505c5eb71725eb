/* vs_actor.h -- host-side dispatch actor over the HNSW engine (libvs_actor.so).
 *
 * Reproduces the concurrency contract of the reference's per-index actor
 * (crates/vector-store/src/vs_index/usearch.rs:688-1177) for callers that do not bring their own:
 *   - two bounded channels, search drained before modify      vs_index/mod.rs:30-45 (`biased` select)
 *   - Operation permits: Insert||Insert, Search||Search, never mixed; Reserve and Remove run alone
 *                                                              usearch.rs:515-624
 *   - capacity growth: when capacity - size < the free threshold (reference: 3 x workers; here 4 x workers + 1,
 *     everything that can be in flight), reserve capacity + 1,000,000 (global index)
 *     or + 1,000 (local, per-partition-key index)              usearch.rs:442-443, 655-665, 908-921
 *   - one engine handle per partition, created lazily on its first AddVector   usearch.rs:757-779
 *   - adds are dropped while the memory guard says Allocate::Cannot           usearch.rs:1156-1177
 *   - adds / removes are fire-and-forget; searches are a round trip; Count is answered by the actor
 *   - a fixed pool of `workers` threads, channel depth 3 x workers            worker.rs:44-118, perf.rs:11-25
 * In a drop-in deployment the Rust actor stays and calls include/vs_hnsw.h directly; this library is
 * for standalone use (vs_bench mixed workloads, tests written like the reference's unit tests).
 */
#ifndef VS_ACTOR_H
#define VS_ACTOR_H

#include "vs_hnsw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_actor vs_actor;

typedef struct vs_actor_options {
    vs_hnsw_options index; /* IndexOptions of every partition's engine handle */
    size_t workers;        /* 0 = hardware concurrency; perf::num_workers() in the reference */
    int local;             /* 0: global index (+1,000,000 per reserve); 1: local (+1,000) */
    size_t reserve_increment; /* 0 = the reference's constants above */
} vs_actor_options;

VS_API int vs_actor_create(const vs_actor_options* options, vs_actor** out);

/* The reference's actor is generic over the index behind it -- `new<I: UsearchIndex + Send + Sync + 'static>`
 * (usearch.rs:688-696); its own benches and tests run it over a Simulator through that parameter.  The same here: the seven
 * methods of `trait UsearchIndex` (usearch.rs:142-160: reserve, capacity, add, remove, search, filtered_search, stop) plus
 * construction (`index_fn`, usearch.rs:689) as a table of C functions.  NULL entries are not allowed.  vs_actor_create binds
 * the HIP engine (vs_hnsw_*); bench.py's cpu_baseline leg and the tests bind the CPU oracle to time / check the SAME actor
 * over it. */
typedef struct vs_actor_index_vtable {
    int (*create)(const vs_hnsw_options* options, void** out);                      /* index_fn */
    void (*stop)(void* index);                                                      /* stop + drop */
    int (*reserve)(void* index, size_t capacity, size_t threads);
    size_t (*capacity)(void* index);
    int (*add)(void* index, uint64_t key, const float* vector, size_t dim);
    int (*remove)(void* index, uint64_t key, int* removed);
    int (*search)(void* index, const float* query, size_t dim, size_t k, uint64_t* keys, float* distances, size_t* found);
    int (*filtered_search)(void* index, const float* query, size_t dim, size_t k, vs_hnsw_predicate predicate, void* ctx,
                           uint64_t* keys, float* distances, size_t* found);
    const char* (*last_error)(void);
    /* optional (NULL: filtered_search serves): the filtered search of a NAMED filter, vs_hnsw_filtered_search_keyed */
    int (*filtered_search_keyed)(void* index, const float* query, size_t dim, size_t k, vs_hnsw_predicate predicate, void* ctx,
                                 uint64_t filter_key, uint64_t* keys, float* distances, size_t* found);
} vs_actor_index_vtable;
VS_API int vs_actor_create_with(const vs_actor_options* options, const vs_actor_index_vtable* index, vs_actor** out);

/* Hands an index that already exists (bulk-built, imported) to the actor as partition `partition`: `size` members, capacity as
 * the index reports it.  The actor does not own it (it is never stopped / freed by the actor).  Before any message for that
 * partition. */
VS_API int vs_actor_adopt_partition(vs_actor* actor, uint64_t partition, void* index, size_t size);
/* Closes both channels, drains what is queued, stops every partition, joins the threads. */
VS_API void vs_actor_stop(vs_actor* actor);

/* VsIndexModify (vs_index/actor.rs:21-44): return once queued; block only while the channel is full. */
VS_API int vs_actor_add_vector(vs_actor* actor, uint64_t partition, uint64_t primary_id, const float* vector, size_t dim);
VS_API int vs_actor_remove_vector(vs_actor* actor, uint64_t partition, uint64_t primary_id);
VS_API int vs_actor_remove_partition(vs_actor* actor, uint64_t partition);
/* The same two messages with the reference's in-progress marker (`AsyncInProgress`, dropped when the worker has processed the
 * message: what benches/pipeline.rs:685-712 waits for per item): return once the index call has run.  *applied: 1 = the vector
 * was indexed / a member was removed, 0 = dropped (memory guard, unknown partition, swallowed error). */
VS_API int vs_actor_add_vector_wait(vs_actor* actor, uint64_t partition, uint64_t primary_id, const float* vector, size_t dim, int* applied);
VS_API int vs_actor_remove_vector_wait(vs_actor* actor, uint64_t partition, uint64_t primary_id, int* applied);

/* VsIndexSearch (vs_index/actor.rs:46-61): blocking round trip.  Unknown / empty partition => found = 0. */
VS_API int vs_actor_ann(vs_actor* actor, uint64_t partition, const float* query, size_t dim, size_t k, uint64_t* keys,
                        float* distances, size_t* found);
VS_API int vs_actor_filtered_ann(vs_actor* actor, uint64_t partition, const float* query, size_t dim, size_t k,
                                 vs_hnsw_predicate predicate, void* ctx, uint64_t* keys, float* distances, size_t* found);
/* FilteredAnn whose filter has a name: a fingerprint of `Filter::restrictions` (vs_index/actor.rs:53-59), see vs_hnsw_filtered_search_keyed. */
VS_API int vs_actor_filtered_ann_keyed(vs_actor* actor, uint64_t partition, const float* query, size_t dim, size_t k,
                                       vs_hnsw_predicate predicate, void* ctx, uint64_t filter_key, uint64_t* keys, float* distances,
                                       size_t* found);
VS_API size_t vs_actor_count(vs_actor* actor);

/* Memory guard (memory.rs Allocate::{Can, Cannot}): can = 0 makes the actor drop AddVector messages. */
VS_API void vs_actor_set_allocate(vs_actor* actor, int can);

/* Introspection for tests. */
VS_API size_t vs_actor_partition_capacity(vs_actor* actor, uint64_t partition);
VS_API size_t vs_actor_partitions(vs_actor* actor);
/* [0] adds executed, [1] adds dropped by the memory guard, [2] reserves, [3] searches, [4] removes,
 * [5] mode switches, [6] max operations in flight, [7] add/remove errors swallowed (logged in the reference) */
VS_API void vs_actor_counters(vs_actor* actor, uint64_t out[8]);
VS_API const char* vs_actor_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
