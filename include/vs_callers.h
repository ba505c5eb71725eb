/* vs_callers.h -- the reference's search load loop against the C ABI, as a library call (libvs_callers.so).
 *
 * crates/benchmark's SearchHttp / search loop (reference crates/benchmark/src/main.rs:435-525): `threads` workers, each
 * issuing ONE query per call and waiting for the answer -- exactly how the service drives `usearch::Index::search` from
 * num_workers() + 1 threads (usearch.rs:203-222, worker.rs:44-118) -- for `seconds`, latencies recorded on the
 * reference's histogram (10,000 linear buckets over 1..100 ms, main.rs:539-604) and QPS = completed / wall.
 * inflight > 1 switches a worker to the non-blocking entry point (vs_hnsw_search_async) with that many queries
 * outstanding, which is what an async runtime binds.
 *
 * bench.py uses it to report, next to the kernel rate, what a drop-in caller gets through the boundary on the SAME
 * index; vs_bench's `search` command is the same loop over fbin / ibin files.
 */
#ifndef VS_CALLERS_H
#define VS_CALLERS_H

#include "vs_actor.h"
#include "vs_hnsw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_callers_result {
    double seconds;
    uint64_t queries;
    double qps;
    int64_t latency_min_ns, latency_max_ns;
    int64_t p01_ns, p10_ns, p25_ns, p50_ns, p75_ns, p90_ns, p99_ns; /* INT64_MAX: beyond the 100 ms window; values below
                                                                         1 ms read as the first bucket (1 ms), as in the reference */
    double recall_avg; /* against `truth`; -1 when none was given */
    uint64_t errors;
    uint64_t launches, team_launches; /* kernel launches of the single-query dispatcher during the run */
} vs_callers_result;

/* queries: nq x dim on the host; truth: nq x k neighbour keys (or NULL). */
VS_API int vs_callers_run(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, const uint64_t* truth,
                          unsigned threads, unsigned inflight, double seconds, vs_callers_result* out);

/* What the callers RECEIVED (round 5: the boundary legs of bench.py compare every answer with the CPU oracle's, not only recall):
 * the first `cap` completed calls of a run, in completion order -- query index, result count, k keys, k distances each.  `n` = calls
 * recorded. */
typedef struct vs_callers_record {
    uint32_t* query;  /* cap */
    uint32_t* found;  /* cap */
    uint64_t* keys;   /* cap x k */
    float* distances; /* cap x k */
    size_t cap, n;
} vs_callers_record;
VS_API int vs_callers_run_recorded(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, const uint64_t* truth,
                                   unsigned threads, unsigned inflight, double seconds, vs_callers_result* out, vs_callers_record* record);
VS_API int vs_callers_run_filtered_recorded(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus,
                                            unsigned threads, double seconds, vs_callers_result* out, uint64_t extra[4],
                                            vs_callers_record* record);

/* ... and with a NAMED filter (vs_hnsw_filtered_search_keyed: the engine remembers the predicate's verdicts across the queries that carry
 * filter_key; 0 = the plain call). */
VS_API int vs_callers_run_filtered_keyed(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus,
                                         uint64_t filter_key, unsigned threads, double seconds, vs_callers_result* out, uint64_t extra[4],
                                         vs_callers_record* record);

/* The same loop over vs_hnsw_filtered_search: the reference runs every filtered query on a blocking thread (spawn_blocking,
 * usearch.rs:937-948) with a predicate that takes a table read-lock per call (usearch.rs:1118-1124).  Predicate here:
 * key % modulus == 0 (selectivity 1 / modulus), counted.  extra: [0] predicate calls, [1] results returned, [2..3] 0.
 * `errors` also counts results the predicate rejects (must stay 0). */
VS_API int vs_callers_run_filtered(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus,
                                   unsigned threads, double seconds, vs_callers_result* out, uint64_t extra[4]);

/* The reference's MIXED workloads (crates/vector-store/benches/pipeline.rs:508-1292: cdc_insert, cdc_update, cdc_delete and
 * search_while_{inserting,updating,deleting}) through a dispatch actor (include/vs_actor.h -- over the HIP engine, or over any
 * other index bound with vs_actor_create_with): `producers` CDC producers (BENCHES_CONCURRENCY, pipeline.rs:158-163; default
 * 1), each sending one item at a time and waiting for its in-progress marker (pipeline.rs:685-712), beside `plain_callers`
 * blocking callers of Ann and `filtered_callers` of FilteredAnn (predicate key % modulus == 0) on the same partition.
 *   insert: AddVector(first_new_key + i, vectors[i % nv])                                   monitor_items.rs:263-284
 *   update: RemoveBeforeAddValue(key) then AddVector(key, vectors[i % nv]), key random in [0, existing_keys)   :301-313
 *   delete: RemoveValue(delete_from + i)                                                     :314-327
 * Runs for `seconds` (or until max_items), then waits until every producer's last item is applied. */
typedef enum vs_mixed_modify { VS_MIXED_NONE = 0, VS_MIXED_INSERT = 1, VS_MIXED_UPDATE = 2, VS_MIXED_DELETE = 3 } vs_mixed_modify;
typedef struct vs_mixed_options {
    unsigned plain_callers, filtered_callers, producers;
    int modify; /* vs_mixed_modify */
    uint64_t modulus, partition, first_new_key, existing_keys, delete_from, max_items;
    size_t k;
    double seconds;
    uint64_t filter_key; /* != 0: the filtered callers name their filter (vs_actor_filtered_ann_keyed) */
} vs_mixed_options;
typedef struct vs_mixed_result {
    double seconds;
    uint64_t items, adds_applied, removes_applied; /* CDC items completed; index calls that took effect */
    uint64_t predicate_calls, filtered_results, errors;
    vs_callers_result item, plain, filtered; /* latency of an item (send .. marker dropped), of a plain / filtered query */
} vs_mixed_result;
VS_API int vs_mixed_run(vs_actor* actor, const vs_mixed_options* options, const float* queries, size_t nq, const float* vectors,
                        size_t nv, size_t dim, vs_mixed_result* out);

#ifdef __cplusplus
}
#endif
#endif
