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

/* The same loop over vs_hnsw_filtered_search: the reference runs every filtered query on a blocking thread (spawn_blocking,
 * usearch.rs:937-948) with a predicate that takes a table read-lock per call (usearch.rs:1118-1124).  Predicate here:
 * key % modulus == 0 (selectivity 1 / modulus), counted.  extra: [0] predicate calls, [1] results returned, [2..3] 0.
 * `errors` also counts results the predicate rejects (must stay 0). */
VS_API int vs_callers_run_filtered(vs_hnsw* index, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus,
                                   unsigned threads, double seconds, vs_callers_result* out, uint64_t extra[4]);

#ifdef __cplusplus
}
#endif
#endif
