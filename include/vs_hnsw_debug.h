/* vs_hnsw_debug.h -- counters, timers and test hooks of libvs_hnsw.so.
 *
 * NOT part of the drop-in boundary: an `impl UsearchIndex for HipIndex` binds include/vs_hnsw.h and nothing of this
 * file.  What is here serves bench.py (the roofline's algorithmic bytes come from vs_hnsw_stats), the tests (A/B hooks that force a
 * code path) and the measurement notes in DESIGN.md.  Same conventions as vs_hnsw.h.
 */
#ifndef VS_HNSW_DEBUG_H
#define VS_HNSW_DEBUG_H

#include "vs_hnsw.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hooks: bits of vs_hnsw_options::reserved (0 in production). */
enum vs_debug_hook {
    VS_DEBUG_TINY_VISITED = 1 << 0,      /* tiny visited table in search (forces the overflow path) */
    VS_DEBUG_EXACT_ON_VALU = 1 << 1,     /* exact search on the VALU tile kernel instead of MFMA */
    VS_DEBUG_ALWAYS_TEAM = 1 << 2,       /* always serve a query with a team of wavefronts */
    VS_DEBUG_NEVER_TEAM = 1 << 3,        /* ... never (default: batches of at most one team per CU) */
    VS_DEBUG_USEARCH_ORDER = 1 << 4,     /* usearch-order walk (two structures, exact tie order) for every search (default: i8 / b1 only) */
    VS_DEBUG_GLOBAL_BITMAP_WALK = 1 << 5,/* always the global-bitmap instance of that walk */
    VS_DEBUG_WIDE_VISITED_TAGS = 1 << 6, /* wide visited tags (the instances for indexes above 2^25 / 2^26 slots) on a small index */
    VS_DEBUG_TINY_WALK_HEAP = 1 << 7,    /* 64-entry global heap for the global-bitmap walk (a flooding walk then reports "outgrew its
                                            workspace", which the host entry points answer by ranking exhaustively) */
    VS_DEBUG_NO_PIPELINED_WALK = 1 << 8  /* lone queries never take the pipelined walk: the team kernels serve them (A/B in tests) */
};

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

/* Unnamed filters (vs_hnsw_filtered_search: an opaque predicate) are ONE walk that asks the caller while it runs (round 6):
 * [0] queries answered that way, [1] walks that handed over to the rounds of rounds 3-5 (an order-relevant tie, a host that did not
 * answer), [2] queries no pod could take, [3] predicate calls of the asking walks; on the device: [4] times a walker stood still for
 * answers, [5] 100 MHz ticks it did, [6] hops, [7] ticks of the walks as a whole. */
VS_API int vs_hnsw_filter_ask_stats(vs_hnsw* index, uint64_t out[8]);

/* Named filters (vs_hnsw_filtered_search_keyed): [0] queries answered with a filter memory, [1] verdicts they still asked the host
 * for, [2] memories created, [3] memories held now, [4] vs_hnsw_filter_forget / _forget_keys calls, [5] members those forgot. */
VS_API int vs_hnsw_filter_memo_stats(vs_hnsw* index, uint64_t out[6]);

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
/* Round 6: the first stage runs over an 8-BIT plane of the rows (int8 + one f32 scale per row: +1 byte per element of HBM) on the int8
 * matrix pipe: [0] batches that took it, [1] of them handed on to the bf16 plane (uncertified), [2] the f32 bits of the largest
 * relative quantisation residual over its rows (the band of its nominations), [3] rows it covers.  exact_stats2's [0] / [1] count the
 * batches that entered the plane stages (either plane, once) and those that left them uncertified. */
VS_API int vs_hnsw_exact_stats3(vs_hnsw* index, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* VS_HNSW_DEBUG_H */
