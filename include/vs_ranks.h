/* vs_ranks.h -- key-range sharded search across the GPUs of one node, ONE PROCESS PER GPU (libvs_ranks.so).
 *
 * BASELINE.json configs[3] / SURVEY.md section 8(e): the index is partitioned by key range -- rank r of `world` owns the
 * rows [r * per, (r + 1) * per), per = ceil(total_rows / world), of the PrimaryId's row index (low 48 bits,
 * reference table/primary_id.rs:35-45) -- every rank holds an independent HNSW graph over its range (include/vs_hnsw.h),
 * every query batch is answered by every shard, and the per-shard top-k lists meet in ONE RCCL ncclAllGather per batch
 * over xGMI, followed by a k-way merge on every rank (topk_merge_kernel, one wavefront per query).
 *
 * The all-gather moves one packed block per rank -- [nq x k keys u64 | nq x k distances f32], 12 B per candidate --
 * in place, on the communicator's own HIP stream; the walk writes straight into this rank's block, so nothing is packed
 * or copied.  With the pipelined entry points the collective and the merge of batch i run while batch i + 1 walks.
 *
 * The reference has no cross-index merge (a query touches exactly one partition index, usearch.rs:787-803); its analogue
 * of the fan-in is the per-partition dispatch at usearch.rs:766-803.  What a Rust service binds: one vs_ranks per worker
 * process, the 128-byte communicator id carried over whatever channel the processes already share.
 *
 * Conventions as in vs_hnsw.h: int status codes, vs_ranks_last_error() on the calling thread, device pointers + stream.
 */
#ifndef VS_RANKS_H
#define VS_RANKS_H

#include "vs_hnsw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_ranks vs_ranks;

#define VS_RANKS_ID_BYTES 128 /* ncclUniqueId */

/* Rank 0 creates the id (ncclGetUniqueId) and hands it to every other rank out of band. */
VS_API int vs_ranks_unique_id(uint8_t id[VS_RANKS_ID_BYTES]);

/* Joins the communicator (ncclCommInitRank; collective: every rank calls it -- a rank that never arrives leaves the others
 * waiting inside RCCL's bootstrap, as with any NCCL program: launch all ranks or none) around this rank's shard.
 * A search call, by contrast, never leaves the others hanging: a rank whose local search fails still enters the all-gather
 * with an empty block and reports its error afterwards.  `shard` stays
 * owned by the caller and must outlive the handle.  total_rows: size of the key space that is split into ranges. */
VS_API int vs_ranks_create(vs_hnsw* shard, int rank, int world, const uint8_t id[VS_RANKS_ID_BYTES], uint64_t total_rows,
                           vs_ranks** out);
/* The same with the exchange named explicitly.  VS_RANKS_RCCL (what vs_ranks_create uses unless the environment says
 * VS_RANKS_EXCHANGE=hostshm): ncclAllGather over xGMI, one rank per GPU -- the product path.  VS_RANKS_HOSTSHM: the
 * all-gather goes through a POSIX shared-memory segment named after the id (D2H, flags, H2D on the communicator stream):
 * for ranks that SHARE a device, which RCCL refuses to pair -- the one-GPU boxes of the test pool run the whole sharded
 * path (in-place blocks, two-slot pipeline, packed merge) with world > 1 that way.  Not a data path for production. */
typedef enum vs_ranks_exchange { VS_RANKS_RCCL = 0, VS_RANKS_HOSTSHM = 1 } vs_ranks_exchange;
VS_API int vs_ranks_create_ex(vs_hnsw* shard, int rank, int world, const uint8_t id[VS_RANKS_ID_BYTES], uint64_t total_rows,
                              int exchange, vs_ranks** out);
VS_API int vs_ranks_exchange_kind(const vs_ranks* r); /* vs_ranks_exchange of the handle */
VS_API void vs_ranks_free(vs_ranks* r);

/* What the handle is part of: this rank, the world it was created with, and comm_ranks = the ranks that have joined its exchange
 * (RCCL: ncclCommCount; host exchange: the processes attached to the segment). */
VS_API int vs_ranks_world(const vs_ranks* r, int* rank, int* world, int* comm_ranks);
/* ncclCommCount of the handle's RCCL communicator; 0 when the handle has none (host exchange).  A world of one has a
 * communicator of one rank and issues the all-gather like any other.  bench.py prints this as `rccl_ranks` and refuses to report
 * an N-GPU run whose value is not N. */
VS_API int vs_ranks_rccl_ranks(const vs_ranks* r, int* n);
/* Queries (cumulative) to which THIS rank's shard contributed no candidates: its walk outgrew its workspace (d_found =
 * 0xFFFFFFFF from vs_hnsw_search_batch_device) or its local search failed to launch.  Such rows enter the all-gather as
 * (free key, +inf), so the merge of every rank stays well defined; a non-zero count means recall was lost.  Synchronises. */
VS_API int vs_ranks_unanswered(vs_ranks* r, uint64_t* queries);

/* Key-range ownership. */
VS_API int vs_ranks_owner(const vs_ranks* r, uint64_t key);
VS_API void vs_ranks_range(const vs_ranks* r, uint64_t* first_row, uint64_t* end_row);
/* The same ingest stream may be offered to every rank: only the keys this rank owns are added, *added counts them. */
VS_API int vs_ranks_add_batch(vs_ranks* r, const uint64_t* keys, const float* vectors, size_t n, size_t dim, size_t* added);

/* One batch: local walk on `hip_stream` -> ncclAllGather + merge on the communicator's stream -> `hip_stream` waits for
 * the merge.  Outputs (nq x k keys / distances, nq found) are complete in `hip_stream` order.  Collective. */
VS_API int vs_ranks_search_batch_device(vs_ranks* r, const float* d_queries, size_t nq, size_t dim, size_t k, uint64_t* d_keys,
                                        float* d_distances, uint32_t* d_found, void* hip_stream);
/* Same over the exact (brute-force) local search: global ground truth. */
VS_API int vs_ranks_exact_search_batch_device(vs_ranks* r, const float* d_queries, size_t nq, size_t dim, size_t k,
                                              uint64_t* d_keys, float* d_distances, uint32_t* d_found, void* hip_stream);
/* Pipelined form: returns once the walk, the all-gather and the merge of this batch are enqueued; `hip_stream` does NOT
 * wait for the merge, so the next batch's walk overlaps it.  slot = 0 / 1 (two batches may be in flight, each with its own
 * output buffers); vs_ranks_wait makes `hip_stream` wait for that slot's merge. */
VS_API int vs_ranks_search_submit(vs_ranks* r, int slot, const float* d_queries, size_t nq, size_t dim, size_t k, uint64_t* d_keys,
                                  float* d_distances, uint32_t* d_found, void* hip_stream);
VS_API int vs_ranks_wait(vs_ranks* r, int slot, void* hip_stream);

VS_API const char* vs_ranks_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
