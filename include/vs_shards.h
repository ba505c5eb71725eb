/* vs_shards.h -- one handle over several GPUs of a node, in ONE process (libvs_shards.so).
 *
 * The reference is a single process per node (tokio service); to use more than one GPU from it the
 * `UsearchIndex` implementation has to span devices itself.  A vs_shards handle owns one engine handle
 * (include/vs_hnsw.h) per listed device and presents the same operations:
 *   - a key belongs to shard ((key & (2^48-1)) / 4096) % n   -- 4096-row key ranges dealt round-robin, so a
 *     dense PrimaryId space (table/primary_id.rs:35-45) is balanced across devices whatever its size;
 *   - add / remove touch exactly one shard; reserve splits the capacity evenly;
 *   - search asks every shard (concurrently, through each device's dispatcher) for its top-k and merges the
 *     n x k candidates on the host -- the in-process twin of the RCCL all-gather + vs_topk_merge_device used
 *     by the one-process-per-GPU benchmark driver (vector_store_amd/sharded.py);
 * Each shard is an independent HNSW graph: recall is at least that of a single graph at equal ef
 * (SURVEY.md section 8e), capacity and build throughput scale with the number of devices.
 * The same device may be listed several times (used by the tests on a 1-GPU box).
 */
#ifndef VS_SHARDS_H
#define VS_SHARDS_H

#include "vs_hnsw.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_shards vs_shards;

VS_API int vs_shards_create(const vs_hnsw_options* options, const int* devices, size_t n_devices, vs_shards** out);
VS_API void vs_shards_free(vs_shards* s);
VS_API size_t vs_shards_count(const vs_shards* s);
VS_API size_t vs_shards_owner(const vs_shards* s, uint64_t key);

VS_API int vs_shards_reserve(vs_shards* s, size_t capacity, size_t threads);
VS_API size_t vs_shards_capacity(const vs_shards* s);
VS_API size_t vs_shards_size(const vs_shards* s);
VS_API int vs_shards_add(vs_shards* s, uint64_t key, const float* vector, size_t dim);
VS_API int vs_shards_add_batch(vs_shards* s, const uint64_t* keys, const float* vectors, size_t n, size_t dim);
VS_API int vs_shards_remove(vs_shards* s, uint64_t key, int* removed);
VS_API int vs_shards_search(vs_shards* s, const float* query, size_t dim, size_t k, uint64_t* keys, float* distances,
                            size_t* found);
VS_API int vs_shards_filtered_search(vs_shards* s, const float* query, size_t dim, size_t k, vs_hnsw_predicate predicate,
                                     void* ctx, uint64_t* keys, float* distances, size_t* found);
VS_API int vs_shards_search_batch(vs_shards* s, const float* queries, size_t nq, size_t dim, size_t k, uint64_t* keys,
                                  float* distances, size_t* found);
VS_API int vs_shards_set_expansion_search(vs_shards* s, size_t expansion_search);
VS_API int vs_shards_stats(vs_shards* s, uint64_t out[8], int reset);
VS_API const char* vs_shards_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
