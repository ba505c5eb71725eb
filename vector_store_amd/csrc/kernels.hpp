// kernels.hpp -- host-callable launchers of the HIP kernels (implemented in kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "hnsw_device.hpp"

namespace vs {

// stats layout == vs_hnsw_stats() (include/vs_hnsw.h)
enum : int { ST_SEARCH_EVALS = 0, ST_SEARCH_HOPS, ST_QUERIES, ST_ADD_EVALS, ST_ADD_HOPS, ST_ADDED, ST_OVERFLOW, ST_LINK_EVALS, ST_COUNT };  // ST_LINK_EVALS: the part of ST_ADD_EVALS spent in hnsw_link_kernel

constexpr int kSearchTeam = 8;  // 512-thread workgroups: one team per CU at the kernel's register footprint
constexpr int kSearchTeamMid = 4;  // search only: two teams per CU, for 257..768 queries on the device

struct SearchArgs {
    IndexView ix;
    const float* queries;  // nq x q_stride floats (unpadded rows)
    uint32_t q_stride;
    uint32_t nq, k, ef;
    uint32_t has_removed;         // some members carry the free key: the beam keeps ef LIVE entries
    uint32_t stress_small_table;  // test hook: 256-bucket visited table (iters == 1 only) to force overflow
    uint32_t team;                // waves per query: 1, kSearchTeamMid or kSearchTeam for batches too small to fill the chip
    uint32_t wide_tags;           // index beyond the plain visited tags' reach (2^25 / 2^26 slots): the wide-tag instance
    uint64_t* out_keys;    // nq x k, padded with kFreeKey
    float* out_dist;       // nq x k, padded with +inf
    uint32_t* out_found;   // nq
    unsigned long long* stats;
    const uint32_t* qlist = nullptr;   // second chance behind the pipelined walk (kernels_pipe.hip): serve queries qlist[0 .. *qcount) of the
    const uint32_t* qcount = nullptr;  // batch (grid = nq: workgroups beyond *qcount leave at once); nullptr: every query
};

struct InsertArgs {
    IndexView ix;
    const uint32_t* slots;    // n
    const int32_t* levels;    // n
    const uint32_t* req_off;  // n: first request index of node b (levels*(M) requests, one block per level)
    uint32_t n, ef_add;
    uint32_t team;            // waves per new node: 1, or kSearchTeam for sub-batches too small to fill the chip
    uint32_t wide_tags;       // see SearchArgs
    uint32_t tie_newest;      // order among equal distances: 0 pseudo-random per node, 1 usearch's (newest first)
    uint32_t req_base;        // req_off[first node of this sub-batch]
    uint64_t* req_key;        // (level << 32) | target, ~0 = unused
    uint64_t* req_val;        // (float bits of d(source,target) << 32) | source
    unsigned long long* stats;
};

struct LinkArgs {
    IndexView ix;
    const uint64_t* req_key;  // sorted
    const uint64_t* req_val;
    uint32_t total;
    uint32_t cache_rows;      // accepted neighbours whose rows the re-selection keeps in LDS (0 = none)
    uint32_t tie_newest;      // see InsertArgs (here: later list members first, new links last, as sorted_buffer_gt orders them)
    unsigned long long* stats;
};

// One round of one lazily filtered query in a batched launch of the pipelined walk (WalkArgs::pipe_qtable).  Lives in pinned host
// memory (device-mapped), like everything in it that is marked pinned.
struct PipeQuery {
    const float* query;        // pinned: dim floats
    uint32_t* allow;           // device: one bit per slot; `known` follows at +words
    uint32_t* known;
    uint32_t words;            // words of each bitmap
    uint32_t zero_bits;        // the query's first round: 1 = both bitmaps are zeroed first; 2 = they are SEEDED from `memo` (round 5)
    uint32_t* list;            // pinned: the slots whose verdict the LAST round missed (input, apply_m of them) -- and this round's (output)
    const uint8_t* verdict;    // pinned: the host's verdicts for list[0 .. apply_m)
    uint32_t apply_m;
    uint32_t slots;            // slots of the index (bound of a listed slot)
    uint32_t cap;              // entries `list` holds
    uint32_t budget;           // WalkArgs::unknown_budget of this round
    uint32_t k;
    uint32_t round_id;         // what the flag (cnt[8]) becomes once everything is visible to the host
    uint32_t* cnt;             // pinned [16]: listed, consulted, found (kPipeRedoFound: not answered), evaluations, 100 MHz ticks the workgroup spent on it; [8] = the flag
    uint64_t* keys;            // pinned: k keys, followed by k distances (f32)
    char* space;               // device: this query's visited bitmap / log / spill slots (all zero between rounds)
    // Verdicts remembered ACROSS queries of one filter (vs_hnsw_filtered_search_keyed, round 5): device bitmaps [allow: memo_stride
    // words | known: memo_stride words] shared by every query that names the filter.  A query seeds its own bitmaps from them
    // (zero_bits 2) and ORs every verdict the host gives it into them (allow before known, so a reader that sees `known` sees the
    // verdict).  nullptr: no memory.
    uint32_t* memo;
    uint32_t memo_stride;
    uint32_t pad_;
};
static_assert(sizeof(PipeQuery) == 112, "a PodSlot is one 128-byte line");

// The usearch-order walk (walk_device.hpp / kernels_walk.hip): persistent workgroups, one WalkSpace each.
enum : uint32_t { WALK_LDS_128 = 0, WALK_LDS_256, WALK_LDS_512, WALK_GLOBAL_512, WALK_GLOBAL_2048, WALK_GLOBAL_10240,
                  WALK_LDS_128_TINY /* test hook: 256-bucket visited table (1-chunk rows only), forces the retry path */,
                  WALK_LDS_128_SMALL /* 512-bucket two-choice table (4,096 entries, slots < 2^24): 14.6 KB of LDS, 10 walks per CU */,
                  WALK_LDS_320 /* beams of 257..288 (`top` of 320): the 256 instance's visited table, 5 walks per CU instead of 3 */,
                  WALK_LDS_256_DENSE /* beams of 129..256 below 2^24 slots: 512 x 12-tag buckets instead of 1,024 x 8: 22.0 KB, 7 walks per CU */ };
constexpr uint32_t kWalkTeamFlag = 0x100;      // instance | flag: one workgroup of kSearchTeam waves per query (LDS instances up to 320, unfiltered; WALK_GLOBAL_512, filtered too)
constexpr uint32_t kWalk320MaxBeam = 288;      // ~23.6 evaluations per beam entry: 6,800 of the table's 8,192 entries
constexpr uint32_t kWalkFailed = 0xFFFFFFFFu;  // out_found: the walk outgrew its workspace, the query was not answered
constexpr uint32_t kMaxWalkBeam = 10240;       // widest `top` of the walk instances

struct WalkArgs {
    IndexView ix;
    const float* queries;  // nq x q_stride floats
    uint32_t q_stride;
    uint32_t nq, k, ef;
    uint32_t has_removed;
    const uint32_t* allow;   // filtered search: bit s = slot s may be a result; nullptr = every live member
    uint32_t allow_stride;   // words between the bitmaps of consecutive queries (0: one bitmap for the batch)
    // Lazy predicate (filtered search on large indexes): `known` says for which slots `allow` is valid.  A slot the walk
    // needs but does not know yet is appended to unknown_list and taken as rejected; the host evaluates the predicate for
    // the listed slots and launches again -- a launch that lists nothing was exact.  nullptr: `allow` is complete.
    const uint32_t* known;
    uint32_t* unknown_list;   // unknown_cap entries per query
    uint32_t* unknown_count;  // one per query; the walk stops once it has listed unknown_budget slots
    uint32_t unknown_cap, unknown_budget;
    uint32_t* consulted;      // nullptr, or one word per query: verdicts the walk asked for (known or not) -- sizes the first lazy round of later queries
    const uint32_t* qlist;   // retry instance: serve queries qlist[0 .. *qcount) instead of 0 .. nq
    const uint32_t* qcount;
    uint32_t* retry_list;    // LDS instances: queries whose visited table or heap ran out are appended here ...
    uint32_t* retry_count;   // ... and served by a global-bitmap instance launched behind (nullptr: flag kWalkFailed)
    uint32_t* work_counter;  // zero at launch: workgroups draw the next query from it (a workgroup that becomes resident
                             // late -- the occupancy query can be one per CU too high -- then simply finds nothing left)
    char* space;             // grid x space_stride bytes: [bitmap_words u32 | vlog_cap u32 | heap_cap uint2] per workgroup
    size_t space_stride;
    uint32_t bitmap_words, vlog_cap, heap_cap;
    uint64_t* out_keys;
    float* out_dist;
    uint32_t* out_found;
    unsigned long long* stats;
    uint32_t* debug;         // nullptr, or nq x 12 words: largest `next`, nodes evaluated, hops, admitted, 8 phase clocks (VS_HNSW_WALK_DEBUG)
    // Batched rounds of lazily filtered queries (engine.hip FilterBatcher): entry blockIdx.x of this table names one query's buffers and
    // the kernel does the round's whole exchange itself -- the host's verdicts of the last round are applied first, the answer, the
    // counters and the list of missing verdicts go to the caller's pinned block, a flag there says when.  nullptr: the strided arrays above.
    const struct PipeQuery* pipe_qtable = nullptr;
    uint32_t pipe_explore = 0;   // pipelined walk, lazy filter: 2 = (pods) walks that ask while they run (round 6); 1 = an exploring round (lists missing verdicts, several candidates at a time; its answer is not one)
    uint32_t pipe_fused_order = 0;  // pipelined walk: the answer must be the fused-list kernel's also among EQUAL distances (plain queries of float indexes): any tie in `top` hands the query over
    uint32_t pipe_lds_visited = 0;  // pipelined walk: the visited set is the LDS tag table (unfiltered; slots < 2^25 at beams <= 256, 2^26 beyond) instead of the bitmap in a.space
    uint32_t pipe_pool_cap = 0;  // pipelined walk (kernels_pipe.hip): entries of `next` behind the front, in LDS (8 B each, <= 16,384)
};
inline size_t walk_space_stride(uint32_t bitmap_words, uint32_t vlog_cap, uint32_t heap_cap) {
    return (((size_t)bitmap_words + vlog_cap) * 4 + (size_t)heap_cap * 8 + 255) / 256 * 256;
}

// per-arithmetic launchers (explicitly specialised in kernels_arith.hip / kernels_walk.hip, one object per arithmetic)
// launch_walk: grid = min(queries, resident workgroups, grid_cap); grid_out != nullptr only reports that grid.
template <int AR> hipError_t launch_walk_ar(const WalkArgs& a, uint32_t iters, uint32_t instance, uint32_t grid_cap, hipStream_t s,
                                            uint32_t* grid_out);
hipError_t launch_walk(const WalkArgs& a, uint32_t iters, uint32_t instance, uint32_t grid_cap, hipStream_t s, uint32_t* grid_out);
// The pipelined walk for lone queries (pipe_device.hpp / kernels_pipe.hip): float arithmetics, M0 <= 64, beams <= 512, rows of up to
// 8 wave-loads; one workgroup and one WalkSpace (a.space) per query.  out_found == kPipeRedo: the usearch-order walk must answer.
template <int AR> hipError_t launch_pipe_walk_ar(const WalkArgs& a, uint32_t iters, hipStream_t s);
bool pipe_walk_supported(const IndexView& ix, uint32_t iters, uint32_t ef);
hipError_t launch_pipe_walk(const WalkArgs& a, uint32_t iters, hipStream_t s);
constexpr uint32_t kPipeRedoFound = 0xFFFFFFFEu;  // == kPipeRedo (pipe_device.hpp)
template <int AR> hipError_t launch_search_ar(const SearchArgs& a, uint32_t iters, hipStream_t s);
template <int AR> hipError_t launch_insert_ar(const InsertArgs& a, uint32_t iters, hipStream_t s);
template <int AR> hipError_t launch_link_ar(const LinkArgs& a, uint32_t iters, hipStream_t s);
int arith_of(int scalar, int metric);

// iters = stride4 / lanes must be one of {1,2,3,4,6,8,12,16}; search ef <= 512, insert ef_add <= 512.
bool search_supported(uint32_t iters, uint32_t ef);
hipError_t launch_search(const SearchArgs& a, uint32_t iters, hipStream_t s);
hipError_t launch_insert(const InsertArgs& a, uint32_t iters, hipStream_t s);
hipError_t launch_link(const LinkArgs& a, uint32_t iters, hipStream_t s);
// bits of slot ids the visited table of the kernel chosen for `ef` can distinguish (wide: the 4-extra-tag-bit instance)
uint32_t visited_domain_bits(uint32_t ef, bool wide = false);
uint32_t walk_small_table_bits();  // slot bits WALK_LDS_128_SMALL can tell apart
uint32_t walk_instance_domain_bits(uint32_t instance);  // slot bits the LDS visited table of walk instance `instance` can tell apart (32: global bitmap)

// f32 rows (dim floats, src_stride apart) -> storage rows (cast as usearch does, zero padded) + aux;
// rows given by slots[] or first + i
hipError_t launch_quantise_rows(const IndexView& ix, uint4* vectors, float* aux, const float* src, uint32_t src_stride,
                                const uint32_t* slots, uint32_t first, uint32_t n, hipStream_t s);
// aux[] of rows 0..n that are already stored (import path)
hipError_t launch_aux_rows(const IndexView& ix, float* aux, uint32_t n, hipStream_t s);
// n rows: dst[r][0..dst_fill_to) = src[r][0..row_bytes) followed by zeros
hipError_t launch_copy_rows(void* dst, uint32_t dst_stride, const void* src, uint32_t src_stride, uint32_t row_bytes,
                            uint32_t dst_fill_to, uint32_t n, hipStream_t s);
hipError_t launch_fill_u32(uint32_t* p, uint32_t value, size_t n, hipStream_t s);
hipError_t launch_fill_rows_u32(uint32_t* base, uint32_t row_words, const uint32_t* rows, uint32_t nrows, uint32_t value,
                                hipStream_t s);
hipError_t launch_scatter_u64(uint64_t* dst, const uint32_t* idx, const uint64_t* src, uint32_t n, hipStream_t s);
hipError_t launch_scatter_u32(uint32_t* dst, const uint32_t* idx, const uint32_t* src, uint32_t n, hipStream_t s);

// Exact (brute-force) top-k, k <= 256.  scratch: see exact_scratch_bytes().
struct ExactArgs {
    IndexView ix;
    const float* queries;
    uint32_t q_stride, nq, k;
    uint32_t slots;  // rows [0, slots) are scanned; removed rows skipped
    uint32_t use_valu;  // 1: dot-product family on the VALU tile kernel instead of MFMA (cross-check)
    uint64_t* out_keys;
    float* out_dist;
    uint32_t* out_found;
};
size_t exact_scratch_bytes(uint32_t nq, uint32_t k, uint32_t dim);
hipError_t launch_exact(const ExactArgs& a, void* scratch, hipStream_t s);
// The same answer through the split-bf16 MFMA nomination pass + exact f32 re-score (kernels_misc.hip "block search"):
// float storage, cos / ip, k <= 64.  *d_uncertified (zeroed by the caller) counts the queries whose certificate failed:
// non-zero after the stream has drained => run launch_exact instead.  max_row_norm: max |row| (inner product; 1 for cosine).
bool block_search_supported(const IndexView& ix, uint32_t k);
size_t block_scratch_bytes(uint32_t nq, uint32_t dim);
hipError_t launch_block_search(const ExactArgs& a, void* scratch, float max_row_norm, uint32_t* d_uncertified, hipStream_t s);
// The one-product form (round 3): a bf16 plane of the rows (block1_plane_rows(slots) x block1_plane_k(ix), kept by the engine),
// ONE v_mfma_f32_16x16x32_bf16 product per score, the same re-score + certificate; rho = max |c - bf16(c)| / |c| over the plane's
// rows (launch_block1_plane_rows accumulates its f32 bits with atomicMax).  Needs >= 65,536 slots, k <= 64, float storage, cos / ip.
bool block1_supported(const IndexView& ix, uint32_t k);
uint32_t block1_plane_k(const IndexView& ix);
uint32_t block1_plane_rows(uint32_t slots);
size_t block1_scratch_bytes(uint32_t nq, uint32_t dim);
// The same over an 8-BIT plane (round 6): block1_plane_rows(slots) x block8_plane_k(ix) int8 + one f32 scale per row; rho8 as rho.
uint32_t block8_plane_k(const IndexView& ix);
size_t block8_scratch_bytes(uint32_t nq, uint32_t dim);
hipError_t launch_block8_plane_rows(const IndexView& ix, uint8_t* plane, float* scale, uint32_t first, uint32_t end, uint32_t slots, uint32_t* d_rho_bits, hipStream_t s);
hipError_t launch_block8_search(const ExactArgs& a, void* scratch, const uint8_t* plane, const float* scale, float rho, float max_row_norm,
                                uint32_t* d_uncertified, hipStream_t s);
hipError_t launch_block1_plane_rows(const IndexView& ix, uint16_t* plane, uint32_t first, uint32_t end, uint32_t slots, uint32_t* d_rho_bits, hipStream_t s);
hipError_t launch_block1_search(const ExactArgs& a, void* scratch, const uint16_t* plane, float rho, float max_row_norm, uint32_t* d_uncertified,
                                hipStream_t s);
hipError_t launch_row_norm_max(const IndexView& ix, uint32_t first, uint32_t n, uint32_t* d_max_bits, hipStream_t s);

// out[s] = distance(query, row s) for s in [0, n): one wave per row; result copied to host_out.
// d_scratch: n + dim + 64 floats.
hipError_t launch_distance_row(const IndexView& ix, const float* d_query, uint32_t n, float* d_scratch, hipStream_t s,
                               float* host_out);

hipError_t launch_topk_merge(const uint64_t* part_keys, const float* part_dist, uint32_t parts, uint32_t nq, uint32_t k,
                             uint64_t* out_keys, float* out_dist, uint32_t* out_found, hipStream_t s, size_t key_stride = 0,
                             size_t dist_stride = 0);  // strides in elements between consecutive parts; 0 = nq * k

// radix sort of (key, value) pairs by key bits [0, end_bit); temp sized by sort_temp_bytes()
size_t sort_temp_bytes(size_t n);
hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint64_t* vals_in,
                      uint64_t* vals_out, size_t n, unsigned end_bit, hipStream_t s);

// keys only, all 64 bits
size_t sort_keys_temp_bytes(size_t n);
hipError_t sort_keys(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, size_t n, hipStream_t s);

// Exhaustive ranking of every member against one query, on the device (k beyond the LDS beam, selective filters):
//   rank_keys   : rank[s] = (order-preserving bits of d[s]) << 32 | s, or ~0 for removed members (they sort last)
//   rank_emit   : sorted rank keys -> (member key, distance) in ascending (distance, slot) order
hipError_t launch_rank_keys(const IndexView& ix, const float* d, uint32_t n, uint64_t* rank, hipStream_t s);
hipError_t launch_rank_emit(const IndexView& ix, const uint64_t* sorted, uint32_t n, uint64_t* out_keys, float* out_dist, hipStream_t s);

}  // namespace vs
