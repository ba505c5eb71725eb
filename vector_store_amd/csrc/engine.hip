// engine.hip -- host side of the MI355X HNSW engine and its C ABI (include/vs_hnsw.h).
//
// Mirrors what `ThreadedUsearchIndex` expects of `usearch::Index`
// (reference crates/vector-store/src/vs_index/usearch.rs:162-251): reserve / capacity /
// add / remove / search / filtered_search on u64 keys and f32 vectors, blocking calls,
// errors as status codes.  Everything that touches vectors or the graph runs on the GPU
// (kernels_*.hip); the host keeps only the key->slot map, the free-slot ring, the level
// RNG and the batching of concurrent single-vector callers.  There is no CPU fallback.
#include <hip/hip_runtime.h>
#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <shared_mutex>
#include <string>
#include <thread>
#include <chrono>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/vs_hnsw_debug.h"
#include "kernels.hpp"
#include "pipe_pod.hpp"
#include "filter_rounds.hpp"

#define VS_VERSION "0.1.0"

#include "engine_base.hpp"
#include "engine_pods.hpp"

namespace vs {

// Workspace of the usearch-order walk kernels (kernels_walk.hip), one per (device, stream): launches on one stream
// run one after the other and may share it; hipFree synchronises the device, so growing it is safe.
struct WalkRes {
    DeviceBuf lds_space;  // LDS-visited instances: the part of `next` beyond LDS, per workgroup
    DeviceBuf g_space;    // global-bitmap instances: bitmap + visited log + heap, per workgroup
    DeviceBuf retry;      // [count u32, 63 pad words | query ids]
    DeviceBuf allow;      // filtered search: the allow-bitmap of the query
    std::mutex mu;        // held while a call sizes the buffers and enqueues its launches
    uint32_t* retry_seen = nullptr;  // pinned: [retried queries of the last LDS-instance launch, its batch size, its instance]
    hipEvent_t retry_seen_ev = nullptr;  // recorded behind the copy of word 0: the record is read only once it has landed, so the
                                         // count always belongs to the (batch size, instance) the host wrote with it
    bool retry_seen_valid = false;
    const void* retry_seen_owner = nullptr;  // the index that launch searched (streams are shared between indexes)
    size_t g_layout[3] = {0, 0, 0};  // (bitmap words, stride, bytes) the bitmaps of g_space are known to be zero for
};
static WalkRes& walk_res(int dev, hipStream_t st) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, std::unique_ptr<WalkRes>> all;  // leaked with the process
    std::lock_guard<std::mutex> g(mu);
    auto& r = all[((uint64_t)(uintptr_t)st << 8) ^ (uint64_t)(uint32_t)dev];
    if (!r) r.reset(new WalkRes());
    return *r;
}

// Sub-batches stay below 1/16 of the graph they are inserted into (nodes of one sub-batch do
// not see each other during their search phase); capped at one staging chunk = 16 rounds of the chip's
// 2048 resident waves, so the last, partly filled round costs ~3 % (8192: 4 rounds, ~12 %; measured at
// 10M x 768: 375k -> 407k vectors/s, recall@10 at ef 200 0.9507 -> 0.9516; 131072 would build 4 % faster
// still but loses 0.002 of recall).
constexpr uint32_t kMaxSubBatch = 32768;
constexpr uint32_t kSubBatchRatio = 16;
constexpr uint32_t kChunk = 32768;  // vectors staged per host->device copy
constexpr uint32_t kMaxBeam = 512;  // widest beam (and k) the LDS search kernel holds

struct Engine {
    // configuration (reference usearch.rs:74-82)
    uint32_t dim = 0, M = 16, M0 = 32, ef_add = 128;
    std::atomic<uint32_t> ef_search{64};
    std::atomic<int> exact_valu{0};  // options.reserved bit 1: exact search without MFMA
    int metric = VS_METRIC_COS;
    int device = 0;
    bool stress_small_table = false;  // vs_hnsw_options.reserved bit 0 (tests only)
    uint32_t max_sub_batch = kMaxSubBatch;
    int nt_policy = 0;                // 0 = by table size; VS_HNSW_NT_ROWS=1 / 0 forces non-temporal row loads on / off
    uint32_t link_cache_rows = 0;     // accepted rows the link kernel's re-selection keeps in LDS (set in init)
    uint32_t chunk_rows = kChunk;
    int order_mode = 0;               // 0 = usearch-order walk for the tie-heavy metrics (i8, b1), fused list otherwise;
                                      // 1 = always the usearch-order walk (reserved bit 4); 2 = never (A/B measurements only)
    bool tie_newest = true;           // build: equal distances ordered as usearch's sorted buffer orders them (newest first; in a
                                      // re-selected row: later members first, the new link last).  VS_HNSW_TIE=random: pseudo-random
                                      // per node (round 1), kept for A/B -- on 50x duplicated data it costs 0.20 of the tied recall
    bool exact_f32_only = false;             // VS_HNSW_EXACT=f32: exact search on the f32-input MFMA path only (A/B, tests)
    std::atomic<uint64_t> block_batches{0}, block_fallbacks{0};  // exact batches on the split-bf16 path / of them re-run in f32
    std::mutex norm_mu;
    float max_norm_value = 0.f;              // max |row| over rows [0, max_norm_slots) (inner product: scales the certificate)
    size_t max_norm_slots = 0;
    uint32_t* d_max_norm = nullptr;
    // One-product block search (kernels_misc.hip "block search, ONE bf16 product"): a bf16 plane of the rows, built lazily by the
    // first exact search that can use it and extended incrementally; rows [0, plane_done) are converted.
    std::mutex plane_mu;                     // held across a search that uses the plane (it may be resized by the next one)
    Arena ar_plane;
    size_t plane_done = 0, plane_rows_cap = 0;
    uint32_t* d_rho = nullptr;
    float plane_rho = 0.f;                   // max |c - bf16(c)| / |c| over the plane's rows
    bool plane_failed = false;               // no HBM left for the plane: the split-bf16 path serves from then on
    int exact_mode = 0;                      // VS_HNSW_EXACT: 0 = bf16 plane -> split bf16 -> f32; 1 = "bf16x3": split bf16 -> f32; 2 = "f32"
    std::atomic<uint64_t> plane_batches{0}, plane_fallbacks{0};
    // ... and over an 8-BIT plane first (round 6: half the bytes, half the LDS traffic, the int8 matrix pipe; kernels_misc.hip "8-BIT plane"):
    // int8 rows + one f32 scale per row, built and extended like the bf16 plane; an uncertified batch goes on to the bf16 plane.
    Arena ar_plane8, ar_p8scale;
    size_t plane8_done = 0, plane8_rows_cap = 0;
    uint32_t* d_rho8 = nullptr;
    float plane8_rho = 0.f;                  // max |c - c^| / |c| over the plane's rows
    bool plane8_failed = false, plane8_off = false;  // (off: VS_HNSW_EXACT=bf16, A/B)
    std::atomic<uint64_t> plane8_batches{0}, plane8_fallbacks{0};
    bool eager_filter = false;               // VS_HNSW_FILTER=eager: always one predicate call per live member (the full bitmap)
    std::atomic<uint64_t> pipe_launches{0};  // launches of the pipelined walk (tests)
    bool no_pipe = false;                    // options.reserved bit 8: lone queries never take the pipelined walk (A/B in tests)
    std::atomic<uint64_t> lazy_rounds{0}, lazy_predicate_calls{0};  // filtered_lazy: rounds / predicate calls so far (tests)
    std::atomic<uint32_t> lazy_sel_hint{0};   // fraction of the asked-about slots the recent filters admitted (1 / 65,536; moving average)
    std::atomic<uint32_t> lazy_need_hint{0};  // verdicts the recent filtered queries of this index needed (moving average): sizes the first round
    std::atomic<bool> lds_walk_bad[16] = {};  // per walk instance: more than a quarter of a batch outgrew its LDS structures -> global-bitmap instance at once
    std::atomic<bool> small_table_ok{true};  // usearch-order walk, beams <= 128: the half-size visited table is paying off
    std::atomic<bool> dense_table_ok{true};  // beams of 129..256: the dense instance (8 walks per CU) is paying off
    bool force_wide_tags = false;     // reserved bit 6 (tests): wide visited tags although the index is small
    bool force_global_walk = false;   // reserved bit 5 (tests): every search takes the global-bitmap walk instance
    uint64_t walk_domain_override = 0;  // VS_HNSW_WALK_DOMAIN_SLOTS (tests): the slot count the walk's instance choice assumes
    std::atomic<uint32_t> last_walk_instance{0xFFFFFFFFu};  // walk instance of the last search_device call (vs_hnsw_walk_info)
    bool tiny_walk_heap = false;      // reserved bit 7 (tests): the global-bitmap walk's `next` gets 64 entries of global memory, so a
                                      // walk that floods the graph (fewer live members than the beam) reports kWalkFailed
    int team_mode = 0;                // 0 = by batch size; reserved bit 2 = always a team per query, bit 3 = never
    uint32_t team_max_nq = 256;       // batches up to one team per CU take the team kernel
    int scalar = VS_SCALAR_F32;
    uint32_t lanes = 64, lanes_log2 = 6, iters = 1, stride4 = 64;
    uint32_t row_bytes = 0;  // payload bytes of one stored vector (usearch bytes_per_vector)

    // HBM arenas (the typed pointers below alias ar_*.base)
    Arena ar_vectors, ar_aux, ar_adj0, ar_upper, ar_upper_off, ar_keys, ar_levels;
    uint4* d_vectors = nullptr;
    float* d_aux = nullptr;
    uint32_t* d_adj0 = nullptr;
    uint32_t* d_upper = nullptr;
    uint32_t* d_upper_off = nullptr;
    uint64_t* d_keys = nullptr;
    int32_t* d_levels = nullptr;
    unsigned long long* d_stats = nullptr;
    size_t capacity = 0, upper_cap = 0;
    std::atomic<size_t> capacity_atomic{0};  // mirror of `capacity` for lock-free reads in add_one
    std::atomic<size_t> slots_atomic{0};     // mirror of `slots` for the search paths (they read it per query: no mod_mu there)
    // What the workspaces of posted / batched walks (visited bitmap | log | spill slots) are laid out for: the CAPACITY, not the slots in
    // use -- so the layout, and with it an open pod, survives adds (round 5); it changes with reserve, which closes the pods.
    size_t layout_slots() const { return capacity_atomic.load(std::memory_order_acquire); }

    // host bookkeeping (guarded by mod_mu)
    std::mutex mod_mu;
    size_t slots = 0, upper_blocks = 0;
    std::atomic<size_t> live{0};
    std::atomic<size_t> removed{0};  // tombstoned slots not yet reused
    size_t linked = 0;  // nodes present in the graph
    std::unordered_map<uint64_t, uint32_t> lookup;
    std::deque<uint32_t> free_slots;  // usearch ring_gt: FIFO
    std::vector<uint8_t> h_levels;
    std::vector<uint32_t> h_upper_off;
    std::vector<uint64_t> h_keys;  // key of every slot (kFreeKey: removed / unused): what a filter's allow-bitmap is built from
    // Searches read the arena pointers (view()) and launch under a shared lock; whatever may MOVE an arena (reserve,
    // growth of `upper` in the middle of an ingest) drains the device and swaps pointers under the exclusive lock.
    mutable std::shared_mutex view_mu;
    std::default_random_engine level_rng;  // same stream as a 1-thread usearch context (oracle orc_level_stream)
    double inv_log_m = 0;
    std::atomic<uint32_t> entry_slot{0};
    std::atomic<int32_t> max_level{-1};

    // Staged single-vector modifications (vs_hnsw_add / vs_hnsw_remove): validated and logged under pend_mu, applied in order and in
    // bulk by flush_pending() -- see add_one() / remove_one().
    struct Pending {
        static constexpr uint32_t kRemove = 0xFFFFFFFFu;
        struct Op {
            uint64_t key;
            uint32_t vec;  // index of the add's vector in `vecs`; kRemove: a remove
        };
        std::vector<Op> ops;
        std::vector<float> vecs;
        std::unordered_map<uint64_t, uint8_t> last;  // per key: 1 = its last staged operation is an add, 2 = a remove
        size_t adds = 0;
    };
    std::mutex pend_mu;
    std::condition_variable flush_cv;
    Pending pend;
    bool flushing = false;                                  // one flush at a time: logs are applied in the order they were taken
    std::unordered_map<uint64_t, uint8_t> flushing_last;   // the log being applied (host state not yet updated): final state per key
    std::atomic<size_t> queued{0};               // operations staged + being applied
    std::atomic<size_t> committed{0};            // live members + staged adds - staged removes: what capacity is checked against
    std::mutex key_mu;                           // guards `lookup` membership (adds validate against it concurrently)
    std::mutex poison_mu;
    std::string poison;                          // set when a deferred insertion failed: every later call reports it

    void use_device() const { HIP_OK(hipSetDevice(device)); }
    // Blocks that were replaced while pods were open are freed once none is (a free synchronises the device).  Called at the top of
    // the host entry points -- never with the pod pool's lock held (pod_submit calls use_device() under it).
    void housekeeping() const { drain_graveyard_if_idle(device); }

    IndexView view() const {
        IndexView v;
        v.vectors = d_vectors;
        v.aux = d_aux;
        v.adj0 = d_adj0;
        v.upper = d_upper;
        v.upper_off = d_upper_off;
        v.keys = d_keys;
        v.dim = dim;
        v.stride4 = stride4;
        v.lanes = lanes;
        v.lanes_log2 = lanes_log2;
        v.M = M;
        v.M0 = M0;
        v.metric = metric;
        v.scalar = scalar;
        v.entry_slot = entry_slot.load();
        v.max_level = max_level.load();
        v.nt_rows = nt_policy == 1 || (nt_policy == 0 && slots * (size_t)stride4 * 16 >= (2ull << 30)) ? 1u : 0u;
        return v;
    }

    ~Engine() {
        (void)hipSetDevice(device);
        // Dropping an index frees its arenas, and a free synchronises the device: with pods open -- this index's or another's -- each
        // free would wait out a pod's life.  Small indexes (the reference keeps thousands of per-partition handles and drops them with
        // their partitions, usearch.rs:766-778, :881-887) only close their own pods and PARK their blocks; large ones (VMM arenas are
        // unmapped, not freed) take the hold.
        try {
            pod_pool(device).quiesce(this);
        } catch (...) {
        }
        bool any_vmm = false;
        for (Arena* a : {&ar_vectors, &ar_aux, &ar_adj0, &ar_upper, &ar_upper_off, &ar_keys, &ar_levels, &ar_plane, &ar_plane8, &ar_p8scale}) any_vmm |= a->vmm;
        if (any_vmm) {
            try {
                PodHold hold(pod_pool(device));
                (void)hipDeviceSynchronize();
                for (Arena* a : {&ar_vectors, &ar_aux, &ar_adj0, &ar_upper, &ar_upper_off, &ar_keys, &ar_levels, &ar_plane, &ar_plane8, &ar_p8scale}) a->release();
            } catch (...) {
            }
        }
        // (plain blocks: nothing of this index runs any more -- its calls have returned, its pods are gone -- so they can wait)
        for (Arena* a : {&ar_vectors, &ar_aux, &ar_adj0, &ar_upper, &ar_upper_off, &ar_keys, &ar_levels, &ar_plane, &ar_plane8, &ar_p8scale}) {
            if (!a->vmm && a->base) graveyard().bury(a->base, nullptr);
            a->base = nullptr;
            a->bytes = 0;
        }
        graveyard().bury(d_rho, nullptr);
        if (d_rho8) graveyard().bury(d_rho8, nullptr);
        graveyard().bury(d_stats, nullptr);
        graveyard().bury(d_max_norm, nullptr);
        housekeeping();
    }

    void init(const vs_hnsw_options& o) {
        if (!o.dimensions) fail(VS_ERR_INVALID_ARGUMENT, "dimensions must be > 0");
        if (o.quantization < VS_SCALAR_F32 || o.quantization > VS_SCALAR_B1)
            fail(VS_ERR_INVALID_ARGUMENT, "unknown quantization");
        scalar = o.quantization;
        metric = o.metric;
        // reference metric_kind() (usearch.rs:450-487): B1 always means Hamming; Hamming needs B1.
        if (scalar == VS_SCALAR_B1) metric = VS_METRIC_HAMMING;
        else if (metric == VS_METRIC_HAMMING) fail(VS_ERR_INVALID_ARGUMENT, "Binary space type requires B1 quantization.");
        if (metric < VS_METRIC_COS || metric > VS_METRIC_HAMMING) fail(VS_ERR_INVALID_ARGUMENT, "unknown metric");
        dim = (uint32_t)o.dimensions;
        stress_small_table = (o.reserved & 1) != 0;
        exact_valu = (o.reserved & 2) ? 1 : 0;
        team_mode = (o.reserved & 4) ? 1 : (o.reserved & 8) ? 2 : 0;
        order_mode = (o.reserved & 16) ? 1 : 0;
        force_global_walk = (o.reserved & 32) != 0;
        force_wide_tags = (o.reserved & 64) != 0;
        tiny_walk_heap = (o.reserved & 128) != 0;
        no_pipe = (o.reserved & 256) != 0;
        if (const char* ff = std::getenv("VS_HNSW_FILTER")) eager_filter = !std::strcmp(ff, "eager");
        if (const char* xf = std::getenv("VS_HNSW_EXACT")) {
            exact_f32_only = !std::strcmp(xf, "f32");
            exact_mode = exact_f32_only ? 2 : !std::strcmp(xf, "bf16x3") ? 1 : 0;
            plane8_off = !std::strcmp(xf, "bf16");  // the bf16 plane first, as before round 6
        }
        if (const char* tn = std::getenv("VS_HNSW_TIE")) tie_newest = std::strcmp(tn, "random") != 0;
        if (const char* fw = std::getenv("VS_HNSW_WALK")) force_global_walk = !std::strcmp(fw, "global");
        if (const char* ds = std::getenv("VS_HNSW_WALK_DOMAIN_SLOTS")) walk_domain_override = std::strtoull(ds, nullptr, 10);
        if (const char* om = std::getenv("VS_HNSW_ORDER")) order_mode = !std::strcmp(om, "usearch") ? 1 : !std::strcmp(om, "fused") ? 2 : order_mode;
        if (const char* cr = std::getenv("VS_HNSW_CHUNK")) chunk_rows = (uint32_t)std::max(1024, std::atoi(cr));                      // build experiments
        if (const char* sb = std::getenv("VS_HNSW_MAX_SUBBATCH")) max_sub_batch = (uint32_t)std::max(1, std::atoi(sb));
        max_sub_batch = std::min(max_sub_batch, chunk_rows);
        if (const char* t = std::getenv("VS_HNSW_TEAM")) team_mode = !std::strcmp(t, "always") ? 1 : !std::strcmp(t, "never") ? 2 : !std::strcmp(t, "mid") ? 3 : team_mode;
        M = o.connectivity ? (uint32_t)o.connectivity : 16;  // usearch default_connectivity
        if (M < 2 || M > 64) fail(VS_ERR_UNSUPPORTED, "connectivity must be in [2, 64]");  // level-0 rows of up to 128 ids: two per lane
        M0 = 2 * M;
        ef_add = o.expansion_add ? (uint32_t)o.expansion_add : 128;
        ef_search = o.expansion_search ? (uint32_t)o.expansion_search : 64;
        if (ef_add > 512) fail(VS_ERR_UNSUPPORTED, "expansion_add > 512 is not supported");
        inv_log_m = 1.0 / std::log((double)M);
        // Row layout: 16-byte chunks, lanes x iters of them.  Least padding first.  Among equals: the FEWEST lanes
        // that still read >= 128 contiguous bytes of a row per load instruction (lanes >= 8) with iters <= 6 --
        // a wave-load then carries 64/lanes rows, so a hop's ~20 neighbours need fewer load/reduce rounds, and
        // each reduction spans fewer lanes (measured at 1M, ef 128: 128-d 2.26M -> 3.11M QPS, 256-d 1.45M -> 2.20M,
        // 768-d unchanged for search, +6 % for build) -- else the most lanes.
        // 768 f32 = 192 chunks = 32 x 6; 128 f32 = 32 = 8 x 4; 768 f16 = 96 = 16 x 6; 768 i8 = 48 = 8 x 6; 1536 f32 = 64 x 6.
        static const uint32_t bits[] = {32, 16, 16, 8, 1};
        row_bytes = (uint32_t)(((uint64_t)dim * bits[scalar] + 7) / 8);
        const uint32_t chunks = (row_bytes + 15) / 16;
        static const uint32_t ok_iters[] = {1, 2, 3, 4, 6, 8, 12, 16};
        uint32_t best = 0;
        bool best_pref = false;
        iters = 0;
        uint32_t max_lg = 6;
        if (const char* ml = std::getenv("VS_HNSW_MAX_LANES_LOG2")) max_lg = std::min<uint32_t>(6, (uint32_t)std::atoi(ml));  // layout experiments
        // b1 rows of 64+ bytes: at least 4 lanes per row (768 bits: 4 x 2 chunks, rows padded to one 128-byte line, instead of
        // 2 x 3) -- one chunk pair per lane keeps the walk kernel at 114 instead of 191 registers (measured: walk +8 %, fused
        // list unchanged; Hamming distances are integers, so the layout cannot change a result)
        uint32_t min_lg = (scalar == VS_SCALAR_B1 && chunks >= 4) ? 2 : 0;
        if (const char* ml = std::getenv("VS_HNSW_MIN_LANES_LOG2")) min_lg = std::min<uint32_t>(max_lg, (uint32_t)std::atoi(ml));  // layout experiments
        for (uint32_t lg = min_lg; lg <= max_lg; ++lg)
            for (uint32_t it : ok_iters) {
                const uint32_t cap = (1u << lg) * it;
                if (cap < chunks) continue;
                const bool pref = lg >= 3 && it <= 6;  // preferred family: fewest lanes first (lg ascends)
                bool take;
                if (!iters || cap < best) take = true;
                else if (cap > best) take = false;
                else if (best_pref) take = false;               // an earlier (fewer-lane) preferred layout stands
                else take = pref || (1u << lg) > lanes;         // first preferred one, or more lanes among the rest
                if (take) {
                    best = cap;
                    best_pref = pref;
                    lanes = 1u << lg;
                    lanes_log2 = lg;
                    iters = it;
                }
            }
        if (!iters) fail(VS_ERR_UNSUPPORTED, "vectors above 16 KiB per row are not supported");
        stride4 = iters * lanes;
        // link kernel: ~18 KiB of accepted rows per wave in LDS (7 waves per CU; measured at 2M x 768: 3 / 4 / 6 / 8 / 11 rows ->
        // link kernel 0.79 / 0.73 / 0.62 / 0.64 / 0.79 s, 0.98 s without), at most the M0 a list can hold
        link_cache_rows = std::min<uint32_t>(2 * M, std::max<uint32_t>(2, (18u << 10) / (stride4 * 16 + 4) + 1));
        if (const char* nt = std::getenv("VS_HNSW_NT_ROWS")) nt_policy = nt[0] == '1' ? 1 : 2;
        if (const char* lc = std::getenv("VS_HNSW_LINK_CACHE")) link_cache_rows = (uint32_t)std::max(0, std::atoi(lc));  // experiments
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count == 0) fail(VS_ERR_DEVICE, "no HIP device available");
        if (o.device >= 0) {
            if (o.device >= count) fail(VS_ERR_DEVICE, "device ordinal out of range");
            device = o.device;
        } else {
            HIP_OK(hipGetDevice(&device));
        }
        use_device();
        HIP_OK(hipMalloc(&d_stats, sizeof(unsigned long long) * ST_COUNT));
        HIP_OK(hipMemset(d_stats, 0, sizeof(unsigned long long) * ST_COUNT));
    }

    template <class T>
    void regrow(Arena& ar, T*& ptr, size_t old_n, size_t new_n, int fill_byte) {
        ptr = (T*)ar.resize(new_n * sizeof(T), old_n * sizeof(T), device);
        if (fill_byte >= 0 && new_n > old_n) HIP_OK(hipMemset(ptr + old_n, fill_byte, (new_n - old_n) * sizeof(T)));
    }

    // usearch reserve_capacity_and_threads (reference usearch.rs:181-185); exclusive by contract.
    void reserve(size_t cap) {
        std::lock_guard<std::mutex> g(mod_mu);
        use_device();
        if (cap < slots) fail(VS_ERR_INVALID_ARGUMENT, "can't reserve less than the current size");
        if (cap >= (1ull << 30)) fail(VS_ERR_UNSUPPORTED, "capacity must be below 2^30 slots per index");
        // the insert kernel's visited table must tell every slot apart (wide tags: 2^30 at expansion_add <= 128, 2^29 above)
        if (cap > (1ull << visited_domain_bits(ef_add, true)))
            fail(VS_ERR_UNSUPPORTED, "capacity above 2^29 slots needs expansion_add <= 128 (or several shards, include/vs_shards.h)");
        if (cap == capacity) return;
        // (the hold BEFORE the view lock: a caller opening a pod holds the pool's lock while it reads the view)
        PodHold hold(pod_pool(device));  // arenas may move: no pod of the device is open, and none opens, until this returns
        std::unique_lock<std::shared_mutex> vg(view_mu);
        HIP_OK(hipDeviceSynchronize());
        memos_drop();  // (remembered verdicts are laid out for the capacity)
        // HBM budget (the GPU analogue of the reference's host-RAM guard, memory.rs): the new arenas
        // coexist with the old ones while rows are copied across.
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        const size_t extra = ar_vectors.extra_needed(cap * (size_t)stride4 * 16, device) + ar_aux.extra_needed(cap * 4, device) +
                             ar_adj0.extra_needed(cap * M0 * 4, device) + ar_upper_off.extra_needed(cap * 4, device) +
                             ar_keys.extra_needed(cap * 8, device) + ar_levels.extra_needed(cap * 4, device);
        if (cap > capacity && extra > free_b)
            fail(VS_ERR_OUT_OF_MEMORY, "not enough HBM to reserve " + std::to_string(cap) + " vectors");
        regrow(ar_vectors, d_vectors, slots * (size_t)stride4, cap * (size_t)stride4, -1);
        regrow(ar_aux, d_aux, slots, cap, 0);
        regrow(ar_adj0, d_adj0, capacity * M0, cap * M0, 0xFF);
        regrow(ar_upper_off, d_upper_off, capacity, cap, 0xFF);
        regrow(ar_keys, d_keys, capacity, cap, 0xFF);
        regrow(ar_levels, d_levels, capacity, cap, 0);
        h_levels.resize(cap, 0);
        h_upper_off.resize(cap, kInvalid);
        h_keys.resize(cap, kFreeKey);
        capacity = cap;
        capacity_atomic = cap;
        // expected upper-level blocks: cap / (M - 1) (sum over l >= 1 of M^-l); sized with a margin here, where moving
        // an arena is free, so that an ingest almost never has to (ensure_upper still can, under the exclusive lock)
        ensure_upper_locked((size_t)((double)cap / (double)(M - 1) * 1.25) + 1024);
    }

    void ensure_upper_locked(size_t blocks) {  // view_mu held exclusively
        if (blocks <= upper_cap) return;
        PodHold hold(pod_pool(device));
        HIP_OK(hipDeviceSynchronize());
        size_t ncap = std::max(blocks, upper_cap * 2);
        regrow(ar_upper, d_upper, upper_cap * M, ncap * M, 0xFF);
        upper_cap = ncap;
    }
    void ensure_upper(size_t blocks) {
        if (blocks <= upper_cap) return;
        PodHold hold(pod_pool(device));                   // (before the view lock, as in reserve)
        std::unique_lock<std::shared_mutex> vg(view_mu);  // no search may hold the old pointer (advisor finding, round 1)
        ensure_upper_locked(blocks);
    }

    int32_t draw_level() {  // usearch choose_random_level_
        std::uniform_real_distribution<double> distribution(0.0, 1.0);
        double r = -std::log(distribution(level_rng)) * inv_log_m;
        return (int32_t)(int16_t)r;
    }

    // ------------------------------------------------------------------ add
    // Returns per-item status (VS_OK / VS_ERR_*), all items attempted.
    // While an index is modified its pods take no posts; afterwards they get the new entry point / top level / removed flag (PodCtl)
    // and go on -- searches do not overlap modifications (usearch.rs:590-612).  mod_mu is held.
    struct PodFreeze {
        Engine& e;
        explicit PodFreeze(Engine& eng) : e(eng) {
            Stopwatch sw(e.m_quiesce_ns);
            e.m_quiesces.fetch_add(1, std::memory_order_relaxed);
            // Raised BEFORE the pool's lock is taken: a search that overlaps the modification (a caller of the raw C ABI breaking the
            // reference's permits) either opened its pod before freeze() got the lock -- freeze() then sees and freezes it -- or finds the
            // flag under that lock and is served by a launch.  Without it an index that had no pod open could get one opened in the
            // middle of the modification, with a half-applied entry point / top level in its control block (round-5 advisor).
            e.being_modified.fetch_add(1, std::memory_order_seq_cst);
            try {
                pod_pool(e.device).freeze(&e);
            } catch (...) {
                e.being_modified.fetch_sub(1, std::memory_order_seq_cst);
                throw;
            }
        }
        ~PodFreeze() {
            pod_pool(e.device).thaw(&e, e.capacity, e.entry_slot.load(), e.max_level.load(), e.removed.load() ? 1u : 0u);
            e.being_modified.fetch_sub(1, std::memory_order_seq_cst);
        }
    };
    std::atomic<int> being_modified{0};

    void add_batch(const uint64_t* keys, const float* vecs, bool on_device, size_t n, std::vector<int>& status,
                   std::string& first_err, bool staged = false) {
        status.assign(n, VS_OK);
        if (!n) return;
        std::lock_guard<std::mutex> g(mod_mu);
        use_device();
        housekeeping();
        PodFreeze freeze(*this);
        add_batch_locked(keys, vecs, on_device, n, status, first_err, staged);
    }

    // Nodes whose slot and level are assigned (the host maps already know them) and whose device pass is still to come.  A slot
    // appears once per run: the device pass stages every node's vector and zeroes the links of re-used slots before the first insert
    // walk, so the same slot twice (removed, re-added, removed and re-added again within one log) needs two passes.
    struct Run {
        std::vector<uint32_t> slot_v, src_row, reuse_rows, reuse_upper;
        std::vector<int32_t> level_v;
        std::vector<uint64_t> key_v;
        std::unordered_set<uint32_t> slots;
        size_t size() const { return slot_v.size(); }
        void clear() {
            slot_v.clear();
            src_row.clear();
            reuse_rows.clear();
            reuse_upper.clear();
            level_v.clear();
            key_v.clear();
            slots.clear();
        }
    };
    static constexpr int kRunFull = 1;  // assign_one: the slot this add would re-use is already in the run -- insert the run first
    // usearch add(): validate the key, take a slot (a removed node's, FIFO, else the next unused one) and a level; the host maps are
    // updated at once (a later remove in the same log finds the key).  VS_OK / VS_ERR_* / kRunFull.  mod_mu held.
    int assign_one(uint64_t key, uint32_t src_index, Run& run, const char** why) {
        if (key == kFreeKey) {
            *why = "Key is reserved for internal use";
            return VS_ERR_INVALID_ARGUMENT;
        }
        {
            std::lock_guard<std::mutex> kg(key_mu);
            if (lookup.count(key)) {
                *why = "Duplicate keys not allowed in high-level wrappers";
                return VS_ERR_DUPLICATE_KEY;
            }
        }
        uint32_t slot;
        int32_t level;
        if (!free_slots.empty()) {  // usearch index_dense: reuse a removed node in place (update path)
            if (run.slots.count(free_slots.front())) return kRunFull;
            slot = free_slots.front();
            free_slots.pop_front();
            --removed;
            level = h_levels[slot];
            run.reuse_rows.push_back(slot);
            for (int l = 0; l < level; ++l) run.reuse_upper.push_back(h_upper_off[slot] + l);
        } else {
            if (slots >= capacity) {
                *why = "Reserve capacity ahead of insertions!";
                return VS_ERR_CAPACITY;
            }
            slot = (uint32_t)slots++;
            level = draw_level();
            h_levels[slot] = (uint8_t)std::min(level, 255);
            if (level > 0) {
                h_upper_off[slot] = (uint32_t)upper_blocks;
                upper_blocks += (size_t)level;
            }
        }
        run.slot_v.push_back(slot);
        run.level_v.push_back(level);
        run.key_v.push_back(key);
        run.src_row.push_back(src_index);
        run.slots.insert(slot);
        {
            std::lock_guard<std::mutex> kg(key_mu);
            lookup.emplace(key, slot);
        }
        h_keys[slot] = key;
        ++live;
        return VS_OK;
    }

    void add_batch_locked(const uint64_t* keys, const float* vecs, bool on_device, size_t n, std::vector<int>& status,
                          std::string& first_err, bool staged) {  // mod_mu held, this index's pods frozen
        status.assign(n, VS_OK);
        if (!n) return;
        Lease w(device);
        Run run;
        for (size_t c0 = 0; c0 < n; c0 += chunk_rows) {
            const size_t cn = std::min<size_t>(chunk_rows, n - c0);
            run.clear();
            for (size_t i = 0; i < cn; ++i) {
                const char* why = "";
                const int rc = assign_one(keys[c0 + i], (uint32_t)i, run, &why);  // (never kRunFull: nothing is removed in between)
                if (rc != VS_OK) {
                    status[c0 + i] = rc == kRunFull ? VS_ERR_DEVICE : rc;
                    if (first_err.empty()) first_err = why;
                }
            }
            insert_run(run, vecs + c0 * dim, on_device, *w.ctx);
            if (!staged) committed += run.size();  // staged vectors were counted when add_one accepted them
        }
    }

    // The device pass of a run: vectors into their (padded, cast) rows, keys / levels / upper offsets, links of re-used slots zeroed,
    // then sub-batches of insert walks against the frozen graph and their links.  `vecs`: rows indexed by the run's src_row.
    void insert_run(Run& run, const float* vecs, bool on_device, WorkCtx& wc) {
        {
            WorkCtx* w = &wc;
            hipStream_t st = w->stream;
            std::vector<uint32_t>&slot_v = run.slot_v, &src_row = run.src_row, &reuse_rows = run.reuse_rows, &reuse_upper = run.reuse_upper;
            std::vector<int32_t>& level_v = run.level_v;
            std::vector<uint64_t>& key_v = run.key_v;
            const size_t c0 = 0;
            bool contiguous = true;  // the run's vectors are consecutive rows of `vecs`: one copy
            for (size_t i = 1; i < src_row.size() && contiguous; ++i) contiguous = src_row[i] == src_row[0] + i;
            const size_t cn = contiguous ? slot_v.size() : (size_t)-1;
            const float* vecs0 = vecs + (src_row.empty() ? 0 : (size_t)src_row[0] * dim);
            const uint32_t m = (uint32_t)slot_v.size();
            if (!m) return;
            ensure_upper(upper_blocks);
            IndexView ix = view();

            // 2. stage: slots / levels / keys / upper offsets, vectors into padded rows, norms
            std::vector<uint32_t> uoff(m);
            for (uint32_t i = 0; i < m; ++i) uoff[i] = h_upper_off[slot_v[i]];
            // request offsets: (min(level, max_level at its turn) + 1) * M per node
            std::vector<uint32_t> req_off(m + 1, 0);
            {
                int32_t ml = max_level.load();
                for (uint32_t i = 0; i < m; ++i) {
                    uint32_t cnt = 0;
                    if (ml < 0) {
                        ml = level_v[i];  // becomes the entry point, no requests
                    } else {
                        cnt = (uint32_t)(std::min(level_v[i], ml) + 1) * M;
                        if (level_v[i] > ml) ml = level_v[i];
                    }
                    req_off[i + 1] = req_off[i] + cnt;
                }
            }
            // A handful of vectors from the host (a flush between two families of searches: the reference alternates them,
            // usearch.rs:590-612, so a mixed workload flushes ONE add at a time): everything the kernels need goes through one
            // pinned block and ONE copy -- eight small copies from pageable memory were a tenth of such a flush.
            const bool small = !on_device && m <= 256;
            uint32_t *d_slots, *d_uoff, *d_reqoff, *d_rr = nullptr, *d_ru = nullptr;
            int32_t* d_lv;
            uint64_t* d_keyv;
            const float* src = nullptr;
            if (small) {
                const size_t o_keys = ((size_t)m * 16 + 7) & ~(size_t)7, o_vec = (o_keys + (size_t)m * 8 + 15) & ~(size_t)15;
                const size_t o_rr = o_vec + (size_t)m * dim * 4, o_ru = o_rr + reuse_rows.size() * 4;
                const size_t total = o_ru + reuse_upper.size() * 4 + 64;
                if (w->pin_bytes < total) {
                    if (w->pin) graveyard().bury(nullptr, w->pin);
                    w->pin = nullptr;
                    w->pin_bytes = 0;
                    HIP_OK(hipHostMalloc((void**)&w->pin, total + total / 2, hipHostMallocDefault));
                    w->pin_bytes = total + total / 2;
                }
                char* hp = w->pin;
                std::memcpy(hp, slot_v.data(), (size_t)m * 4);
                std::memcpy(hp + (size_t)m * 4, level_v.data(), (size_t)m * 4);
                std::memcpy(hp + (size_t)m * 8, uoff.data(), (size_t)m * 4);
                std::memcpy(hp + (size_t)m * 12, req_off.data(), (size_t)m * 4);
                std::memcpy(hp + o_keys, key_v.data(), (size_t)m * 8);
                for (uint32_t i = 0; i < m; ++i) std::memcpy(hp + o_vec + (size_t)i * dim * 4, vecs + (c0 + src_row[i]) * dim, (size_t)dim * 4);
                if (!reuse_rows.empty()) std::memcpy(hp + o_rr, reuse_rows.data(), reuse_rows.size() * 4);
                if (!reuse_upper.empty()) std::memcpy(hp + o_ru, reuse_upper.data(), reuse_upper.size() * 4);
                char* dp = (char*)w->a.ensure(total);
                HIP_OK(hipMemcpyAsync(dp, hp, total - 64, hipMemcpyHostToDevice, st));
                d_slots = (uint32_t*)dp;
                d_lv = (int32_t*)(dp + (size_t)m * 4);
                d_uoff = (uint32_t*)(dp + (size_t)m * 8);
                d_reqoff = (uint32_t*)(dp + (size_t)m * 12);
                d_keyv = (uint64_t*)(dp + o_keys);
                src = (const float*)(dp + o_vec);
                d_rr = (uint32_t*)(dp + o_rr);
                d_ru = (uint32_t*)(dp + o_ru);
            } else {
                d_slots = (uint32_t*)w->a.ensure((size_t)m * 4 * 4 + 64);
                d_lv = (int32_t*)(d_slots + m);
                d_uoff = (uint32_t*)(d_lv + m);
                d_reqoff = d_uoff + m;
                d_keyv = (uint64_t*)w->b.ensure((size_t)m * 8);
                HIP_OK(hipMemcpyAsync(d_slots, slot_v.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
                HIP_OK(hipMemcpyAsync(d_lv, level_v.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
                HIP_OK(hipMemcpyAsync(d_uoff, uoff.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
                HIP_OK(hipMemcpyAsync(d_reqoff, req_off.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
                HIP_OK(hipMemcpyAsync(d_keyv, key_v.data(), (size_t)m * 8, hipMemcpyHostToDevice, st));
                if (on_device) {
                    if (m == cn) {
                        src = vecs0;
                    } else {  // some rows were rejected: compact on the host side list of source rows
                        float* stg = (float*)w->c.ensure((size_t)m * dim * 4);
                        for (uint32_t i = 0; i < m; ++i)
                            HIP_OK(hipMemcpyAsync(stg + (size_t)i * dim, vecs + (c0 + src_row[i]) * dim, (size_t)dim * 4,
                                                  hipMemcpyDeviceToDevice, st));
                        src = stg;
                    }
                } else {
                    float* stg = (float*)w->c.ensure((size_t)m * dim * 4);
                    if (m == cn) {
                        HIP_OK(hipMemcpyAsync(stg, vecs0, (size_t)m * dim * 4, hipMemcpyHostToDevice, st));
                    } else {
                        for (uint32_t i = 0; i < m; ++i)
                            HIP_OK(hipMemcpyAsync(stg + (size_t)i * dim, vecs + (c0 + src_row[i]) * dim, (size_t)dim * 4,
                                                  hipMemcpyHostToDevice, st));
                    }
                    src = stg;
                }
            }
            const uint32_t* d_src_slots = d_slots;
            HIP_OK(launch_quantise_rows(ix, d_vectors, d_aux, src, dim, d_src_slots, 0, m, st));
            HIP_OK(launch_scatter_u64(d_keys, d_slots, d_keyv, m, st));
            HIP_OK(launch_scatter_u32((uint32_t*)d_levels, d_slots, (const uint32_t*)d_lv, m, st));
            HIP_OK(launch_scatter_u32(d_upper_off, d_slots, d_uoff, m, st));
            if (!reuse_rows.empty()) {  // a reused slot lies below max_norm_slots: the certificate's norm bound is recomputed over every row
                {
                    std::lock_guard<std::mutex> ng(norm_mu);
                    max_norm_slots = 0;
                }
                std::lock_guard<std::mutex> pg(plane_mu);  // ... and below plane_done: the bf16 plane is converted again
                plane_done = 0;
                plane8_done = 0;
            }
            if (!reuse_rows.empty()) {  // usearch update(): the reused node's links are zeroed first
                if (!small) {
                    d_rr = (uint32_t*)w->d.ensure((reuse_rows.size() + reuse_upper.size()) * 4 + 64);
                    HIP_OK(hipMemcpyAsync(d_rr, reuse_rows.data(), reuse_rows.size() * 4, hipMemcpyHostToDevice, st));
                    d_ru = d_rr + reuse_rows.size();
                    if (!reuse_upper.empty()) HIP_OK(hipMemcpyAsync(d_ru, reuse_upper.data(), reuse_upper.size() * 4, hipMemcpyHostToDevice, st));
                }
                HIP_OK(launch_fill_rows_u32(d_adj0, M0, d_rr, (uint32_t)reuse_rows.size(), kInvalid, st));
                if (!reuse_upper.empty()) HIP_OK(launch_fill_rows_u32(d_upper, M, d_ru, (uint32_t)reuse_upper.size(), kInvalid, st));
            }

            // 3. sub-batches against the frozen graph
            size_t max_req = 0;  // most requests any run of <= max_sub_batch consecutive nodes can emit (req_off = prefix sums)
            for (uint32_t i = 0; i < m; ++i)
                max_req = std::max<size_t>(max_req, req_off[std::min<uint32_t>(i + max_sub_batch, m)] - req_off[i]);
            uint64_t* rk_in = (uint64_t*)w->e.ensure(std::max<size_t>(max_req, 1) * 8 * 4);
            uint64_t* rv_in = rk_in + max_req;
            uint64_t* rk_out = rv_in + max_req;
            uint64_t* rv_out = rk_out + max_req;
            size_t temp_bytes = sort_temp_bytes(std::max<size_t>(max_req, 1));
            void* temp = w->f.ensure(temp_bytes);

            uint32_t pos = 0;
            while (pos < m) {
                if (max_level.load() < 0) {  // first member: entry point, no links (usearch add(), `!new_slot`)
                    entry_slot = slot_v[pos];
                    max_level = level_v[pos];
                    ++pos;
                    ++linked;
                    continue;
                }
                const int32_t ml = max_level.load();
                uint32_t limit = (uint32_t)std::min<size_t>(max_sub_batch, std::max<size_t>(1, linked / kSubBatchRatio));
                uint32_t take = 0;
                while (take < limit && pos + take < m && level_v[pos + take] <= ml) ++take;
                bool promote = false;
                if (take == 0) {  // a node above the current top level is inserted alone, then becomes the entry
                    take = 1;
                    promote = true;
                }
                const uint32_t total_req = req_off[pos + take] - req_off[pos];
                InsertArgs ia;
                ia.ix = view();
                ia.slots = d_slots + pos;
                ia.levels = d_lv + pos;
                ia.req_off = d_reqoff + pos;
                ia.n = take;
                ia.ef_add = ef_add;
                ia.team = (team_mode == 1 || (team_mode == 0 && take <= team_max_nq)) ? (uint32_t)kSearchTeam : 1u;
                ia.tie_newest = tie_newest ? 1u : 0u;
                ia.wide_tags = (slots > (1ull << visited_domain_bits(ef_add)) || force_wide_tags) ? 1u : 0u;
                ia.req_base = req_off[pos];
                ia.req_key = rk_in;
                ia.req_val = rv_in;
                ia.stats = d_stats;
                HIP_OK(launch_insert(ia, iters, st));
                if (total_req) {
                    HIP_OK(sort_pairs(temp, temp_bytes, rk_in, rk_out, rv_in, rv_out, total_req, 40, st));
                    LinkArgs la;
                    la.ix = ia.ix;
                    la.req_key = rk_out;
                    la.req_val = rv_out;
                    la.total = total_req;
                    la.cache_rows = link_cache_rows;
                    la.tie_newest = tie_newest ? 1u : 0u;
                    la.stats = d_stats;
                    HIP_OK(launch_link(la, iters, st));
                }
                if (promote) {
                    entry_slot = slot_v[pos];
                    max_level = level_v[pos];
                }
                pos += take;
                linked += take;
            }
            HIP_OK(hipStreamSynchronize(st));
            (void)key_v;
            slots_atomic.store(slots, std::memory_order_release);
        }
    }

    // Single-vector callers (one add / remove per FFI call from <= num_workers()+1 threads, reference worker.rs:44-118).
    // An insert is one graph walk (~1 ms of latency for a lone team of waves), so inserting per call would cap the build
    // at the walk latency.  The reference treats adds and removes as fire-and-forget messages (usearch.rs:1028-1049), which
    // allows the same "as-if" here: add_one / remove_one validate synchronously against the state every EARLIER call left
    // (reserved / duplicate / unknown key, capacity), log the operation and return; the log is applied IN ORDER and in bulk
    // when kFlushThreshold operations are waiting and, at the latest, before any other call observes the index (search,
    // reserve, size, stats, export), so every call still sees the effect of everything that returned before it.
    // Round 5: removes are logged too.  Until then each remove flushed the adds before it, so the reference's update --
    // RemoveBeforeAddValue then AddVector, monitor_items.rs:301-313 -- paid one lone insert walk per item (1.1k updates/s
    // through the actor at 10M x 768, scripts/mixed_probe.py; the bulk insert path does hundreds of thousands).
    static constexpr size_t kFlushThreshold = 4096;

    // pend_mu held.  Is `key` a member once everything logged so far has been applied?
    bool key_present_locked(uint64_t key) {
        auto it = pend.last.find(key);
        if (it != pend.last.end()) return it->second == 1;
        auto jt = flushing_last.find(key);
        if (jt != flushing_last.end()) return jt->second == 1;
        std::lock_guard<std::mutex> kg(key_mu);
        return lookup.count(key) != 0;
    }
    bool poisoned() {
        std::lock_guard<std::mutex> pg(poison_mu);
        if (poison.empty()) return false;
        g_err = poison;
        return true;
    }

    int add_one(uint64_t key, const float* v) {
        if (poisoned()) return VS_ERR_DEVICE;
        std::unique_lock<std::mutex> lk(pend_mu);
        if (key == kFreeKey) {
            g_err = error_text(VS_ERR_INVALID_ARGUMENT);
            return VS_ERR_INVALID_ARGUMENT;
        }
        if (key_present_locked(key)) {
            g_err = error_text(VS_ERR_DUPLICATE_KEY);
            return VS_ERR_DUPLICATE_KEY;
        }
        if (committed.load() >= capacity_atomic.load()) {
            g_err = error_text(VS_ERR_CAPACITY);
            return VS_ERR_CAPACITY;
        }
        pend.ops.push_back({key, (uint32_t)pend.adds});
        pend.vecs.insert(pend.vecs.end(), v, v + dim);
        ++pend.adds;
        pend.last[key] = 1;
        ++queued;
        ++committed;
        if (pend.ops.size() < kFlushThreshold) return VS_OK;
        return flush_locked(lk, false);
    }

    // usearch index_dense::remove: true when the key was a member.
    int remove_one(uint64_t key, bool* was_member) {
        Stopwatch sw(m_remove_ns);
        m_removes.fetch_add(1, std::memory_order_relaxed);
        *was_member = false;
        if (poisoned()) return VS_ERR_DEVICE;
        std::unique_lock<std::mutex> lk(pend_mu);
        if (key == kFreeKey || !key_present_locked(key)) return VS_OK;
        pend.ops.push_back({key, Pending::kRemove});
        pend.last[key] = 2;
        ++queued;
        --committed;
        *was_member = true;
        if (pend.ops.size() < kFlushThreshold) return VS_OK;
        return flush_locked(lk, false);
    }

    // Applies everything logged so far, in order.  Called with pend_mu held; releases it while the GPU works so that
    // other callers keep logging.  One flush at a time (logs are applied in the order they were taken): a caller that
    // finds one running waits for it when it needs the barrier (`barrier`), and otherwise leaves its operations staged.
    int flush_locked(std::unique_lock<std::mutex>& lk, bool barrier) {
        while (flushing) {
            if (!barrier) return VS_OK;
            flush_cv.wait(lk);
        }
        if (pend.ops.empty()) return VS_OK;
        Stopwatch sw(m_flush_ns);
        m_flushes.fetch_add(1, std::memory_order_relaxed);
        m_flushed.fetch_add(pend.adds, std::memory_order_relaxed);
        Pending take;
        std::swap(take, pend);
        flushing_last.swap(take.last);
        flushing = true;
        lk.unlock();
        int rc = VS_OK;
        size_t lost = 0;
        try {
            apply_log(take, lost);
        } catch (const Fail& f) {
            rc = f.code;
            g_err = f.msg;
        } catch (const std::exception& e) {
            rc = VS_ERR_DEVICE;
            g_err = e.what();
        }
        if (rc != VS_OK) {
            // Staged vectors are lost although their vs_hnsw_add calls returned VS_OK, and the log may have stopped between the
            // host bookkeeping and the GPU work: the index is not trustworthy any more.  It says so to every later call (a
            // device failure is not recoverable in this process anyway) instead of reporting a false "Reserve capacity" to an
            // unrelated caller (advisor finding, round 1).
            std::lock_guard<std::mutex> pg(poison_mu);
            if (poison.empty()) poison = "index unusable after a failed flush of " + std::to_string(take.ops.size()) + " staged operations: " + g_err;
        }
        lk.lock();
        flushing = false;
        flushing_last.clear();
        queued -= take.ops.size();
        flush_cv.notify_all();
        return rc;
    }

    // The staged log, in order.  HOST state follows the calls one by one: an add takes the slot a remove EARLIER in the log freed
    // (usearch's free ring is FIFO), a remove finds the key an earlier add of the same log brought.  The DEVICE work is batched across
    // the removes: assigned adds collect in a run that goes through the bulk insert in one pass (interleaved with removes, as the
    // reference's update is -- RemoveBeforeAddValue + AddVector per item -- runs of one add each made a log of 2,048 updates 2,048
    // lone insert walks: 1.3k updates/s), cut only where an add would re-use a slot that is already in the run; removes are
    // tombstoned on the device in one launch at the end (a slot that was removed and re-added in this log carries its new key by then).
    void apply_log(Pending& log, size_t& lost) {
        std::lock_guard<std::mutex> g(mod_mu);
        use_device();
        housekeeping();
        PodFreeze freeze(*this);
        Lease w(device);
        std::vector<uint32_t> tomb;
        Run run;
        const float* vecs = log.vecs.data();
        for (const Pending::Op& op : log.ops) {
            if (op.vec == Pending::kRemove) {
                uint32_t slot = kInvalid;
                {
                    std::lock_guard<std::mutex> kg(key_mu);
                    auto it = lookup.find(op.key);
                    if (it != lookup.end()) {
                        slot = it->second;
                        lookup.erase(it);
                    }
                }
                if (slot != kInvalid) {  // (always: the call was validated against the state this log produces)
                    h_keys[slot] = kFreeKey;
                    free_slots.push_back(slot);
                    ++removed;
                    --live;
                    tomb.push_back(slot);
                } else {
                    ++committed;  // the remove was counted when it was staged
                }
                continue;
            }
            const char* why = "";
            int rc = assign_one(op.key, op.vec, run, &why);
            if (rc == kRunFull) {
                insert_run(run, vecs, false, *w.ctx);
                run.clear();
                rc = assign_one(op.key, op.vec, run, &why);
            }
            if (rc != VS_OK) {  // accepted at staging time, rejected at insertion (cannot normally happen)
                --committed;
                ++lost;
            }
            if (run.size() >= chunk_rows) {
                insert_run(run, vecs, false, *w.ctx);
                run.clear();
            }
        }
        insert_run(run, vecs, false, *w.ctx);
        if (!tomb.empty()) {
            memos_forget(tomb, *w.ctx);  // (every removed slot, re-used or not: its member changed)
            size_t m = 0;
            for (uint32_t s : tomb)
                if (h_keys[s] == kFreeKey) tomb[m++] = s;  // not re-added later in this log
            if (m) {
                uint32_t* d_t = (uint32_t*)w->d.ensure(m * 4 + 64);
                HIP_OK(hipMemcpyAsync(d_t, tomb.data(), m * 4, hipMemcpyHostToDevice, w->stream));
                HIP_OK(launch_fill_rows_u32((uint32_t*)d_keys, 2, d_t, (uint32_t)m, kInvalid, w->stream));  // key := ~0 (usearch free_key)
                HIP_OK(hipStreamSynchronize(w->stream));
            }
        }
    }

    // Barrier used by every operation that observes the index.
    void flush_pending() {
        if (queued.load(std::memory_order_acquire) == 0) {  // nothing staged, nothing being applied
            std::lock_guard<std::mutex> pg(poison_mu);
            if (!poison.empty()) fail(VS_ERR_DEVICE, poison);
            return;
        }
        {
            std::lock_guard<std::mutex> pg(poison_mu);
            if (!poison.empty()) fail(VS_ERR_DEVICE, poison);
        }
        {
            std::unique_lock<std::mutex> lk(pend_mu);
            int rc = flush_locked(lk, true);
            if (rc != VS_OK) fail(rc, g_err);
        }
        std::lock_guard<std::mutex> g(mod_mu);  // a bulk add started by another caller has finished too
    }

    static const char* error_text(int code) {
        switch (code) {
            case VS_ERR_DUPLICATE_KEY: return "Duplicate keys not allowed in high-level wrappers";
            case VS_ERR_CAPACITY: return "Reserve capacity ahead of insertions!";
            case VS_ERR_INVALID_ARGUMENT: return "Key is reserved for internal use";
            default: return "add failed";
        }
    }

    // ------------------------------------------------------------------ search
    // Which kernel serves a search (usearch: expansion = max(expansion_search, wanted)):
    //   fused-list kernel (hnsw_search_kernel)   float metrics, beam <= 512, index within the LDS tags' reach;
    //   usearch-order walk, LDS visited table     i8 / b1 (ties are the rule there) or order_mode 1, beam <= 512;
    //   usearch-order walk, global visited bitmap filtered search, beams 513..10,240, indexes beyond the LDS tags.
    // lone queries on float indexes: the pipelined walk (kernels_pipe.hip) serves them
    bool pipe_usable(uint32_t ef) const {
        static const bool pipe_off = std::getenv("VS_HNSW_PIPE") && std::getenv("VS_HNSW_PIPE")[0] == '0';  // A/B measurements
        IndexView v{};
        v.scalar = scalar;
        v.M0 = (uint32_t)M0;
        return !pipe_off && !no_pipe && !tl_no_pipe && !tiny_walk_heap && iters < 12 && pipe_walk_supported(v, iters, ef);
    }
    bool needs_global_walk_beyond_pipe(uint32_t ef) const { return ef > 512; }
    bool usearch_order() const { return order_mode == 1 || (order_mode == 0 && (scalar == VS_SCALAR_I8 || scalar == VS_SCALAR_B1)); }
    void check_search(size_t k, uint32_t& ef) const {
        if (k == 0) fail(VS_ERR_INVALID_ARGUMENT, "k must be > 0");
        ef = (uint32_t)std::min<size_t>(std::max<size_t>(ef_search.load(), k), 0x7FFFFFFFu);
        if (ef > kMaxWalkBeam)
            fail(VS_ERR_UNSUPPORTED, "k / expansion_search above 10240 needs the exhaustive path (vs_hnsw_search)");
    }
    bool needs_global_walk(uint32_t ef) const {
        return ef > kMaxBeam || force_global_walk || beyond_lds_tags(ef, true) || (beyond_lds_tags(ef) && usearch_order());
    }
    bool beyond_lds_tags(uint32_t ef, bool wide = false) const { return slots > (1ull << visited_domain_bits(std::min<uint32_t>(ef, kMaxBeam), wide)); }

    // `load`: queries that will be on the device together with this batch (other pipeline slots included);
    // the team kernel only pays while the chip has idle CUs.
    // allow: device bitmap(s) over slots for filtered search (allow_stride words apart, 0 = one for the whole batch).
    struct LazyFilter {  // device side of a lazily evaluated predicate (WalkArgs::known ...)
        const uint32_t* known = nullptr;
        uint32_t* unknown_list = nullptr;
        uint32_t* unknown_count = nullptr;
        uint32_t cap = 0, budget = 0;
        uint32_t* consulted = nullptr;
        bool explore = false;  // an exploring round of the pipelined walk (kernels_pipe.hip): lists missing verdicts, answers nothing
    };
    void search_device(const float* d_q, size_t nq, size_t k, uint64_t* d_keys_out, float* d_dist_out, uint32_t* d_found,
                       hipStream_t st, size_t load = 0, const uint32_t* allow = nullptr, uint32_t allow_stride = 0,
                       const LazyFilter* lazy = nullptr) {
        uint32_t ef;
        check_search(k, ef);
        if (!nq) return;
        std::shared_lock<std::shared_mutex> vg(view_mu);
        // plain tags -> wide tags (fused kernel only) -> global bitmap, as the index outgrows what each can tell apart
        const bool global = allow || needs_global_walk(ef);
        // Lone queries on float indexes (the reference issues one query per FFI call, usearch.rs:212; the dispatcher hands over what
        // queued since the last launch): the pipelined walk (kernels_pipe.hip) instead of the team form of the fused-list kernel --
        // the walker decides on registers while the other waves measure candidates ahead of it.
        // (up to 48 queries on the device: beyond that the team form of the fused-list kernel has the better throughput -- measured at
        // 10M x 768 on one box, blocking callers 17 / 33 / 65: 13.7k / 26.5k / 47.3k QPS pipelined against 12.2k / 24.5k / 48.6k, and 16 x 16
        // queries in flight 115k against 167k)
        constexpr size_t kLonePipeMaxLoad = 48;
        static const bool lone_pipe_off = std::getenv("VS_HNSW_PIPE_LONE") && std::getenv("VS_HNSW_PIPE_LONE")[0] == '0';  // A/B measurements
        const bool lone_pipe = !allow && !global && !usearch_order() && !lone_pipe_off && pipe_usable(ef) && team_mode != 2 && team_mode != 3 &&
                               std::max(nq, load) <= kLonePipeMaxLoad && !stress_small_table && !force_wide_tags;
        if (global || usearch_order() || lone_pipe) {
            WalkArgs a;
            a.ix = view();
            a.queries = d_q;
            a.q_stride = dim;
            a.nq = (uint32_t)nq;
            a.k = (uint32_t)k;
            a.ef = ef;
            a.has_removed = removed.load() ? 1u : 0u;
            a.allow = allow;
            a.allow_stride = allow_stride;
            a.known = lazy ? lazy->known : nullptr;
            a.unknown_list = lazy ? lazy->unknown_list : nullptr;
            a.unknown_count = lazy ? lazy->unknown_count : nullptr;
            a.unknown_cap = lazy ? lazy->cap : 0u;
            a.unknown_budget = lazy ? lazy->budget : 0u;
            a.consulted = lazy ? lazy->consulted : nullptr;
            a.pipe_explore = (lazy && lazy->explore) ? 1u : 0u;
            a.qlist = nullptr;
            a.qcount = nullptr;
            a.retry_list = nullptr;
            a.retry_count = nullptr;
            a.out_keys = d_keys_out;
            a.out_dist = d_dist_out;
            a.out_found = d_found;
            a.stats = d_stats;
            a.debug = nullptr;
            WalkRes& wr = walk_res(device, st);
            std::lock_guard<std::mutex> wl(wr.mu);
            const uint32_t g_inst = ef <= 512 ? WALK_GLOBAL_512 : ef <= 2048 ? WALK_GLOBAL_2048 : WALK_GLOBAL_10240;
            // workspace of a global-bitmap launch of `grid` workgroups; the bitmaps must be zero on entry (the kernel
            // leaves them zero), so a change of layout re-zeroes the block
            auto global_space = [&](WalkArgs& w, uint32_t inst, uint32_t want_grid, bool deep_heap) {
                w.bitmap_words = (uint32_t)((slots + 31) / 32);
                w.vlog_cap = (uint32_t)std::max<size_t>(64, std::min<size_t>(slots, 1u << 16));
                w.heap_cap = tiny_walk_heap ? 64u : (uint32_t)std::max<size_t>(64, std::min<size_t>(slots, deep_heap ? (1u << 18) : (1u << 16)));
                w.space_stride = walk_space_stride(w.bitmap_words, w.vlog_cap, w.heap_cap);
                const uint32_t budget_grid = (uint32_t)std::max<size_t>(16, (4ull << 30) / w.space_stride);
                uint32_t grid = 0;
                HIP_OK(launch_walk(w, iters, inst, std::min(want_grid, budget_grid), st, &grid));
                const size_t bytes = (size_t)grid * w.space_stride;
                const bool grown = bytes > wr.g_space.bytes;
                w.space = (char*)wr.g_space.ensure(bytes);
                if (grown || wr.g_layout[0] != w.bitmap_words || wr.g_layout[1] != w.space_stride || wr.g_layout[2] < bytes) {
                    HIP_OK(hipMemsetAsync(w.space, 0, bytes, st));
                    wr.g_layout[0] = w.bitmap_words;
                    wr.g_layout[1] = w.space_stride;
                    wr.g_layout[2] = bytes;
                }
                return grid;
            };
            uint32_t* retry = (uint32_t*)wr.retry.ensure((64 + nq) * 4);  // [0] retried queries, [1] / [2] work counters, [64..] their ids
            HIP_OK(hipMemsetAsync(retry, 0, 256, st));
            // How the previous launch on this stream fared: the half-size table goes back to the full one when > 5 % of a
            // batch outgrew it; an instance most of whose queries outgrow its LDS structures (structureless data: `next`
            // holds thousands of equal-distance entries) is skipped from then on -- the retry launch is exact but narrow.
            if (wr.retry_seen_valid && wr.retry_seen_owner == this && hipEventQuery(wr.retry_seen_ev) == hipSuccess) {
                const uint32_t seen = wr.retry_seen[0], of = wr.retry_seen[1], which = wr.retry_seen[2] & 15u;
                if (of >= 64 && which == WALK_LDS_128_SMALL && seen * 20 > of) small_table_ok = false;
                else if (of >= 64 && which == WALK_LDS_256_DENSE && seen * 20 > of) dense_table_ok = false;  // > 5 % of a batch outgrew it
                else if (of >= 64 && which != WALK_LDS_128_SMALL && which != WALK_LDS_128_TINY && seen * 4 > of) lds_walk_bad[which] = true;
            }
            wr.retry_seen_valid = false;
            const bool small = ef <= 128 && small_table_ok.load() && slots <= (1ull << walk_small_table_bits()) && !stress_small_table;
            static const bool no_team_global = std::getenv("VS_HNSW_TEAM_GLOBAL") && std::getenv("VS_HNSW_TEAM_GLOBAL")[0] == '0';  // A/B measurements
            static const bool dense_off = std::getenv("VS_HNSW_WALK_DENSE") && std::getenv("VS_HNSW_WALK_DENSE")[0] == '0';  // A/B measurements
            const bool dense = ef > 128 && ef <= 256 && !dense_off && dense_table_ok.load() && std::max<uint64_t>(slots, walk_domain_override) <= (1ull << walk_instance_domain_bits(WALK_LDS_256_DENSE));
            uint32_t inst = (stress_small_table && iters == 1 && ef <= 128) ? WALK_LDS_128_TINY : small ? WALK_LDS_128_SMALL : ef <= 128 ? WALK_LDS_128 : dense ? WALK_LDS_256_DENSE : ef <= 256 ? WALK_LDS_256 : ef <= kWalk320MaxBeam ? WALK_LDS_320 : WALK_LDS_512;
            // The visited tags of the instance actually chosen must tell every slot apart (needs_global_walk asks by beam, and
            // the 320-entry instance carries the 256 instance's 25-bit table, not the 512 instance's 26 bits): a wider
            // instance first, the global bitmap beyond that.  walk_domain_override: test hook (pretends the index is larger).
            const uint64_t domain_slots = std::max<uint64_t>(slots, walk_domain_override);
            if (inst == WALK_LDS_320 && domain_slots > (1ull << walk_instance_domain_bits(inst))) inst = WALK_LDS_512;
            const bool out_of_domain = inst != WALK_LDS_128_TINY && domain_slots > (1ull << walk_instance_domain_bits(inst));
            last_walk_instance = (global || lone_pipe || out_of_domain || lds_walk_bad[inst].load()) ? g_inst : inst;
            if (global || lone_pipe || out_of_domain || lds_walk_bad[inst].load()) {
                // a lone query (every filtered one is: the predicate is the caller's) takes a team of waves here too -- one
                // workgroup, one workspace per query
                const bool team_g = g_inst == WALK_GLOBAL_512 && iters < 12 && nq <= (lone_pipe ? team_max_nq : (size_t)32) && !no_team_global &&
                                    (team_mode == 1 || (team_mode == 0 && std::max(nq, load) <= team_max_nq));
                const uint32_t gi = team_g ? (g_inst | kWalkTeamFlag) : g_inst;
                const uint32_t grid = global_space(a, gi, (uint32_t)std::min<size_t>(nq, 1u << 20), nq <= 256 && !lone_pipe);  // (lone_pipe: `next` lives in LDS; the heap only serves the rare second chance)
                if (team_g && grid != nq) fail(VS_ERR_DEVICE, "team walk: workspace");
                a.work_counter = retry + 1;
                // Lone queries on float indexes: the PIPELINED walk first (kernels_pipe.hip: the walker decides on registers, the other
                // waves measure candidates ahead of it -- a fifth of the team walk's chain per hop); a query in which two equal
                // distances meet is handed to the team form of the usearch-order walk right behind (same stream, same workspace).
                static const uint32_t pipe_pool = std::getenv("VS_HNSW_PIPE_POOL") ? (uint32_t)std::atoi(std::getenv("VS_HNSW_PIPE_POOL")) : 12288u;
                if (team_g && pipe_usable(ef)) {
                    WalkArgs p = a;
                    p.retry_count = retry;
                    p.retry_list = retry + 64;
                    p.pipe_pool_cap = std::min<uint32_t>(std::max<uint32_t>(pipe_pool, 256u), 16384u);
                    p.pipe_fused_order = lone_pipe ? 1u : 0u;
                    static const bool walk_debug_p = std::getenv("VS_HNSW_WALK_DEBUG") != nullptr;
                    uint32_t* d_dbg_p = nullptr;
                    if (walk_debug_p) {
                        HIP_OK(hipMalloc((void**)&d_dbg_p, nq * 48));
                        HIP_OK(hipMemsetAsync(d_dbg_p, 0, nq * 48, st));
                        p.debug = d_dbg_p;
                    }
                    static const int dbg_second = std::getenv("VS_HNSW_PIPE_SECOND") ? std::atoi(std::getenv("VS_HNSW_PIPE_SECOND")) : 1;  // measurement aid: 0 = no second-chance launch (redone queries stay unanswered)
                    static const bool dbg_fused = !(std::getenv("VS_HNSW_PIPE_FUSED") && std::getenv("VS_HNSW_PIPE_FUSED")[0] == '0');  // measurement aid
                    if (!dbg_fused) p.pipe_fused_order = 0u;
                    HIP_OK(launch_pipe_walk(p, iters, st));
                    if (!dbg_second || (lone_pipe && tl_pipe_no_second)) {
                        // (the caller reads out_found on the host and re-runs the queries marked kPipeRedo itself)
                    } else if (lone_pipe) {
                        // second chance for plain queries of a float index: the team form of the fused-list kernel, i.e. exactly what a
                        // batch of them gets -- a lone query and a batched one never differ, however many distances tie
                        SearchArgs sa;
                        sa.ix = a.ix;
                        sa.queries = d_q;
                        sa.q_stride = dim;
                        sa.nq = (uint32_t)nq;
                        sa.k = (uint32_t)k;
                        sa.ef = ef;
                        sa.has_removed = a.has_removed;
                        sa.stress_small_table = 0u;
                        sa.wide_tags = beyond_lds_tags(ef) ? 1u : 0u;
                        sa.team = (uint32_t)kSearchTeam;
                        sa.out_keys = d_keys_out;
                        sa.out_dist = d_dist_out;
                        sa.out_found = d_found;
                        sa.stats = d_stats;
                        sa.qlist = p.retry_list;
                        sa.qcount = p.retry_count;
                        HIP_OK(launch_search(sa, iters, st));
                    } else {
                        WalkArgs r = a;  // second chance: only the queries the pipelined walk listed, on the usearch-order walk
                        r.qlist = p.retry_list;
                        r.qcount = p.retry_count;
                        HIP_OK(launch_walk(r, iters, gi, grid, st, nullptr));
                    }
                    if (walk_debug_p) {
                        std::vector<uint32_t> h(nq * 12);
                        uint32_t redone = 0;
                        HIP_OK(hipMemcpyAsync(h.data(), d_dbg_p, nq * 48, hipMemcpyDeviceToHost, st));
                        HIP_OK(hipMemcpyAsync(&redone, retry, 4, hipMemcpyDeviceToHost, st));
                        HIP_OK(hipStreamSynchronize(st));
                        (void)hipFree(d_dbg_p);
                        static const char* names[12] = {"max_next (prof: clk/16 entry wait)", "evals (prof: hops that waited)", "hops", "early+windows<<16", "clk/16 pop", "clk/16 read+issue+early", "clk/16 atomics",
                                                        "clk/16 verdicts", "clk/16 push+top", "clk/16 schedule", "misses (prof: clk per job part)", "refills"};
                        for (size_t i = 0; i < std::min<size_t>(nq, 4); ++i) {
                            fprintf(stderr, "[walk pipe] query %zu:", i);
                            for (int c = 0; c < 12; ++c) fprintf(stderr, " %s %u;", names[c], h[i * 12 + c]);
                            fprintf(stderr, " redone %u of %zu\n", redone, nq);
                        }
                    }
                    pipe_launches.fetch_add(1, std::memory_order_relaxed);
                    return;
                }
                static const bool walk_debug_g = std::getenv("VS_HNSW_WALK_DEBUG") != nullptr;  // measurement aid, as below
                uint32_t* d_dbg_g = nullptr;
                if (walk_debug_g) {
                    HIP_OK(hipMalloc((void**)&d_dbg_g, nq * 48));
                    HIP_OK(hipMemsetAsync(d_dbg_g, 0, nq * 48, st));
                    a.debug = d_dbg_g;
                }
                HIP_OK(launch_walk(a, iters, gi, grid, st, nullptr));
                if (walk_debug_g) {
                    std::vector<uint32_t> h(nq * 12);
                    HIP_OK(hipMemcpyAsync(h.data(), d_dbg_g, nq * 48, hipMemcpyDeviceToHost, st));
                    HIP_OK(hipStreamSynchronize(st));
                    (void)hipFree(d_dbg_g);
                    static const char* names[12] = {"max_next", "evals", "hops", "pushed", "clk/16 startup", "clk/16 pop", "clk/16 visited",
                                                    "clk/16 distances", "clk/16 admission", "clk/16 pushes", "clk/16 merge", "clk/16 -"};
                    for (size_t i = 0; i < std::min<size_t>(nq, 4); ++i) {
                        fprintf(stderr, "[walk global%s] query %zu:", team_g ? " team" : "", i);
                        for (int c = 0; c < 12; ++c) fprintf(stderr, " %s %u;", names[c], h[i * 12 + c]);
                        fprintf(stderr, "\n");
                    }
                }
                return;
            }
            // LDS visited table; queries that exhaust it (or the heap workspace) go to a global-bitmap launch behind
            // Beams up to 128 visit ~3,000 nodes: a 4,096-entry two-choice table holds them in half the LDS (11 instead of 6
            // walks per CU).  Queries that outgrow it are retried exactly (global bitmap); should a data set make that
            // common -- the previous launch's retry count is read back with every launch -- the index goes back to the
            // 8,192-entry table for good.
            a.bitmap_words = 0;
            a.vlog_cap = 0;
            a.heap_cap = 8192;
            a.space_stride = walk_space_stride(0, 0, a.heap_cap);
            // batches too small to fill the chip (lone callers): a team of waves per query, as the fused-list kernel does
            if (inst == WALK_LDS_256_DENSE && (team_mode == 1 || (team_mode == 0 && std::max(nq, load) <= team_max_nq))) inst = WALK_LDS_256;  // (lone callers: the team form)
            const bool team_walk = inst != WALK_LDS_512 && inst != WALK_LDS_128_TINY &&
                                   (team_mode == 1 || (team_mode == 0 && std::max(nq, load) <= team_max_nq));
            const uint32_t launch_inst = team_walk ? (inst | kWalkTeamFlag) : inst;
            uint32_t grid = 0;
            static const uint32_t grid_limit = std::getenv("VS_HNSW_WALK_GRID") ? (uint32_t)std::max(1, std::atoi(std::getenv("VS_HNSW_WALK_GRID"))) : (1u << 20);  // residency experiments
            HIP_OK(launch_walk(a, iters, launch_inst, grid_limit, st, &grid));
            a.space = (char*)wr.lds_space.ensure((size_t)grid * a.space_stride);
            a.retry_count = retry;
            a.retry_list = retry + 64;
            a.work_counter = retry + 1;
            WalkArgs r = a;  // the retry launch: same queries, same outputs, global bitmap, only the listed queries
            r.qlist = a.retry_list;
            r.qcount = a.retry_count;
            r.retry_list = nullptr;
            r.retry_count = nullptr;
            r.work_counter = retry + 2;
            const uint32_t rgrid = global_space(r, WALK_GLOBAL_512, 256, true);
            static const bool walk_debug = std::getenv("VS_HNSW_WALK_DEBUG") != nullptr;  // measurement aid: per-query walk sizes to stderr
            uint32_t* d_dbg = nullptr;
            if (walk_debug) {
                HIP_OK(hipMalloc((void**)&d_dbg, nq * 48));
                HIP_OK(hipMemsetAsync(d_dbg, 0, nq * 48, st));
                a.debug = r.debug = d_dbg;
            }
            HIP_OK(launch_walk(a, iters, launch_inst, grid, st, nullptr));
            HIP_OK(launch_walk(r, iters, WALK_GLOBAL_512, rgrid, st, nullptr));
            {  // how many queries of this launch had to be retried: looked at by the next launch on this stream
                if (!wr.retry_seen) {
                    HIP_OK(hipHostMalloc((void**)&wr.retry_seen, 16, hipHostMallocDefault));
                    HIP_OK(hipEventCreateWithFlags(&wr.retry_seen_ev, hipEventDisableTiming));
                }
                // a copy of an earlier launch may still be in flight (pipelined callers): it lands before this one's (stream
                // order), and the record is not read until this launch's event has fired
                wr.retry_seen[1] = (uint32_t)nq;
                wr.retry_seen[2] = inst;
                wr.retry_seen_owner = this;
                HIP_OK(hipMemcpyAsync(&wr.retry_seen[0], retry, 4, hipMemcpyDeviceToHost, st));
                HIP_OK(hipEventRecord(wr.retry_seen_ev, st));
                wr.retry_seen_valid = true;
            }
            if (walk_debug) {
                std::vector<uint32_t> h(nq * 12);
                uint32_t retried = 0;
                HIP_OK(hipMemcpyAsync(h.data(), d_dbg, nq * 48, hipMemcpyDeviceToHost, st));
                HIP_OK(hipMemcpyAsync(&retried, retry, 4, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                (void)hipFree(d_dbg);
                for (int c = 0; c < 12; ++c) {
                    std::vector<uint32_t> v(nq);
                    for (size_t i = 0; i < nq; ++i) v[i] = h[i * 12 + c];
                    std::sort(v.begin(), v.end());
                    static const char* names[12] = {"max_next", "evals", "hops", "pushed", "clk/16 startup", "clk/16 pop", "clk/16 visited",
                                                    "clk/16 distances", "clk/16 admission", "clk/16 pushes", "clk/16 merge", "clk/16 -"};
                    fprintf(stderr, "[walk] %s: p50 %u p90 %u p99 %u max %u\n", names[c], v[nq / 2], v[nq * 9 / 10], v[nq * 99 / 100], v[nq - 1]);
                }
                fprintf(stderr, "[walk] ef %u grid %u retried %u of %zu\n", ef, grid, retried, nq);
            }
            return;
        }
        SearchArgs a;
        a.ix = view();
        a.queries = d_q;
        a.q_stride = dim;
        a.nq = (uint32_t)nq;
        a.k = (uint32_t)k;
        a.ef = ef;
        a.has_removed = removed.load() ? 1u : 0u;
        a.stress_small_table = stress_small_table ? 1u : 0u;
        a.wide_tags = (beyond_lds_tags(ef) || force_wide_tags) ? 1u : 0u;
        const size_t on_device = std::max(nq, load);
        a.team = (team_mode == 1 || (team_mode == 0 && on_device <= team_max_nq)) ? (uint32_t)kSearchTeam
                 : (team_mode == 0 && on_device <= 3 * team_max_nq)               ? (uint32_t)kSearchTeamMid  // measured: 4 waves win up to ~800
                 : (team_mode == 3)                                               ? (uint32_t)kSearchTeamMid
                                                                                  : 1u;
        a.out_keys = d_keys_out;
        a.out_dist = d_dist_out;
        a.out_found = d_found;
        a.stats = d_stats;
        HIP_OK(launch_search(a, iters, st));
    }

    // max |row| of the stored vectors (removed rows included: an upper bound is all the certificate needs); rows added since
    // the last call are folded in.  view_mu is held (shared) by the caller.
    float max_row_norm(const IndexView& ix, hipStream_t st) {
        std::lock_guard<std::mutex> g(norm_mu);
        if (!d_max_norm) {
            HIP_OK(hipMalloc((void**)&d_max_norm, 4));
            HIP_OK(hipMemset(d_max_norm, 0, 4));
        }
        const size_t n = slots;
        if (n > max_norm_slots) {
            HIP_OK(launch_row_norm_max(ix, (uint32_t)max_norm_slots, (uint32_t)(n - max_norm_slots), d_max_norm, st));
            uint32_t bits = 0;
            HIP_OK(hipMemcpyAsync(&bits, d_max_norm, 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            std::memcpy(&max_norm_value, &bits, 4);
            max_norm_slots = n;
        }
        return max_norm_value;
    }

    // The bf16 plane covers rows [0, slots) (+ zero rows up to a whole tile).  plane_mu and view_mu (shared) are held.
    bool ensure_plane(const IndexView& ix, hipStream_t st) {
        const size_t kp = block1_plane_k(ix);
        const size_t rows_cap = block1_plane_rows((uint32_t)std::max(capacity, slots));
        try {
            if (rows_cap != plane_rows_cap) {
                HIP_OK(hipStreamSynchronize(st));
                size_t free_b = 0, total_b = 0;
                HIP_OK(hipMemGetInfo(&free_b, &total_b));
                const size_t want = rows_cap * kp * 2;
                if (ar_plane.extra_needed(want, device) + (1ull << 30) > free_b) fail(VS_ERR_OUT_OF_MEMORY, "no HBM for the bf16 plane");
                // The plane is tile-major (a 256-row tile is kp / 64 blocks of 256 x 128 B): the bytes of the first r rows are the WHOLE
                // tiles below r -- a byte count of r rows would cover only the first k steps of a partial last tile, and a copying
                // resize would leave the rest of that tile uninitialised.  Whole tiles are kept, and the partial tile is converted
                // again (plane_done rounded down), so nothing depends on what a copy carried over.
                const size_t whole = std::min(plane_done, rows_cap) / 256 * 256;
                const size_t keep = whole * kp * 2;
                ar_plane.resize(want, keep, device);
                plane_rows_cap = rows_cap;
                plane_done = whole;
            }
            if (!d_rho) {
                HIP_OK(hipMalloc((void**)&d_rho, 4));
                HIP_OK(hipMemset(d_rho, 0, 4));
            }
            const size_t n = slots, end = block1_plane_rows((uint32_t)n);
            if (plane_done < n || plane_done == 0) {
                // rows added since the last search, and the zero rows that pad the last tile (they may since have become real rows)
                const size_t first = plane_done / 256 * 256;
                HIP_OK(launch_block1_plane_rows(ix, (uint16_t*)ar_plane.base, (uint32_t)first, (uint32_t)end, (uint32_t)n, d_rho, st));
                uint32_t bits = 0;
                HIP_OK(hipMemcpyAsync(&bits, d_rho, 4, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                std::memcpy(&plane_rho, &bits, 4);
                plane_done = n;
            }
            return true;
        } catch (const Fail& f) {
            if (f.code != VS_ERR_OUT_OF_MEMORY) throw;
            plane_failed = true;  // the other exact paths serve
            return false;
        }
    }

    // The int8 plane and its scales cover rows [0, slots) (+ zero rows up to a whole tile).  plane_mu and view_mu (shared) are held.
    bool ensure_plane8(const IndexView& ix, hipStream_t st) {
        const size_t kp = block8_plane_k(ix);
        const size_t rows_cap = block1_plane_rows((uint32_t)std::max(capacity, slots));
        try {
            if (rows_cap != plane8_rows_cap) {
                HIP_OK(hipStreamSynchronize(st));
                size_t free_b = 0, total_b = 0;
                HIP_OK(hipMemGetInfo(&free_b, &total_b));
                const size_t want = rows_cap * kp;
                if (ar_plane8.extra_needed(want, device) + ar_p8scale.extra_needed(rows_cap * 4, device) + (1ull << 30) > free_b)
                    fail(VS_ERR_OUT_OF_MEMORY, "no HBM for the int8 plane");
                const size_t whole = std::min(plane8_done, rows_cap) / 256 * 256;  // (tile-major: whole tiles are kept, see ensure_plane)
                ar_plane8.resize(want, whole * kp, device);
                ar_p8scale.resize(rows_cap * 4, whole * 4, device);
                plane8_rows_cap = rows_cap;
                plane8_done = whole;
            }
            if (!d_rho8) {
                HIP_OK(hipMalloc((void**)&d_rho8, 4));
                HIP_OK(hipMemset(d_rho8, 0, 4));
            }
            const size_t n = slots, end = block1_plane_rows((uint32_t)n);
            if (plane8_done < n || plane8_done == 0) {
                const size_t first = plane8_done / 256 * 256;
                HIP_OK(launch_block8_plane_rows(ix, (uint8_t*)ar_plane8.base, (float*)ar_p8scale.base, (uint32_t)first, (uint32_t)end, (uint32_t)n, d_rho8, st));
                uint32_t bits = 0;
                HIP_OK(hipMemcpyAsync(&bits, d_rho8, 4, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                std::memcpy(&plane8_rho, &bits, 4);
                plane8_done = n;
            }
            return plane8_rho < 0.25f;  // (rows that int8 cannot represent -- an infinity, a NaN -- leave the bf16 plane to serve)
        } catch (const Fail& f) {
            if (f.code != VS_ERR_OUT_OF_MEMORY) throw;
            plane8_failed = true;
            return false;
        }
    }

    void exact_device(const float* d_q, size_t nq, size_t k, uint64_t* d_keys_out, float* d_dist_out, uint32_t* d_found,
                      hipStream_t st, WorkCtx& w) {
        if (k == 0 || k > 256) fail(VS_ERR_UNSUPPORTED, "exact search supports 1 <= k <= 256");
        std::shared_lock<std::shared_mutex> vg(view_mu);
        ExactArgs a;
        a.ix = view();
        a.queries = d_q;
        a.q_stride = dim;
        a.nq = (uint32_t)nq;
        a.k = (uint32_t)k;
        a.slots = (uint32_t)slots;
        a.use_valu = exact_valu.load() ? 1u : 0u;
        a.out_keys = d_keys_out;
        a.out_dist = d_dist_out;
        a.out_found = d_found;
        // Large float indexes, cos / ip, k <= 64: nominate with ONE bf16 product per score over the bf16 plane of the rows, re-score
        // the nominees exactly, certify; an uncertified batch goes on to the split-bf16 pass, and from there to the f32 path.
        // Round 6: first over the int8 plane (half the bytes); an uncertified batch goes on to the bf16 plane.
        bool took_plane8 = false;
        if (block1_supported(a.ix, a.k) && slots >= (1u << 16) && !a.use_valu && exact_mode == 0 && !plane8_failed && !plane8_off) {
            std::lock_guard<std::mutex> pg(plane_mu);
            if (ensure_plane8(a.ix, st)) {
                const float mx = metric == VS_METRIC_IP ? max_row_norm(a.ix, st) : 1.f;
                char* scratch = (char*)w.f.ensure(block8_scratch_bytes((uint32_t)nq, dim) + 256);
                uint32_t* d_unc = (uint32_t*)scratch;
                HIP_OK(hipMemsetAsync(d_unc, 0, 4, st));
                HIP_OK(launch_block8_search(a, scratch + 256, (const uint8_t*)ar_plane8.base, (const float*)ar_p8scale.base, plane8_rho, mx, d_unc, st));
                uint32_t unc = 0;
                HIP_OK(hipMemcpyAsync(&unc, d_unc, 4, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                plane8_batches += 1;
                plane_batches += 1;  // (batches that entered the plane stages, whichever plane: counted once)
                took_plane8 = true;
                if (unc == 0) return;
                plane8_fallbacks += 1;
            }
        }
        if (block1_supported(a.ix, a.k) && slots >= (1u << 16) && !a.use_valu && exact_mode == 0 && !plane_failed) {
            std::lock_guard<std::mutex> pg(plane_mu);
            if (ensure_plane(a.ix, st)) {
                const float mx = metric == VS_METRIC_IP ? max_row_norm(a.ix, st) : 1.f;
                char* scratch = (char*)w.f.ensure(block1_scratch_bytes((uint32_t)nq, dim) + 256);
                uint32_t* d_unc = (uint32_t*)scratch;
                HIP_OK(hipMemsetAsync(d_unc, 0, 4, st));
                HIP_OK(launch_block1_search(a, scratch + 256, (const uint16_t*)ar_plane.base, plane_rho, mx, d_unc, st));
                uint32_t unc = 0;
                HIP_OK(hipMemcpyAsync(&unc, d_unc, 4, hipMemcpyDeviceToHost, st));
                HIP_OK(hipStreamSynchronize(st));
                if (!took_plane8) plane_batches += 1;
                if (unc == 0) return;
                plane_fallbacks += 1;
                took_plane8 = false;
            }
        }
        if (took_plane8) plane_fallbacks += 1;  // (no bf16 plane to go on to: the batch left the plane stages uncertified)
        if (block_search_supported(a.ix, a.k) && slots >= (1u << 16) && !a.use_valu && !exact_f32_only) {
            const float mx = metric == VS_METRIC_IP ? max_row_norm(a.ix, st) : 1.f;
            char* scratch = (char*)w.f.ensure(block_scratch_bytes((uint32_t)nq, dim) + 256);
            uint32_t* d_unc = (uint32_t*)scratch;
            HIP_OK(hipMemsetAsync(d_unc, 0, 4, st));
            HIP_OK(launch_block_search(a, scratch + 256, mx, d_unc, st));
            uint32_t unc = 0;
            HIP_OK(hipMemcpyAsync(&unc, d_unc, 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            block_batches += 1;
            if (unc == 0) return;
            block_fallbacks += 1;
        }
        void* scratch = w.f.ensure(exact_scratch_bytes((uint32_t)nq, (uint32_t)k, dim));
        HIP_OK(launch_exact(a, scratch, st));
    }

    // allow (host, optional): one allow-bitmap over slots for the whole batch (filtered search).
    void search_host(const float* q, size_t nq, size_t k, uint64_t* keys, float* dist, size_t* found, bool exact,
                     const std::vector<uint32_t>* allow = nullptr) {
        if (!nq) return;
        use_device();
        housekeeping();
        Lease w(device);
        hipStream_t st = w->stream;
        float* d_q = (float*)w->a.ensure(nq * dim * 4);
        uint64_t* d_k = (uint64_t*)w->b.ensure(nq * k * 8);
        float* d_d = (float*)w->c.ensure(nq * k * 4);
        uint32_t* d_f = (uint32_t*)w->d.ensure(nq * 4);
        HIP_OK(hipMemcpyAsync(d_q, q, nq * dim * 4, hipMemcpyHostToDevice, st));
        if (exact) {
            exact_device(d_q, nq, k, d_k, d_d, d_f, st, *w.ctx);
        } else if (allow) {
            uint32_t* d_allow = (uint32_t*)w->e.ensure(allow->size() * 4);
            HIP_OK(hipMemcpyAsync(d_allow, allow->data(), allow->size() * 4, hipMemcpyHostToDevice, st));
            search_device(d_q, nq, k, d_k, d_d, d_f, st, 0, d_allow, 0);
        } else {
            search_device(d_q, nq, k, d_k, d_d, d_f, st);
        }
        std::vector<uint32_t> f32(nq);
        HIP_OK(hipMemcpyAsync(keys, d_k, nq * k * 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(dist, d_d, nq * k * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipMemcpyAsync(f32.data(), d_f, nq * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        for (size_t i = 0; i < nq; ++i) found[i] = f32[i] == kWalkFailed ? (size_t)-1 : f32[i];  // (size_t)-1: rank exhaustively
    }

    // One query per FFI call (reference usearch.rs:212): handled by the per-device SearchService below.
    int search_one(const float* q, size_t k, uint64_t* keys, float* dist, size_t* found);
    bool search_one_pod(const float* q, size_t k, uint64_t* keys, float* dist, size_t* found);
    void search_async(const float* q, size_t k, uint64_t* keys, float* dist, size_t* found,
                      void (*cb)(void*, int), void* ctx);

    // Exhaustive ranking of every member against one query (filtered search with selective predicates, k beyond
    // the LDS beam), on the device: all distances (one wave per row), a 64-bit radix sort of
    // (order-preserving distance bits, slot) -- removed members last --, then (key, distance) in ascending order.
    // The host walks that order in chunks and stops as soon as it has what it needs, so a query moves a few hundred
    // KB over PCIe instead of every distance and every key (40 + 80 MB at 10M members) and sorts nothing itself.
    struct Ranked {
        Engine& e;
        Lease w;
        size_t n = 0;
        uint64_t* d_keys_sorted = nullptr;
        float* d_dist_sorted = nullptr;
        Ranked(Engine& eng, const float* q) : e(eng), w(eng.device) {
            e.use_device();
            hipStream_t st = w->stream;
            n = e.slots;
            if (!n) return;
            std::shared_lock<std::shared_mutex> vg(e.view_mu);
            float* d_q = (float*)w->a.ensure((size_t)e.dim * 4);
            HIP_OK(hipMemcpyAsync(d_q, q, (size_t)e.dim * 4, hipMemcpyHostToDevice, st));
            float* d_d = (float*)w->f.ensure((n + e.dim + 64) * 4);
            const IndexView ix = e.view();
            HIP_OK(launch_distance_row(ix, d_q, (uint32_t)n, d_d, st, nullptr));
            uint64_t* rank = (uint64_t*)w->b.ensure(n * 8);
            uint64_t* sorted = (uint64_t*)w->c.ensure(n * 8);
            HIP_OK(launch_rank_keys(ix, d_d, (uint32_t)n, rank, st));
            const size_t tb = sort_keys_temp_bytes(n);
            HIP_OK(sort_keys(w->d.ensure(tb), tb, rank, sorted, n, st));
            d_keys_sorted = rank;  // the unsorted ranks are dead: reuse as the key output
            d_dist_sorted = d_d;   // likewise the raw distances
            HIP_OK(launch_rank_emit(ix, sorted, (uint32_t)n, d_keys_sorted, d_dist_sorted, st));
            HIP_OK(hipStreamSynchronize(st));  // nothing reads the arenas after the lock is gone
        }
        // members [from, from + count) of the ascending order; removed members (free key) mark the end
        void fetch(size_t from, size_t count, uint64_t* keys, float* dist) {
            hipStream_t st = w->stream;
            HIP_OK(hipMemcpyAsync(keys, d_keys_sorted + from, count * 8, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(dist, d_dist_sorted + from, count * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
        }
    };

    // usearch filtered_search (reference usearch.rs:224-248; wrapper filtered_ann :1107-1154): the predicate gates
    // admission to `top` INSIDE the traversal, rejected nodes are still expanded.  The predicate is host state (a table
    // read-lock + restriction evaluation, usearch.rs:1118-1124) and cannot run in the kernel, so it is evaluated once
    // per live member into an allow-bitmap over slots (in parallel above 256k members), which the usearch-order walk
    // tests at admission (walk_device.hpp): same result set, same order as the CPU algorithm on the same graph.
    // Beams beyond the widest walk instance, or a walk that outgrows its workspace, fall back to ranking every
    // member exhaustively (exact; all matches are found, as vs_index.rs:1119-1158 requires: 9 of 30 with limit 100).
    std::vector<uint32_t> allow_bitmap(vs_hnsw_predicate pred, void* pctx) {
        size_t n;
        {
            std::lock_guard<std::mutex> g(mod_mu);  // searches do not overlap adds / reserve (usearch.rs:590-612): h_keys is stable
            n = slots;
        }
        const size_t words = (n + 31) / 32;
        std::vector<uint32_t> bits(words ? words : 1, 0u);
        auto run = [&](size_t w0, size_t w1) {
            for (size_t w = w0; w < w1; ++w) {
                uint32_t v = 0;
                const size_t s1 = std::min(n, (w + 1) * 32);
                for (size_t s = w * 32; s < s1; ++s) {
                    const uint64_t key = h_keys[s];
                    if (key != kFreeKey && pred(key, pctx)) v |= 1u << (s & 31);
                }
                bits[w] = v;
            }
        };
        const size_t threads = n >= (1u << 18) ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
        if (threads <= 1) {
            run(0, words);
        } else {
            std::vector<std::thread> th;
            const size_t per = (words + threads - 1) / threads;
            for (size_t t = 0; t < threads; ++t)
                th.emplace_back(run, std::min(words, t * per), std::min(words, (t + 1) * per));
            for (auto& x : th) x.join();
        }
        return bits;
    }

    // Filtered search on a large index without asking the predicate for every member (the reference's predicate takes a
    // table read-lock per call, usearch.rs:1118-1124; usearch itself only asks for the candidates a walk admits).  Rounds:
    // the walk runs with the verdicts known so far (`known` / `allow` bitmaps), lists the slots whose verdict it needs and
    // does not have -- taking them as rejected, which can only make it explore further --, the host evaluates the predicate
    // for exactly those, and the walk runs again.  A round that lists nothing used only true verdicts: it IS the walk
    // with the full predicate, so its result is exact.  The per-round budget doubles, predicate calls total a small
    // multiple of what the final walk needs.  (size_t)-1: gave up (caller falls back to the full bitmap).
    // ---- batched rounds (round 4) -----------------------------------------------------------------------------------------------------
    // Every lazily filtered query is a chain of rounds, and the reference puts each query on a blocking thread of its own (usearch.rs:937-948:
    // as many as there are requests).  One launch per caller and round means one small kernel per stream at a time -- 16 streams, 16 walks
    // in flight on 256 CUs.  Here the callers' rounds share launches: a caller queues its round (one PipeQuery: its buffers, the verdicts
    // of its last round, its budget) and waits on a flag in its own pinned block; whoever finds no launch in progress takes everything
    // queued -- exact walks and exploring rounds are different kernels -- and launches it, one workgroup per query, the device's streams
    // in turn.  The kernel does the round's whole exchange (kernels_pipe.hip): no export / apply launches, no copy engine, no
    // hipStreamSynchronize -- and as many walks in flight as callers.
    struct FilterBatcher {
        struct Req {
            PipeQuery pq;
            int explore;
            uint32_t ef;  // max(expansion_search, k) of the caller: one launch serves one beam
        };
        std::mutex mu;
        std::vector<Req> pending;
        bool launching = false;
        static constexpr int kTables = 64, kMaxBatch = 128;  // (a table is reused 64 launches later: not even a 1 % filter's walk is still running then)
        PipeQuery* table[kTables] = {};
        hipEvent_t ev[kTables] = {};
        bool used[kTables] = {};
        int next_table = 0;
        unsigned next_stream = 0;
        hipEvent_t stream_ev[DeviceStreams::kMax] = {};  // the last batch launched on each of the device's streams
        std::atomic<uint64_t> launches{0}, rounds{0};
        ~FilterBatcher() {
            for (int i = 0; i < kTables; ++i) {
                if (table[i]) graveyard().bury(nullptr, table[i]);
                if (ev[i]) (void)hipEventDestroy(ev[i]);
            }
            for (auto& e : stream_ev)
                if (e) (void)hipEventDestroy(e);
        }
    };
    FilterBatcher batcher;
    static constexpr uint32_t kBatchVlogCap = 1u << 16, kBatchHeapCap = 1u << 18;
    size_t batch_space_bytes(size_t n_slots) const { return walk_space_stride((uint32_t)((n_slots + 31) / 32), kBatchVlogCap, kBatchHeapCap); }

    // ---- pods (pipe_pod.hpp): the round is posted to a resident workgroup instead of being launched ---------------------------------
    std::atomic<uint64_t> pod_rounds{0}, pod_opens{0};
    std::atomic<uint64_t> batched_done{0}, batched_handed_over{0}, batched_no_pod{0}, batched_second_chances{0};  // filtered queries answered by posted / batched rounds; handed over to rounds of their own; rounds no pod could take
    std::atomic<uint32_t> wait_typical_us[4] = {};  // how long this index's callers lately waited for a plain query / an exact filtered walk / an exploring round
    // Post one query / round (mode: 0 plain, 1 exact filtered walk, 2 exploring round) to a pod of this index; no ticket: no pod can
    // take it now (the caller launches as before).  `n`: the index's slots, as the caller's workspace is laid out for.
    // `n`: what the caller's workspace is laid out for (layout_slots(): the index's capacity).
    PodTicket pod_submit(int mode, uint32_t ef, size_t n, const PipeQuery& pq) {
        PodPool& pp = pod_pool(device);
        if (!pp.enabled || ef > 512) return {};
        if (mode == 4 && ef > 256) return {};  // (walk pods, round 6: the usearch-order team walk's LDS instances for beams up to 256)
        const uint32_t efcap = mode == 4 ? (ef <= 128 ? 128u : 256u) : ef <= 256 ? 256u : 512u;
        const uint32_t walk_kind = mode == 2 ? 1u : mode == 3 ? 2u : 0u;  // PodSlot::explore: 0 exact, 1 exploring, 2 the walk that asks (round 6)
        if (mode == 2) mode = 1;  // (one kind of pod serves both kinds of round of a filtered query; walks that ask -- mode 3 -- have pods of their own)
        std::lock_guard<std::mutex> g(pp.mu);
        if (pp.holds > 0) return {};  // somebody is about to synchronise the device: no pod opens, none takes a post
        if (being_modified.load(std::memory_order_seq_cst) > 0) return {};  // (the index is being modified -- PodFreeze: the caller breaks the reference's contract, and is served by a launch)
        int use = -1, free_pod = -1;
        for (int i = 0; i < PodPool::kPods && use < 0; ++i) {
            Pod& p = pp.pods[i];
            if (p.state != Pod::kFree && p.owner == this && p.frozen) return {};  // (the index is being modified: the caller breaks the reference's contract, and is served by a launch)
            if (p.state == Pod::kOpen && p.owner == this && p.mode == mode && p.efcap == efcap && p.index_slots == n && p.n_busy < p.n) use = i;
            if (p.state == Pod::kFree && free_pod < 0) free_pod = i;
        }
        if (use < 0) {
            if (free_pod < 0) return {};
            Pod& p = pp.pods[free_pod];
            use_device();
            try {  // (a pod that cannot be opened is no error: the caller launches as before)
            if (!p.st) {
                HIP_OK(hipStreamCreateWithFlags(&p.st, hipStreamNonBlocking));
                g_streams_created.fetch_add(1, std::memory_order_relaxed);
            }
            if (!p.ctl) HIP_OK(hipHostMalloc((void**)&p.ctl, sizeof(PodCtl), hipHostMallocDefault));
            if (!p.slots) HIP_OK(hipHostMalloc((void**)&p.slots, sizeof(PodSlot) * Pod::kSlots, hipHostMallocDefault));
            if (!p.stage) HIP_OK(hipMalloc((void**)&p.stage, sizeof(PipeQuery) * Pod::kSlots));
            std::memset(p.ctl, 0, sizeof(PodCtl));
            p.frozen = false;
            std::memset(p.slots, 0, sizeof(PodSlot) * Pod::kSlots);
            std::memset(p.busy, 0, sizeof(p.busy));
            std::memset(p.seq, 0, sizeof(p.seq));
            p.n = pp.n_slots;
            p.n_busy = 0;
            ++p.gen;
            WalkArgs a{};
            {
                std::shared_lock<std::shared_mutex> vg(view_mu);
                a.ix = view();
            }
            a.nq = p.n;
            a.ef = efcap;  // (the instance; a query's own beam travels in its slot)
            a.k = 0;
            a.has_removed = removed.load() ? 1u : 0u;
            // (what adds and removes change is read per query from the pod's control block: PodFreeze rewrites it after every modification)
            p.ctl->entry_slot = a.ix.entry_slot;
            p.ctl->max_level = a.ix.max_level;
            p.ctl->has_removed = a.has_removed;
            a.bitmap_words = (uint32_t)((n + 31) / 32);
            a.vlog_cap = kBatchVlogCap;
            a.heap_cap = kBatchHeapCap;
            a.stats = d_stats;
            a.pipe_qtable = p.stage;
            a.pipe_explore = mode == 3 ? 2u : 0u;
            a.pipe_fused_order = mode == 0 ? 1u : 0u;
            a.pipe_pool_cap = 12288u;
            std::atomic_thread_fence(std::memory_order_seq_cst);
            if (mode == 4) HIP_OK(launch_walk_pod(a, iters, p.st, p.slots, p.ctl));  // (b1: kernels_walk.hip)
            else HIP_OK(launch_pipe_pod(a, iters, p.st, p.slots, p.ctl));
            } catch (...) {
                (void)hipGetLastError();
                return {};
            }
            p.state = Pod::kOpen;
            p.owner = this;
            p.mode = mode;
            p.efcap = efcap;
            p.index_slots = n;
            p.opened = p.last_used = std::chrono::steady_clock::now();
            pp.n_opened.fetch_add(1, std::memory_order_relaxed);
            pod_opens.fetch_add(1, std::memory_order_relaxed);
            if (!pp.keeper_started) {
                pp.keeper_started = true;
                std::thread([&pp] { pp.keep(); }).detach();
            }
            pp.keeper_cv.notify_all();
            use = free_pod;
        }
        Pod& p = pp.pods[use];
        uint32_t slot = 0;
        while (slot < p.n && p.busy[slot]) ++slot;
        if (slot >= p.n) return {};
        if (__atomic_load_n(&p.slots[slot].left, __ATOMIC_ACQUIRE)) {  // its workgroup has gone (the host stood still for seconds): no more posts
            pp.close_locked(p);
            return {};
        }
        PodSlot& sl = p.slots[slot];
        sl.q = pq;
        sl.ef = ef;
        sl.explore = walk_kind;
        __atomic_store_n(&sl.posted, ++p.seq[slot], __ATOMIC_RELEASE);
        p.busy[slot] = true;
        ++p.n_busy;
        p.last_used = std::chrono::steady_clock::now();
        pp.n_served.fetch_add(1, std::memory_order_relaxed);
        pod_rounds.fetch_add(1, std::memory_order_relaxed);
        pipe_launches.fetch_add(1, std::memory_order_relaxed);  // (walks of the pipelined kernel: launched or posted)
        PodTicket t;
        t.pod = use;
        t.slot = slot;
        t.gen = p.gen;
        return t;
    }
    struct PodRelease {  // the slot is free for the next caller once this one has its answer (or has given up on it)
        int device;
        PodTicket t;
        void done() {
            if (t) pod_pool(device).release(t);
            t = PodTicket{};
        }
        ~PodRelease() { done(); }
    };
    // where a modification's time goes (vs_hnsw_modify_stats): flushes of staged adds, closing of this index's pods, removes
    std::atomic<uint64_t> m_flushes{0}, m_flushed{0}, m_flush_ns{0}, m_quiesces{0}, m_quiesce_ns{0}, m_removes{0}, m_remove_ns{0};
    // ... and a search call's (vs_hnsw_call_stats): single-query calls and ns inside them, filtered calls / ns / ns waiting for the device / ns asking the predicate
    std::atomic<uint64_t> c_searches{0}, c_search_ns{0}, c_filtered{0}, c_filtered_ns{0}, c_filtered_wait_ns{0}, c_filtered_pred_ns{0}, c_flush_wait_ns{0};
    struct Stopwatch {
        std::atomic<uint64_t>& ns;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        explicit Stopwatch(std::atomic<uint64_t>& total) : ns(total) {}
        ~Stopwatch() { ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed); }
    };
    void pods_quiesce() {  // before anything that changes the view (entry point, arenas, removed flag)
        Stopwatch sw(m_quiesce_ns);
        m_quiesces.fetch_add(1, std::memory_order_relaxed);
        pod_pool(device).quiesce(this);
    }

    void launch_rounds(std::vector<FilterBatcher::Req>& take, size_t n_slots) {
        FilterBatcher& b = batcher;
        std::vector<uint32_t> efs;
        for (auto& r : take)
            if (std::find(efs.begin(), efs.end(), r.ef) == efs.end()) efs.push_back(r.ef);
        for (uint32_t ef : efs)
        for (int explore = 0; explore < 2; ++explore) {
            std::vector<const PipeQuery*> group;
            for (auto& r : take)
                if (r.explore == explore && r.ef == ef) group.push_back(&r.pq);
            for (size_t off = 0; off < group.size(); off += FilterBatcher::kMaxBatch) {
                const size_t n = std::min<size_t>(FilterBatcher::kMaxBatch, group.size() - off);
                const int t = b.next_table++ % FilterBatcher::kTables;
                if (!b.table[t]) {
                    HIP_OK(hipHostMalloc((void**)&b.table[t], sizeof(PipeQuery) * FilterBatcher::kMaxBatch, hipHostMallocDefault));
                    HIP_OK(hipEventCreateWithFlags(&b.ev[t], hipEventDisableTiming));
                }
                if (b.used[t]) HIP_OK(hipEventSynchronize(b.ev[t]));  // (the launch that read this table is long done)
                for (size_t i = 0; i < n; ++i) b.table[t][i] = *group[off + i];
                WalkArgs a{};
                {
                    std::shared_lock<std::shared_mutex> vg(view_mu);
                    a.ix = view();
                }
                a.nq = (uint32_t)n;
                a.ef = ef;
                a.k = 0;
                a.has_removed = removed.load() ? 1u : 0u;
                a.bitmap_words = (uint32_t)((n_slots + 31) / 32);
                a.vlog_cap = kBatchVlogCap;
                a.heap_cap = kBatchHeapCap;
                a.stats = d_stats;
                a.pipe_qtable = b.table[t];
                a.pipe_explore = explore ? 1u : 0u;
                a.pipe_pool_cap = 12288u;
                // A stream whose last batch has finished: a launch behind a running batch would wait for ALL of that batch's walks (stream
                // order), however many streams sit idle.  When every stream is busy the launcher waits for the first to finish -- the rounds
                // that queue meanwhile join the next launch.
                DeviceStreams& ds = device_streams(device);
                int si = -1;
                for (uint32_t spins = 0; si < 0; ++spins) {
                    for (int probe = 0; probe < ds.count && si < 0; ++probe) {
                        const int c = (int)((b.next_stream + (unsigned)probe) % (unsigned)ds.count);
                        if (!b.stream_ev[c] || hipEventQuery(b.stream_ev[c]) == hipSuccess) si = c;
                    }
                    if (si < 0) {
                        if (spins > 2000000u) si = (int)(b.next_stream % (unsigned)ds.count);  // (seconds: something is stuck; stream order will do)
                        else if (spins > 50) std::this_thread::sleep_for(std::chrono::microseconds(20));
                        else __builtin_ia32_pause();
                    }
                }
                b.next_stream = (unsigned)si + 1u;
                hipStream_t st;
                {
                    std::lock_guard<std::mutex> g(ds.mu);
                    st = ds.at(si);
                }
                if (!b.stream_ev[si]) HIP_OK(hipEventCreateWithFlags(&b.stream_ev[si], hipEventDisableTiming));
                HIP_OK(launch_pipe_walk(a, iters, st));
                HIP_OK(hipEventRecord(b.ev[t], st));
                HIP_OK(hipEventRecord(b.stream_ev[si], st));
                b.used[t] = true;
                b.launches.fetch_add(1, std::memory_order_relaxed);
                b.rounds.fetch_add(n, std::memory_order_relaxed);
                pipe_launches.fetch_add(1, std::memory_order_relaxed);
            }
        }
    }

    void wait_idle_batch_stream() {
        FilterBatcher& b = batcher;
        DeviceStreams& ds = device_streams(device);
        for (uint32_t spins = 0; spins < 2000000u; ++spins) {
            for (int c = 0; c < ds.count; ++c)
                if (!b.stream_ev[c] || hipEventQuery(b.stream_ev[c]) == hipSuccess) return;
            if (spins > 50) std::this_thread::sleep_for(std::chrono::microseconds(20));
            else __builtin_ia32_pause();
        }
    }

    // queue one round; on return it has been launched (by this thread or the one that was launching) or failed (its flag says redo)
    void submit_round(const PipeQuery& pq, bool explore, uint32_t ef, size_t n_slots) {
        FilterBatcher& b = batcher;
        std::unique_lock<std::mutex> lk(b.mu);
        b.pending.push_back({pq, explore ? 1 : 0, ef});
        if (b.launching) return;  // the thread that is launching takes it along
        b.launching = true;
        for (;;) {
            if (b.pending.empty()) {
                b.launching = false;
                return;
            }
            // (everything queued until a stream is free goes into one launch)
            lk.unlock();
            wait_idle_batch_stream();
            lk.lock();
            std::vector<FilterBatcher::Req> take;
            take.swap(b.pending);
            lk.unlock();
            try {
                use_device();
                launch_rounds(take, n_slots);
            } catch (...) {
                // whatever could not be launched is handed back as "not answered": its caller falls back to rounds of its own
                for (auto& r : take) {
                    if (__atomic_load_n(r.pq.cnt + 8, __ATOMIC_ACQUIRE) != r.pq.round_id) {
                        r.pq.cnt[0] = r.pq.cnt[1] = 0;
                        r.pq.cnt[2] = kPipeRedoFound;
                        __atomic_store_n(r.pq.cnt + 8, r.pq.round_id, __ATOMIC_RELEASE);
                    }
                }
            }
            lk.lock();
        }
    }

    // ---- verdicts remembered across the queries of ONE filter (vs_hnsw_filtered_search_keyed, round 5) ----------------------------------
    // usearch asks the predicate for every candidate of every query; the reference's predicate is a table read-lock + a restriction
    // evaluation (usearch.rs:1118-1124), the same function of the key for every query that carries the same restrictions.  A caller
    // that NAMES its filter (a fingerprint of the restrictions) lets the engine remember verdicts: device bitmaps [allow | known] per
    // filter, shared by its queries -- each seeds its own bitmaps from them and ORs what it learns into them (kernels_pipe.hip).  Once
    // the filter's neighbourhoods are known a query is ONE exact walk with no predicate call and no exploring round.  A slot whose
    // member changes (remove, re-use) is forgotten by every filter; reserve (the layout follows the capacity) drops the memories.
    struct FilterMemo {
        uint64_t key = 0;
        uint32_t* bits = nullptr;  // device: [allow: stride words | known: stride words]
        uint32_t stride = 0;
        uint64_t last_use = 0;
        std::atomic<uint32_t> asked_avg{0xFFFFFFFFu};  // verdicts the recent queries of this filter still had to ask the host for (moving average)
        std::atomic<uint64_t> queries{0}, asked{0};
        ~FilterMemo() {
            if (bits) graveyard().bury(bits, nullptr);
        }
    };
    static constexpr size_t kMaxMemos = 4;
    std::mutex memo_mu;
    std::vector<std::shared_ptr<FilterMemo>> memos;
    uint64_t memo_clock = 0;
    std::atomic<uint64_t> memo_queries{0}, memo_asked{0}, memo_created{0};
    std::shared_ptr<FilterMemo> memo_for(uint64_t filter_key) {
        if (!filter_key) return nullptr;
        std::lock_guard<std::mutex> g(memo_mu);
        for (auto& m : memos)
            if (m->key == filter_key) {
                m->last_use = ++memo_clock;
                return m;
            }
        const uint32_t stride = (uint32_t)((layout_slots() + 31) / 32);
        if (!stride) return nullptr;
        auto m = std::make_shared<FilterMemo>();
        if (hipMalloc((void**)&m->bits, (size_t)stride * 8) != hipSuccess) {  // (no HBM for it: the query runs without a memory)
            (void)hipGetLastError();
            m->bits = nullptr;
            return nullptr;
        }
        Lease w(device);
        HIP_OK(hipMemsetAsync(m->bits, 0, (size_t)stride * 8, w->stream));
        HIP_OK(hipStreamSynchronize(w->stream));
        m->key = filter_key;
        m->stride = stride;
        m->last_use = ++memo_clock;
        if (memos.size() >= kMaxMemos) {  // the least recently used one goes (queries that hold it finish with it)
            size_t lru = 0;
            for (size_t i = 1; i < memos.size(); ++i)
                if (memos[i]->last_use < memos[lru]->last_use) lru = i;
            memos.erase(memos.begin() + (long)lru);
        }
        memos.push_back(m);
        memo_created.fetch_add(1, std::memory_order_relaxed);
        return m;
    }
    void memos_forget(const std::vector<uint32_t>& changed, WorkCtx& w) {  // mod_mu held; no search runs (usearch.rs:590-612)
        std::lock_guard<std::mutex> g(memo_mu);
        if (memos.empty() || changed.empty()) return;
        uint32_t* d = (uint32_t*)w.f.ensure(changed.size() * 4 + 64);
        HIP_OK(hipMemcpyAsync(d, changed.data(), changed.size() * 4, hipMemcpyHostToDevice, w.stream));
        for (auto& m : memos) HIP_OK(launch_memo_forget(m->bits, m->stride, d, (uint32_t)changed.size(), w.stream));
        HIP_OK(hipStreamSynchronize(w.stream));
    }
    void memos_drop() {
        std::lock_guard<std::mutex> g(memo_mu);
        memos.clear();
    }
    // ---- invalidation (round 6).  The reference's predicate reads FILTERING COLUMNS (`table.is_valid_for`, usearch.rs:1118-1124), which
    // `Table::upsert` rewrites in place under a fixed PrimaryId (`update_columns`, table/mod.rs:676-695, called at :1053-1061 BEFORE
    // `update_index`, which returns no operation when the vector's timestamp is not newer, :905-910): a CQL UPDATE of such a column
    // changes a verdict with no remove and no add, so nothing above ever forgets it.  The host side says so:
    //   * memo_forget_filter(key): that filter's memory is dropped (0: every filter's) -- its next query asks afresh;
    //   * memo_forget_keys(keys): every filter forgets what it knows about those members (what `Table::upsert` calls for the row it
    //     rewrote): copy-on-forget -- each memory is replaced by a copy with those members' bits cleared.  Queries in flight keep the
    //     memory they started with (detached now: they finish with it, nobody reads it afterwards), so a verdict evaluated BEFORE the
    //     change can never reach the memory the queries that start AFTER this call returns are seeded from.  The copy takes `known`
    //     first and `allow` after it (a writer sets `allow` first): a known bit in the copy has its verdict in the copy.
    // A query that overlaps the call may use either verdict -- as a usearch query that overlaps the `table.write()` of the upsert may.
    std::atomic<uint64_t> memo_forgets{0}, memo_forgotten_keys{0};
    size_t memo_forget_filter(uint64_t filter_key) {
        std::lock_guard<std::mutex> g(memo_mu);
        size_t dropped = 0;
        for (size_t i = memos.size(); i-- > 0;)
            if (!filter_key || memos[i]->key == filter_key) {
                memos.erase(memos.begin() + (long)i);
                ++dropped;
            }
        memo_forgets.fetch_add(1, std::memory_order_relaxed);
        return dropped;
    }
    size_t memo_forget_keys(const uint64_t* keys_in, size_t n) {
        if (!n) return 0;
        std::vector<uint32_t> changed;
        {
            std::lock_guard<std::mutex> g(mod_mu);  // (the host map; a key that is only staged has no slot and no remembered verdict yet)
            for (size_t i = 0; i < n; ++i) {
                auto it = lookup.find(keys_in[i]);
                if (it != lookup.end()) changed.push_back(it->second);
            }
        }
        memo_forgets.fetch_add(1, std::memory_order_relaxed);
        memo_forgotten_keys.fetch_add(changed.size(), std::memory_order_relaxed);
        if (changed.empty()) return 0;
        use_device();
        std::lock_guard<std::mutex> g(memo_mu);
        if (memos.empty()) return changed.size();
        Lease w(device);
        uint32_t* d = (uint32_t*)w->f.ensure(changed.size() * 4 + 64);
        HIP_OK(hipMemcpyAsync(d, changed.data(), changed.size() * 4, hipMemcpyHostToDevice, w->stream));
        std::vector<std::shared_ptr<FilterMemo>> fresh;
        for (auto& m : memos) {
            auto c = std::make_shared<FilterMemo>();
            if (hipMalloc((void**)&c->bits, (size_t)m->stride * 8) != hipSuccess) {  // no HBM for the copy: the filter starts over
                (void)hipGetLastError();
                c->bits = nullptr;
                continue;
            }
            HIP_OK(hipMemcpyAsync(c->bits + m->stride, m->bits + m->stride, (size_t)m->stride * 4, hipMemcpyDeviceToDevice, w->stream));  // known ...
            HIP_OK(hipMemcpyAsync(c->bits, m->bits, (size_t)m->stride * 4, hipMemcpyDeviceToDevice, w->stream));                          // ... then allow
            HIP_OK(launch_memo_forget(c->bits, m->stride, d, (uint32_t)changed.size(), w->stream));
            c->key = m->key;
            c->stride = m->stride;
            c->last_use = m->last_use;
            c->asked_avg.store(m->asked_avg.load(std::memory_order_relaxed), std::memory_order_relaxed);
            c->queries.store(m->queries.load(std::memory_order_relaxed), std::memory_order_relaxed);
            c->asked.store(m->asked.load(std::memory_order_relaxed), std::memory_order_relaxed);
            fresh.push_back(std::move(c));
        }
        HIP_OK(hipStreamSynchronize(w->stream));
        memos.swap(fresh);  // (the old memories die with the last query that holds them)
        return changed.size();
    }

    // filtered_lazy through the batcher.  (size_t)-1: hand the query to the unbatched rounds (a tie where order matters, or a failure).
    size_t filtered_batched(const float* q, size_t k, vs_hnsw_predicate pred, void* pctx, uint64_t* keys, float* dist, uint32_t ef, FilterMemo* memo = nullptr) {
        use_device();
        housekeeping();
        const size_t n = slots_atomic.load(std::memory_order_acquire), lay = layout_slots();
        const size_t words = (n + 31) / 32, lay_words = (lay + 31) / 32;
        const uint32_t cap = 1u << 17;
        // (at most 128 of these queries hold a context -- a workspace of a few MB each -- at a time)
        static std::mutex gate_mu;
        static std::condition_variable gate_cv;
        static int gate_free = 128;
        {
            std::unique_lock<std::mutex> gl(gate_mu);
            gate_cv.wait(gl, [&] { return gate_free > 0; });
            --gate_free;
        }
        struct Permit {
            ~Permit() {
                {
                    std::lock_guard<std::mutex> gl(gate_mu);
                    ++gate_free;
                }
                gate_cv.notify_one();
            }
        } permit;
        Lease w(device);
        uint32_t* d_bits = (uint32_t*)w->e.ensure(words * 8);
        const size_t space = batch_space_bytes(lay);
        if (w->ws.bytes < space || w->ws_zeroed != lay_words) {
            char* p = (char*)w->ws.ensure(space);
            HIP_OK(hipMemsetAsync(p, 0, w->ws.bytes, w->stream));
            HIP_OK(hipStreamSynchronize(w->stream));
            w->ws_zeroed = lay_words;
        }
        // pinned, device-mapped: [flag, counters 64 B | list cap x 4 | verdicts cap | keys k x 8 | dist k x 4 | the query]
        const size_t pin_need = 64 + (size_t)cap * 5 + k * 12 + 64 + (size_t)dim * 4;
        if (w->pin_bytes < pin_need) {
            if (w->pin) graveyard().bury(nullptr, w->pin);
            w->pin = nullptr;
            w->pin_bytes = 0;
            HIP_OK(hipHostMalloc((void**)&w->pin, pin_need, hipHostMallocDefault));
            w->pin_bytes = pin_need;
        }
        uint32_t* h_cnt = (uint32_t*)w->pin;         // [0..3] counters
        uint32_t* h_done = (uint32_t*)w->pin + 8;    // the flag
        uint32_t* h_list = (uint32_t*)(w->pin + 64);
        uint8_t* h_verdict = (uint8_t*)(w->pin + 64 + (size_t)cap * 4);
        w->ask_dirty = cap;  // (this path writes verdicts wherever its rounds list slots)
        uint64_t* h_k = (uint64_t*)(w->pin + 64 + (size_t)cap * 5);
        float* h_d = (float*)(h_k + k);
        float* h_q = (float*)(w->pin + ((64 + (size_t)cap * 5 + k * 12 + 63) & ~(size_t)63));
        std::memcpy(h_q, q, (size_t)dim * 4);
        const uint32_t hint = lazy_need_hint.load();
        const uint32_t sel_hint = lazy_sel_hint.load();
        const uint32_t first_budget = std::max<uint32_t>(2048u, std::min<uint32_t>(cap / 2, hint + hint / 2));
        static const int guess_pct = std::getenv("VS_HNSW_FILTER_GUESS") ? std::atoi(std::getenv("VS_HNSW_FILTER_GUESS")) : 50;
        uint64_t n_known = 0, n_allowed = 0;
        // (a filter whose recent queries found nearly every verdict in its memory starts with the exact walk: no exploring round)
        if (memo && memo->stride < words) memo = nullptr;
        bool explored = memo && memo->asked_avg.load(std::memory_order_relaxed) < 1024u;
        int exact_rounds = 0, explore_rounds = 0;
        uint32_t apply_m = 0;
        if (!memo && !tl_ask_known.empty()) {  // an asking walk handed this query over (filtered_ask): its verdicts are the first round's input
            const uint32_t m = (uint32_t)std::min<size_t>(tl_ask_known.size(), cap);
            for (uint32_t i = 0; i < m; ++i) {
                h_list[i] = tl_ask_known[i] >> 1;
                h_verdict[i] = (uint8_t)(tl_ask_known[i] & 1u);
                n_allowed += tl_ask_known[i] & 1u;
            }
            n_known = m;
            apply_m = m;
            explored = true;
        }
        tl_ask_known.clear();
        for (int round = 0; round < 24; ++round) {
            uint32_t guess = 0;  // of 256
            if (n_known > 0 && guess_pct > 0) guess = std::min<uint32_t>(255u, (uint32_t)((uint64_t)256 * n_allowed * (uint64_t)guess_pct / (100 * n_known)));
            else if (sel_hint > 0 && guess_pct > 0) guess = std::min<uint32_t>(255u, (uint32_t)((uint64_t)256 * sel_hint * (uint64_t)guess_pct / (100ull * 65536ull)));
            const bool explore = !explored && guess_pct > 0;
            uint32_t budget;
            if (explore) {
                budget = (guess ? cap / 2 : first_budget) | (guess << 24);
                explored = guess != 0 || ++explore_rounds >= 2;
            } else {
                budget = (uint32_t)std::min<size_t>(cap, (size_t)first_budget << exact_rounds);
                if (n_known > 0) budget |= guess << 24;
                ++exact_rounds;
            }
            PipeQuery pq{};
            pq.query = h_q;
            pq.allow = d_bits;
            pq.known = d_bits + words;
            pq.words = (uint32_t)words;
            pq.zero_bits = round == 0 ? (memo ? 2u : 1u) : 0u;
            pq.list = h_list;
            pq.verdict = h_verdict;
            pq.apply_m = apply_m;
            pq.slots = (uint32_t)n;
            pq.cap = cap;
            pq.budget = budget;
            pq.k = (uint32_t)k;
            pq.round_id = ++w->round_seq ? w->round_seq : ++w->round_seq;
            pq.cnt = h_cnt;
            pq.keys = h_k;
            pq.space = (char*)w->ws.p;
            pq.memo = memo ? memo->bits : nullptr;
            pq.memo_stride = memo ? memo->stride : 0u;
            __atomic_store_n(h_done, 0u, __ATOMIC_RELEASE);
            PodRelease pod{device, pod_submit(explore ? 2 : 1, ef, lay, pq)};
            if (!pod.t) {
                batched_no_pod.fetch_add(1, std::memory_order_relaxed);
                submit_round(pq, explore, ef, lay);
            }
            // wait for the kernel's flag (the two kinds of round take different times: one moving average each)
            static std::atomic<int> waiting{0};
            {
            Stopwatch wait_sw(c_filtered_wait_ns);
            for (int attempt = 0;; ++attempt) {
                bool lost = false;
                uint32_t looks = 0;
                if (!wait_for_device_flag(
                        [&] {
                            if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == pq.round_id) return true;
                            if (pod.t && (++looks & 1023u) == 0u && pod_pool(device).lost_post(pod.t)) lost = __atomic_load_n(h_done, __ATOMIC_ACQUIRE) != pq.round_id;
                            return lost;
                        },
                        waiting, wait_typical_us[explore ? 2 : 1], 20.0)) {
                    w.retire();
                    fail(VS_ERR_DEVICE, "a batched filtered round did not finish");
                }
                pod.done();
                if (!lost || attempt) break;
                submit_round(pq, explore, ef, lay);  // the workgroup had left before it saw the post: the round goes into a batched launch
            }
            }
            uint32_t count = h_cnt[0], consulted = h_cnt[1], found = h_cnt[2];
            if (found == kPipeRedoFound && explore) {
                // (an exploring round keeps no order, so nothing it meets hands it over -- but if one ever is: the exact walk lists what it needs)
                explored = true;
                apply_m = 0;
                continue;
            }
            if (found == kPipeRedoFound) {
                // Two equal distances met where their order matters (one filtered walk in 15 at 10M x 768: thousands of candidates wait in
                // `next` together): THIS round is walked again in usearch's order -- the team form of the usearch-order walk, on the caller's
                // own stream, with the same device-resident verdicts --, and the query goes on from its outcome.
                batched_second_chances.fetch_add(1, std::memory_order_relaxed);
                hipStream_t st = w->stream;
                uint64_t* d_k = (uint64_t*)w->b.ensure(k * 8);
                float* d_d = (float*)w->c.ensure(k * 4);
                uint32_t* d_f = (uint32_t*)w->d.ensure(64);
                uint32_t* d_unknown = (uint32_t*)w->f.ensure(((size_t)cap + 64) * 4);  // [count, consulted, 62 pad | list]
                HIP_OK(hipMemsetAsync(d_unknown, 0, 8, st));
                LazyFilter lf;
                lf.known = d_bits + words;
                lf.unknown_list = d_unknown + 64;
                lf.unknown_count = d_unknown;
                lf.cap = cap;
                lf.budget = budget;
                lf.consulted = d_unknown + 1;
                {
                    struct NoPipe {
                        NoPipe() { tl_no_pipe = true; }
                        ~NoPipe() { tl_no_pipe = false; }
                    } no_pipe_here;
                    search_device(h_q, 1, k, d_k, d_d, d_f, st, 0, d_bits, 0, &lf);
                }
                HIP_OK(launch_export_round(d_unknown, cap, d_k, d_d, d_f, (uint32_t)k, h_cnt, h_list, h_k, h_d, st));
                if (!w->ev) HIP_OK(hipEventCreateWithFlags(&w->ev, hipEventDisableTiming));
                HIP_OK(hipEventRecord(w->ev, st));
                static std::atomic<int> waiting_sc{0};
                if (!wait_for_device_flag(
                        [&] {
                            const hipError_t e = hipEventQuery(w->ev);
                            if (e != hipSuccess && e != hipErrorNotReady) HIP_OK(e);
                            return e == hipSuccess;
                        },
                        waiting_sc, wait_typical_us[3], 60.0))
                    fail(VS_ERR_DEVICE, "a filtered round's second walk did not finish");
                count = h_cnt[0];
                consulted = h_cnt[1];
                found = h_cnt[2];
                if (found == kPipeRedoFound) return (size_t)-1;  // (cannot happen: the usearch-order walk answers or fails)
            }
            if (count == 0 && !explore) {
                if (std::getenv("VS_HNSW_ASK_DEBUG"))  // (profile builds: the phases of a hop of the exact walk, as filtered_ask prints them)
                    std::fprintf(stderr, "[exact] hops %u ticks %u | clk/16: head %u decide %u pop %u entry %u atomics %u verdicts %u pushes %u schedule %u wait_entry %u\n", h_cnt[9],
                                 h_cnt[4], h_cnt[7], h_cnt[1], h_cnt[10], h_cnt[11], h_cnt[12], h_cnt[13], h_cnt[14], h_cnt[15], h_cnt[6]);
                if (found == kWalkFailed) return (size_t)-1;
                std::memcpy(keys, h_k, (size_t)std::min<size_t>(found, k) * 8);
                std::memcpy(dist, h_d, (size_t)std::min<size_t>(found, k) * 4);
                lazy_rounds += (uint64_t)round + 1;
                if (memo) {
                    // (clipped: ONE query in an unvisited neighbourhood must not send the filter's next dozen queries back to exploring rounds)
                    const uint32_t asked = (uint32_t)std::min<uint64_t>(n_known, 2048u), avg = memo->asked_avg.load(std::memory_order_relaxed);
                    memo->asked_avg.store(avg == 0xFFFFFFFFu ? asked : (uint32_t)(((uint64_t)avg * 3 + asked) / 4), std::memory_order_relaxed);
                    memo->queries.fetch_add(1, std::memory_order_relaxed);
                    memo->asked.fetch_add(n_known, std::memory_order_relaxed);
                    memo_queries.fetch_add(1, std::memory_order_relaxed);
                    memo_asked.fetch_add(n_known, std::memory_order_relaxed);
                }
                lazy_need_hint = hint ? (3 * hint + consulted) / 4 : consulted;
                if (n_known) {
                    const uint32_t sel = (uint32_t)std::min<uint64_t>(65536, n_allowed * 65536ull / n_known);
                    lazy_sel_hint = sel_hint ? (3 * sel_hint + sel) / 4 : sel;
                }
                return found;
            }
            Stopwatch pred_sw(c_filtered_pred_ns);
            const uint32_t m = std::min(count, cap);
            uint64_t asked = 0;  // (counted here, added once: every caller's every call on one shared counter is a cache line passed round 50M times a second)
            for (uint32_t i = 0; i < m; ++i) {
                if (i + 8 < m && h_list[i + 8] < n) __builtin_prefetch(&h_keys[h_list[i + 8]]);
                const uint32_t s = h_list[i];
                uint8_t v = 0;
                if (s < n) {
                    const uint64_t key = h_keys[s];
                    ++asked;
                    v = key != kFreeKey && pred(key, pctx) ? 1 : 0;
                }
                h_verdict[i] = v;
                n_allowed += v;
            }
            lazy_predicate_calls += asked;
            n_known += m;
            apply_m = m;
        }
        return (size_t)-1;
    }
    // ---- an opaque predicate, ONE walk (round 6) -----------------------------------------------------------------------------------------
    // The trait's own signature (usearch.rs:224-248: a closure, no name) gives the engine nothing to remember: every query asks afresh.
    // Rounds 3-5 did that in ROUNDS (an exploring walk that lists the verdicts the exact walk will need, the answers, the exact walk
    // from scratch: 2.2 walks, twice the CPU's predicate calls).  Here the query is posted ONCE, to a pod, as a walk that asks while it
    // runs (pipe_device.hpp, "asks"): this thread -- it would only be waiting -- answers: the kernel's courier wave publishes the number
    // of asks in h_cnt[5] and their slots in h_list, one verdict byte per ask goes back (1 rejected, 2 admitted).  A member is asked
    // about when it is new to the walk's visited set and passes the radius test, i.e. when usearch would ask, once.
    // (size_t)-1: not served here (no pod free, the walk handed over at an order-relevant tie, the device gave up on a host that did not
    // answer) -- the rounds serve the query.
    static inline thread_local std::vector<uint32_t> tl_ask_known;  // (slot << 1 | admitted) of a walk that handed over, for the rounds that follow on this thread
    std::atomic<uint64_t> ask_queries{0}, ask_handed_over{0}, ask_no_pod{0}, ask_calls{0}, ask_waits{0}, ask_wait_ticks{0}, ask_hops{0}, ask_walk_ticks{0};
    size_t filtered_ask(const float* q, size_t k, vs_hnsw_predicate pred, void* pctx, uint64_t* keys, float* dist, uint32_t ef) {
        use_device();
        housekeeping();
        const size_t n = slots_atomic.load(std::memory_order_acquire), lay = layout_slots();
        const size_t lay_words = (lay + 31) / 32;
        const uint32_t cap = 1u << 17;
        Lease w(device);
        const size_t space = batch_space_bytes(lay);
        if (w->ws.bytes < space || w->ws_zeroed != lay_words) {
            char* p = (char*)w->ws.ensure(space);
            HIP_OK(hipMemsetAsync(p, 0, w->ws.bytes, w->stream));
            HIP_OK(hipStreamSynchronize(w->stream));
            w->ws_zeroed = lay_words;
        }
        const size_t pin_need = 64 + (size_t)cap * 5 + k * 12 + 64 + (size_t)dim * 4;
        if (w->pin_bytes < pin_need) {
            if (w->pin) graveyard().bury(nullptr, w->pin);
            w->pin = nullptr;
            w->pin_bytes = 0;
            w->ask_dirty = cap;
            HIP_OK(hipHostMalloc((void**)&w->pin, pin_need, hipHostMallocDefault));
            w->pin_bytes = pin_need;
        }
        uint32_t* h_cnt = (uint32_t*)w->pin;
        uint32_t* h_done = (uint32_t*)w->pin + 8;
        uint32_t* h_list = (uint32_t*)(w->pin + 64);
        uint8_t* h_verdict = (uint8_t*)(w->pin + 64 + (size_t)cap * 4);
        uint64_t* h_k = (uint64_t*)(w->pin + 64 + (size_t)cap * 5);
        float* h_d = (float*)(h_k + k);
        float* h_q = (float*)(w->pin + ((64 + (size_t)cap * 5 + k * 12 + 63) & ~(size_t)63));
        std::memcpy(h_q, q, (size_t)dim * 4);
        // (the verdict bytes double as "answered" flags: whatever an earlier query of this context -- of either path -- left there goes)
        std::memset(h_verdict, 0, std::min<size_t>(cap, w->ask_dirty));
        w->ask_dirty = cap;  // (until this query's own count is known)
        for (int i = 0; i < 16; ++i)
            if (i != 8) h_cnt[i] = 0u;
        PipeQuery pq{};
        pq.query = h_q;
        pq.allow = nullptr;
        pq.known = nullptr;
        pq.words = 0u;
        pq.zero_bits = 0u;
        pq.list = h_list;
        pq.verdict = h_verdict;
        pq.apply_m = 0u;
        pq.slots = (uint32_t)n;
        pq.cap = cap;
        pq.budget = cap;
        pq.k = (uint32_t)k;
        pq.round_id = ++w->round_seq ? w->round_seq : ++w->round_seq;
        pq.cnt = h_cnt;
        pq.keys = h_k;
        pq.space = (char*)w->ws.p;
        pq.memo = nullptr;
        pq.memo_stride = 0u;
        __atomic_store_n(h_done, 0u, __ATOMIC_RELEASE);
        PodRelease pod{device, pod_submit(3, ef, lay, pq)};
        if (!pod.t) {
            w->ask_dirty = 0;
            ask_no_pod.fetch_add(1, std::memory_order_relaxed);
            return (size_t)-1;
        }
        static std::atomic<int> waiting{0};
        struct Count {
            explicit Count() { waiting.fetch_add(1, std::memory_order_relaxed); }
            ~Count() { waiting.fetch_sub(1, std::memory_order_relaxed); }
        } count;
        const int cores = usable_cores();
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t answered = 0;
        uint64_t calls = 0;
        bool lost = false, timed_out = false;
        static const long crowd_sleep_ns = std::getenv("VS_HNSW_ASK_SLEEP_US") ? 1000L * std::atol(std::getenv("VS_HNSW_ASK_SLEEP_US")) : 30000L;
        bool slack_cut = false;
        long slack_was = 0;
        struct SlackBack {
            bool& cut;
            long& was;
            ~SlackBack() {
                if (cut) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)(was > 0 ? was : 50000), 0, 0, 0);
            }
        } slack_back{slack_cut, slack_was};
        for (uint32_t it = 0;; ++it) {
            const uint32_t asked = __atomic_load_n(h_cnt + 5, __ATOMIC_ACQUIRE);
            if (asked > answered) {
                for (uint32_t i = answered; i < asked && i < cap; ++i) {
                    const uint32_t s = h_list[i];
                    uint8_t v = 1;
                    if (s < n) {
                        const uint64_t key = h_keys[s];
                        if (key != kFreeKey) {
                            ++calls;
                            v = pred(key, pctx) ? 2 : 1;
                        }
                    }
                    __atomic_store_n(h_verdict + i, v, __ATOMIC_RELEASE);  // (in order: the courier takes the answered PREFIX)
                }
                answered = asked;
                it = 0;
                continue;
            }
            if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == pq.round_id) break;
            // A caller per core spins.  A crowd beyond the cores SLEEPS between looks, a few tens of microseconds at a time (its timer
            // slack cut to a microsecond for the length of the call): the walk does not wait for an answer until its 64 pending lanes are
            // full -- twenty-odd hops, a hundred microseconds -- so a caller that looks every 40-50 us keeps its walk going, and 128 of
            // them cost two cores.  (sched_yield does not do this: spinning callers burn whole time slices while the one whose walk
            // waits is runnable and not running -- measured with 64 / 128 callers on 16 cores: 1.4k / 1.2k queries/s, p99 93 ms.)
            if (waiting.load(std::memory_order_relaxed) <= cores) {
                for (int p = 0; p < 4; ++p) __builtin_ia32_pause();
            } else {
                if (!slack_cut) {
                    slack_was = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
                    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
                    slack_cut = true;
                }
                struct timespec ts = {0, crowd_sleep_ns};
                nanosleep(&ts, nullptr);
            }
            if ((it & 1023u) == 1023u) {
                if (pod_pool(device).lost_post(pod.t) && __atomic_load_n(h_done, __ATOMIC_ACQUIRE) != pq.round_id && __atomic_load_n(h_cnt + 5, __ATOMIC_ACQUIRE) == 0u) {
                    lost = true;  // the workgroup had left before it saw the post
                    break;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
                    timed_out = true;
                    break;
                }
            }
        }
        w->ask_dirty = std::min<uint32_t>(cap, answered);
        lazy_predicate_calls += calls;
        ask_calls.fetch_add(calls, std::memory_order_relaxed);
        if (timed_out) {
            w.retire();
            fail(VS_ERR_DEVICE, "a filtered walk that asks did not finish");
        }
        pod.done();
        if (lost) return (size_t)-1;
        const uint32_t found = h_cnt[2];
        if (found == kPipeRedoFound || found == kWalkFailed) {
            ask_handed_over.fetch_add(1, std::memory_order_relaxed);
            // what the walk learned goes with the query: the rounds that serve it start from these verdicts (no exploring round, no second
            // call of the predicate for them)
            tl_ask_known.clear();
            for (uint32_t i = 0; i < answered && i < cap; ++i)
                if (h_list[i] < n) tl_ask_known.push_back((h_list[i] << 1) | (h_verdict[i] == 2 ? 1u : 0u));  // (slots are below 2^30)
            return (size_t)-1;
        }
        std::memcpy(keys, h_k, (size_t)std::min<size_t>(found, k) * 8);
        std::memcpy(dist, h_d, (size_t)std::min<size_t>(found, k) * 4);
        ask_queries.fetch_add(1, std::memory_order_relaxed);
        ask_waits.fetch_add(h_cnt[6], std::memory_order_relaxed);
        ask_wait_ticks.fetch_add(h_cnt[7], std::memory_order_relaxed);
        ask_hops.fetch_add(h_cnt[9], std::memory_order_relaxed);
        ask_walk_ticks.fetch_add(h_cnt[4], std::memory_order_relaxed);
        if (std::getenv("VS_HNSW_ASK_DEBUG"))  // (profile builds: the phases of a hop, see pipe_device.hpp)
            std::fprintf(stderr, "[ask] hops %u ticks %u | clk/16: head %u decide %u pop %u entry %u atomics %u verdicts %u pushes %u schedule %u wait_entry %u\n", h_cnt[9], h_cnt[4],
                         h_cnt[7], h_cnt[1], h_cnt[10], h_cnt[11], h_cnt[12], h_cnt[13], h_cnt[14], h_cnt[15], h_cnt[6]);
        lazy_rounds += 1;
        return found;
    }
    static inline std::atomic<int> filtered_active_callers{0};

    static constexpr size_t kLazyFilterAbove = 1u << 16;
    size_t filtered_lazy(const float* q, size_t k, vs_hnsw_predicate pred, void* pctx, uint64_t* keys, float* dist, uint64_t filter_key = 0) {
        struct ActiveCaller {
            ActiveCaller() { filtered_active_callers.fetch_add(1, std::memory_order_relaxed); }
            ~ActiveCaller() { filtered_active_callers.fetch_sub(1, std::memory_order_relaxed); }
        } active_caller;
        {
            // Rounds through the batcher when the callers outnumber the device's streams by a margin AND this index's filtered walks are
            // short: a batch holds its stream until its slowest walk ends, so a crowd of 10 %-selective queries (6 ms a round) gains --
            // 64 callers 1.77k queries/s against 0.93k one stream each -- while 1 %-selective ones (100 ms a round) keep every stream busy
            // and the next launch waiting (17 callers: 28 against 62), and a handful of callers is served best one stream each (17
            // callers at 10 %: 945 against 850).  VS_HNSW_FILTER_BATCH = 0 never, 1 whenever the pipelined walk can serve.
            static const int batch_env = std::getenv("VS_HNSW_FILTER_BATCH") ? std::atoi(std::getenv("VS_HNSW_FILTER_BATCH")) : -1;
            uint32_t ef_b = 0;
            check_search(k, ef_b);
            const bool crowd = filtered_active_callers.load(std::memory_order_relaxed) > device_streams(device).count + 8;
            // (what the recent filtered queries of this index needed: few verdicts -- or, once the share their filters admit is known, a share
            // of 3 % or more: the count of verdicts of a 1 % filter hovers around the threshold, its selectivity does not)
            const uint32_t sel_now = lazy_sel_hint.load();
            const bool short_walks = lazy_need_hint.load() != 0 && (sel_now ? sel_now >= 1966u : lazy_need_hint.load() < 20000u);
            // With pods (pipe_pod.hpp) a round is posted to a resident workgroup -- no launch, no stream to wait for -- whatever the crowd.
            // (Filters whose walks are long -- 1 % selective: 100 ms a round, `next` spilling to global memory -- take pods when several
            // callers are at it: measured at 10M x 768, 1 / 17 / 64 / 128 callers: 6.3 / 90 / 310 / 597 queries/s through pods against
            // 9 / 77 / 66 / 66 on rounds of their own.)
            const bool several = filtered_active_callers.load(std::memory_order_relaxed) > 8;
            const bool pods = pod_pool(device).enabled;
            // (a NAMED filter always takes the posted / batched rounds: that is where its remembered verdicts are used)
            // An unnamed filter -- the trait's own signature -- takes ONE walk that asks while it runs (round 6) whenever pods can serve;
            // VS_HNSW_FILTER_ASK=0: the rounds of rounds 3-5 (A/B).
            static const int ask_env = std::getenv("VS_HNSW_FILTER_ASK") ? std::atoi(std::getenv("VS_HNSW_FILTER_ASK")) : 1;
            // (every asking walk needs its caller to look in every few tens of microseconds for its whole length -- filtered_ask: callers
            // beyond the cores sleep between looks, 64 / 128 callers on 16 cores: 6.1k / 12.0k queries/s against the rounds' 3.8k / 7.1k;
            // a crowd of more than eight callers per core is left to the rounds)
            // callers per core the asking walks serve (VS_HNSW_ASK_CROWD; beyond: the rounds, whose callers sleep through two long waits)
            static const int ask_crowd = std::getenv("VS_HNSW_ASK_CROWD") ? std::max(1, std::atoi(std::getenv("VS_HNSW_ASK_CROWD"))) : 8;
            const bool awake = ask_env == 2 || filtered_active_callers.load(std::memory_order_relaxed) <= ask_crowd * usable_cores();
            if (!filter_key && ask_env != 0 && awake && pods && batch_env != 0 && pipe_usable(ef_b) && !needs_global_walk_beyond_pipe(ef_b)) {
                tl_ask_known.clear();
                const size_t f = filtered_ask(q, k, pred, pctx, keys, dist, ef_b);
                if (f != (size_t)-1) return f;
            }
            std::shared_ptr<FilterMemo> memo = (filter_key && batch_env != 0 && pipe_usable(ef_b) && !needs_global_walk_beyond_pipe(ef_b)) ? memo_for(filter_key) : nullptr;
            if (batch_env != 0 && pipe_usable(ef_b) && !needs_global_walk_beyond_pipe(ef_b) && (batch_env == 1 || memo || (short_walks && (pods || crowd)) || (pods && several))) {
                const size_t f = filtered_batched(q, k, pred, pctx, keys, dist, ef_b, memo.get());
                (f != (size_t)-1 ? batched_done : batched_handed_over).fetch_add(1, std::memory_order_relaxed);
                if (f != (size_t)-1) return f;
                // (two equal distances met where their order matters, or the launch failed: the query starts over on rounds of its own,
                // whose second-chance launch is the usearch-order walk)
            }
        }
        use_device();
        size_t n;
        {
            std::lock_guard<std::mutex> g(mod_mu);
            n = slots;
        }
        const size_t words = (n + 31) / 32;
        const uint32_t cap = 1u << 17;
        // Admission: a lazily filtered query is a chain of small launches on its context's stream, and contexts beyond the device's
        // streams share one -- their walks then take turns on it while each still holds a caller (64 callers on 16 streams measured
        // 505 queries/s where 17 got 717).  At most one query per stream is let through; the others queue here, in arrival order
        // as the condition variable wakes them, and lose nothing: the device is as busy as it gets.
        struct Gate {
            std::mutex mu;
            std::condition_variable cv;
            int free_permits = -1;
        };
        static Gate gates[16];
        Gate& gate = gates[device & 15];
        {
            std::unique_lock<std::mutex> gl(gate.mu);
            if (gate.free_permits < 0) {
                static const int gate_env = std::getenv("VS_HNSW_FILTER_GATE") ? std::atoi(std::getenv("VS_HNSW_FILTER_GATE")) : 0;  // 0: one per stream
                gate.free_permits = gate_env > 0 ? gate_env : device_streams(device).count;
            }
            gate.cv.wait(gl, [&] { return gate.free_permits > 0; });
            --gate.free_permits;
        }
        struct Permit {
            Gate& g;
            ~Permit() {
                {
                    std::lock_guard<std::mutex> gl(g.mu);
                    ++g.free_permits;
                }
                g.cv.notify_one();
            }
        } permit{gate};
        Lease w(device);  // (taken behind the gate: the contexts in use -- each with its own stream -- are as many as the permits)
        hipStream_t st = w->stream;
        uint64_t* d_k = (uint64_t*)w->b.ensure(k * 8);
        float* d_d = (float*)w->c.ensure(k * 4);
        uint32_t* d_f = (uint32_t*)w->d.ensure(64);
        uint32_t* d_bits = (uint32_t*)w->e.ensure(words * 8);                           // [allow | known], on the device for the whole query
        uint32_t* d_unknown = (uint32_t*)w->f.ensure(((size_t)cap + 64) * 4);  // [count, consulted, 62 pad | list]
        // pinned, device-mapped: [counters 64 B | list cap x 4 | verdicts cap | keys k x 8 | dist k x 4 | the query]: the device reads the
        // query and the verdicts from it and writes each round's outcome into it -- no copy engine on this path
        const size_t pin_need = 64 + (size_t)cap * 5 + k * 12 + 64 + (size_t)dim * 4;
        if (w->pin_bytes < pin_need) {
            if (w->pin) graveyard().bury(nullptr, w->pin);
            w->pin = nullptr;
            w->pin_bytes = 0;
            HIP_OK(hipHostMalloc((void**)&w->pin, pin_need, hipHostMallocDefault));
            w->pin_bytes = pin_need;
        }
        uint32_t* h_cnt = (uint32_t*)w->pin;
        uint32_t* h_list = (uint32_t*)(w->pin + 64);
        uint8_t* h_verdict = (uint8_t*)(w->pin + 64 + (size_t)cap * 4);
        w->ask_dirty = cap;
        uint64_t* h_k = (uint64_t*)(w->pin + 64 + (size_t)cap * 5);
        float* h_d = (float*)(h_k + k);
        float* h_q = (float*)(w->pin + ((64 + (size_t)cap * 5 + k * 12 + 63) & ~(size_t)63));
        std::memcpy(h_q, q, (size_t)dim * 4);
        HIP_OK(hipMemsetAsync(d_bits, 0, words * 8, st));
        HIP_OK(hipMemsetAsync(d_unknown, 0, 8, st));
        // Waiting for a round: hipStreamSynchronize polls, which is the fastest wake-up while every caller has a core; with
        // more filtered calls in flight than cores (the reference runs each on a spawn_blocking thread, usearch.rs:937-948:
        // as many as there are requests) polling callers starve the ones that have verdicts to compute, so the crowd sleeps
        // in 100 us steps on the round's event instead (measured with 64 callers on 16 cores: 70 QPS polling, 17 callers: 241).
        static std::atomic<int> filtered_active{0};
        struct Active {
            std::atomic<int>& n;
            explicit Active(std::atomic<int>& c) : n(c) { n.fetch_add(1, std::memory_order_relaxed); }
            ~Active() { n.fetch_sub(1, std::memory_order_relaxed); }
        } active(filtered_active);

        if (!w->ev) HIP_OK(hipEventCreateWithFlags(&w->ev, hipEventDisableTiming));
        auto wait_round = [&] {
            static const int cores = (int)std::max(1u, std::thread::hardware_concurrency());
            if (filtered_active.load(std::memory_order_relaxed) < cores) {
                HIP_OK(hipStreamSynchronize(st));
                return;
            }
            HIP_OK(hipEventRecord(w->ev, st));
            for (;;) {
                const hipError_t e = hipEventQuery(w->ev);
                if (e == hipSuccess) break;
                if (e != hipErrorNotReady) HIP_OK(e);
                std::this_thread::sleep_for(std::chrono::microseconds(100));
            }
        };
        const uint32_t hint = lazy_need_hint.load();
        const uint32_t sel_hint = lazy_sel_hint.load();  // selectivity of recent filtered queries, in 1 / 65,536
        const uint32_t first_budget = std::max<uint32_t>(2048u, std::min<uint32_t>(cap / 2, hint + hint / 2));
        uint64_t n_known = 0, n_allowed = 0;  // verdicts of this query so far
        // Rounds, round 4: on float indexes the pipelined walk (kernels_pipe.hip) first EXPLORES -- several candidates at a time, no
        // order kept, verdicts it misses guessed at half the selectivity seen so far (of this query, else of recent queries of the
        // index) and listed --, the host answers the list, and the next round is the exact walk, which as a rule meets no slot
        // without a verdict.  An index that has seen no filtered query yet has no selectivity to guess with: its first exploring
        // round takes every unknown slot as rejected and stops after first_budget of them; the second one guesses.
        uint32_t ef_now = 0;
        check_search(k, ef_now);
        const bool can_explore = pipe_usable(ef_now);
        bool explored = false;  // an exploring round that could guess has run (or two that could not)
        int exact_rounds = 0, explore_rounds = 0;
        for (int round = 0; round < 24; ++round) {
            LazyFilter lf;
            lf.known = d_bits + words;
            lf.unknown_list = d_unknown + 64;
            lf.unknown_count = d_unknown;
            lf.cap = cap;
            lf.consulted = d_unknown + 1;  // (a pad word of the count block)
            static const int guess_pct = std::getenv("VS_HNSW_FILTER_GUESS") ? std::atoi(std::getenv("VS_HNSW_FILTER_GUESS")) : 50;  // % of the observed selectivity; 0 = off
            static const bool guess_first = std::getenv("VS_HNSW_FILTER_GUESS_FIRST") && std::getenv("VS_HNSW_FILTER_GUESS_FIRST")[0] == '1';  // experiment
            static const bool explore_off = std::getenv("VS_HNSW_FILTER_EXPLORE") && std::getenv("VS_HNSW_FILTER_EXPLORE")[0] == '0';  // A/B measurements
            uint32_t guess = 0;  // of 256
            if (n_known > 0 && guess_pct > 0) guess = std::min<uint32_t>(255u, (uint32_t)((uint64_t)256 * n_allowed * (uint64_t)guess_pct / (100 * n_known)));
            else if (sel_hint > 0 && guess_pct > 0) guess = std::min<uint32_t>(255u, (uint32_t)((uint64_t)256 * sel_hint * (uint64_t)guess_pct / (100ull * 65536ull)));
            lf.explore = can_explore && !explore_off && !explored && guess_pct > 0;
            if (lf.explore) {
                lf.budget = guess ? cap / 2 : first_budget;
                lf.budget |= guess << 24;
                explored = guess != 0 || ++explore_rounds >= 2;
            } else {
                // The first exact round's budget follows the number of verdicts the exact walks of recent filtered queries of this index
                // consulted (+ 50 %): filters of one workload tend to be alike.  It doubles from there.
                lf.budget = (uint32_t)std::min<size_t>(cap, (size_t)first_budget << exact_rounds);
                // a round that meets unknown slots after all guesses them at half the selectivity seen so far (walk_device.hpp `guess_t`)
                if (n_known > 0) {
                    lf.budget |= guess << 24;
                } else if (round == 0 && guess_first && guess) {
                    lf.budget = (uint32_t)std::min<size_t>(cap, std::max<size_t>(lf.budget, 3 * (size_t)hint));
                    lf.budget |= guess << 24;
                }
                ++exact_rounds;
            }
            search_device(h_q, 1, k, d_k, d_d, d_f, st, 0, d_bits, 0, &lf);
            // counters, answer and list reach the pinned block by a kernel: one wait per round
            HIP_OK(launch_export_round(d_unknown, cap, d_k, d_d, d_f, (uint32_t)k, h_cnt, h_list, h_k, h_d, st));
            wait_round();
            const uint32_t count = h_cnt[0], consulted = h_cnt[1], found = h_cnt[2];
            if (count == 0 && !lf.explore) {
                if (found == kWalkFailed) return (size_t)-1;
                std::memcpy(keys, h_k, (size_t)std::min<size_t>(found, k) * 8);
                std::memcpy(dist, h_d, (size_t)std::min<size_t>(found, k) * 4);
                lazy_rounds += (uint64_t)round + 1;
                // what the exact walk (this last round) consulted, smoothed over the recent queries of this index
                lazy_need_hint = hint ? (3 * hint + consulted) / 4 : consulted;
                if (n_known) {
                    const uint32_t sel = (uint32_t)std::min<uint64_t>(65536, n_allowed * 65536ull / n_known);
                    lazy_sel_hint = sel_hint ? (3 * sel_hint + sel) / 4 : sel;
                }
                return found;
            }
            if (count == 0) continue;  // (an exploring round that found every verdict it wanted)
            const uint32_t m = std::min(count, cap);
            // a walk evaluates a node once, so a list names a slot once, and slots with a verdict are never listed again
            uint64_t asked = 0;  // (counted here, added once: every caller's every call on one shared counter is a cache line passed round 50M times a second)
            for (uint32_t i = 0; i < m; ++i) {
                if (i + 8 < m && h_list[i + 8] < n) __builtin_prefetch(&h_keys[h_list[i + 8]]);
                const uint32_t s = h_list[i];
                uint8_t v = 0;
                if (s < n) {
                    const uint64_t key = h_keys[s];
                    ++asked;
                    v = key != kFreeKey && pred(key, pctx) ? 1 : 0;
                }
                h_verdict[i] = v;
                n_allowed += v;
            }
            lazy_predicate_calls += asked;
            n_known += m;
            HIP_OK(launch_apply_verdicts(d_unknown, h_verdict, m, (uint32_t)n, d_bits, d_bits + words, st));
        }
        return (size_t)-1;
    }

    size_t rank_all(const float* q, size_t k, uint64_t* keys, float* dist) {  // exhaustive ranking, no predicate
        struct All {
            static int yes(uint64_t, void*) { return 1; }
        };
        return filtered(q, k, &All::yes, nullptr, keys, dist, true);
    }

    size_t filtered(const float* q, size_t k, vs_hnsw_predicate pred, void* pctx, uint64_t* keys, float* dist, bool exhaustive = false, uint64_t filter_key = 0) {
        const size_t n_live = live.load();
        if (!n_live) return 0;
        if (!exhaustive && std::max<size_t>(k, ef_search.load()) <= kMaxWalkBeam) {
            if (slots > kLazyFilterAbove && !eager_filter) {
                const size_t f = filtered_lazy(q, k, pred, pctx, keys, dist, filter_key);
                if (f != (size_t)-1) return f;
            }
            const std::vector<uint32_t> bits = allow_bitmap(pred, pctx);
            size_t f = 0;
            search_host(q, 1, k, keys, dist, &f, false, &bits);
            if (f != (size_t)-1) return f;
        }
        // Walk the members in ascending (distance, slot) order and ask the predicate lazily: about k / selectivity calls
        // instead of one per member (the reference's predicate takes a table read-lock per call).
        Ranked ranked(*this, q);
        std::vector<uint64_t> kk;
        std::vector<float> dd;
        size_t out = 0, from = 0, chunk = std::max<size_t>(4096, 2 * k);
        while (out < k && from < ranked.n) {
            const size_t m = std::min(chunk, ranked.n - from);
            kk.resize(m);
            dd.resize(m);
            ranked.fetch(from, m, kk.data(), dd.data());
            bool end = false;
            for (size_t i = 0; i < m && out < k; ++i) {
                if (kk[i] == kFreeKey) {  // removed members sort last: nothing live beyond this point
                    end = true;
                    break;
                }
                if (pred(kk[i], pctx)) {
                    keys[out] = kk[i];
                    dist[out] = dd[i];
                    ++out;
                }
            }
            if (end) break;
            from += m;
            chunk = std::min<size_t>(chunk * 4, 1u << 20);
        }
        return out;
    }
};



}  // namespace vs

#include "engine_service.hpp"
#include "engine_abi.hpp"
