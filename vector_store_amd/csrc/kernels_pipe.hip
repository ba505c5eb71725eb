// kernels_pipe.hip -- hnsw_pipe_walk_kernel for ONE float arithmetic (-DVS_AR=0..5): the pipelined walk of pipe_device.hpp
// behind usearch::Index::search / filtered_search for LONE queries (the reference issues one query per FFI call:
// crates/vector-store/src/vs_index/usearch.rs:210-212, :233-236; filtered queries each on their own blocking thread, :937-948).
// One workgroup of kPipeTeam waves per query: wave 0 walks, the others evaluate candidates ahead of it.  A query this kernel
// cannot answer exactly as usearch orders it (two equal distances met, or `next` outgrew the pool) is flagged kPipeRedo and
// served by the usearch-order walk (kernels_walk.hip).
#include <atomic>
#include <mutex>

#include "kernels.hpp"
#include "pipe_device.hpp"
#include "pipe_pod.hpp"

#ifndef VS_AR
#error "compile with -DVS_AR=<arithmetic>"
#endif

namespace vs {

// A value every lane of the wave holds alike, moved to scalar registers (a pod's staging entry is written by the kernel itself, so its
// fields come through vector loads: left in vector registers, a dozen uniform pointers push the walker's working set into scratch).
template <class T>
__device__ __forceinline__ T uni(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two registers");
    if constexpr (sizeof(T) == 4) {
        union {
            T v;
            int w;
        } u;
        u.v = v;
        u.w = __builtin_amdgcn_readfirstlane(u.w);
        return u.v;
    } else {
        union {
            T v;
            int w[2];
        } u;
        u.v = v;
        u.w[0] = __builtin_amdgcn_readfirstlane(u.w[0]);
        u.w[1] = __builtin_amdgcn_readfirstlane(u.w[1]);
        return u.v;
    }
}

// One query, one workgroup.  Its buffers: the strided arrays of WalkArgs (query `qi`), or -- `pq` -- its entry of the batch table.
template <int AR, int I, int EFCAP, int MODE, class Sh>
__device__ __forceinline__ void pipe_query(const WalkArgs& a, const PipeQuery* pq, uint32_t qi, uint32_t ef, Sh& sh, uint2* pipe_pool, const uint32_t tid,
                                           const uint32_t bid, const uint64_t t_begin, const uint32_t entry_slot, const int32_t max_level, const bool tomb) {
    // The fields every distance's address arithmetic needs, held in scalar registers: left to itself the compiler re-reads them from the
    // kernel's argument block wherever they are used (constants to it: cheaper to load again than to keep) -- six scalar loads and as
    // many waits in the chain row -> vectors -> distance that the walker waits for (round 5: 845 -> 808 us per lone walk).
    IndexView ix = a.ix;
    asm volatile("" : "+s"(ix.vectors), "+s"(ix.aux), "+s"(ix.adj0), "+s"(ix.stride4), "+s"(ix.lanes), "+s"(ix.lanes_log2), "+s"(ix.M0));
    const int lane = (int)(tid & 63u);
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const float* query = pq ? uni(pq->query) : a.queries + (size_t)qi * a.q_stride;
    const uint32_t k = pq ? uni(pq->k) : a.k;
    uint64_t* ok = pq ? uni(pq->keys) : a.out_keys + (size_t)qi * a.k;
    float* od = pq ? reinterpret_cast<float*>(ok + k) : a.out_dist + (size_t)qi * a.k;  // (a posted query's distances follow its keys)
    uint32_t* found_out = pq ? uni(pq->cnt) + 2 : a.out_found + qi;
    if (max_level < 0) {  // empty index
        if (w == 0) {
            for (uint32_t i = lane; i < k; i += kWave) {
                ok[i] = kFreeKey;
                od[i] = __builtin_inff();
            }
            if (lane == 0) {
                *found_out = 0;
                if (pq) {
                    uni(pq->cnt)[0] = uni(pq->cnt)[1] = uni(pq->cnt)[3] = 0u;
                    __threadfence_system();
                    __hip_atomic_store(uni(pq->cnt) + 8, uni(pq->round_id), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        return;
    }
    if (tid < (uint32_t)kPipeTeam) {
        sh.job_state[tid] = 0u;
        if (tid == 0) {
            sh.stop = 0u;
            sh.aq_tail = 0u;
            sh.courier_done = 0u;
            sh.tw_req = sh.tw_done = 0u;
            for (int i = 0; i < 8; ++i) sh.prof_jobs[i] = 0u;
        }
    }
    if (tid < (uint32_t)kPipeCache) sh.c_ready[tid] = 0u;
    WalkSpace ws = {nullptr, nullptr, nullptr, 0u, 0u, 0u};
    {
        char* base = pq ? uni(pq->space) : a.space + (size_t)bid * a.space_stride;
        ws.bitmap = reinterpret_cast<uint32_t*>(base);
        ws.vlog = ws.bitmap + a.bitmap_words;
        ws.heap = reinterpret_cast<uint2*>(ws.vlog + a.vlog_cap);
        ws.bitmap_words = a.bitmap_words;
        ws.vlog_cap = a.vlog_cap;
        ws.heap_cap = a.heap_cap;
    }
    const uint32_t* allow = pq ? uni(pq->allow) : a.allow ? a.allow + (size_t)qi * a.allow_stride : nullptr;
    const uint32_t* known = pq ? uni(pq->known) : a.known ? a.known + (size_t)qi * a.allow_stride : nullptr;
    if (pq) {
        // The round's exchange with the host, first half: the verdicts it gave for the slots the last round listed become bits of the
        // query's device-resident bitmaps (the first round zeroes them instead).  Every wave takes part; the barriers of the descent
        // that follows order it before the first verdict is read.
        uint32_t* allow_w = uni(pq->allow);
        uint32_t* known_w = uni(pq->known);
        uint32_t* memo = uni(pq->memo);  // verdicts remembered across the queries of one filter: [allow | known], memo_stride words each
        const uint32_t mstride = uni(pq->memo_stride);
        if (uni(pq->zero_bits) == 2u && memo) {
            // seeded from the filter's memory.  Other queries of the filter add verdicts meanwhile (allow first, then known): `known`
            // is read first, and `allow` behind a fence -- a known bit seen here has its verdict in the allow word read after it
            for (uint32_t i = tid; i < uni(pq->words); i += 64u * kPipeTeam) known_w[i] = __hip_atomic_load(&memo[mstride + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            __syncthreads();
            for (uint32_t i = tid; i < uni(pq->words); i += 64u * kPipeTeam) allow_w[i] = __hip_atomic_load(&memo[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (uni(pq->zero_bits)) {
            for (uint32_t i = tid; i < uni(pq->words); i += 64u * kPipeTeam) {
                allow_w[i] = 0u;
                known_w[i] = 0u;
            }
        }
        const uint32_t m = uni(pq->apply_m);
        if (m && uni(pq->zero_bits) == 1u) {  // (round 6: a first round that starts from verdicts an asking walk left -- zeroed words before the first bit)
            __threadfence();
            __syncthreads();
        }
        for (uint32_t i = tid; i < m; i += 64u * kPipeTeam) {
            const uint32_t s = uni(pq->list)[i];
            if (s < uni(pq->slots)) {
                atomicOr(&known_w[s >> 5], 1u << (s & 31u));
                if (uni(pq->verdict)[i]) {
                    atomicOr(&allow_w[s >> 5], 1u << (s & 31u));
                    if (memo) atomicOr(&memo[s >> 5], 1u << (s & 31u));
                }
            }
        }
        __threadfence();
        __syncthreads();
        if (memo && m) {  // ... and the filter's memory learns them: `known` only once every `allow` bit of the list is out
            for (uint32_t i = tid; i < m; i += 64u * kPipeTeam) {
                const uint32_t s = uni(pq->list)[i];
                if (s < uni(pq->slots)) atomicOr(&memo[mstride + (s >> 5)], 1u << (s & 31u));
            }
        }
    }
    Query<AR, I> q;
    query_from_f32<AR, I>(ix, query, q, lane);
    if (w != 0) {
        team_helper_loop<AR, I>(ix, q, sh, lane, w);  // the descent through the upper levels: the team form, with barriers
#ifdef VS_PIPE_SOLO
        if ((w & 3u) == 0u) {  // (experiment: the walker has its SIMD to itself)
            if (lane == 0) sh.job_state[w] = 1u;
            return;
        }
#endif
        if constexpr (MODE == kPipePlain) {
            if (w == kPipeTopWave) {  // plain walks: this wave keeps `top` (pipe_device.hpp, "the top wave")
                pipe_top_loop<EFCAP / 64>(sh, ef, a.pipe_fused_order != 0u, lane);
                return;
            }
        }
        if constexpr (MODE == kPipeAsk) {
            if (w == kPipeCourierWave) {  // asks: this wave carries the walker's questions to the caller and the answers back (pipe_device.hpp)
                pipe_courier_loop(sh, uni(pq->list), uni(pq->verdict), uni(pq->cnt), lane);
                return;
            }
        }
        pipe_helper_loop<AR, I>(ix, q, sh, ws, tomb, allow, known, lane, w);
        return;
    }
    Counters cnt = {0, 0, 0};
    float start_d = 0.f;
    uint32_t start;
    // (s_setprio 3 for the walker was measured in round 5: 925 -> 920 us per lone walk, 17 callers 17.5-17.9k -> 17.6-18.1k queries/s --
    // within the noise: the walker is bound by its own instruction count, one issue per four clocks, not by the helpers on its SIMD)
    // (inlined whatever the size of the kernel: an out-of-line call passes the query's registers through scratch memory)
    [[clang::always_inline]] start = greedy_descent<AR, I>(ix, sh, q, entry_slot, max_level, 0, cnt, lane, &start_d);
    team_release(sh, lane);  // the last barrier: from here on the waves meet through LDS words only
    PipeTop<EFCAP / 64> top;
    const PipeOut r = pipe_walk<AR, I, MODE>(ix, sh, pipe_pool, a.pipe_pool_cap, ws, start, start_d, ef, tomb, allow, known,
                                       pq ? uni(pq->list) : a.unknown_list ? a.unknown_list + (size_t)qi * a.unknown_cap : nullptr,
                                       pq ? uni(pq->cnt) : a.unknown_count ? a.unknown_count + qi : nullptr, pq ? uni(pq->cap) : a.unknown_cap,
                                       pq ? uni(pq->budget) : a.unknown_budget, pq ? uni(pq->cnt) + 1 : a.consulted ? a.consulted + qi : nullptr, cnt, lane, top,
                                       a.debug ? a.debug + (size_t)qi * 12 : nullptr, a.pipe_fused_order != 0u);
    if constexpr (MODE == kPipeAsk) {  // (the walk has set sh.stop: the courier leaves within one look)
        for (uint32_t spins = 0; lds_load_acquire(&sh.courier_done) == 0u && spins < (1u << 24); ++spins) __builtin_amdgcn_s_sleep(1);
    }
    if (r.status == 1u) {  // the usearch-order walk answers it (and lists the verdicts IT misses: this walk's list is dropped)
        if (lane == 0) {
            if (a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1u)] = qi;
            *found_out = kPipeRedo;
            if (pq) {
                uni(pq->cnt)[0] = uni(pq->cnt)[1] = 0u;
                __threadfence_system();
                __hip_atomic_store(uni(pq->cnt) + 8, uni(pq->round_id), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                if (a.unknown_count) a.unknown_count[qi] = 0u;
                if (a.consulted) a.consulted[qi] = 0u;
            }
        }
        return;
    }
    // top.shrink(wanted): `top` holds admitted, live members only (a round that ran out of budget reports what it has; the host
    // looks at the listed slots first and walks again)
    constexpr uint32_t R = EFCAP / 64;
    const uint32_t found = r.sz < k ? r.sz : k;
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) {
        const uint32_t pos = (uint32_t)lane * R + j;
        if (pos < found) {
            ok[pos] = ix.keys[top.s[j]];
            od[pos] = top.d[j];
        }
    }
    for (uint32_t i = found + (uint32_t)lane; i < k; i += kWave) {
        ok[i] = kFreeKey;
        od[i] = __builtin_inff();
    }
    if (lane == 0) {
        *found_out = found;
        atomicAdd(&a.stats[ST_SEARCH_EVALS], cnt.evals);
        atomicAdd(&a.stats[ST_SEARCH_HOPS], cnt.hops);
        atomicAdd(&a.stats[ST_QUERIES], 1ull);
        if (pq) {
            uni(pq->cnt)[3] = (uint32_t)cnt.evals;
            uni(pq->cnt)[4] = (uint32_t)(wall_clock64() - t_begin);  // 100 MHz ticks from the moment the query was seen (a launch: from its start)
        }
    }
    if (pq) {
        // second half of the exchange: the answer, the counters and the list are in the caller's pinned block -- every lane's stores are
        // out (the list's as well: pipe_walk drained them) before the flag says so
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(uni(pq->cnt) + 8, uni(pq->round_id), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// `slots` == nullptr: one query per workgroup, then done.  Else a POD (pipe_pod.hpp): workgroup b serves slot b -- a line of pinned host
// memory a caller posts its query to -- until the host closes the pod; a.pipe_qtable is then the pod's staging table in device memory
// (entry b: this workgroup's copy of the posted PipeQuery, which the walk reads exactly as it reads a batch table's entry).
constexpr int kPipeBoth = 3;  // (kernel instances only: a pod whose workgroups serve exact filtered walks and exploring rounds alike)

struct PipeKernArgs {
    WalkArgs a;
    PodSlot* slots;
    PodCtl* ctl;
};

template <int AR, int I, int EFCAP, int MODE>
__global__ __launch_bounds__(64 * kPipeTeam) void hnsw_pipe_walk_kernel(PipeKernArgs ka_unused) {
    constexpr bool VISG = true;
    using Sh = PipeShared<EFCAP, kPipeTeam, false, VISG>;
    __shared__ Sh sh;
    extern __shared__ uint2 pipe_pool[];
    __shared__ uint32_t pod_cmd[6];
    uint32_t seen = 0;
    for (;;) {
        // The kernel's arguments, read from the argument segment INSIDE the loop (the offset is opaque to the compiler): hoisted out of
        // it, every argument and every address derived from one stays live through the whole walk -- 150 more scalar registers than
        // there are, spilled through vector registers into scratch memory, and a walk twice as slow.
        uint32_t zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        const PipeKernArgs& ka = *reinterpret_cast<const PipeKernArgs*>((const char*)__builtin_amdgcn_kernarg_segment_ptr() + zero);
        const WalkArgs& a = ka.a;
        PodSlot* const slots = ka.slots;
        PodCtl* const ctl = ka.ctl;
        // (the same for everything derived from the thread's and the workgroup's number: lane masks, offsets, predicates)
        uint32_t tid = threadIdx.x, bid = blockIdx.x;
        asm volatile("" : "+v"(tid));
        asm volatile("" : "+s"(bid));
        const PipeQuery* pq = a.pipe_qtable ? a.pipe_qtable + bid : nullptr;
        uint32_t qi = bid, ef = a.ef;
        uint64_t t_begin = wall_clock64();
        [[maybe_unused]] uint32_t explore = 0u;  // a pod of filtered queries: 0 the exact walk, 1 an exploring round, 2 the asking walk (round 6)
        // what adds and removes change: a launch carries it in its arguments, a pod reads it per query (pipe_pod.hpp: PodCtl)
        uint32_t entry_slot = a.ix.entry_slot;
        int32_t max_level = a.ix.max_level;
        bool tomb = a.has_removed != 0;
        if (!slots) {
            if (a.qlist) {  // second-chance launches name their queries
                if (qi >= *a.qcount) return;
                qi = a.qlist[qi];
            }
        } else {
            PodSlot* slot = slots + bid;
            if (tid == 0) {  // one thread polls the slot; the other waves sleep at the barrier
                uint32_t p = 0, polls = 0, beat = 0;
                uint64_t beat_at = 0;
                for (;;) {
                    // (relaxed: an acquire at system scope invalidates the caches, and every idle workgroup looks every few microseconds)
                    p = __hip_atomic_load(&slot->posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (p != seen) break;
                    bool leave = (polls & 7u) == 0u && __hip_atomic_load(&ctl->closed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
                    if (!leave && (polls & 255u) == 0u) {  // the host's heartbeat: a pod nobody looks after any more (2 s at 100 MHz) ends by itself
                        const uint32_t b = __hip_atomic_load(&ctl->heartbeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        const uint64_t now = wall_clock64();
                        if (b != beat || beat_at == 0) {
                            beat = b;
                            beat_at = now;
                        } else if (now - beat_at > 200000000ull) {
                            leave = true;
                        }
                    }
                    if (leave) {
                        // (a query posted before the pod was closed is still answered: the host posts and closes under one lock)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                        p = __hip_atomic_load(&slot->posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (p == seen) {
                            p = 0u;
                            __hip_atomic_store(&slot->left, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        break;
                    }
                    ++polls;
                    // ~1.5 us between looks while a caller is between two rounds, ~8 us once the slot has been idle for a while
                    __builtin_amdgcn_s_sleep(60);
                    if (polls > 256u) {
                        __builtin_amdgcn_s_sleep(127);
                        __builtin_amdgcn_s_sleep(127);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the query's line, and whatever the caller wrote for it, after its number
                // ... and the SCALAR cache: the compiler turns wave-uniform loads of the index (upper_off[cur], a level's block) into
                // scalar loads, whose cache no fence invalidates -- a launch starts with it clean, a pod that survives an add (round 5)
                // must clean it itself or it walks with a stale upper_off of a slot that has been filled since
                __builtin_amdgcn_s_dcache_inv();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                pod_cmd[0] = p;
                pod_cmd[1] = __hip_atomic_load(&slot->ef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                pod_cmd[2] = __hip_atomic_load(&slot->explore, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                pod_cmd[3] = __hip_atomic_load(&ctl->entry_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                pod_cmd[4] = (uint32_t)__hip_atomic_load(&ctl->max_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                pod_cmd[5] = __hip_atomic_load(&ctl->has_removed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            // the posted query: read once, past the caches (the host rewrites the line between queries), into this workgroup's entry of
            // the staging table
            __syncthreads();
            const uint32_t p = pod_cmd[0];
            if (p == 0u) return;  // (every thread reads the same word: the workgroup leaves together)
            seen = p;
            t_begin = wall_clock64();
            ef = (uint32_t)__builtin_amdgcn_readfirstlane((int)pod_cmd[1]);
            explore = (uint32_t)__builtin_amdgcn_readfirstlane((int)pod_cmd[2]);
            entry_slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)pod_cmd[3]);
            max_level = (int32_t)__builtin_amdgcn_readfirstlane((int)pod_cmd[4]);
            tomb = __builtin_amdgcn_readfirstlane((int)pod_cmd[5]) != 0;
            if (tid < sizeof(PipeQuery) / 4) {
                const uint32_t v = __hip_atomic_load(reinterpret_cast<uint32_t*>(&slot->q) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                reinterpret_cast<uint32_t*>(const_cast<PipeQuery*>(pq))[tid] = v;
            }
            __threadfence();
            __syncthreads();
        }
        if constexpr (MODE == kPipeBoth) {
            // a pod of filtered queries: the two kinds of round alternate for every caller, so one workgroup serves either
            if (explore) pipe_query<AR, I, EFCAP, kPipeExplore>(a, pq, qi, ef, sh, pipe_pool, tid, bid, t_begin, entry_slot, max_level, tomb);
            else pipe_query<AR, I, EFCAP, kPipeFiltered>(a, pq, qi, ef, sh, pipe_pool, tid, bid, t_begin, entry_slot, max_level, tomb);
        } else {
            pipe_query<AR, I, EFCAP, MODE>(a, pq, qi, ef, sh, pipe_pool, tid, bid, t_begin, entry_slot, max_level, tomb);
        }
        if (!slots) return;
        __syncthreads();  // every wave is done with this query's LDS before the next one's is laid out
    }
}

template <int AR, int I, int EFCAP, int MODE>
static hipError_t pipe_launch(const WalkArgs& a, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    auto kernel = hnsw_pipe_walk_kernel<AR, I, EFCAP, MODE>;
    const size_t dyn = (size_t)a.pipe_pool_cap * sizeof(uint2);
    // the attribute is per device; a failed attempt is tried again by the next launch (a once_flag would be spent on it, and every later
    // launch would fail at the launch itself with an unrelated error: advisor finding, round 4)
    static std::atomic<int> attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 15].load(std::memory_order_acquire)) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (attr != hipSuccess) return attr;
        attr_set[dev & 15].store(1, std::memory_order_release);
    }
    PipeKernArgs ka{a, slots, ctl};
    hipLaunchKernelGGL(kernel, dim3(a.nq), dim3(64 * kPipeTeam), dyn, s, ka);
    return hipGetLastError();
}

template <int AR, int I>
static hipError_t pipe_ef(const WalkArgs& a, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    // (the LDS tag table as the visited set -- PipeShared<..., VISG = false> -- was measured for unfiltered lone walks and dropped: its
    // test-and-set costs the walker as much as the returning global atomic, which runs under the early post: 0.80 against 0.74 ms)
    if (a.pipe_lds_visited) return hipErrorInvalidValue;
    // one instance per purpose (pipe_device.hpp `MODE`): plain lone queries, the exact walk of a filtered query, its exploring rounds
    if (a.ef > 512) return hipErrorInvalidValue;
    // (round 6: a pod of walks that ask while they run is a kernel of its own -- as a third body of the kPipeBoth kernel its walker's
    // scalar registers spilled, 311 spill instructions against 38-59 in the one-purpose kernels, and every phase of a hop ran twice as slow)
    if (slots && a.pipe_explore == 2u) return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipeAsk>(a, s, slots, ctl) : pipe_launch<AR, I, 512, kPipeAsk>(a, s, slots, ctl);
    if (slots && !a.pipe_fused_order) return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipeBoth>(a, s, slots, ctl) : pipe_launch<AR, I, 512, kPipeBoth>(a, s, slots, ctl);
    if (a.pipe_explore) return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipeExplore>(a, s, slots, ctl) : pipe_launch<AR, I, 512, kPipeExplore>(a, s, slots, ctl);
    // (a pod of plain queries has a staging table too: pipe_fused_order tells it from a pod of filtered ones)
    if (a.allow || (a.pipe_qtable && !(slots && a.pipe_fused_order))) return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipeFiltered>(a, s, slots, ctl) : pipe_launch<AR, I, 512, kPipeFiltered>(a, s, slots, ctl);
    return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipePlain>(a, s, slots, ctl) : pipe_launch<AR, I, 512, kPipePlain>(a, s, slots, ctl);
}

template <>
hipError_t launch_pipe_pod_ar<VS_AR>(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    if (!a.nq) return hipSuccess;
    if (a.ix.M0 > 64u || (!a.space && !a.pipe_qtable) || a.pipe_pool_cap < 256u || (slots && (!ctl || !a.pipe_qtable))) return hipErrorInvalidValue;
    switch (iters) {
#ifndef VS_PIPE_DEV
        case 1: return pipe_ef<VS_AR, 1>(a, s, slots, ctl);
        case 2: return pipe_ef<VS_AR, 2>(a, s, slots, ctl);
        case 3: return pipe_ef<VS_AR, 3>(a, s, slots, ctl);
        case 4: return pipe_ef<VS_AR, 4>(a, s, slots, ctl);
        case 8: return pipe_ef<VS_AR, 8>(a, s, slots, ctl);
#endif
        case 6: return pipe_ef<VS_AR, 6>(a, s, slots, ctl);
        default: return hipErrorInvalidValue;
    }
}

template <>
hipError_t launch_pipe_walk_ar<VS_AR>(const WalkArgs& a, uint32_t iters, hipStream_t s) {
    return launch_pipe_pod_ar<VS_AR>(a, iters, s, nullptr, nullptr);
}

}  // namespace vs
