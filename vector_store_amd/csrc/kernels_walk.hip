// kernels_walk.hip -- hnsw_walk_kernel for ONE arithmetic (-DVS_AR=0..7): the usearch-order search
// (walk_device.hpp) behind usearch::Index::search / filtered_search as the reference calls them
// (crates/vector-store/src/vs_index/usearch.rs:210-212, :233-236).
//
// Instances per arithmetic and row layout:
//   LDS visited table  (EFCAP 128 / 256 / 512)     -- tie-heavy metrics (i8, Hamming) and "usearch order" indexes at the
//                                                      usual beams; a query whose table or heap runs out is handed to
//   global visited bitmap (EFCAP 512 / 2048 / 10240) -- the retry instance; filtered search (allow-bitmap tested at admission
//                                                      to `top`, rejected nodes still expanded); beams of 513..10,240
//                                                      (the reference passes any `limit` through, httproutes.rs:842-847);
//                                                      indexes beyond what the LDS tags can tell apart.
// Workgroups are persistent: each owns one WalkSpace and draws queries from a shared counter until none is left.
#include <atomic>

#include "kernels.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "walk_device.hpp"
#include "pipe_pod.hpp"

#ifndef VS_AR
#error "compile with -DVS_AR=<arithmetic>"
#endif

namespace vs {

#ifndef VS_WALK_LCAP
#define VS_WALK_LCAP 512
#endif
constexpr int kWalkPodHeapLds = VS_WALK_LCAP;           // (= kWalkHeapLds below)

template <int AR, int I, int EFCAP, int LCAP, int NB, int CH, bool VISG, int BS = 8>
__global__ __launch_bounds__(64) void hnsw_walk_kernel(WalkArgs a) {
    using Sh = WalkShared<EFCAP, LCAP, NB, CH, VISG, 1, false, false, BS>;
    __shared__ Sh sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    WalkSpace ws;
    {
        char* base = a.space + (size_t)blockIdx.x * a.space_stride;
        ws.bitmap = reinterpret_cast<uint32_t*>(base);
        ws.vlog = ws.bitmap + a.bitmap_words;
        ws.heap = reinterpret_cast<uint2*>(ws.vlog + a.vlog_cap);
        ws.bitmap_words = a.bitmap_words;
        ws.vlog_cap = a.vlog_cap;
        ws.heap_cap = a.heap_cap;
    }
    const uint32_t total = a.qlist ? *a.qcount : a.nq;
    if (ix.max_level < 0) {  // empty index: nothing to walk
        for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
            const uint32_t qi = a.qlist ? a.qlist[t] : t;
            for (uint32_t i = lane; i < a.k; i += kWave) {
                a.out_keys[(size_t)qi * a.k + i] = kFreeKey;
                a.out_dist[(size_t)qi * a.k + i] = __builtin_inff();
            }
            if (lane == 0) a.out_found[qi] = 0;
        }
        return;
    }
    __shared__ uint32_t next_query;
    for (;;) {
        // The next query comes from a shared counter; it reaches the wave through LDS between two barriers.  (Lane 0's
        // atomic followed by v_readfirstlane is NOT enough: on a path without a barrier the compiler peeled the other 63
        // lanes into a loop of their own that never re-ran the atomic -- an endless loop on an empty index.)
        if (lane == 0) next_query = atomicAdd(a.work_counter, 1u);
        __syncthreads();
        const uint32_t t = next_query;
        __syncthreads();
        if (t >= total) break;
        const uint32_t qi = a.qlist ? a.qlist[t] : t;
        uint64_t* ok = a.out_keys + (size_t)qi * a.k;
        float* od = a.out_dist + (size_t)qi * a.k;
        Query<AR, I> q;
        query_from_f32<AR, I>(ix, a.queries + (size_t)qi * a.q_stride, q, lane);
        Counters cnt = {0, 0, 0};
        const uint32_t start = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, 0, cnt, lane);
        bool exhausted = false;
        const uint32_t* allow = a.allow ? a.allow + (size_t)qi * a.allow_stride : nullptr;
        const uint32_t sz = walk_usearch<AR, I>(ix, sh, ws, q, start, 0, a.ef, kInvalid, a.has_removed != 0, allow, cnt, lane, exhausted,
                                         a.debug ? a.debug + (size_t)qi * 12 : nullptr,
                                         a.known ? a.known + (size_t)qi * a.allow_stride : nullptr,
                                         a.unknown_list ? a.unknown_list + (size_t)qi * a.unknown_cap : nullptr,
                                         a.unknown_count ? a.unknown_count + qi : nullptr, a.unknown_cap, a.unknown_budget,
                                         a.consulted ? a.consulted + qi : nullptr);
        if (exhausted) {
            if (lane == 0) {
                if (a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1u)] = qi;  // the global-bitmap instance takes it
                else a.out_found[qi] = kWalkFailed;                                // workspace too small: the host ranks exhaustively
            }
            __syncthreads();
            continue;
        }
        // top.shrink(wanted): `top` holds admitted, live members only
        const uint32_t found = sz < a.k ? sz : a.k;
        for (uint32_t i = lane; i < a.k; i += kWave) {
            const bool in = i < found;
            ok[i] = in ? ix.keys[sh.lst_s[i]] : kFreeKey;
            od[i] = in ? sh.lst_d[i] : __builtin_inff();
        }
        if (lane == 0) {
            a.out_found[qi] = found;
            atomicAdd(&a.stats[ST_SEARCH_EVALS], cnt.evals);
            atomicAdd(&a.stats[ST_SEARCH_HOPS], cnt.hops);
            atomicAdd(&a.stats[ST_QUERIES], 1ull);
        }
        __syncthreads();
    }
}

// (Round 6, measured and not kept: this loop with TWO waves per walk -- the walker and the heap wave of the walk pods, walk_device.hpp.  Both
// roles are one kernel, so a workgroup's waves take the larger register count of the two: 179 VGPRs for i8 rows of 768 B (one-wave walk:
// 152), 153 for b1 (119) -- two / three waves per SIMD, i.e. 4 / 6 walks per CU where LDS would hold 7.  i8 batch 845k -> 377k queries/s,
// b1 574k -> 575k.  The pods, one workgroup to a CU, have the registers to spare.)
// Small batches (lone callers: the reference issues one query per FFI call): one workgroup of TEAM waves per query, as
// hnsw_search_kernel does -- wave 0 walks exactly as above, every wave evaluates its share of each hop's neighbours, so a
// lone walk has TEAM times the row loads in flight.  Same decisions, same ids.  LDS instances only; a query that outgrows
// its structures goes to the retry list like any other.
template <int AR, int I, int EFCAP, int LCAP, int NB, int CH, int TEAM>
__global__ __launch_bounds__(64 * TEAM) void hnsw_walk_team_kernel(WalkArgs a) {
    using Sh = WalkShared<EFCAP, LCAP, NB, CH, false, TEAM>;
    __shared__ Sh sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t qi = blockIdx.x, w = threadIdx.x >> 6;
    uint64_t* ok = a.out_keys + (size_t)qi * a.k;
    float* od = a.out_dist + (size_t)qi * a.k;
    if (ix.max_level < 0) {  // empty index
        if (w == 0) {
            for (uint32_t i = lane; i < a.k; i += kWave) {
                ok[i] = kFreeKey;
                od[i] = __builtin_inff();
            }
            if (lane == 0) a.out_found[qi] = 0;
        }
        return;
    }
    Query<AR, I> q;
    query_from_f32<AR, I>(ix, a.queries + (size_t)qi * a.q_stride, q, lane);
    if (w != 0) {
        team_helper_loop<AR, I>(ix, q, sh, lane, w);
        return;
    }
    WalkSpace ws = {nullptr, nullptr, nullptr, 0u, 0u, 0u};  // LDS instance: nothing lives in global memory
    Counters cnt = {0, 0, 0};
    const uint32_t start = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, 0, cnt, lane);
    bool exhausted = false;
    const uint32_t sz = walk_usearch<AR, I>(ix, sh, ws, q, start, 0, a.ef, kInvalid, a.has_removed != 0, nullptr, cnt, lane, exhausted,
                                     a.debug ? a.debug + (size_t)qi * 12 : nullptr);
    team_release(sh, lane);
    wsync<Sh>();
    if (exhausted) {
        if (lane == 0) {
            if (a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1u)] = qi;
            else a.out_found[qi] = kWalkFailed;
        }
        return;
    }
    const uint32_t found = sz < a.k ? sz : a.k;
    for (uint32_t i = lane; i < a.k; i += kWave) {
        const bool in = i < found;
        ok[i] = in ? ix.keys[sh.lst_s[i]] : kFreeKey;
        od[i] = in ? sh.lst_d[i] : __builtin_inff();
    }
    if (lane == 0) {
        a.out_found[qi] = found;
        atomicAdd(&a.stats[ST_SEARCH_EVALS], cnt.evals);
        atomicAdd(&a.stats[ST_SEARCH_HOPS], cnt.hops);
        atomicAdd(&a.stats[ST_QUERIES], 1ull);
    }
}

// The global-bitmap walk (filtered search, beams and indexes beyond the LDS tables) for a lone query: the same team of waves.
// Wave 0 walks with the visited bitmap, the visited log and the deep levels of `next` in its workspace (a.space, one per
// workgroup) and consults the predicate's verdicts exactly as the one-wave kernel does; the helpers only measure distances.
template <int AR, int I, int EFCAP, int LCAP, int TEAM>
__global__ __launch_bounds__(64 * TEAM) void hnsw_walk_team_global_kernel(WalkArgs a) {
    using Sh = WalkShared<EFCAP, LCAP, 256, 1, true, TEAM>;
    __shared__ Sh sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    uint32_t qi = blockIdx.x;
    const uint32_t w = threadIdx.x >> 6;
    if (a.qlist) {  // second chance behind the pipelined walk (kernels_pipe.hip): only the queries it handed over
        if (qi >= *a.qcount) return;
        qi = a.qlist[qi];
    }
    uint64_t* ok = a.out_keys + (size_t)qi * a.k;
    float* od = a.out_dist + (size_t)qi * a.k;
    if (ix.max_level < 0) {  // empty index
        if (w == 0) {
            for (uint32_t i = lane; i < a.k; i += kWave) {
                ok[i] = kFreeKey;
                od[i] = __builtin_inff();
            }
            if (lane == 0) a.out_found[qi] = 0;
        }
        return;
    }
    Query<AR, I> q;
    query_from_f32<AR, I>(ix, a.queries + (size_t)qi * a.q_stride, q, lane);
    if (w != 0) {
        team_helper_loop<AR, I>(ix, q, sh, lane, w);
        return;
    }
    WalkSpace ws;
    {
        char* base = a.space + (size_t)blockIdx.x * a.space_stride;
        ws.bitmap = reinterpret_cast<uint32_t*>(base);
        ws.vlog = ws.bitmap + a.bitmap_words;
        ws.heap = reinterpret_cast<uint2*>(ws.vlog + a.vlog_cap);
        ws.bitmap_words = a.bitmap_words;
        ws.vlog_cap = a.vlog_cap;
        ws.heap_cap = a.heap_cap;
    }
    Counters cnt = {0, 0, 0};
    const uint32_t start = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, 0, cnt, lane);
    bool exhausted = false;
    const uint32_t* allow = a.allow ? a.allow + (size_t)qi * a.allow_stride : nullptr;
    const uint32_t sz = walk_usearch<AR, I>(ix, sh, ws, q, start, 0, a.ef, kInvalid, a.has_removed != 0, allow, cnt, lane, exhausted,
                                     a.debug ? a.debug + (size_t)qi * 12 : nullptr,
                                     a.known ? a.known + (size_t)qi * a.allow_stride : nullptr,
                                     a.unknown_list ? a.unknown_list + (size_t)qi * a.unknown_cap : nullptr,
                                     a.unknown_count ? a.unknown_count + qi : nullptr, a.unknown_cap, a.unknown_budget,
                                     a.consulted ? a.consulted + qi : nullptr);
    team_release(sh, lane);
    wsync<Sh>();
    if (exhausted) {
        if (lane == 0) a.out_found[qi] = kWalkFailed;  // workspace too small: the host ranks exhaustively
        return;
    }
    const uint32_t found = sz < a.k ? sz : a.k;
    for (uint32_t i = lane; i < a.k; i += kWave) {
        const bool in = i < found;
        ok[i] = in ? ix.keys[sh.lst_s[i]] : kFreeKey;
        od[i] = in ? sh.lst_d[i] : __builtin_inff();
    }
    if (lane == 0) {
        a.out_found[qi] = found;
        atomicAdd(&a.stats[ST_SEARCH_EVALS], cnt.evals);
        atomicAdd(&a.stats[ST_SEARCH_HOPS], cnt.hops);
        atomicAdd(&a.stats[ST_QUERIES], 1ull);
    }
}

template <int AR, int I, int EFCAP, int LCAP>
static hipError_t walk_team_global_launch(const WalkArgs& a, hipStream_t s, uint32_t* grid_out) {
    if (grid_out) {
        *grid_out = a.nq ? a.nq : 1;
        return hipSuccess;
    }
    if (!a.nq) return hipSuccess;
    hipLaunchKernelGGL((hnsw_walk_team_global_kernel<AR, I, EFCAP, LCAP, kSearchTeam>), dim3(a.nq), dim3(64 * kSearchTeam), 0, s, a);
    return hipGetLastError();
}

template <int AR, int I, int EFCAP, int LCAP, int NB, int CH>
static hipError_t walk_team_launch(const WalkArgs& a, hipStream_t s, uint32_t* grid_out) {
    if (grid_out) {
        *grid_out = a.nq ? a.nq : 1;
        return hipSuccess;
    }
    if (!a.nq || a.qlist || a.allow) return a.nq ? hipErrorInvalidValue : hipSuccess;
    hipLaunchKernelGGL((hnsw_walk_team_kernel<AR, I, EFCAP, LCAP, NB, CH, kSearchTeam>), dim3(a.nq), dim3(64 * kSearchTeam), 0, s, a);
    return hipGetLastError();
}

// ---- walk PODS (round 6): the team form above as a resident kernel that callers post lone queries to (pipe_pod.hpp: the same slots,
// the same control block, the same waiter on the host).  For the storage whose ties the pipelined walk cannot order -- b1: a few hundred
// distinct Hamming distances, nearly every pipelined walk handed over -- a lone query was a launch of its own
// through the single-query dispatcher: 17 blocking callers 5.8-9.8k queries/s at 10M x 768.  Workgroup b serves slot b: one thread polls
// the slot, the other waves wait at the barrier; the posted PipeQuery names the query (pinned), the beam, the caller's pinned block for
// keys / distances / counters / flag.  Entry point, top level and "has removed members" are read per query from the pod's control
// block, so the pod survives adds and removes like the others.  Same walk, same ids and distance bits as the launch.
struct WalkPodArgs {
    WalkArgs a;
    PodSlot* slots;
    PodCtl* ctl;
};
// HWV: the second wave of a two-wave team owns `next` (walk_device.hpp, HeapWaveBox) from the end of the descent to the end of the walk.
template <int AR, int I, int EFCAP, int LCAP, int NB, int CH, int TEAM, bool HWV = false>
__global__ __launch_bounds__(64 * TEAM) void hnsw_walk_team_pod_kernel(WalkPodArgs ka) {
    using Sh = WalkShared<EFCAP, LCAP, NB, CH, false, TEAM, false, false, 8, HWV>;
    __shared__ Sh sh;
    __shared__ uint32_t pod_cmd[6];
    __shared__ PipeQuery pq_s;
    const WalkArgs& a = ka.a;
    IndexView ix = a.ix;
    const int lane = lane_id();
    const uint32_t tid = threadIdx.x, w = tid >> 6;
    PodSlot* const slot = ka.slots + blockIdx.x;
    PodCtl* const ctl = ka.ctl;
    uint32_t seen = 0;
    for (;;) {
        if (tid == 0) {  // (the polling loop of hnsw_pipe_walk_kernel)
            uint32_t p = 0, polls = 0, beat = 0;
            uint64_t beat_at = 0;
            for (;;) {
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
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                    p = __hip_atomic_load(&slot->posted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (p == seen) {
                        p = 0u;
                        __hip_atomic_store(&slot->left, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    break;
                }
                ++polls;
                __builtin_amdgcn_s_sleep(60);
                if (polls > 256u) {
                    __builtin_amdgcn_s_sleep(127);
                    __builtin_amdgcn_s_sleep(127);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            __builtin_amdgcn_s_dcache_inv();  // (scalar loads of the index survive no fence: a pod that outlives an add must clean the scalar cache itself)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pod_cmd[0] = p;
            pod_cmd[1] = __hip_atomic_load(&slot->ef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pod_cmd[3] = __hip_atomic_load(&ctl->entry_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pod_cmd[4] = (uint32_t)__hip_atomic_load(&ctl->max_level, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pod_cmd[5] = __hip_atomic_load(&ctl->has_removed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        const uint32_t p = pod_cmd[0];
        if (p == 0u) return;  // (every thread reads the same word: the workgroup leaves together)
        seen = p;
        const uint64_t t_begin = wall_clock64();
        const uint32_t ef = pod_cmd[1];
        ix.entry_slot = pod_cmd[3];
        ix.max_level = (int32_t)pod_cmd[4];
        const bool tomb = pod_cmd[5] != 0u;
        if (tid < sizeof(PipeQuery) / 4) reinterpret_cast<uint32_t*>(&pq_s)[tid] = __hip_atomic_load(reinterpret_cast<uint32_t*>(&slot->q) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        const uint32_t k = pq_s.k;
        uint64_t* const ok = pq_s.keys;
        float* const od = reinterpret_cast<float*>(ok + k);  // (a posted query's distances follow its keys)
        uint32_t* const cnt_out = pq_s.cnt;
        uint32_t found = 0;
        Counters cnt = {0, 0, 0};
        if (ix.max_level >= 0) {
            Query<AR, I> q;
            query_from_f32<AR, I>(ix, pq_s.query, q, lane);
            bool helper = false;
            if constexpr (TEAM > 1) {
                helper = w != 0;
                if (helper) {
                    team_helper_loop<AR, I>(ix, q, sh, lane, w);  // (the descent through the upper levels)
                    if constexpr (HWV) walk_heap_wave_loop<AR, I>(ix, q, sh, lane);
                }
            }
            if (!helper) {
                WalkSpace ws = {nullptr, nullptr, nullptr, 0u, 0u, 0u};  // LDS instance: nothing lives in global memory
                const uint32_t start = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, 0, cnt, lane);
                bool exhausted = false;
                if constexpr (HWV) team_release(sh, lane);  // the helper becomes the heap wave; the walk itself sends it home
                const uint32_t sz = walk_usearch<AR, I>(ix, sh, ws, q, start, 0, ef, kInvalid, tomb, nullptr, cnt, lane, exhausted, nullptr);
                if constexpr (!HWV) team_release(sh, lane);
                wsync<Sh>();
                if (exhausted) {
                    found = kPipeRedoFound;  // its structures ran out: the dispatcher's launch (and its retry instance) serves the query
                } else {
                    found = sz < k ? sz : k;
                    for (uint32_t i = lane; i < k; i += kWave) {
                        const bool in = i < found;
                        ok[i] = in ? ix.keys[sh.lst_s[i]] : kFreeKey;
                        od[i] = in ? sh.lst_d[i] : __builtin_inff();
                    }
                }
            }
        } else if (w == 0) {
            for (uint32_t i = lane; i < k; i += kWave) {
                ok[i] = kFreeKey;
                od[i] = __builtin_inff();
            }
        }
        if (w == 0) {
            if (lane == 0) {
                cnt_out[0] = cnt_out[1] = 0u;
                cnt_out[2] = found;
                cnt_out[3] = (uint32_t)cnt.evals;
                cnt_out[4] = (uint32_t)(wall_clock64() - t_begin);
                if (found != kPipeRedoFound) {
                    atomicAdd(&a.stats[ST_SEARCH_EVALS], cnt.evals);
                    atomicAdd(&a.stats[ST_SEARCH_HOPS], cnt.hops);
                    atomicAdd(&a.stats[ST_QUERIES], 1ull);
                }
            }
            __threadfence_system();
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(cnt_out + 8, pq_s.round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();  // every wave is done with this query's LDS before the next one's is laid out
    }
}

template <int AR, int I>
static hipError_t walk_pod_ef(const WalkArgs& a, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    if constexpr (AR == AR_B1 && I < 12) {  // (the one storage these pods are for: one more set of instances in one of the eight files)
        if (!a.nq || !slots || !ctl) return hipErrorInvalidValue;
        WalkPodArgs ka{a, slots, ctl};
        // (VS_HNSW_B1_POD_TEAM: waves per posted query.  A b1 row is 96 bytes, a hop's 32 rows one wave's load, and a lone walk is the walker's own
        // chain of LDS round trips and scalar code -- 14.7k clocks per hop at 10M x 768, ef 200, of which the memory is a small part
        // (profiles/r06_b1_walk_phases.txt): one / two / eight waves that only SHARE THE ROWS measured 1.37 ms per walk each.  Round 6: the
        // second wave of two owns `next` instead (2, the default) -- pop_heap under the visited test, the push_heaps under the merge into `top`;
        // 1 = one wave does everything, 8 = the row-sharing team.  A pod's workgroup has a CU nearly to itself, so the visited table is the wide
        // one-choice form (4,096 buckets: one bucket read per test instead of two and an overflow list).)
        static const int team = std::getenv("VS_HNSW_B1_POD_TEAM") ? std::atoi(std::getenv("VS_HNSW_B1_POD_TEAM")) : 2;
        if (team == 1) {
            if (a.ef <= 128) hipLaunchKernelGGL((hnsw_walk_team_pod_kernel<AR, I, 128, kWalkPodHeapLds, 4096, 1, 1>), dim3(a.nq), dim3(64), 0, s, ka);
            else if (a.ef <= 256) hipLaunchKernelGGL((hnsw_walk_team_pod_kernel<AR, I, 256, 796, 4096, 1, 1>), dim3(a.nq), dim3(64), 0, s, ka);
            else return hipErrorInvalidValue;
        } else if (team == 2) {
            if (a.ef <= 128) hipLaunchKernelGGL((hnsw_walk_team_pod_kernel<AR, I, 128, kWalkPodHeapLds, 4096, 1, 2, true>), dim3(a.nq), dim3(128), 0, s, ka);
            else if (a.ef <= 256) hipLaunchKernelGGL((hnsw_walk_team_pod_kernel<AR, I, 256, 796, 4096, 1, 2, true>), dim3(a.nq), dim3(128), 0, s, ka);
            else return hipErrorInvalidValue;
        } else {
            if (a.ef <= 128) hipLaunchKernelGGL((hnsw_walk_team_pod_kernel<AR, I, 128, kWalkPodHeapLds, 1024, 1, kSearchTeam>), dim3(a.nq), dim3(64 * kSearchTeam), 0, s, ka);
            else if (a.ef <= 256) hipLaunchKernelGGL((hnsw_walk_team_pod_kernel<AR, I, 256, 796, 1024, 2, kSearchTeam>), dim3(a.nq), dim3(64 * kSearchTeam), 0, s, ka);
            else return hipErrorInvalidValue;
        }
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

template <class K>
static uint32_t resident_workgroups(K kernel, int device, int block = 64) {  // workgroups of `kernel` the chip holds at once
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    if (const char* pc = std::getenv("VS_HNSW_WALK_PER_CU")) per_cu = std::max(1, std::atoi(pc));  // residency experiments
    if (std::getenv("VS_HNSW_WALK_DEBUG")) {
        hipFuncAttributes fa{};
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kernel));
        fprintf(stderr, "[walk] occupancy: %d workgroups per CU x %d CUs (registers %d, LDS %zu B, scratch %zu B)\n", per_cu, cus, fa.numRegs,
                fa.sharedSizeBytes, fa.localSizeBytes);
    }
    return (uint32_t)per_cu * (uint32_t)cus;
}

template <int AR, int I, int EFCAP, int LCAP, int NB, int CH, bool VISG, int BS = 8>
static hipError_t walk_launch(const WalkArgs& a, uint32_t grid_cap, hipStream_t s, uint32_t* grid_out) {
    auto kernel = hnsw_walk_kernel<AR, I, EFCAP, LCAP, NB, CH, VISG, BS>;
    static std::atomic<uint32_t> resident_cache{0};  // per instance; the engine serves one device model
    uint32_t resident = resident_cache.load(std::memory_order_relaxed);
    if (!resident) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        resident = resident_workgroups(kernel, dev);
        resident_cache.store(resident, std::memory_order_relaxed);
    }
    uint32_t want = a.qlist ? grid_cap : a.nq;
    uint32_t g = want < resident ? want : resident;
    g = g < grid_cap ? g : grid_cap;
    if (grid_out) {  // sizing query only
        *grid_out = g ? g : 1;
        return hipSuccess;
    }
    if (!g) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(g), dim3(64), 0, s, a);
    return hipGetLastError();
}

#ifndef VS_WALK_LCAP
#define VS_WALK_LCAP 512
#endif
constexpr int kWalkHeapLds = VS_WALK_LCAP;  // entries of `next` in LDS (LDS instances); deeper levels live in global memory

template <int AR, int I>
static hipError_t walk_ef(const WalkArgs& a, uint32_t instance, uint32_t grid_cap, hipStream_t s, uint32_t* grid_out) {
    if constexpr (I >= 12) {  // rows of 12 / 16 KiB: a reduced set of instances (build time), each request served by the next larger one
        switch (instance & ~kWalkTeamFlag) {
            case WALK_LDS_128:
            case WALK_LDS_128_SMALL:
            case WALK_LDS_128_TINY: return walk_launch<AR, I, 128, kWalkHeapLds, 1024, 1, false>(a, grid_cap, s, grid_out);
            case WALK_LDS_256:
            case WALK_LDS_256_DENSE:
            case WALK_LDS_320:
            case WALK_LDS_512: return walk_launch<AR, I, 512, 1690, 2048, 2, false>(a, grid_cap, s, grid_out);
            case WALK_GLOBAL_512: return walk_launch<AR, I, 512, 1024, 256, 1, true>(a, grid_cap, s, grid_out);
            case WALK_GLOBAL_2048: return walk_launch<AR, I, 2048, 1024, 256, 1, true>(a, grid_cap, s, grid_out);
            case WALK_GLOBAL_10240: return walk_launch<AR, I, 10240, 1024, 256, 1, true>(a, grid_cap, s, grid_out);
            default: return hipErrorInvalidValue;
        }
    }
    switch (instance) {
        // `next` in LDS: 4 x the beam (largest heap seen at 1M x 768: 469 / 792 at beams of 128 / 256); beyond -> retry instance
        case WALK_LDS_128: return walk_launch<AR, I, 128, kWalkHeapLds, 1024, 1, false>(a, grid_cap, s, grid_out);
        // 796 entries of `next` (largest seen at a beam of 256: 792; beyond -> retry instance): 26,612 B of LDS, what still
        // fits 6 per CU (LDS is handed out in 2 KiB steps: 26 KiB x 6 = 156 KiB)
        case WALK_LDS_256: return walk_launch<AR, I, 256, 796, 1024, 2, false>(a, grid_cap, s, grid_out);
        // beams of 257..288: 990 entries of `next`, 28,664 B of LDS, 5 per CU
        case WALK_LDS_320: return walk_launch<AR, I, 320, 990, 1024, 2, false>(a, grid_cap, s, grid_out);
        // 1,690 entries of `next` (largest seen at a beam of 512: 1,341): 53,220 B of LDS, what still fits 3 per CU
        case WALK_LDS_512: return walk_launch<AR, I, 512, 1690, 2048, 2, false>(a, grid_cap, s, grid_out);
        case WALK_GLOBAL_512: return walk_launch<AR, I, 512, 1024, 256, 1, true>(a, grid_cap, s, grid_out);
        case WALK_GLOBAL_2048: return walk_launch<AR, I, 2048, 1024, 256, 1, true>(a, grid_cap, s, grid_out);
        case WALK_GLOBAL_10240: return walk_launch<AR, I, 10240, 1024, 256, 1, true>(a, grid_cap, s, grid_out);
        case WALK_LDS_128_SMALL: return walk_launch<AR, I, 128, kWalkHeapLds, 512, 2, false>(a, grid_cap, s, grid_out);
        // beams of 129..256 below 2^24 slots: 512 buckets x 12 tags (6,144 entries, two choices; a beam of 208 visits ~4,900
        // nodes, 5,500 at most) instead of 1,024 x 8: 22,004 B of LDS = 7 walks per CU instead of 6.  (`next` keeps its 796
        // entries: with 600 -- 8 walks per CU -- a quarter of the headline's queries outgrew it, p50 569 / p99 679 / max 772.)
        // A query that outgrows the table goes to the retry instance, and an index on which many do goes back to the 256
        // instance (engine.hip).
        case WALK_LDS_256_DENSE: return walk_launch<AR, I, 256, 796, 512, 2, false, 12>(a, grid_cap, s, grid_out);
        // team forms (small batches) of the instances above
        case WALK_LDS_128 | kWalkTeamFlag: return walk_team_launch<AR, I, 128, kWalkHeapLds, 1024, 1>(a, s, grid_out);
        case WALK_LDS_128_SMALL | kWalkTeamFlag: return walk_team_launch<AR, I, 128, kWalkHeapLds, 512, 2>(a, s, grid_out);
        case WALK_LDS_256 | kWalkTeamFlag: return walk_team_launch<AR, I, 256, 796, 1024, 2>(a, s, grid_out);
        case WALK_LDS_320 | kWalkTeamFlag: return walk_team_launch<AR, I, 320, 990, 1024, 2>(a, s, grid_out);
        // lone filtered queries: `next` grows to thousands of entries (rejected nodes are expanded too); a team workgroup has
        // its CU's LDS nearly to itself, so 6,144 entries of it (12 levels) stay there: 54 KB
        case WALK_GLOBAL_512 | kWalkTeamFlag: return walk_team_global_launch<AR, I, 512, 6144>(a, s, grid_out);
        case WALK_LDS_128_TINY:
            if constexpr (I == 1) return walk_launch<AR, 1, 128, kWalkHeapLds, 256, 1, false>(a, grid_cap, s, grid_out);
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

template <>
hipError_t launch_walk_pod_ar<VS_AR>(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    switch (iters) {
        case 1: return walk_pod_ef<VS_AR, 1>(a, s, slots, ctl);
        case 2: return walk_pod_ef<VS_AR, 2>(a, s, slots, ctl);
        case 3: return walk_pod_ef<VS_AR, 3>(a, s, slots, ctl);
        case 4: return walk_pod_ef<VS_AR, 4>(a, s, slots, ctl);
        case 6: return walk_pod_ef<VS_AR, 6>(a, s, slots, ctl);
        case 8: return walk_pod_ef<VS_AR, 8>(a, s, slots, ctl);
        default: return hipErrorInvalidValue;
    }
}

template <>
hipError_t launch_walk_ar<VS_AR>(const WalkArgs& a, uint32_t iters, uint32_t instance, uint32_t grid_cap, hipStream_t s,
                                 uint32_t* grid_out) {
    switch (iters) {
        case 1: return walk_ef<VS_AR, 1>(a, instance, grid_cap, s, grid_out);
        case 2: return walk_ef<VS_AR, 2>(a, instance, grid_cap, s, grid_out);
        case 3: return walk_ef<VS_AR, 3>(a, instance, grid_cap, s, grid_out);
        case 4: return walk_ef<VS_AR, 4>(a, instance, grid_cap, s, grid_out);
        case 6: return walk_ef<VS_AR, 6>(a, instance, grid_cap, s, grid_out);
        case 8: return walk_ef<VS_AR, 8>(a, instance, grid_cap, s, grid_out);
        case 12: return walk_ef<VS_AR, 12>(a, instance, grid_cap, s, grid_out);
        case 16: return walk_ef<VS_AR, 16>(a, instance, grid_cap, s, grid_out);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vs
