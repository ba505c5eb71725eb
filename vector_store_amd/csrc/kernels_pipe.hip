// kernels_pipe.hip -- hnsw_pipe_walk_kernel for ONE float arithmetic (-DVS_AR=0..5): the pipelined walk of pipe_device.hpp
// behind usearch::Index::search / filtered_search for LONE queries (the reference issues one query per FFI call:
// crates/vector-store/src/vs_index/usearch.rs:210-212, :233-236; filtered queries each on their own blocking thread, :937-948).
// One workgroup of kPipeTeam waves per query: wave 0 walks, the others evaluate candidates ahead of it.  A query this kernel
// cannot answer exactly as usearch orders it (two equal distances met, or `next` outgrew the pool) is flagged kPipeRedo and
// served by the usearch-order walk (kernels_walk.hip).
#include <mutex>

#include "kernels.hpp"
#include "pipe_device.hpp"

#ifndef VS_AR
#error "compile with -DVS_AR=<arithmetic>"
#endif

namespace vs {

template <int AR, int I, int EFCAP, int MODE>
__global__ __launch_bounds__(64 * kPipeTeam) void hnsw_pipe_walk_kernel(WalkArgs a) {
    constexpr bool VISG = true;
    using Sh = PipeShared<EFCAP, kPipeTeam, false, VISG>;
    __shared__ Sh sh;
    extern __shared__ uint2 pipe_pool[];
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t w = threadIdx.x >> 6;
    uint32_t qi = blockIdx.x;
    if (a.qlist) {  // second-chance launches name their queries
        if (qi >= *a.qcount) return;
        qi = a.qlist[qi];
    }
    // this query's buffers: the strided arrays of WalkArgs, or its entry of the batch table
    const PipeQuery* pq = a.pipe_qtable ? a.pipe_qtable + blockIdx.x : nullptr;
    const float* query = pq ? pq->query : a.queries + (size_t)qi * a.q_stride;
    const uint32_t k = pq ? pq->k : a.k;
    uint64_t* ok = pq ? pq->keys : a.out_keys + (size_t)qi * a.k;
    float* od = pq ? pq->dist : a.out_dist + (size_t)qi * a.k;
    uint32_t* found_out = pq ? pq->cnt + 2 : a.out_found + qi;
    if (ix.max_level < 0) {  // empty index
        if (w == 0) {
            for (uint32_t i = lane; i < k; i += kWave) {
                ok[i] = kFreeKey;
                od[i] = __builtin_inff();
            }
            if (lane == 0) {
                *found_out = 0;
                if (pq) {
                    pq->cnt[0] = pq->cnt[1] = pq->cnt[3] = 0u;
                    __threadfence_system();
                    __hip_atomic_store(pq->done, pq->round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        return;
    }
    if (threadIdx.x < (uint32_t)kPipeTeam) {
        sh.job_state[threadIdx.x] = 0u;
        if (threadIdx.x == 0) {
            sh.stop = 0u;
            sh.prof_jobs[0] = sh.prof_jobs[1] = 0u;
        }
    }
    if (threadIdx.x < (uint32_t)kPipeCache) sh.c_ready[threadIdx.x] = 0u;
    WalkSpace ws = {nullptr, nullptr, nullptr, 0u, 0u, 0u};
    {
        char* base = pq ? pq->space : a.space + (size_t)blockIdx.x * a.space_stride;
        ws.bitmap = reinterpret_cast<uint32_t*>(base);
        ws.vlog = ws.bitmap + a.bitmap_words;
        ws.heap = reinterpret_cast<uint2*>(ws.vlog + a.vlog_cap);
        ws.bitmap_words = a.bitmap_words;
        ws.vlog_cap = a.vlog_cap;
        ws.heap_cap = a.heap_cap;
    }
    const bool tomb = a.has_removed != 0;
    const uint32_t* allow = pq ? pq->allow : a.allow ? a.allow + (size_t)qi * a.allow_stride : nullptr;
    const uint32_t* known = pq ? pq->known : a.known ? a.known + (size_t)qi * a.allow_stride : nullptr;
    if (pq) {
        // The round's exchange with the host, first half: the verdicts it gave for the slots the last round listed become bits of the
        // query's device-resident bitmaps (the first round zeroes them instead).  Every wave takes part; the barriers of the descent
        // that follows order it before the first verdict is read.
        uint32_t* allow_w = pq->allow;
        uint32_t* known_w = pq->known;
        if (pq->zero_bits) {
            for (uint32_t i = threadIdx.x; i < pq->words; i += 64u * kPipeTeam) {
                allow_w[i] = 0u;
                known_w[i] = 0u;
            }
        }
        const uint32_t m = pq->apply_m;
        for (uint32_t i = threadIdx.x; i < m; i += 64u * kPipeTeam) {
            const uint32_t s = pq->list[i];
            if (s < pq->slots) {
                atomicOr(&known_w[s >> 5], 1u << (s & 31u));
                if (pq->verdict[i]) atomicOr(&allow_w[s >> 5], 1u << (s & 31u));
            }
        }
        __threadfence();
        __syncthreads();
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
        pipe_helper_loop<AR, I>(ix, q, sh, ws, tomb, allow, known, lane, w);
        return;
    }
    Counters cnt = {0, 0, 0};
    float start_d = 0.f;
    uint32_t start;
    // (inlined whatever the size of the kernel: an out-of-line call passes the query's registers through scratch memory)
    [[clang::always_inline]] start = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, 0, cnt, lane, &start_d);
    team_release(sh, lane);  // the last barrier: from here on the waves meet through LDS words only
    PipeTop<EFCAP / 64> top;
    const PipeOut r = pipe_walk<AR, I, MODE>(ix, sh, pipe_pool, a.pipe_pool_cap, ws, start, start_d, a.ef, tomb, allow, known,
                                       pq ? pq->list : a.unknown_list ? a.unknown_list + (size_t)qi * a.unknown_cap : nullptr,
                                       pq ? pq->cnt : a.unknown_count ? a.unknown_count + qi : nullptr, pq ? pq->cap : a.unknown_cap,
                                       pq ? pq->budget : a.unknown_budget, pq ? pq->cnt + 1 : a.consulted ? a.consulted + qi : nullptr, cnt, lane, top,
                                       a.debug ? a.debug + (size_t)qi * 12 : nullptr, a.pipe_fused_order != 0u);
    if (r.status == 1u) {  // the usearch-order walk answers it (and lists the verdicts IT misses: this walk's list is dropped)
        if (lane == 0) {
            if (a.retry_list) a.retry_list[atomicAdd(a.retry_count, 1u)] = qi;
            *found_out = kPipeRedo;
            if (pq) {
                pq->cnt[0] = pq->cnt[1] = 0u;
                __threadfence_system();
                __hip_atomic_store(pq->done, pq->round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
        if (pq) pq->cnt[3] = (uint32_t)cnt.evals;
    }
    if (pq) {
        // second half of the exchange: the answer, the counters and the list are in the caller's pinned block -- every lane's stores are
        // out (the list's as well: pipe_walk drained them) before the flag says so
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(pq->done, pq->round_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int AR, int I, int EFCAP, int MODE>
static hipError_t pipe_launch(const WalkArgs& a, hipStream_t s) {
    auto kernel = hnsw_pipe_walk_kernel<AR, I, EFCAP, MODE>;
    const size_t dyn = (size_t)a.pipe_pool_cap * sizeof(uint2);
    static std::once_flag once[16];  // the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t attr = hipSuccess;
    std::call_once(once[dev & 15], [&] {
        attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(kernel, dim3(a.nq), dim3(64 * kPipeTeam), dyn, s, a);
    return hipGetLastError();
}

template <int AR, int I>
static hipError_t pipe_ef(const WalkArgs& a, hipStream_t s) {
    // (the LDS tag table as the visited set -- PipeShared<..., VISG = false> -- was measured for unfiltered lone walks and dropped: its
    // test-and-set costs the walker as much as the returning global atomic, which runs under the early post: 0.80 against 0.74 ms)
    if (a.pipe_lds_visited) return hipErrorInvalidValue;
    // one instance per purpose (pipe_device.hpp `MODE`): plain lone queries, the exact walk of a filtered query, its exploring rounds
    if (a.ef > 512) return hipErrorInvalidValue;
    if (a.pipe_explore) return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipeExplore>(a, s) : pipe_launch<AR, I, 512, kPipeExplore>(a, s);
    if (a.allow || a.pipe_qtable) return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipeFiltered>(a, s) : pipe_launch<AR, I, 512, kPipeFiltered>(a, s);
    return a.ef <= 256 ? pipe_launch<AR, I, 256, kPipePlain>(a, s) : pipe_launch<AR, I, 512, kPipePlain>(a, s);
}

template <>
hipError_t launch_pipe_walk_ar<VS_AR>(const WalkArgs& a, uint32_t iters, hipStream_t s) {
    if (!a.nq) return hipSuccess;
    if (a.ix.M0 > 64u || (!a.space && !a.pipe_qtable) || a.pipe_pool_cap < 256u) return hipErrorInvalidValue;
    switch (iters) {
        case 1: return pipe_ef<VS_AR, 1>(a, s);
        case 2: return pipe_ef<VS_AR, 2>(a, s);
        case 3: return pipe_ef<VS_AR, 3>(a, s);
        case 4: return pipe_ef<VS_AR, 4>(a, s);
        case 6: return pipe_ef<VS_AR, 6>(a, s);
        case 8: return pipe_ef<VS_AR, 8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vs
