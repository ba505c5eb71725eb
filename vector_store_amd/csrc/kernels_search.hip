// kernels_search.hip -- hnsw_search: one wavefront per query (K5 of SURVEY.md section 2.3).
// Replaces usearch::Index::search as called at reference vs_index/usearch.rs:210-212.
#include "kernels.hpp"

namespace vs {

template <int KIND, int I, int EFCAP, int NB, int CH>
__global__ __launch_bounds__(64) void hnsw_search_kernel(SearchArgs a) {
    __shared__ BeamShared<EFCAP, NB, false, CH> sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t qi = blockIdx.x;
    uint64_t* ok = a.out_keys + (size_t)qi * a.k;
    float* od = a.out_dist + (size_t)qi * a.k;
    if (ix.max_level < 0) {
        for (uint32_t i = lane; i < a.k; i += kWave) {
            ok[i] = kFreeKey;
            od[i] = __builtin_inff();
        }
        if (lane == 0) a.out_found[qi] = 0;
        return;
    }
    float4 q[I];
    load_query<I>(ix, a.queries + (size_t)qi * a.q_stride, q, lane);
    const float q_inv = ix.metric == COS ? inv_norm_of<I>(ix, q) : 0.f;
    Counters cnt = {0, 0, 0};
    // usearch index_gt::search: search_for_one_ down to level 1, then the base-level beam.
    uint32_t start = greedy_descent<KIND, I>(ix, sh, q, q_inv, ix.entry_slot, ix.max_level, 0, cnt, lane);
    int cur = 0;
    uint32_t sz = beam_search<KIND, I>(ix, sh, q, q_inv, start, 0, a.ef, kInvalid, cnt, lane, cur, a.has_removed != 0);
    __syncthreads();
    // top.sort_ascending(); top.shrink(wanted); removed members (free key) are never results.
    uint32_t written = 0;
#pragma unroll
    for (int r = 0; r < EFCAP / kWave; ++r) {
        uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
        bool okp = p < sz;
        uint32_t slot = okp ? (sh.lst_s[cur][p] & kSlotMask) : 0u;
        uint64_t key = okp ? ix.keys[slot] : kFreeKey;
        okp = okp && key != kFreeKey;
        uint64_t mask = __ballot(okp);
        uint32_t pos = written + mbcnt(mask);
        if (okp && pos < a.k) {
            ok[pos] = key;
            od[pos] = sh.lst_d[cur][p];
        }
        written += (uint32_t)__popcll(mask);
    }
    uint32_t found = written < a.k ? written : a.k;
    for (uint32_t i = found + lane; i < a.k; i += kWave) {
        ok[i] = kFreeKey;
        od[i] = __builtin_inff();
    }
    if (lane == 0) {
        a.out_found[qi] = found;
        atomicAdd(&a.stats[ST_SEARCH_EVALS], cnt.evals);
        atomicAdd(&a.stats[ST_SEARCH_HOPS], cnt.hops);
        atomicAdd(&a.stats[ST_QUERIES], 1ull);
        if (cnt.overflow) atomicAdd(&a.stats[ST_OVERFLOW], cnt.overflow);
    }
}

template <int KIND, int I>
static hipError_t launch_ef(const SearchArgs& a, hipStream_t s) {
    dim3 grid(a.nq), block(64);
    if (I == 1 && a.stress_small_table && a.ef <= 128)
        hipLaunchKernelGGL((hnsw_search_kernel<KIND, 1, 128, 256, 1>), grid, block, 0, s, a);
    else if (a.ef <= 128)
        hipLaunchKernelGGL((hnsw_search_kernel<KIND, I, 128, 1024, 1>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((hnsw_search_kernel<KIND, I, 256, 1024, 2>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <int KIND>
static hipError_t launch_iters(const SearchArgs& a, uint32_t iters, hipStream_t s) {
    switch (iters) {
        case 1: return launch_ef<KIND, 1>(a, s);
        case 2: return launch_ef<KIND, 2>(a, s);
        case 3: return launch_ef<KIND, 3>(a, s);
        case 4: return launch_ef<KIND, 4>(a, s);
        case 6: return launch_ef<KIND, 6>(a, s);
        case 8: return launch_ef<KIND, 8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

bool search_supported(uint32_t iters, uint32_t ef) {
    return (iters == 1 || iters == 2 || iters == 3 || iters == 4 || iters == 6 || iters == 8) && ef >= 1 && ef <= 256;
}

uint32_t visited_domain_bits(uint32_t ef) { return ef <= 128 ? VisitedCfg<1024, 1>::domain_bits : VisitedCfg<1024, 2>::domain_bits; }

hipError_t launch_search(const SearchArgs& a, uint32_t iters, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (!search_supported(iters, a.ef)) return hipErrorInvalidValue;
    return a.ix.metric == L2SQ ? launch_iters<KL2>(a, iters, s) : launch_iters<KDOT>(a, iters, s);
}

}  // namespace vs
