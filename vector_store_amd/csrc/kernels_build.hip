// kernels_build.hip -- batched HNSW insertion (K7 of SURVEY.md section 2.3).
// Replaces usearch::Index::add as called at reference vs_index/usearch.rs:194-196.
//
// A sub-batch of new nodes is inserted against the graph frozen at the start of the
// sub-batch (the host keeps sub-batches small relative to the index, see engine.cpp):
//   1. hnsw_insert_kernel : one wave per new node -- greedy descent, per-level beam search
//      (expansion_add), neighbour-selection heuristic, forward links, reverse-link requests;
//   2. radix sort of the requests by (level, target);
//   3. hnsw_link_kernel   : one wave per (level, target) group -- append or re-run the
//      heuristic on the target's list (usearch reconnect_neighbor_nodes_).
// No locks: every adjacency row is written by exactly one wave per kernel.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.hpp"

namespace vs {

template <int KIND, int I, int EFCAP, int NB, int CH>
__global__ __launch_bounds__(64) void hnsw_insert_kernel(InsertArgs a) {
    __shared__ BeamShared<EFCAP, NB, true, CH> sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t b = blockIdx.x;
    const uint32_t slot = a.slots[b];
    const int level = a.levels[b];
    float4 q[I];
    load_row<I>(ix, slot, q, lane);
    const float q_inv = ix.metric == COS ? ix.inv_norm[slot] : 0.f;
    Counters cnt = {0, 0, 0};
    uint32_t closest = ix.entry_slot;
    if (ix.max_level > level)
        closest = greedy_descent<KIND, I>(ix, sh, q, q_inv, ix.entry_slot, ix.max_level, level, cnt, lane);
    const int top = level < ix.max_level ? level : ix.max_level;
    uint32_t req = a.req_off[b] - a.req_base;
    for (int l = top; l >= 0; --l) {
        int cur = 0;
        uint32_t sz = beam_search<KIND, I>(ix, sh, q, q_inv, closest, l, a.ef_add, slot, cnt, lane, cur);
        // usearch connect_new_node_: forward links are refined to `connectivity` on every level
        uint32_t nsel = refine<KIND, I>(ix, sh, cur, sz, ix.M, cnt, lane);
        uint32_t cap;
        uint32_t* row = const_cast<uint32_t*>(adjacency(ix, slot, l, cap));
        if ((uint32_t)lane < cap) row[lane] = (uint32_t)lane < nsel ? sh.sel_s[lane] : kInvalid;
        if ((uint32_t)lane < ix.M) {
            bool on = (uint32_t)lane < nsel;
            a.req_key[req + lane] = on ? (((uint64_t)(uint32_t)l << 32) | sh.sel_s[lane]) : ~0ull;
            a.req_val[req + lane] = on ? (((uint64_t)__float_as_uint(sh.sel_d[lane]) << 32) | slot) : 0ull;
        }
        req += ix.M;
        if (nsel) closest = sh.sel_s[0];
        __syncthreads();
    }
    if (lane == 0) {
        atomicAdd(&a.stats[ST_ADD_EVALS], cnt.evals);
        atomicAdd(&a.stats[ST_ADD_HOPS], cnt.hops);
        atomicAdd(&a.stats[ST_ADDED], 1ull);
        if (cnt.overflow) atomicAdd(&a.stats[ST_OVERFLOW], cnt.overflow);
    }
}

struct LinkShared {
    float lst_d[1][128];
    uint32_t lst_s[1][128];
    float t_d[128];
    uint32_t t_s[128];
    uint32_t u_slot[64];
    float u_dist[64];
    uint32_t sel_s[64];
    float sel_d[64];
};

constexpr uint32_t kMaxNewPerTarget = 64;

template <int KIND, int I>
__global__ __launch_bounds__(64) void hnsw_link_kernel(LinkArgs a) {
    __shared__ LinkShared sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t r = blockIdx.x;
    const uint64_t key = a.req_key[r];
    if (key == ~0ull) return;
    if (r > 0 && a.req_key[r - 1] == key) return;  // not the head of its (level, target) group
    uint32_t n_new;
    {
        uint32_t idx = r + (uint32_t)lane;
        bool same = idx < a.total && a.req_key[idx] == key;
        uint64_t mask = __ballot(same);
        n_new = ~mask ? (uint32_t)__builtin_ctzll(~mask) : kMaxNewPerTarget;
    }
    const uint32_t target = (uint32_t)key;
    const int level = (int)(key >> 32);
    uint32_t cap;
    uint32_t* row = const_cast<uint32_t*>(adjacency(ix, target, level, cap));
    uint32_t src = kInvalid;
    float src_d = 0.f;
    if ((uint32_t)lane < n_new) {
        uint64_t v = a.req_val[r + lane];
        src = (uint32_t)v;
        src_d = __uint_as_float((uint32_t)(v >> 32));
    }
    // existing links; a link to a source being (re)inserted is superseded by the new request
    uint32_t ex = (uint32_t)lane < cap ? row[lane] : kInvalid;
    for (uint32_t j = 0; j < n_new; ++j) {
        uint32_t sj = (uint32_t)__shfl((int)src, (int)j);
        if (ex == sj) ex = kInvalid;
    }
    uint64_t emask = __ballot(ex != kInvalid);
    const uint32_t cnt = (uint32_t)__popcll(emask);
    if (ex != kInvalid) sh.u_slot[mbcnt(emask)] = ex;
    __syncthreads();
    if (cnt + n_new <= cap) {  // usearch: close_header.push_back(new_slot)
        const bool is_new = lane >= (int)cnt && lane < (int)(cnt + n_new);
        const uint32_t from_new = (uint32_t)__shfl((int)src, is_new ? lane - (int)cnt : 0);
        if ((uint32_t)lane < cap) row[lane] = (uint32_t)lane < cnt ? sh.u_slot[lane] : (is_new ? from_new : kInvalid);
        return;
    }
    // usearch: top = {new} U existing, all measured from `close_slot`; refine_(connectivity_max)
    Counters c = {0, 0, 0};
    float4 q[I];
    load_row<I>(ix, target, q, lane);
    const float q_inv = ix.metric == COS ? ix.inv_norm[target] : 0.f;
    eval_batch<KIND, I>(ix, q, q_inv, sh.u_slot, sh.u_dist, cnt, lane);
    __syncthreads();
    c.evals += cnt;
    const uint32_t total = cnt + n_new;  // <= 32 + 64
    if ((uint32_t)lane < cnt) {
        sh.t_d[lane] = sh.u_dist[lane];
        sh.t_s[lane] = sh.u_slot[lane];
    }
    if ((uint32_t)lane < n_new) {
        sh.t_d[cnt + lane] = src_d;
        sh.t_s[cnt + lane] = src;
    }
    __syncthreads();
    for (uint32_t e = (uint32_t)lane; e < total; e += kWave) {  // rank sort, ascending (distance, slot)
        float ed = sh.t_d[e];
        uint32_t es = sh.t_s[e];
        uint32_t rank = 0;
        for (uint32_t f = 0; f < total; ++f) rank += key_less(sh.t_d[f], sh.t_s[f], ed, es) ? 1u : 0u;
        sh.lst_d[0][rank] = ed;
        sh.lst_s[0][rank] = es;
    }
    __syncthreads();
    uint32_t nsel = refine<KIND, I>(ix, sh, 0, total, cap, c, lane);
    if ((uint32_t)lane < cap) row[lane] = (uint32_t)lane < nsel ? sh.sel_s[lane] : kInvalid;
    if (lane == 0) atomicAdd(&a.stats[ST_ADD_EVALS], c.evals);
}

template <int KIND, int I>
static hipError_t insert_ef(const InsertArgs& a, hipStream_t s) {
    dim3 grid(a.n), block(64);
    if (a.ef_add <= 128)
        hipLaunchKernelGGL((hnsw_insert_kernel<KIND, I, 128, 1024, 1>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((hnsw_insert_kernel<KIND, I, 256, 1024, 2>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <int KIND>
static hipError_t insert_iters(const InsertArgs& a, uint32_t iters, hipStream_t s) {
    switch (iters) {
        case 1: return insert_ef<KIND, 1>(a, s);
        case 2: return insert_ef<KIND, 2>(a, s);
        case 3: return insert_ef<KIND, 3>(a, s);
        case 4: return insert_ef<KIND, 4>(a, s);
        case 6: return insert_ef<KIND, 6>(a, s);
        case 8: return insert_ef<KIND, 8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_insert(const InsertArgs& a, uint32_t iters, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    if (!search_supported(iters, a.ef_add)) return hipErrorInvalidValue;
    return a.ix.metric == L2SQ ? insert_iters<KL2>(a, iters, s) : insert_iters<KDOT>(a, iters, s);
}

template <int KIND>
static hipError_t link_iters(const LinkArgs& a, uint32_t iters, hipStream_t s) {
    dim3 grid(a.total), block(64);
    switch (iters) {
        case 1: hipLaunchKernelGGL((hnsw_link_kernel<KIND, 1>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((hnsw_link_kernel<KIND, 2>), grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL((hnsw_link_kernel<KIND, 3>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((hnsw_link_kernel<KIND, 4>), grid, block, 0, s, a); break;
        case 6: hipLaunchKernelGGL((hnsw_link_kernel<KIND, 6>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((hnsw_link_kernel<KIND, 8>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_link(const LinkArgs& a, uint32_t iters, hipStream_t s) {
    if (a.total == 0) return hipSuccess;
    return a.ix.metric == L2SQ ? link_iters<KL2>(a, iters, s) : link_iters<KDOT>(a, iters, s);
}

size_t sort_temp_bytes(size_t n) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, n ? n : 1, 0, 64, (hipStream_t)0);
    return bytes;
}

hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint64_t* vals_in,
                      uint64_t* vals_out, size_t n, unsigned end_bit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, s);
}

}  // namespace vs
