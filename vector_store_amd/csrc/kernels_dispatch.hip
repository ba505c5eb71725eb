// kernels_dispatch.hip -- picks the arithmetic (storage type x metric family) and forwards to the per-arithmetic
// translation units (kernels_arith.hip, one object per -DVS_AR); also hosts the request sort of the build path.
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.hpp"
#include "pipe_pod.hpp"

namespace vs {

int arith_of(int scalar, int metric) {
    switch (scalar) {
        case SC_F32: return metric == L2SQ ? AR_F32_L2 : AR_F32_DOT;
        case SC_F16: return metric == L2SQ ? AR_F16_L2 : AR_F16_DOT;
        case SC_BF16: return metric == L2SQ ? AR_BF16_L2 : AR_BF16_DOT;
        case SC_I8: return AR_I8;
        default: return AR_B1;
    }
}

bool search_supported(uint32_t iters, uint32_t ef) {
    return (iters == 1 || iters == 2 || iters == 3 || iters == 4 || iters == 6 || iters == 8 || iters == 12 || iters == 16) && ef >= 1 && ef <= 512;
}

uint32_t walk_small_table_bits() { return VisitedCfg<512, 2>::domain_bits; }

// must name the same VisitedCfg as the instance table in kernels_walk.hip (walk_ef)
uint32_t walk_instance_domain_bits(uint32_t instance) {
    switch (instance & ~kWalkTeamFlag) {
        case WALK_LDS_128: return VisitedCfg<1024, 1>::domain_bits;
        case WALK_LDS_128_SMALL:
        case WALK_LDS_256_DENSE: return VisitedCfg<512, 2>::domain_bits;
        case WALK_LDS_128_TINY: return VisitedCfg<256, 1>::domain_bits;
        case WALK_LDS_256:
        case WALK_LDS_320: return VisitedCfg<1024, 2>::domain_bits;
        case WALK_LDS_512: return VisitedCfg<2048, 2>::domain_bits;
        default: return 32;
    }
}

uint32_t visited_domain_bits(uint32_t ef, bool wide) {
    // (the insert kernel's instance for expansion_add of 257..512 always carries the wide tags: engine.hip passes wide = true there)
    if (wide) return ef <= 128 ? VisitedCfg<1024, 1, true>::domain_bits : ef <= 256 ? VisitedCfg<1024, 2, true>::domain_bits : VisitedCfg<2048, 2, true>::domain_bits;
    return ef <= 128 ? VisitedCfg<1024, 1>::domain_bits : ef <= 256 ? VisitedCfg<1024, 2>::domain_bits : VisitedCfg<2048, 2>::domain_bits;
}

#define VS_DISPATCH(fn, args)                                  \
    switch (arith_of(a.ix.scalar, a.ix.metric)) {              \
        case AR_F32_DOT: return fn<AR_F32_DOT> args;           \
        case AR_F32_L2: return fn<AR_F32_L2> args;             \
        case AR_F16_DOT: return fn<AR_F16_DOT> args;           \
        case AR_F16_L2: return fn<AR_F16_L2> args;             \
        case AR_BF16_DOT: return fn<AR_BF16_DOT> args;         \
        case AR_BF16_L2: return fn<AR_BF16_L2> args;           \
        case AR_I8: return fn<AR_I8> args;                     \
        default: return fn<AR_B1> args;                        \
    }

hipError_t launch_search(const SearchArgs& a, uint32_t iters, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (!search_supported(iters, a.ef)) return hipErrorInvalidValue;
    VS_DISPATCH(launch_search_ar, (a, iters, s))
}

hipError_t launch_walk(const WalkArgs& a, uint32_t iters, uint32_t instance, uint32_t grid_cap, hipStream_t s, uint32_t* grid_out) {
    if (!grid_out && !a.qlist && a.nq == 0) return hipSuccess;
    if (!search_supported(iters, 1) || a.ef < 1 || a.ef > kMaxWalkBeam) return hipErrorInvalidValue;
    VS_DISPATCH(launch_walk_ar, (a, iters, instance, grid_cap, s, grid_out))
}

bool pipe_walk_supported(const IndexView& ix, uint32_t iters, uint32_t ef) {
    // Integer storage: ties are the rule there.  i8 (cosine over 8-bit components: thousands of distinct distances): most rounds of a
    // filtered walk still pass the window rule (pipe_device.hpp), the others are walked again in usearch's order, as on float storage --
    // 10M x 768, 10 % selective, 17 / 64 / 128 callers: 423 / 425 / 422 -> 1,163 / 4,350 / 8,444 queries/s.  b1 (a few hundred distinct
    // distances): nearly every round is walked again, the attempt is wasted (10 %: 344 -> 341-369, 1 %: 34.5 -> 17.6): not by default.
    // VS_HNSW_PIPE_INT=0: neither (the state before the end of round 4); =1: both.
    static const char* int_env = std::getenv("VS_HNSW_PIPE_INT");
    const bool i8_ok = !(int_env && int_env[0] == '0'), b1_ok = int_env && int_env[0] == '1';
    return (ix.scalar != SC_I8 || i8_ok) && (ix.scalar != SC_B1 || b1_ok) && ix.M0 <= 64u && ef >= 1 && ef <= 512 &&
           (iters == 1 || iters == 2 || iters == 3 || iters == 4 || iters == 6 || iters == 8);
}

hipError_t launch_pipe_walk(const WalkArgs& a, uint32_t iters, hipStream_t s) {
    if (!pipe_walk_supported(a.ix, iters, a.ef)) return hipErrorInvalidValue;
    switch (arith_of(a.ix.scalar, a.ix.metric)) {
        case AR_F32_DOT: return launch_pipe_walk_ar<AR_F32_DOT>(a, iters, s);
        case AR_F32_L2: return launch_pipe_walk_ar<AR_F32_L2>(a, iters, s);
        case AR_F16_DOT: return launch_pipe_walk_ar<AR_F16_DOT>(a, iters, s);
        case AR_F16_L2: return launch_pipe_walk_ar<AR_F16_L2>(a, iters, s);
        case AR_BF16_DOT: return launch_pipe_walk_ar<AR_BF16_DOT>(a, iters, s);
        case AR_BF16_L2: return launch_pipe_walk_ar<AR_BF16_L2>(a, iters, s);
        case AR_I8: return launch_pipe_walk_ar<AR_I8>(a, iters, s);
        case AR_B1: return launch_pipe_walk_ar<AR_B1>(a, iters, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pipe_pod(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    if (!pipe_walk_supported(a.ix, iters, a.ef)) return hipErrorInvalidValue;
    switch (arith_of(a.ix.scalar, a.ix.metric)) {
        case AR_F32_DOT: return launch_pipe_pod_ar<AR_F32_DOT>(a, iters, s, slots, ctl);
        case AR_F32_L2: return launch_pipe_pod_ar<AR_F32_L2>(a, iters, s, slots, ctl);
        case AR_F16_DOT: return launch_pipe_pod_ar<AR_F16_DOT>(a, iters, s, slots, ctl);
        case AR_F16_L2: return launch_pipe_pod_ar<AR_F16_L2>(a, iters, s, slots, ctl);
        case AR_BF16_DOT: return launch_pipe_pod_ar<AR_BF16_DOT>(a, iters, s, slots, ctl);
        case AR_BF16_L2: return launch_pipe_pod_ar<AR_BF16_L2>(a, iters, s, slots, ctl);
        case AR_I8: return launch_pipe_pod_ar<AR_I8>(a, iters, s, slots, ctl);
        case AR_B1: return launch_pipe_pod_ar<AR_B1>(a, iters, s, slots, ctl);
        default: return hipErrorInvalidValue;
    }
}

// (round 6) the usearch-order team walk as a pod: b1 storage only (kernels_walk.hip)
hipError_t launch_walk_pod(const WalkArgs& a, uint32_t iters, hipStream_t s, PodSlot* slots, PodCtl* ctl) {
    if (arith_of(a.ix.scalar, a.ix.metric) != AR_B1) return hipErrorInvalidValue;
    return launch_walk_pod_ar<AR_B1>(a, iters, s, slots, ctl);
}

hipError_t launch_insert(const InsertArgs& a, uint32_t iters, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    if (!search_supported(iters, a.ef_add)) return hipErrorInvalidValue;
    VS_DISPATCH(launch_insert_ar, (a, iters, s))
}

hipError_t launch_link(const LinkArgs& a, uint32_t iters, hipStream_t s) {
    if (a.total == 0) return hipSuccess;
    VS_DISPATCH(launch_link_ar, (a, iters, s))
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

size_t sort_keys_temp_bytes(size_t n) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, bytes, k, k, n ? n : 1, 0, 64, (hipStream_t)0);
    return bytes;
}

hipError_t sort_keys(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, n, 0u, 64u, s);
}

}  // namespace vs
