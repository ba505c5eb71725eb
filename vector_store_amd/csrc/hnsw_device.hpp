// hnsw_device.hpp -- device-side building blocks of the MI355X HNSW engine (gfx950, wave64).
//
// One 64-lane wavefront owns one query (or one node being inserted).  All control flow
// is wave-uniform; per-lane work is (a) one neighbour id per lane for the visited test,
// (b) 16 B of a vector row per lane per load for distances.
//
// Replaces what the reference delegates to usearch::Index::{search,add}
// (reference crates/vector-store/src/vs_index/usearch.rs:196,212); the algorithm is the
// one restated in oracle/cpu_hnsw.cpp (search_for_one / search_to_insert /
// search_to_find_in_base / refine / reconnect_neighbor_nodes).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace vs {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kExpanded = 0x80000000u;  // bit 31 of a list entry's slot: already expanded
constexpr uint32_t kDead = 0x40000000u;      // bit 30: removed member (free key): traversed, never a result
constexpr uint32_t kSlotMask = 0x3FFFFFFFu;
constexpr uint64_t kFreeKey = ~0ull;
constexpr int kWave = 64;

enum : int { COS = 0, L2SQ = 1, IP = 2, HAMMING = 3 };                     // vs_metric_kind
enum : int { SC_F32 = 0, SC_F16 = 1, SC_BF16 = 2, SC_I8 = 3, SC_B1 = 4 };  // vs_scalar_kind (storage type)
// Arithmetic of one (storage type, metric family) pair -- the kernels' template parameter.
// cos and ip share the dot product; i8 derives every metric from the integer dot product and the stored
// sums of squares; b1 is popcount(xor).
enum : int { AR_F32_DOT = 0, AR_F32_L2 = 1, AR_F16_DOT = 2, AR_F16_L2 = 3, AR_BF16_DOT = 4, AR_BF16_L2 = 5, AR_I8 = 6, AR_B1 = 7 };
constexpr int KDOT = AR_F32_DOT, KL2 = AR_F32_L2;  // the f32 pair under its old name

// HBM layout (see DESIGN.md "Data layout"):
//  vectors  [capacity][stride4] 16-byte chunks, row = iters*lanes chunks, zero padded; a chunk holds
//           4 f32 | 8 f16 | 8 bf16 | 16 i8 | 128 bits, elements in order
//  aux      [capacity] f32: f32/f16/bf16 + cosine: 1/|row| (0 marks the zero vector);
//           i8: sum of squares of the stored bytes (exact integer); unused otherwise
//  adj0     [capacity][M0] u32 slots, kInvalid padded (128 B at M0=32: one cache line/expansion)
//  upper    [blocks][M] u32; node s, level l>=1 lives in block upper_off[s]+l-1
//  keys     [capacity] u64, kFreeKey = removed / never used
struct IndexView {
    const uint4* vectors;
    const float* aux;
    uint32_t* adj0;
    uint32_t* upper;
    const uint32_t* upper_off;
    const uint64_t* keys;
    uint32_t dim;
    uint32_t stride4;     // float4 per row = iters * lanes
    uint32_t lanes;       // lanes cooperating on one vector (power of two <= 64)
    uint32_t lanes_log2;
    uint32_t M, M0;
    int32_t metric;       // COS / L2SQ / IP / HAMMING
    int32_t scalar;       // SC_*
    uint32_t entry_slot;
    int32_t max_level;    // -1: empty index
    uint32_t nt_rows;     // 1: vector rows are loaded non-temporally (table far larger than the caches)
};

struct Counters {
    unsigned long long evals;
    unsigned long long hops;
    unsigned long long overflow;
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// (distance, slot) strict total order; slot compared without the expanded flag.
__device__ __forceinline__ bool key_less(float ad, uint32_t as, float bd, uint32_t bs) {
    return ad < bd || (ad == bd && (as & kSlotMask) < (bs & kSlotMask));
}

// Sum over each aligned group of `lanes` consecutive lanes (lanes = power of two), result in every lane
// of the group.  Rows of 16 lanes are all-reduced with four DPP rotations (v_add_f32_dpp row_ror, a few
// cycles each) instead of a six-deep chain of ds_bpermute (~100+ cycles each): the reduction sits on the
// critical path of every hop, so this is what sets single-query latency.
#define VS_DPP_ROR(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xF, 0xF, true))
__device__ __forceinline__ float group_sum(float v, uint32_t lanes) {
    if (lanes >= 16) {
        v += VS_DPP_ROR(v, 8);
        v += VS_DPP_ROR(v, 4);
        v += VS_DPP_ROR(v, 2);
        v += VS_DPP_ROR(v, 1);
        if (lanes == 64) {
            float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
            float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
            float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
            v = (a + b) + (c + d);
        } else if (lanes == 32) {
            v += __shfl_xor(v, 16);
        }
        return v;
    }
    for (uint32_t o = lanes >> 1; o; o >>= 1) v += __shfl_xor(v, (int)o);
    return v;
}

__device__ __forceinline__ int group_sum(int v, uint32_t lanes) {  // integer twin (i8 dot products, popcounts)
    if (lanes >= 16) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 8, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 4, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 2, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x120 + 1, 0xF, 0xF, true);
        if (lanes == 64) {
            v = (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
                (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
        } else if (lanes == 32) {
            v += __shfl_xor(v, 16);
        }
        return v;
    }
    for (uint32_t o = lanes >> 1; o; o >>= 1) v += __shfl_xor(v, (int)o);
    return v;
}
__device__ __forceinline__ double group_sum(double v, uint32_t lanes) {  // query / row preparation only
    for (uint32_t o = lanes >> 1; o; o >>= 1) v += __shfl_xor(v, (int)o);
    return v;
}

template <int AR>
struct Arith {
    static constexpr bool is_int = AR == AR_I8 || AR == AR_B1;
    static constexpr bool is_l2 = AR == AR_F32_L2 || AR == AR_F16_L2 || AR == AR_BF16_L2;
    static constexpr int scalar = AR <= AR_F32_L2 ? SC_F32 : AR <= AR_F16_L2 ? SC_F16 : AR <= AR_BF16_L2 ? SC_BF16 : AR == AR_I8 ? SC_I8 : SC_B1;
    static constexpr uint32_t epc = scalar == SC_F32 ? 4 : scalar == SC_I8 ? 16 : scalar == SC_B1 ? 128 : 8;  // elements per chunk
    using acc_t = typename std::conditional<is_int, int, float>::type;
};

__device__ __forceinline__ float half_bits_to_float(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }
__device__ __forceinline__ uint32_t float_to_half_bits(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float bf16_bits_to_float(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t float_to_bf16_bits(float f) {  // round to nearest even; NaN stays NaN
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

template <int AR>
__device__ __forceinline__ void unpack8(const uint4 c, float (&f)[8]) {  // f16 / bf16 chunk -> 8 floats
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (Arith<AR>::scalar == SC_F16) {
            f[2 * j] = half_bits_to_float(w[j] & 0xFFFFu);
            f[2 * j + 1] = half_bits_to_float(w[j] >> 16);
        } else {
            f[2 * j] = bf16_bits_to_float(w[j] & 0xFFFFu);
            f[2 * j + 1] = bf16_bits_to_float(w[j] >> 16);
        }
    }
}

// acc += partial metric of one 16-byte chunk of the query (a) against one chunk of a row (b).
template <int AR>
__device__ __forceinline__ typename Arith<AR>::acc_t accumulate(typename Arith<AR>::acc_t acc, const uint4 a, const uint4 b) {
    if constexpr (AR == AR_F32_DOT || AR == AR_F32_L2) {
        const float ax = __uint_as_float(a.x), ay = __uint_as_float(a.y), az = __uint_as_float(a.z), aw = __uint_as_float(a.w);
        const float bx = __uint_as_float(b.x), by = __uint_as_float(b.y), bz = __uint_as_float(b.z), bw = __uint_as_float(b.w);
        if constexpr (AR == AR_F32_L2) {
            float dx = ax - bx, dy = ay - by, dz = az - bz, dw = aw - bw;
            acc = fmaf(dx, dx, acc);
            acc = fmaf(dy, dy, acc);
            acc = fmaf(dz, dz, acc);
            acc = fmaf(dw, dw, acc);
        } else {
            acc = fmaf(ax, bx, acc);
            acc = fmaf(ay, by, acc);
            acc = fmaf(az, bz, acc);
            acc = fmaf(aw, bw, acc);
        }
    } else if constexpr (AR == AR_I8) {
        acc = __builtin_amdgcn_sdot4((int)a.x, (int)b.x, acc, false);
        acc = __builtin_amdgcn_sdot4((int)a.y, (int)b.y, acc, false);
        acc = __builtin_amdgcn_sdot4((int)a.z, (int)b.z, acc, false);
        acc = __builtin_amdgcn_sdot4((int)a.w, (int)b.w, acc, false);
    } else if constexpr (AR == AR_B1) {
        acc += __popc(a.x ^ b.x) + __popc(a.y ^ b.y) + __popc(a.z ^ b.z) + __popc(a.w ^ b.w);
    } else {  // f16 / bf16: every element widened to f32, arithmetic in f32 (products are exact)
        float fa[8], fb[8];
        unpack8<AR>(a, fa);
        unpack8<AR>(b, fb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (Arith<AR>::is_l2) {
                float d = fa[j] - fb[j];
                acc = fmaf(d, d, acc);
            } else {
                acc = fmaf(fa[j], fb[j], acc);
            }
        }
    }
    return acc;
}

// usearch metric_cos_gt / l2sq / ip with the SimSIMD zero rules (oracle dist_cos): both zero -> 0, one zero
// or ab == 0 -> 1, clamp to [0, 2] so Distance::try_from (reference distance.rs:65-70) never rejects a
// result.  Float storage: aux = 1/|v| (cosine).  i8: aux = sum of squares; cos = 1 - ab/sqrt(a2*b2),
// l2sq = a2 + b2 - 2ab, ip = 1 - ab, all on the stored integers.  b1: the popcount itself.
template <int AR>
__device__ __forceinline__ float finalize(int metric, typename Arith<AR>::acc_t acc, float a_aux, float b_aux) {
    if constexpr (AR == AR_B1) {
        return (float)acc;
    } else if constexpr (AR == AR_I8) {
        const float ab = (float)acc;
        if (metric == L2SQ) return a_aux + b_aux - 2.0f * ab;
        if (metric == IP) return 1.0f - ab;
        if (a_aux == 0.f && b_aux == 0.f) return 0.f;
        if (a_aux == 0.f || b_aux == 0.f || acc == 0) return 1.f;
        float r = 1.0f - ab / (sqrtf(a_aux) * sqrtf(b_aux));
        return fminf(fmaxf(r, 0.f), 2.f);
    } else {
        if (metric == L2SQ) return acc;
        if (metric == IP) return 1.0f - acc;
        if (a_aux == 0.f && b_aux == 0.f) return 0.f;
        if (a_aux == 0.f || b_aux == 0.f || acc == 0.f) return 1.f;
        float r = 1.0f - acc * (a_aux * b_aux);
        return fminf(fmaxf(r, 0.f), 2.f);
    }
}

template <int AR>
__device__ __forceinline__ bool needs_aux(int metric) {
    return AR == AR_I8 || (AR != AR_B1 && metric == COS);
}

// A vector held in registers in STORAGE format: lane li of a `lanes`-wide group holds chunks li, li+lanes, ...
template <int AR, int I>
struct Query {
    uint4 c[I];
    float aux;  // see finalize
};

// Vector rows are read once per evaluation and are ~99 % of the traffic.  When the table is far larger than the
// caches (IndexView::nt_rows, set by the host above 2 GiB of vectors) they are loaded non-temporally so that they do
// not displace the adjacency rows and upper levels, which are re-used across queries, from L2 / Infinity Cache
// (10M x 768 f32: 24.4 -> 22.4 ms per 10,000 queries; a 0.5-1.5 GB table that partly lives in the 256 MB Infinity
// Cache loses 4 % with the hint, hence the threshold).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 load_row_chunk(const uint4* p) {
    if constexpr (NT) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}

template <int AR, int I>
__device__ __forceinline__ void query_from_row(const IndexView& ix, uint32_t slot, Query<AR, I>& q, int lane) {
    const uint32_t li = lane & (ix.lanes - 1);
    const uint4* row = ix.vectors + (size_t)slot * ix.stride4 + li;
#pragma unroll
    for (int i = 0; i < I; ++i) q.c[i] = row[(size_t)i * ix.lanes];
    q.aux = needs_aux<AR>(ix.metric) ? ix.aux[slot] : 0.f;
}

// Quantise `EPC` consecutive f32 elements starting at e0 into one chunk (the casts usearch applies when
// quantization != F32: f16 IEEE round-to-nearest, bf16 round-to-nearest-even, i8 = trunc(x*127/|x|) clamped
// to [-127,127], b1 = (x > 0), LSB first as reference usearch.rs:1179-1205).  `scale` is 127/|x| for i8.
template <int AR>
__device__ __forceinline__ uint4 quantise_chunk(const float* v, uint32_t e0, uint32_t dim, float mag) {
    auto at = [&](uint32_t e) { return e < dim ? v[e] : 0.f; };
    uint32_t w[4] = {0, 0, 0, 0};
    if constexpr (Arith<AR>::scalar == SC_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = __float_as_uint(at(e0 + j));
    } else if constexpr (Arith<AR>::scalar == SC_F16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = float_to_half_bits(at(e0 + 2 * j)) | (float_to_half_bits(at(e0 + 2 * j + 1)) << 16);
    } else if constexpr (Arith<AR>::scalar == SC_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = float_to_bf16_bits(at(e0 + 2 * j)) | (float_to_bf16_bits(at(e0 + 2 * j + 1)) << 16);
    } else if constexpr (Arith<AR>::scalar == SC_I8) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float t = mag > 0.f ? (at(e0 + j) * 127.0f) / mag : 0.f;
            t = fminf(fmaxf(t, -127.f), 127.f);
            w[j >> 2] |= ((uint32_t)(int)t & 0xFFu) << ((j & 3) * 8);
        }
    } else {
        for (int j = 0; j < 128; ++j)
            if (at(e0 + j) > 0.0f) w[j >> 5] |= 1u << (j & 31);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// |v| for the i8 cast, accumulated in f64 so that the value does not depend on the summation order
// (the oracle sums sequentially, the GPU across lanes): (float)sqrt(sum x^2).
__device__ __forceinline__ float magnitude_f64(const float* v, uint32_t dim, uint32_t lanes, uint32_t li) {
    double s = 0.0;
    for (uint32_t e = li; e < dim; e += lanes) s += (double)v[e] * (double)v[e];
    s = group_sum(s, lanes);
    return (float)sqrt(s);
}

// Sum of squares (float kinds: of the dequantised values, in f32; i8: of the bytes, exact) of the chunks in
// registers, reduced over the lane group -> the aux value of finalize().
template <int AR, int I>
__device__ __forceinline__ float aux_of(const IndexView& ix, const Query<AR, I>& q) {
    if constexpr (AR == AR_B1) {
        return 0.f;
    } else if constexpr (AR == AR_I8) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < I; ++i) s = accumulate<AR_I8>(s, q.c[i], q.c[i]);
        return (float)group_sum(s, ix.lanes);
    } else {
        constexpr int DOT = Arith<AR>::scalar == SC_F32 ? AR_F32_DOT : Arith<AR>::scalar == SC_F16 ? AR_F16_DOT : AR_BF16_DOT;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < I; ++i) s = accumulate<DOT>(s, q.c[i], q.c[i]);
        s = group_sum(s, ix.lanes);
        return s > 0.f ? 1.0f / sqrtf(s) : 0.f;
    }
}

// A caller-provided (unpadded, possibly unaligned) f32 vector -> storage format in registers.
template <int AR, int I>
__device__ __forceinline__ void query_from_f32(const IndexView& ix, const float* v, Query<AR, I>& q, int lane) {
    const uint32_t li = lane & (ix.lanes - 1);
    float mag = 0.f;
    if constexpr (AR == AR_I8) mag = magnitude_f64(v, ix.dim, ix.lanes, li);
#pragma unroll
    for (int i = 0; i < I; ++i) {
        q.c[i] = quantise_chunk<AR>(v, ((uint32_t)i * ix.lanes + li) * Arith<AR>::epc, ix.dim, mag);
        // keep each chunk's loads next to their use: without it the scheduler hoists all 16*I element loads
        // of the i8 cast and the kernel spills
        if constexpr (Arith<AR>::epc > 8) __builtin_amdgcn_sched_barrier(0);
    }
    q.aux = needs_aux<AR>(ix.metric) ? aux_of<AR, I>(ix, q) : 0.f;
}

// Distances from q to u_slot[0..m).  64/lanes vectors per wave-load, U wave-loads per group, and the
// groups software-pipelined two deep: while group g is reduced, the 16-byte loads of group g+1 are
// already in flight, so a wave keeps U*I..2*U*I loads outstanding through the whole phase
// (memory-level parallelism is what bounds this kernel: MI355X_MICROARCH "Indexed rows").
template <int I, int U>
struct RowGroup {
    uint4 buf[U][I];
    uint32_t slot[U];
    float aux[U];
};

// A TEAM of waves (one workgroup) may share one batch: wave-load L of wave w covers vectors
// ((L * TEAM + w) * V + grp); TEAM == 1, w == 0 is the one-wave-per-query layout.
template <int AR, int I, int U, int TEAM, bool NT>
__device__ __forceinline__ void group_issue(const IndexView& ix, RowGroup<I, U>& g, const uint32_t* u_slot, uint32_t m,
                                            uint32_t L, uint32_t w, uint32_t vshift, uint32_t grp, uint32_t li) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        uint32_t idx = (((L + (uint32_t)u) * (uint32_t)TEAM + w) << vshift) + grp;
        g.slot[u] = idx < m ? u_slot[idx] : kInvalid;
        g.aux[u] = 0.f;
        if (g.slot[u] != kInvalid) {
            const uint4* row = ix.vectors + (size_t)g.slot[u] * ix.stride4 + li;
#pragma unroll
            for (int i = 0; i < I; ++i) g.buf[u][i] = load_row_chunk<NT>(row + (size_t)i * ix.lanes);
            if (needs_aux<AR>(ix.metric)) g.aux[u] = ix.aux[g.slot[u]];
        } else {
#pragma unroll
            for (int i = 0; i < I; ++i) g.buf[u][i] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

template <int AR, int I, int U, int TEAM>
__device__ __forceinline__ void group_reduce(const IndexView& ix, const RowGroup<I, U>& g, const Query<AR, I>& q,
                                             float* u_dist, uint32_t L, uint32_t w, uint32_t vshift, uint32_t grp, uint32_t li) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        typename Arith<AR>::acc_t acc = 0;
#pragma unroll
        for (int i = 0; i < I; ++i) acc = accumulate<AR>(acc, q.c[i], g.buf[u][i]);
        acc = group_sum(acc, ix.lanes);
        if (g.slot[u] != kInvalid && li == 0) {
            // NaN (a non-finite query or stored vector) would break the total order every list relies on -- ranks collide and
            // leave holes with stale slot ids -- so it is ranked as +inf, behind everything finite
            const float d = finalize<AR>(ix.metric, acc, q.aux, g.aux[u]);
            u_dist[(((L + (uint32_t)u) * (uint32_t)TEAM + w) << vshift) + grp] = d == d ? d : __builtin_inff();
        }
    }
}

// NT: the cache policy of the row loads is a property of the kernel instance (a run-time choice inside the unrolled
// load loops, or two copies of this function in one kernel, cost registers and spill).
template <int AR, int I, int TEAM = 1, bool NT = false>
__device__ __forceinline__ void eval_batch(const IndexView& ix, const Query<AR, I>& q, const uint32_t* u_slot,
                                           float* u_dist, uint32_t m, int lane, uint32_t w = 0) {
    constexpr int U = I >= 12 ? 1 : (I >= 6 || (AR == AR_I8 && I >= 3)) ? 2 : 4;  // more loads per group at I <= 2 measured no gain; rows of
                                                                                  // 12 / 16 KiB (I = 12 / 16): one wave-load group of 12-16 KiB in flight per buffer
    const uint32_t lg = ix.lanes_log2;
    const uint32_t vshift = 6u - lg;  // V = 64 >> lg vectors per wave-load
    const uint32_t grp = (uint32_t)lane >> lg, li = (uint32_t)lane & (ix.lanes - 1);
    const uint32_t nl_all = (m + (1u << vshift) - 1u) >> vshift;  // wave-loads in the batch
    const uint32_t nl = TEAM == 1 ? nl_all : (nl_all > w ? (nl_all - w + (uint32_t)TEAM - 1u) / (uint32_t)TEAM : 0u);  // mine
    if (nl == 0) return;
    RowGroup<I, U> a, b;
    group_issue<AR, I, U, TEAM, NT>(ix, a, u_slot, m, 0, w, vshift, grp, li);
    uint32_t L = 0;
    for (;;) {
        if (L + U < nl) group_issue<AR, I, U, TEAM, NT>(ix, b, u_slot, m, L + U, w, vshift, grp, li);
        group_reduce<AR, I, U, TEAM>(ix, a, q, u_dist, L, w, vshift, grp, li);
        L += U;
        if (L >= nl) break;
        if (L + U < nl) group_issue<AR, I, U, TEAM, NT>(ix, a, u_slot, m, L + U, w, vshift, grp, li);
        group_reduce<AR, I, U, TEAM>(ix, b, q, u_dist, L, w, vshift, grp, li);
        L += U;
        if (L >= nl) break;
    }
}

// ---------------------------------------------------------------------------------------
// Per-wave LDS state of one beam search.
//   list     : the `top` buffer of usearch (ascending, <= ef entries) fused with `next`:
//              an entry's MSB says whether it has been expanded.  Entries that fall out
//              of the top-ef can never be expanded (their distance exceeds the radius and
//              the radius only shrinks), so dropping them is equivalent to the CPU heaps.
//   visited  : exact set of slots, NB buckets x 8 x 16-bit tags.  (slot * odd) mod
//              2^(16+log2 NB) is a bijection, so (bucket, tag) identifies the slot as long
//              as capacity <= 2^(16+log2 NB) (enforced by the host).
// ---------------------------------------------------------------------------------------
constexpr int kOvf = 128;

template <bool ON>
struct SelArrays {  // heuristic output (insert / link kernels only)
    uint32_t sel_s[64];
    float sel_d[64];
    uint32_t tie_salt;  // insert kernel: order among equal distances is pseudo-random per new node (see key_less_in)
    uint32_t tie_newest;  // 1: usearch's own order instead -- a newly found entry precedes the equal ones already listed
};
template <>
struct SelArrays<false> {};

// TEAM > 1: a workgroup of TEAM waves serves ONE query (small batches, where most of the chip would idle):
// wave 0 walks the graph, all waves evaluate each hop's neighbour batch.  team_m is the mailbox.
constexpr uint32_t kTeamExit = 0xFFFFFFFFu;
template <int TM>
struct TeamBox {
    uint32_t team_m;
    uint32_t team_q;  // kInvalid: distances from the walker's own query to u_slot[]; else from stored row team_q to sel_s[] (refine)
};
template <>
struct TeamBox<1> {};

// Wide tags: 4 more tag bits per visited entry (one dword of nibbles per bucket), for indexes whose slot ids the
// 15 / 16-bit tags cannot tell apart: 2^26 -> 2^30 slots (one choice), 2^25 -> 2^29 (two choices).  +4 B per bucket.
template <bool WIDE, int NB>
struct VisitedHi {
    uint32_t vis_hi[NB];
};
template <int NB>
struct VisitedHi<false, NB> {};

// EFCAP=128, NB=1024, no SelArrays: 20,480 B -> 8 single-wave workgroups per CU (160 KiB LDS).
template <int EFCAP, int NB, bool SEL = false, int CH = 1, int TM = 1, bool NT = false, bool WT = false>
struct BeamShared : SelArrays<SEL>, TeamBox<TM>, VisitedHi<WT, NB> {
    static constexpr bool kWideTags = WT;
    static constexpr bool kNT = NT;  // walk evaluations load vector rows non-temporally (IndexView::nt_rows, host-chosen)
    static constexpr int kChoices = CH;
    static constexpr int kEfCap = EFCAP;
    static constexpr int kNB = NB;
    static constexpr int kBucket = 8;  // tags per bucket of the visited table
    static constexpr int kTeam = TM;
    static constexpr bool kSel = SEL;
    static constexpr uint32_t kOvfCap = (uint32_t)kOvf - 2u;  // entries of vis_ovf
    float lst_d[1][EFCAP];  // one buffer: list_merge works in place
    uint32_t lst_s[1][EFCAP];
    alignas(16) uint16_t vis_tag[NB * 8];
    uint32_t vis_cnt[NB / 4];  // one byte per bucket
    uint32_t vis_ovf[kOvf - 2];
    uint32_t ovf_cnt;
    uint32_t overflowed;
    uint32_t u_slot[64];
    float u_dist[64];
};

// The order of a kernel's candidate list.  Search (and everything that reports results): (distance, slot).  Insert: among
// EQUAL distances a pseudo-random order per new node -- with "lowest slot first" every node of a sub-batch (they all see
// the same frozen graph) picks the same members of a group of exact duplicates, whose later copies then get no links
// at all; multiplying by an odd constant is a bijection on the 30 slot bits, so this is still a strict total order.
template <class Sh>
__device__ __forceinline__ bool key_less_in(const Sh& sh, float ad, uint32_t as, float bd, uint32_t bs) {
    if constexpr (Sh::kSel) {
        const uint32_t ta = (((as & kSlotMask) ^ sh.tie_salt) * 0x85EBCA6Bu) & kSlotMask;
        const uint32_t tb = (((bs & kSlotMask) ^ sh.tie_salt) * 0x85EBCA6Bu) & kSlotMask;
        return ad < bd || (ad == bd && ta < tb);
    } else {
        return key_less(ad, as, bd, bs);
    }
}

// LDS hand-over between the lanes of the walking wave.  One wave per workgroup: the workgroup barrier (free).
// Team: only a wave-level ordering point -- the helper waves are parked at the team barrier and must not be released.
template <class Sh>
__device__ __forceinline__ void wsync() {
    if constexpr (Sh::kTeam == 1) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// Distances of sh.u_slot[0..m) -> sh.u_dist, visible to every lane of the walking wave on return.
template <int AR, int I, class Sh>
__device__ __forceinline__ void eval_shared(const IndexView& ix, const Query<AR, I>& q, Sh& sh, uint32_t m, int lane) {
    if constexpr (Sh::kTeam == 1) {
        eval_batch<AR, I, 1, Sh::kNT>(ix, q, sh.u_slot, sh.u_dist, m, lane);
        __syncthreads();
    } else {
        if (lane == 0) {
            sh.team_m = m;
            sh.team_q = kInvalid;
        }
        __syncthreads();  // releases the helpers (see team_helper_loop)
        eval_batch<AR, I, Sh::kTeam, Sh::kNT>(ix, q, sh.u_slot, sh.u_dist, m, lane, 0);
        __syncthreads();  // every wave's distances are in LDS
    }
}

// refine's inner step: distances from stored row `cs` (already in `cv`) to sh.sel_s[0..m) -> sh.u_dist.
template <int AR, int I, class Sh>
__device__ __forceinline__ void eval_selected(const IndexView& ix, uint32_t cs, const Query<AR, I>& cv, Sh& sh, uint32_t m, int lane) {
    if constexpr (Sh::kTeam == 1) {
        eval_batch<AR, I>(ix, cv, sh.sel_s, sh.u_dist, m, lane);
        __syncthreads();
    } else {
        if (lane == 0) {
            sh.team_m = m;
            sh.team_q = cs;
        }
        __syncthreads();
        eval_batch<AR, I, Sh::kTeam>(ix, cv, sh.sel_s, sh.u_dist, m, lane, 0);
        __syncthreads();
    }
}

// Waves 1..TEAM-1 of a team workgroup: evaluate their share of every batch until the walker says stop.
template <int AR, int I, class Sh>
__device__ __forceinline__ void team_helper_loop(const IndexView& ix, const Query<AR, I>& q, Sh& sh, int lane, uint32_t w) {
    for (;;) {
        __syncthreads();
        const uint32_t m = sh.team_m;
        if (m == kTeamExit) return;
        if constexpr (Sh::kSel) {  // insert kernel: the heuristic measures from a stored row against sel_s[]
            const uint32_t qs = sh.team_q;
            const uint32_t vshift = 6u - ix.lanes_log2;
            const bool mine = w < ((m + (1u << vshift) - 1u) >> vshift);  // only waves that get a share fetch the row
            Query<AR, I> use = q;
            if (qs != kInvalid && mine) query_from_row<AR, I>(ix, qs, use, lane);
            const uint32_t* list = qs == kInvalid ? sh.u_slot : sh.sel_s;
            if (mine) eval_batch<AR, I, Sh::kTeam>(ix, use, list, sh.u_dist, m, lane, w);
        } else {
            {
                eval_batch<AR, I, Sh::kTeam>(ix, q, sh.u_slot, sh.u_dist, m, lane, w);
            }
        }
        __syncthreads();
    }
}

template <class Sh>
__device__ __forceinline__ void team_release(Sh& sh, int lane) {  // walker: no more batches
    if constexpr (Sh::kTeam > 1) {
        if (lane == 0) sh.team_m = kTeamExit;
        __syncthreads();
    }
}

template <int NB, int CH = 1, bool WT = false>
struct VisitedCfg {
    static constexpr int log2nb = NB == 256 ? 8 : NB == 512 ? 9 : NB == 1024 ? 10 : NB == 2048 ? 11 : 12;
    static constexpr uint32_t tag_bits = CH == 2 ? 15 : 16;  // two-choice: bit 15 marks "stored in its alternate bucket"
    static constexpr uint32_t hi_bits = WT ? 4 : 0;          // wide tags: kept in vis_hi, one nibble per entry
    static constexpr uint32_t domain_bits = tag_bits + hi_bits + log2nb;
    static constexpr uint32_t domain_mask = (1u << domain_bits) - 1u;
};

template <class Sh>
__device__ __forceinline__ void visited_clear(Sh& sh, int lane) {
    for (int i = lane; i < Sh::kNB / 4; i += kWave) sh.vis_cnt[i] = 0;
    if constexpr (Sh::kWideTags)
        for (int i = lane; i < Sh::kNB; i += kWave) sh.vis_hi[i] = 0;  // nibbles are OR-ed in
    if (lane == 0) {
        sh.ovf_cnt = 0;
        sh.overflowed = 0;
    }
}

// Tags of one bucket that match `want` among its first min(cnt, 8) entries.
template <class Sh>
__device__ __forceinline__ bool bucket_has(const Sh& sh, uint32_t b, uint32_t cnt, uint32_t want, uint32_t want_hi) {
    constexpr int BS = Sh::kBucket;  // 8 tags = one 16-byte read; 12 tags (the dense table of the walk) = three 8-byte reads
    static_assert(BS == 8 || (BS == 12 && !Sh::kWideTags), "bucket sizes");
    uint32_t w[BS / 2];
    if constexpr (BS == 8) {
        const uint4 t4 = *reinterpret_cast<const uint4*>(&sh.vis_tag[b * 8]);
        w[0] = t4.x; w[1] = t4.y; w[2] = t4.z; w[3] = t4.w;
    } else {
#pragma unroll
        for (int i = 0; i < BS / 4; ++i) {
            const uint2 t2 = *reinterpret_cast<const uint2*>(&sh.vis_tag[b * BS + 4 * i]);
            w[2 * i] = t2.x;
            w[2 * i + 1] = t2.y;
        }
    }
    const uint32_t n = cnt < (uint32_t)BS ? cnt : (uint32_t)BS;
    uint32_t hw = 0;
    if constexpr (Sh::kWideTags) hw = sh.vis_hi[b];
    bool found = false;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
        uint32_t tj = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
        bool eq = tj == want;
        if constexpr (Sh::kWideTags) eq = eq && ((hw >> (4 * j)) & 15u) == want_hi;
        found |= ((uint32_t)j < n) && eq;
    }
    return found;
}

// Per-lane test-and-set; returns true when `slot` was already in the set.
// CH == 2 (two-choice): a slot may live in bucket b1 (tag as is) or in b2 = b1 ^ alt(tag) (tag | 0x8000);
// it goes to the emptier one, which keeps 8-entry buckets overflow-free up to ~85 % load.
// (bucket, stored tag) still identifies the slot: b1 is recovered from b2 and the tag.
template <class Sh>
__device__ __forceinline__ bool visited_test_and_set(Sh& sh, uint32_t slot) {
    constexpr int NB = Sh::kNB, CH = Sh::kChoices;
    using C = VisitedCfg<NB, CH, Sh::kWideTags>;
    const uint32_t m = (slot * 0x9E3779B1u) & C::domain_mask;
    const uint32_t b1 = m >> (C::tag_bits + C::hi_bits);
    const uint32_t tag = m & ((1u << C::tag_bits) - 1u);
    const uint32_t hi = (m >> C::tag_bits) & ((1u << C::hi_bits) - 1u);
    const uint32_t s1 = (b1 & 3u) * 8u;
    const uint32_t c1 = (sh.vis_cnt[b1 >> 2] >> s1) & 0xFFu;
    bool found = bucket_has(sh, b1, c1, tag, hi);
    uint32_t b = b1, sb = s1, cb = c1, stored = tag;
    if (CH == 2) {
        const uint32_t alt = ((tag * 0x5BD1u) >> 3) & (uint32_t)(NB - 1);
        const uint32_t b2 = b1 ^ alt;
        if (b2 != b1) {
            const uint32_t s2 = (b2 & 3u) * 8u;
            const uint32_t c2 = (sh.vis_cnt[b2 >> 2] >> s2) & 0xFFu;
            found |= bucket_has(sh, b2, c2, tag | 0x8000u, hi);
            if (c2 < c1) {
                b = b2;
                sb = s2;
                cb = c2;
                stored = tag | 0x8000u;
            }
        }
    }
    constexpr uint32_t ovf_cap = Sh::kOvfCap;
    if (!found) {
        // Members that found their bucket full live in the overflow list.  With two choices a member can land there
        // while its OTHER bucket still has room (several lanes of one hop pick the same nearly-full bucket from the
        // counts they read before any of them inserted), so the list is searched whenever it is not empty, not only
        // when both buckets are full -- else such a member is evaluated, and listed, twice.
        uint32_t oc = sh.ovf_cnt < ovf_cap ? sh.ovf_cnt : ovf_cap;
        for (uint32_t j = 0; j < oc; ++j) found |= sh.vis_ovf[j] == slot;
    }
    if (found) return true;
    // a full bucket is not incremented further: the byte counter can never wrap
    constexpr uint32_t BS = (uint32_t)Sh::kBucket;
    uint32_t pos = cb >= BS ? BS : (atomicAdd(&sh.vis_cnt[b >> 2], 1u << sb) >> sb) & 0xFFu;
    if (pos < BS) {
        sh.vis_tag[b * BS + pos] = (uint16_t)stored;
        if constexpr (Sh::kWideTags) atomicOr(&sh.vis_hi[b], hi << (4u * pos));
        return false;
    }
    uint32_t o = atomicAdd(&sh.ovf_cnt, 1u);
    if (o < ovf_cap) {
        sh.vis_ovf[o] = slot;
        return false;
    }
    // Table exhausted (counted in stats[6]): report "fresh".  The node may then be evaluated a second
    // time; beam_search drops it again by an exact (distance, slot) lookup in the sorted list, so
    // results never contain duplicates and no node is ever lost -- only evaluations are repeated.
    sh.overflowed = 1;
    return false;
}

// The lookup half of visited_test_and_set: is `slot` in the set?  Nothing is inserted (speculative evaluation must not mark).
template <class Sh>
__device__ __forceinline__ bool visited_contains(const Sh& sh, uint32_t slot) {
    constexpr int NB = Sh::kNB, CH = Sh::kChoices;
    using C = VisitedCfg<NB, CH, Sh::kWideTags>;
    const uint32_t m = (slot * 0x9E3779B1u) & C::domain_mask;
    const uint32_t b1 = m >> (C::tag_bits + C::hi_bits);
    const uint32_t tag = m & ((1u << C::tag_bits) - 1u);
    const uint32_t hi = (m >> C::tag_bits) & ((1u << C::hi_bits) - 1u);
    const uint32_t c1 = (sh.vis_cnt[b1 >> 2] >> ((b1 & 3u) * 8u)) & 0xFFu;
    bool found = bucket_has(sh, b1, c1, tag, hi);
    if (CH == 2) {
        const uint32_t alt = ((tag * 0x5BD1u) >> 3) & (uint32_t)(NB - 1);
        const uint32_t b2 = b1 ^ alt;
        if (b2 != b1) {
            const uint32_t c2 = (sh.vis_cnt[b2 >> 2] >> ((b2 & 3u) * 8u)) & 0xFFu;
            found |= bucket_has(sh, b2, c2, tag | 0x8000u, hi);
        }
    }
    if (!found) {
        constexpr uint32_t ovf_cap = Sh::kOvfCap;
        const uint32_t oc = sh.ovf_cnt < ovf_cap ? sh.ovf_cnt : ovf_cap;
        for (uint32_t j = 0; j < oc; ++j) found |= sh.vis_ovf[j] == slot;
    }
    return found;
}

__device__ __forceinline__ const uint32_t* adjacency(const IndexView& ix, uint32_t slot, int level, uint32_t& cap) {
    if (level == 0) {
        cap = ix.M0;
        return ix.adj0 + (size_t)slot * ix.M0;
    }
    cap = ix.M;
    return ix.upper + ((size_t)ix.upper_off[slot] + (uint32_t)(level - 1)) * ix.M;
}

// usearch search_for_one_: greedy walk on levels (from_level .. to_level+1].
template <int AR, int I, class Sh>
__device__ uint32_t greedy_descent(const IndexView& ix, Sh& sh, const Query<AR, I>& q,
                                   uint32_t start, int from_level, int to_level, Counters& cnt, int lane, float* out_d = nullptr) {
    uint32_t cur = start;
    if (lane == 0) sh.u_slot[0] = cur;
    wsync<Sh>();
    eval_shared<AR, I>(ix, q, sh, 1, lane);
    float cur_d = sh.u_dist[0];
    cnt.evals += 1;
    for (int level = from_level; level > to_level; --level) {
        for (;;) {
            uint32_t cap;
            const uint32_t* row = adjacency(ix, cur, level, cap);
            uint32_t n = (uint32_t)lane < cap ? row[lane] : kInvalid;
            uint64_t mask = __ballot(n != kInvalid);
            uint32_t m = (uint32_t)__popcll(mask);
            wsync<Sh>();
            if (n != kInvalid) sh.u_slot[mbcnt(mask)] = n;
            wsync<Sh>();
            eval_shared<AR, I>(ix, q, sh, m, lane);
            cnt.evals += m;
            cnt.hops += 1;
            float d = (uint32_t)lane < m ? sh.u_dist[lane] : __builtin_inff();
            uint32_t idx = (uint32_t)lane;
            // argmin over (d, position): first minimum in adjacency order, as the CPU's strict '<' scan.
            for (int o = 32; o; o >>= 1) {
                float od = __shfl_xor(d, o);
                uint32_t oi = (uint32_t)__shfl_xor((int)idx, o);
                if (od < d || (od == d && oi < idx)) {
                    d = od;
                    idx = oi;
                }
            }
            if (m && d < cur_d) {
                cur_d = d;
                cur = sh.u_slot[idx];
            } else {
                break;
            }
        }
    }
    if (out_d) *out_d = cur_d;
    return cur;
}

// Merge m new (distance, slot) pairs held by lanes 0..m-1 into the sorted list (size sz), keeping the
// ef best.  Rank-based and in place: every lane first reads what it owns into registers and computes
// final positions (binary search among the old entries, all-pairs among the <= 64 new ones), then,
// after a barrier, everything is scattered to its final position.  O(m + log sz) per lane, no
// data-dependent divergence.  Returns the new size.  `cur` is always 0 (kept for call-site symmetry).
template <class Sh>
__device__ __forceinline__ uint32_t list_merge(Sh& sh, int cur, uint32_t sz, uint32_t ef,
                                               float nd, uint32_t ns, uint32_t m, int lane) {
    constexpr int EFCAP = Sh::kEfCap;
    float* od = sh.lst_d[cur];
    uint32_t* os = sh.lst_s[cur];
    constexpr int R = EFCAP / kWave;
    nd = nd == nd ? nd : __builtin_inff();  // see group_reduce: (distance, slot) must be a total order
    // own old entries -> registers
    float keep_d[R];
    uint32_t keep_s[R], keep_p[R], shift[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
        keep_d[r] = p < sz ? od[p] : __builtin_inff();
        keep_s[r] = p < sz ? os[p] : kInvalid;
        shift[r] = 0;
    }
    // One pass over the new elements (wave-uniform j, broadcast by v_readlane): its rank among the old
    // entries by ballot + popcount, among the new ones by comparison; every old entry counts the new
    // elements that precede it.  No dependent LDS chain (a binary search costs log2(ef) LDS latencies).
    uint32_t r_old = 0, r_new = 0;
    for (uint32_t j = 0; j < m; ++j) {
        const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)j));
        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)ns, (int)j);
        uint32_t below = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool valid = (uint32_t)lane + (uint32_t)r * kWave < sz;
            bool less;
            if constexpr (Sh::kSel) less = valid && (sh.tie_newest ? keep_d[r] < dj : key_less_in(sh, keep_d[r], keep_s[r], dj, sj));
            else less = valid && key_less_in(sh, keep_d[r], keep_s[r], dj, sj);
            below += (uint32_t)__popcll(__ballot(less));
            shift[r] += (valid && !less) ? 1u : 0u;
        }
        if ((uint32_t)lane == j) r_old = below;
        bool before;  // new element j precedes this lane's new element
        if constexpr (Sh::kSel) before = sh.tie_newest ? (dj < nd || (dj == nd && j > (uint32_t)lane)) : key_less_in(sh, dj, sj, nd, ns);
        else before = key_less_in(sh, dj, sj, nd, ns);
        r_new += ((uint32_t)lane < m && before) ? 1u : 0u;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
        keep_p[r] = (p < sz && p + shift[r] < ef) ? p + shift[r] : kInvalid;
    }
    wsync<Sh>();  // every read of the old list is done
    if ((uint32_t)lane < m) {
        uint32_t pos = r_old + r_new;
        if (pos < ef) {
            od[pos] = nd;
            os[pos] = ns;
        }
    }
#pragma unroll
    for (int r = 0; r < EFCAP / kWave; ++r)
        if (keep_p[r] != kInvalid) {
            od[keep_p[r]] = keep_d[r];
            os[keep_p[r]] = keep_s[r];
        }
    uint32_t nsz = sz + m;
    return nsz < ef ? nsz : ef;
}

// usearch search_to_insert_ / search_to_find_in_base_ (unfiltered): beam search on one level.
// On return the sorted candidates are in sh.lst_*[cur] (cur returned through `out_cur`).
// `self` (or kInvalid): slot that is never evaluated, expanded nor returned.
template <int AR, int I, class Sh>
__device__ uint32_t beam_search(const IndexView& ix, Sh& sh, const Query<AR, I>& q,
                                uint32_t start, int level, uint32_t ef, uint32_t self, Counters& cnt, int lane,
                                int& out_cur, bool tomb = false) {
    constexpr int EFCAP = Sh::kEfCap;
    visited_clear(sh, lane);
    wsync<Sh>();
    int cur = 0;
    uint32_t sz = 0;
    uint32_t live = 0;  // members of the list that can be results (== sz unless the index has removed members)
    if (lane == 0) {
        if (self != kInvalid) visited_test_and_set(sh, self);
        if (start != self) visited_test_and_set(sh, start);
        sh.u_slot[0] = start;
    }
    wsync<Sh>();
    if (start != self) {
        eval_shared<AR, I>(ix, q, sh, 1, lane);
        cnt.evals += 1;
        const bool start_dead = tomb && ix.keys[start] == kFreeKey;
        if (lane == 0) {
            sh.lst_d[0][0] = sh.u_dist[0];
            sh.lst_s[0][0] = start | (start_dead ? kDead : 0u);
        }
        sz = 1;
        live = start_dead ? 0 : 1;
    }
    wsync<Sh>();
    // Adjacency prefetch: while hop h evaluates its neighbours, the row of the runner-up candidate is
    // already on its way; it is used when that candidate is indeed expanded next (no closer one arrived).
    uint32_t pf_slot = kInvalid, pf_n = kInvalid;
    for (;;) {
        // closest unexpanded entry (and the runner-up, for the prefetch)
        int pick = -1, pick2 = -1;
#pragma unroll
        for (int r = 0; r < EFCAP / kWave; ++r) {
            uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
            bool un = p < sz && !(sh.lst_s[cur][p] & kExpanded);
            uint64_t mask = __ballot(un);
            if (pick < 0 && mask) {
                pick = r * kWave + (int)__builtin_ctzll(mask);
                mask &= mask - 1;
            }
            if (pick >= 0 && pick2 < 0 && mask) pick2 = r * kWave + (int)__builtin_ctzll(mask);
        }
        if (pick < 0) break;
        const uint32_t c_entry = sh.lst_s[cur][pick];
        const uint32_t c_slot = c_entry & kSlotMask;
        uint32_t c2_slot = pick2 >= 0 ? (sh.lst_s[cur][pick2] & kSlotMask) : kInvalid;
        wsync<Sh>();
        if (lane == 0) sh.lst_s[cur][pick] = c_entry | kExpanded;
        cnt.hops += 1;
        // neighbours: one id per lane, exact visited test-and-set, compaction
        uint32_t cap;
        const uint32_t* row = adjacency(ix, c_slot, level, cap);
        uint32_t n;
        if (c_slot == pf_slot) {
            n = pf_n;
        } else {
            n = (uint32_t)lane < cap ? row[lane] : kInvalid;
        }
        pf_slot = c2_slot;
        if (c2_slot != kInvalid) {
            uint32_t cap2;
            const uint32_t* row2 = adjacency(ix, c2_slot, level, cap2);
            pf_n = (uint32_t)lane < cap2 ? row2[lane] : kInvalid;
        }
        // connectivity above 32: a level-0 row holds up to 128 ids -- two per lane -- and is taken 64 at a time, in adjacency order
        // (the CPU loop takes the neighbours one at a time, so two batches are as equivalent to it as one)
        const uint32_t n_hi = cap > (uint32_t)kWave && (uint32_t)kWave + (uint32_t)lane < cap ? row[kWave + lane] : kInvalid;
        for (uint32_t half = 0; half < (cap > (uint32_t)kWave ? 2u : 1u); ++half) {
        if (half) n = n_hi;
        bool fresh = n != kInvalid && !visited_test_and_set(sh, n);
        uint64_t fmask = __ballot(fresh);
        uint32_t m = (uint32_t)__popcll(fmask);
        if (fresh) sh.u_slot[mbcnt(fmask)] = n;
        wsync<Sh>();
        if (m == 0) continue;
        eval_shared<AR, I>(ix, q, sh, m, lane);
        cnt.evals += m;
        // admission: top not full, or closer than the current radius (usearch: `top.size() < top_limit || d < radius`)
        float nd = (uint32_t)lane < m ? sh.u_dist[lane] : __builtin_inff();
        uint32_t ns = (uint32_t)lane < m ? sh.u_slot[lane] : kInvalid;
        bool admit = (uint32_t)lane < m;
        if (tomb) {
            // usearch: `top` holds ef LIVE members; removed ones only live in `next`.  Here both share the
            // list, which is cut right after its ef-th live entry, so that entry's distance is the radius.
            if (admit && ix.keys[ns] == kFreeKey) ns |= kDead;
            if (live >= ef) admit = admit && nd < sh.lst_d[cur][sz - 1];
        } else if (sz + m > ef) {
            // only elements that can land inside the top-ef matter; prune against the radius when full
            if (sz == ef) admit = admit && nd < sh.lst_d[cur][ef - 1];
        }
        if (sh.overflowed) {  // wave-uniform; see visited_test_and_set: re-evaluated nodes are dropped here
            bool positional = false;
            if constexpr (Sh::kSel) positional = sh.tie_newest != 0;
            uint32_t lo = 0, hi = admit ? sz : 0;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                const bool lt = positional ? sh.lst_d[cur][mid] < nd : key_less_in(sh, sh.lst_d[cur][mid], sh.lst_s[cur][mid], nd, ns);
                if (lt) lo = mid + 1; else hi = mid;
            }
            if (positional) {  // equal distances are not ordered by slot: scan the run
                for (; admit && lo < sz && sh.lst_d[cur][lo] == nd; ++lo)
                    if ((sh.lst_s[cur][lo] & kSlotMask) == (ns & kSlotMask)) admit = false;
            } else if (admit && lo < sz && (sh.lst_s[cur][lo] & kSlotMask) == (ns & kSlotMask) && sh.lst_d[cur][lo] == nd) {
                admit = false;
            }
        }
        uint64_t amask = __ballot(admit);
        uint32_t ma = (uint32_t)__popcll(amask);
        if (ma == 0) continue;
        wsync<Sh>();
        if (admit) {
            uint32_t r = mbcnt(amask);
            sh.u_dist[r] = nd;
            sh.u_slot[r] = ns;
        }
        wsync<Sh>();
        nd = (uint32_t)lane < ma ? sh.u_dist[lane] : __builtin_inff();
        ns = (uint32_t)lane < ma ? sh.u_slot[lane] : kInvalid;
        sz = list_merge(sh, cur, sz, tomb ? (uint32_t)EFCAP : ef, nd, ns, ma, lane);
        wsync<Sh>();
        if (tomb) {  // cut after the ef-th live entry
            uint32_t cum = 0, cut = sz;
            bool found = false;
#pragma unroll
            for (int r = 0; r < EFCAP / kWave; ++r) {
                uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
                bool lv = p < sz && !(sh.lst_s[cur][p] & kDead);
                uint64_t mask = __ballot(lv);
                uint32_t c = (uint32_t)__popcll(mask);
                if (!found && cum + c >= ef) {
                    uint32_t need = ef - cum;  // the need-th (1-based) live entry of this round
                    for (uint32_t i = 1; i < need; ++i) mask &= mask - 1;
                    cut = (uint32_t)r * kWave + (uint32_t)__builtin_ctzll(mask) + 1u;
                    found = true;
                }
                cum += c;
            }
            live = found ? ef : cum;
            sz = cut;
        } else {
            live = sz;
        }
        }  // half
    }
    if (sh.overflowed) cnt.overflow += 1;
    out_cur = cur;
    return sz;
}


// usearch refine_: neighbour-selection heuristic over the sorted candidates in
// sh.lst_*[cur][0..sz).  Accept c iff for every already accepted a: d(c, a) >= d(c, centre).
// Result in sh.sel_s / sh.sel_d (ascending); returns the number selected (<= needed).
template <int AR, int I, class Sh>
__device__ uint32_t refine(const IndexView& ix, Sh& sh, int cur, uint32_t sz, uint32_t needed,
                           Counters& cnt, int lane) {
    if (sz < needed || sz == 0) {
        wsync<Sh>();
        for (uint32_t e = (uint32_t)lane; e < sz; e += kWave) {
            sh.sel_s[e] = sh.lst_s[cur][e] & kSlotMask;
            sh.sel_d[e] = sh.lst_d[cur][e];
        }
        wsync<Sh>();
        return sz;
    }
    wsync<Sh>();
    if (lane == 0) {
        sh.sel_s[0] = sh.lst_s[cur][0] & kSlotMask;
        sh.sel_d[0] = sh.lst_d[cur][0];
    }
    wsync<Sh>();
    uint32_t nsel = 1;
    for (uint32_t c = 1; c < sz && nsel < needed; ++c) {
        const uint32_t cs = sh.lst_s[cur][c] & kSlotMask;
        const float cd = sh.lst_d[cur][c];
        Query<AR, I> cv;
        query_from_row<AR, I>(ix, cs, cv, lane);
        eval_selected<AR, I>(ix, cs, cv, sh, nsel, lane);
        cnt.evals += nsel;
        bool bad = false;
        for (uint32_t b0 = 0; b0 < nsel; b0 += kWave) bad = bad || (b0 + (uint32_t)lane < nsel && sh.u_dist[b0 + lane] < cd);  // (more than 64 accepted: link kernel, connectivity above 32)
        bool reject = __ballot(bad) != 0ull;
        wsync<Sh>();
        if (!reject) {
            if (lane == 0) {
                sh.sel_s[nsel] = cs;
                sh.sel_d[nsel] = cd;
            }
            ++nsel;
            wsync<Sh>();
        }
    }
    return nsel;
}

}  // namespace vs
