// kernels_misc.hip -- row staging, norms, exact (brute-force) top-k and the multi-GPU top-k merge.
#include <atomic>

#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "filter_rounds.hpp"

namespace vs {

// ---------------------------------------------------------------- row staging / small utilities
// f32 rows (dim floats, src_stride apart) -> storage rows (quantised, zero padded) + aux.  One wave per row.
template <int AR>
__global__ void quantise_rows_kernel(IndexView ix, uint4* vectors, float* aux, const float* src, uint32_t src_stride,
                                     const uint32_t* slots, uint32_t first, uint32_t n) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= n) return;
    const uint32_t slot = slots ? slots[w] : first + w;
    const float* v = src + (size_t)w * src_stride;
    float mag = 0.f;
    if constexpr (AR == AR_I8) mag = magnitude_f64(v, ix.dim, 64, (uint32_t)lane);
    uint4* row = vectors + (size_t)slot * ix.stride4;
    typename Arith<AR>::acc_t sq = 0;
    constexpr int SELF = Arith<AR>::scalar == SC_F32 ? AR_F32_DOT : Arith<AR>::scalar == SC_F16 ? AR_F16_DOT
                         : Arith<AR>::scalar == SC_BF16 ? AR_BF16_DOT : AR;
    for (uint32_t c = lane; c < ix.stride4; c += kWave) {
        uint4 q = quantise_chunk<AR>(v, c * Arith<AR>::epc, ix.dim, mag);
        row[c] = q;
        if constexpr (AR != AR_B1) sq = accumulate<SELF>(sq, q, q);
    }
    if (needs_aux<AR>(ix.metric)) {
        sq = group_sum(sq, 64);
        if (lane == 0) {
            if constexpr (AR == AR_I8) aux[slot] = (float)sq;
            else aux[slot] = (float)sq > 0.f ? 1.0f / sqrtf((float)sq) : 0.f;
        }
    }
}

#define VS_AR_SWITCH(ar, CALL)                       \
    switch (ar) {                                    \
        case AR_F32_DOT: case AR_F32_L2: { constexpr int A = AR_F32_DOT; CALL; } break;   \
        case AR_F16_DOT: case AR_F16_L2: { constexpr int A = AR_F16_DOT; CALL; } break;   \
        case AR_BF16_DOT: case AR_BF16_L2: { constexpr int A = AR_BF16_DOT; CALL; } break; \
        case AR_I8: { constexpr int A = AR_I8; CALL; } break;                              \
        default: { constexpr int A = AR_B1; CALL; } break;                                 \
    }

hipError_t launch_quantise_rows(const IndexView& ix, uint4* vectors, float* aux, const float* src, uint32_t src_stride,
                                const uint32_t* slots, uint32_t first, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    const int ar = arith_of(ix.scalar, ix.metric);
    VS_AR_SWITCH(ar, hipLaunchKernelGGL((quantise_rows_kernel<A>), dim3((n + 3) / 4), dim3(256), 0, s, ix, vectors, aux, src,
                                        src_stride, slots, first, n))
    return hipGetLastError();
}

// aux of rows that are already in storage format (import path).
template <int AR>
__global__ void aux_rows_kernel(IndexView ix, float* aux, uint32_t n) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= n) return;
    const uint4* row = ix.vectors + (size_t)w * ix.stride4;
    typename Arith<AR>::acc_t sq = 0;
    for (uint32_t c = lane; c < ix.stride4; c += kWave) sq = accumulate<AR>(sq, row[c], row[c]);
    sq = group_sum(sq, 64);
    if (lane == 0) {
        if constexpr (AR == AR_I8) aux[w] = (float)sq;
        else aux[w] = (float)sq > 0.f ? 1.0f / sqrtf((float)sq) : 0.f;
    }
}

hipError_t launch_aux_rows(const IndexView& ix, float* aux, uint32_t n, hipStream_t s) {
    const int ar = arith_of(ix.scalar, ix.metric);
    if (!n || ar == AR_B1) return hipSuccess;
    VS_AR_SWITCH(ar, hipLaunchKernelGGL((aux_rows_kernel<A>), dim3((n + 3) / 4), dim3(256), 0, s, ix, aux, n))
    return hipGetLastError();
}

// storage rows <-> unpadded raw rows (export / import): row_bytes payload, stride_bytes padded row.
__global__ void copy_rows_kernel(uint8_t* dst, uint32_t dst_stride, const uint8_t* src, uint32_t src_stride,
                                 uint32_t row_bytes, uint32_t dst_fill_to) {
    const uint32_t r = blockIdx.x;
    uint8_t* d = dst + (size_t)r * dst_stride;
    const uint8_t* sp = src + (size_t)r * src_stride;
    for (uint32_t c = threadIdx.x; c < dst_fill_to; c += blockDim.x) d[c] = c < row_bytes ? sp[c] : (uint8_t)0;
}

hipError_t launch_copy_rows(void* dst, uint32_t dst_stride, const void* src, uint32_t src_stride, uint32_t row_bytes,
                            uint32_t dst_fill_to, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(n), dim3(256), 0, s, (uint8_t*)dst, dst_stride, (const uint8_t*)src, src_stride,
                       row_bytes, dst_fill_to);
    return hipGetLastError();
}

__global__ void fill_u32_kernel(uint32_t* p, uint32_t v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) p[i] = v;
}

hipError_t launch_fill_u32(uint32_t* p, uint32_t value, size_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, p, value, n);
    return hipGetLastError();
}

__global__ void fill_rows_u32_kernel(uint32_t* base, uint32_t row_words, const uint32_t* rows, uint32_t value) {
    uint32_t* p = base + (size_t)rows[blockIdx.x] * row_words;
    for (uint32_t c = threadIdx.x; c < row_words; c += blockDim.x) p[c] = value;
}

hipError_t launch_fill_rows_u32(uint32_t* base, uint32_t row_words, const uint32_t* rows, uint32_t nrows, uint32_t value,
                                hipStream_t s) {
    if (!nrows) return hipSuccess;
    hipLaunchKernelGGL(fill_rows_u32_kernel, dim3(nrows), dim3(64), 0, s, base, row_words, rows, value);
    return hipGetLastError();
}

__global__ void scatter_u64_kernel(uint64_t* dst, const uint32_t* idx, const uint64_t* src, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
__global__ void scatter_u32_kernel(uint32_t* dst, const uint32_t* idx, const uint32_t* src, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[i];
}
hipError_t launch_scatter_u64(uint64_t* dst, const uint32_t* idx, const uint64_t* src, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(scatter_u64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, idx, src, n);
    return hipGetLastError();
}
hipError_t launch_scatter_u32(uint32_t* dst, const uint32_t* idx, const uint32_t* src, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(scatter_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, idx, src, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------- exact (brute-force) search
// Ground truth for recall and the `exact` search of usearch.  Two kernels per
// (query block, base chunk): a 64x64-tile distance kernel (f32 FMA, exact formulas) and a
// one-wave-per-query streaming top-k select that keeps the running k best in LDS.
constexpr uint32_t kExactQB = 1024;   // queries per pass
constexpr uint32_t kExactCH = 65536;  // base rows per pass

// Element k..k+3 of a stored row as floats (f16/bf16 widened, i8 as integers, b1 as 0/1).
__device__ __forceinline__ void load4_dequant(const IndexView& ix, size_t row, uint32_t k, float (&o)[4]) {
    const uint8_t* base = reinterpret_cast<const uint8_t*>(ix.vectors) + row * (size_t)ix.stride4 * 16;
    const uint32_t epc = ix.scalar == SC_F32 ? 4 : ix.scalar == SC_I8 ? 16 : ix.scalar == SC_B1 ? 128 : 8;
    if (k >= ix.stride4 * epc) {  // the k extent of a tile may run past a short row: zeros, like the row's own padding
        o[0] = o[1] = o[2] = o[3] = 0.f;
        return;
    }
    switch (ix.scalar) {
        case SC_F32: {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)k * 4);
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        } break;
        case SC_F16: {
            const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)k * 2);
            o[0] = half_bits_to_float(v.x & 0xFFFFu); o[1] = half_bits_to_float(v.x >> 16);
            o[2] = half_bits_to_float(v.y & 0xFFFFu); o[3] = half_bits_to_float(v.y >> 16);
        } break;
        case SC_BF16: {
            const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)k * 2);
            o[0] = bf16_bits_to_float(v.x & 0xFFFFu); o[1] = bf16_bits_to_float(v.x >> 16);
            o[2] = bf16_bits_to_float(v.y & 0xFFFFu); o[3] = bf16_bits_to_float(v.y >> 16);
        } break;
        case SC_I8: {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(base + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (float)(int)(int8_t)((v >> (8 * j)) & 0xFFu);
        } break;
        default: {
            const uint32_t v = base[k >> 3] >> (k & 7);  // k is a multiple of 4
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (float)((v >> j) & 1u);
        } break;
    }
}

// The same in two steps for a kernel compiled for ONE storage type: the raw load carries no wait (it stays in flight
// under other work), the widening happens when the value is consumed.
template <int SC>
__device__ __forceinline__ uint4 load4_raw(const IndexView& ix, size_t row, uint32_t k, bool ok) {
    constexpr uint32_t epc = SC == SC_F32 ? 4 : SC == SC_I8 ? 16 : 8;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (!ok || k >= ix.stride4 * epc) return r;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(ix.vectors) + row * (size_t)ix.stride4 * 16;
    if constexpr (SC == SC_F32) {
        r = *reinterpret_cast<const uint4*>(base + (size_t)k * 4);
    } else if constexpr (SC == SC_I8) {
        r.x = *reinterpret_cast<const uint32_t*>(base + k);
    } else {
        const uint2 v = *reinterpret_cast<const uint2*>(base + (size_t)k * 2);
        r.x = v.x;
        r.y = v.y;
    }
    return r;
}
template <int SC>
__device__ __forceinline__ void dequant4(const uint4 r, float (&o)[4]) {
    if constexpr (SC == SC_F32) {
        o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
    } else if constexpr (SC == SC_F16) {
        o[0] = half_bits_to_float(r.x & 0xFFFFu); o[1] = half_bits_to_float(r.x >> 16);
        o[2] = half_bits_to_float(r.y & 0xFFFFu); o[3] = half_bits_to_float(r.y >> 16);
    } else if constexpr (SC == SC_BF16) {
        o[0] = bf16_bits_to_float(r.x & 0xFFFFu); o[1] = bf16_bits_to_float(r.x >> 16);
        o[2] = bf16_bits_to_float(r.y & 0xFFFFu); o[3] = bf16_bits_to_float(r.y >> 16);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (float)(int)(int8_t)((r.x >> (8 * j)) & 0xFFu);
    }
}

// Exact path, step 0: queries -> the values the metric really sees (quantised, then widened back to f32),
// kpad floats per query, plus the query-side aux of finalize().
template <int AR>
__global__ void prepare_queries_kernel(IndexView ix, const float* q, uint32_t q_stride, uint32_t nq, uint32_t kpad, float* qd,
                                       float* q_aux) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= nq) return;
    const float* v = q + (size_t)w * q_stride;
    float mag = 0.f;
    if constexpr (AR == AR_I8) mag = magnitude_f64(v, ix.dim, 64, (uint32_t)lane);
    float sq = 0.f;
    for (uint32_t c = lane; c * Arith<AR>::epc < kpad; c += kWave) {
        const uint4 ch = quantise_chunk<AR>(v, c * Arith<AR>::epc, ix.dim, mag);
        const uint32_t wds[4] = {ch.x, ch.y, ch.z, ch.w};
        for (uint32_t j = 0; j < Arith<AR>::epc; ++j) {
            const uint32_t e = c * Arith<AR>::epc + j;
            if (e >= kpad) break;
            float f;
            if constexpr (Arith<AR>::scalar == SC_F32) f = __uint_as_float(wds[j]);
            else if constexpr (Arith<AR>::scalar == SC_F16) f = half_bits_to_float((wds[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
            else if constexpr (Arith<AR>::scalar == SC_BF16) f = bf16_bits_to_float((wds[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
            else if constexpr (Arith<AR>::scalar == SC_I8) f = (float)(int)(int8_t)((wds[j >> 2] >> ((j & 3) * 8)) & 0xFFu);
            else f = (float)((wds[j >> 5] >> (j & 31)) & 1u);
            qd[(size_t)w * kpad + e] = f;
            sq = fmaf(f, f, sq);
        }
    }
    for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) q_aux[w] = AR == AR_I8 ? sq : (sq > 0.f ? 1.0f / sqrtf(sq) : 0.f);
}

__device__ __forceinline__ float finalize_exact(const IndexView& ix, float acc, float q_aux, float r_aux) {
    switch (ix.scalar) {
        case SC_I8: return finalize<AR_I8>(ix.metric, (int)lrintf(acc), q_aux, r_aux);
        case SC_B1: return acc;  // sum of (a-b)^2 over 0/1 values == popcount(xor)
        default: return finalize<AR_F32_DOT>(ix.metric, acc, q_aux, r_aux);
    }
}
__device__ __forceinline__ bool exact_needs_aux(const IndexView& ix) {
    return ix.scalar == SC_I8 || (ix.scalar != SC_B1 && ix.metric == COS);
}

// KIND: KDOT accumulates a*b, KL2 accumulates (a-b)^2 (l2sq on float storage, hamming on b1).
template <int KIND>
__global__ __launch_bounds__(256) void exact_dist_kernel(IndexView ix, const float* qd, uint32_t kpad, const float* q_aux,
                                                         uint32_t q0, uint32_t nq_blk, uint32_t n0, uint32_t n_blk, float* D) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const uint32_t qt = blockIdx.y * 64, nt = blockIdx.x * 64;
    const uint32_t lrow = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
    float acc[4][4] = {};
    for (uint32_t k0 = 0; k0 < kpad; k0 += 16) {
        {
            const uint32_t qi = qt + lrow;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi < nq_blk) a = *reinterpret_cast<const float4*>(qd + (size_t)(q0 + qi) * kpad + k0 + lk);
            As[lk + 0][lrow] = a.x;
            As[lk + 1][lrow] = a.y;
            As[lk + 2][lrow] = a.z;
            As[lk + 3][lrow] = a.w;
            const uint32_t ni = nt + lrow;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (ni < n_blk) load4_dequant(ix, (size_t)(n0 + ni), k0 + lk, v);
            Bs[lk + 0][lrow] = v[0];
            Bs[lk + 1][lrow] = v[1];
            Bs[lk + 2][lrow] = v[2];
            Bs[lk + 3][lrow] = v[3];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[kk][ty * 4 + i];
                b[i] = Bs[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (KIND == KL2) {
                        float d = a[i] - b[j];
                        acc[i][j] = fmaf(d, d, acc[i][j]);
                    } else {
                        acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
                    }
                }
        }
        __syncthreads();
    }
    const bool aux = exact_needs_aux(ix);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t qi = qt + ty * 4 + i;
        if (qi >= nq_blk) continue;
        float qa = aux ? q_aux[q0 + qi] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t ni = nt + tx * 4 + j;
            if (ni >= n_blk) continue;
            float ra = aux ? ix.aux[n0 + ni] : 0.f;
            D[(size_t)qi * kExactCH + ni] = finalize_exact(ix, acc[i][j], qa, ra);
        }
    }
}

// K8 of SURVEY.md section 2.3 -- block distances on the matrix cores: a dense Q[128 x K] . C^T[K x 128] tile per
// workgroup with v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate: bit-for-bit a k-ordered fmaf chain, so the
// scores equal the VALU kernel's).  This is the one place of the path that really is a dense contraction
// (exact search, ground truth, the q = 256 batched inner-product configuration); the graph walk is a gather.
// Dot-product family only (cos / ip / every i8 metric); l2sq and hamming keep the (a-b)^2 VALU kernel.
// 4 waves, each a 64 x 64 quadrant = 2 x 2 MFMA tiles (64 accumulator registers); K staged 16 deep through two LDS buffers.
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kMfmaKC = 16;

// 128 VGPRs (amdgpu_waves_per_eu 4): four workgroups per CU instead of three (86.9 -> 97.7 TFLOP/s at q = 256).
// K is staged 16 deep through TWO LDS buffers: chunk c+1 is written to the other buffer after the MFMAs of chunk c
// have been issued, so a chunk costs one barrier and its global loads are in flight under the MFMAs.
template <int SC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void exact_dist_mfma_kernel(
    IndexView ix, const float* qd, uint32_t kpad, const float* q_aux, uint32_t q0, uint32_t nq_blk, uint32_t n0, uint32_t n_blk,
    float* D) {
    __shared__ float As[2][kMfmaKC][133];  // 133: rows e and 8 + e (written by lanes t, t ^ 1) fall in different banks
    __shared__ float Bs[2][kMfmaKC][133];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t wy = w >> 1, wx = w & 1;
    // query tile on the fast grid index: the workgroups that share a 128-row base tile run back to back, so the tile is
    // read from HBM once per launch and served from L2 to the other query tiles (the queries, <= 3 MB, stay in L2 anyway)
    const uint32_t qt = blockIdx.x * 128, nt = blockIdx.y * 128;
    const uint32_t lrow = t >> 1, lf4 = (t & 1) * 2;  // staging: row of the tile, first of 2 float4 of the K chunk
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 a[2];
    uint4 braw[2];  // storage bits of the base row: widened when they are staged, so the loads carry no wait
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const uint32_t k = k0 + (lf4 + f) * 4;
            a[f] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qt + lrow < nq_blk && k < kpad) a[f] = *reinterpret_cast<const float4*>(qd + (size_t)(q0 + qt + lrow) * kpad + k);
            braw[f] = load4_raw<SC>(ix, (size_t)(n0 + nt + lrow), k, nt + lrow < n_blk && k < kpad);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float bv[4];
            dequant4<SC>(braw[f], bv);
            const float av[4] = {a[f].x, a[f].y, a[f].z, a[f].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[buf][(lf4 + f) * 4 + e][lrow] = av[e];
                Bs[buf][(lf4 + f) * 4 + e][lrow] = bv[e];
            }
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (uint32_t k0 = 0; k0 < kpad; k0 += kMfmaKC) {
        const bool more = k0 + kMfmaKC < kpad;
        if (more) fetch(k0 + kMfmaKC);
#pragma unroll
        for (int kk = 0; kk < kMfmaKC / 2; ++kk) {
            const uint32_t kr = 2 * kk + (lane >> 5), c = lane & 31;
            const float a0 = As[cur][kr][wy * 64 + c], a1 = As[cur][kr][wy * 64 + 32 + c];
            const float b0 = Bs[cur][kr][wx * 64 + c], b1 = Bs[cur][kr][wx * 64 + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stage(cur ^ 1);  // nobody reads that buffer: its last readers passed the previous barrier
        __syncthreads();
        cur ^= 1;
    }
    const bool aux = exact_needs_aux(ix);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t ni = nt + wx * 64 + j * 32 + (lane & 31);  // C/D layout: column on the lane,
            const float ra = (aux && ni < n_blk) ? ix.aux[n0 + ni] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t qi = qt + wy * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // row in the registers
                if (qi < nq_blk && ni < n_blk)
                    D[(size_t)qi * kExactCH + ni] = finalize_exact(ix, acc[i][j][r], aux ? q_aux[q0 + qi] : 0.f, ra);
            }
        }
}

using SelectShared = BeamShared<256, 256>;

// Grid (queries, S): with few queries a score row is cut into S segments, each with its own running top-k
// (state index qg * S + seg), merged by exact_finish_kernel at the end -- 256 queries alone would put one wave
// on each CU.  S == 1: the single state is written out directly by the last tile.
__global__ __launch_bounds__(64) void exact_select_kernel(IndexView ix, const float* D, uint32_t q0, uint32_t n0,
                                                          uint32_t n_all, uint32_t k, int first, int last, float* st_d,
                                                          uint32_t* st_s, uint32_t* st_n, uint64_t* out_keys,
                                                          float* out_dist, uint32_t* out_found, int raw_slots = 0) {
    __shared__ SelectShared sh;
    const int lane = lane_id();
    const uint32_t ql = blockIdx.x, qg = q0 + ql;
    const uint32_t S = gridDim.y, seg = blockIdx.y;
    const size_t sg = (size_t)qg * S + seg;
    const uint32_t per = ((n_all + S - 1) / S + kWave - 1) / kWave * kWave;  // columns of one segment
    const uint32_t c0 = seg * per;
    const uint32_t n_blk = c0 >= n_all ? 0u : (n_all - c0 < per ? n_all - c0 : per);
    n0 += c0;
    int cur = 0;
    uint32_t sz = 0;
    if (!first) {
        sz = st_n[sg];
        for (uint32_t i = lane; i < sz; i += kWave) {
            sh.lst_d[0][i] = st_d[sg * k + i];
            sh.lst_s[0][i] = st_s[sg * k + i];
        }
    }
    __syncthreads();
    const float* row = D + (size_t)ql * kExactCH + c0;
    // UN independent score loads in flight per lane (a lone wave per query is latency-bound otherwise: with 256
    // queries the chip holds one wave per CU); the member check (removed keys) is only paid by scores that beat
    // the current k-th -- a stale threshold merely admits candidates that list_merge drops again.
    constexpr uint32_t UN = 8;
    for (uint32_t j0 = 0; j0 < n_blk; j0 += kWave * UN) {
        float dv[UN];
#pragma unroll
        for (uint32_t u = 0; u < UN; ++u) {
            const uint32_t j = j0 + u * kWave + (uint32_t)lane;
            dv[u] = j < n_blk ? row[j] : __builtin_inff();
        }
#pragma unroll
        for (uint32_t u = 0; u < UN; ++u) {
            const uint32_t j = j0 + u * kWave + (uint32_t)lane;
            const float d = dv[u];
            const uint32_t slot = n0 + j;
            bool ok = j < n_blk;
            if (ok && sz == k) ok = key_less(d, slot, sh.lst_d[cur][k - 1], sh.lst_s[cur][k - 1]);
            if (!__ballot(ok)) continue;
            ok = ok && ix.keys[ok ? slot : 0] != kFreeKey;
            uint64_t mask = __ballot(ok);
            if (!mask) continue;
            uint32_t ma = (uint32_t)__popcll(mask);
            __syncthreads();
            if (ok) {
                uint32_t r = mbcnt(mask);
                sh.u_dist[r] = d;
                sh.u_slot[r] = slot;
            }
            __syncthreads();
            float nd = (uint32_t)lane < ma ? sh.u_dist[lane] : __builtin_inff();
            uint32_t ns = (uint32_t)lane < ma ? sh.u_slot[lane] : kInvalid;
            sz = list_merge(sh, cur, sz, k, nd, ns, ma, lane);
            __syncthreads();
        }
    }
    if (!last || S > 1) {
        for (uint32_t i = lane; i < sz; i += kWave) {
            st_d[sg * k + i] = sh.lst_d[cur][i];
            st_s[sg * k + i] = sh.lst_s[cur][i];
        }
        if (lane == 0) st_n[sg] = sz;
        return;
    }
    for (uint32_t i = lane; i < k; i += kWave) {
        const uint32_t slot = i < sz ? (sh.lst_s[cur][i] & kSlotMask) : 0u;
        out_keys[(size_t)qg * k + i] = i < sz ? (raw_slots ? (uint64_t)slot : ix.keys[slot]) : kFreeKey;
        out_dist[(size_t)qg * k + i] = i < sz ? sh.lst_d[cur][i] : __builtin_inff();
    }
    if (lane == 0) out_found[qg] = sz;
}

// Merges the S per-segment states of one query into its result.
__global__ __launch_bounds__(64) void exact_finish_kernel(IndexView ix, uint32_t q0, uint32_t S, uint32_t k, const float* st_d,
                                                          const uint32_t* st_s, const uint32_t* st_n, uint64_t* out_keys,
                                                          float* out_dist, uint32_t* out_found, int raw_slots = 0) {
    __shared__ SelectShared sh;
    const int lane = lane_id();
    const uint32_t qg = q0 + blockIdx.x;
    const size_t s0 = (size_t)qg * S;
    uint32_t sz = st_n[s0];
    for (uint32_t i = lane; i < sz; i += kWave) {
        sh.lst_d[0][i] = st_d[s0 * k + i];
        sh.lst_s[0][i] = st_s[s0 * k + i];
    }
    __syncthreads();
    for (uint32_t seg = 1; seg < S; ++seg) {
        const size_t sg = s0 + seg;
        const uint32_t n = st_n[sg];
        for (uint32_t i0 = 0; i0 < n; i0 += kWave) {
            const uint32_t m = n - i0 < (uint32_t)kWave ? n - i0 : (uint32_t)kWave;
            float nd = (uint32_t)lane < m ? st_d[sg * k + i0 + lane] : __builtin_inff();
            uint32_t ns = (uint32_t)lane < m ? st_s[sg * k + i0 + lane] : kInvalid;
            sz = list_merge(sh, 0, sz, k, nd, ns, m, lane);
            __syncthreads();
        }
    }
    for (uint32_t i = lane; i < k; i += kWave) {
        const uint32_t slot = i < sz ? (sh.lst_s[0][i] & kSlotMask) : 0u;
        out_keys[(size_t)qg * k + i] = i < sz ? (raw_slots ? (uint64_t)slot : ix.keys[slot]) : kFreeKey;
        out_dist[(size_t)qg * k + i] = i < sz ? sh.lst_d[0][i] : __builtin_inff();
    }
    if (lane == 0) out_found[qg] = sz;
}

// Segments per query: enough waves (~4096) to cover the select pass's load latency, at most 16.
static uint32_t exact_segments(uint32_t nq) {
    const uint32_t nqb = nq < kExactQB ? nq : kExactQB;
    uint32_t S = 1;
    while (S < 16 && nqb * (S * 2) <= 4096) S *= 2;
    return S;
}

static uint32_t exact_kpad(const IndexView& ix) {
    return (ix.dim + 15u) & ~15u;  // k extent of the tiles; load4_dequant pads short rows with zeros
}
// the dot product is what cos / ip / every i8 metric are made of; l2sq and hamming accumulate (a-b)^2
static bool exact_uses_l2(const IndexView& ix) { return ix.scalar == SC_B1 || (ix.scalar != SC_I8 && ix.metric == L2SQ); }

size_t exact_scratch_bytes(uint32_t nq, uint32_t k, uint32_t dim) {
    const size_t S = exact_segments(nq);
    return (size_t)kExactQB * kExactCH * 4 + (size_t)nq * S * k * 8 + (size_t)nq * S * 4 + (size_t)nq * 4 + (size_t)nq * (dim + 16) * 4 + 1024;
}

static hipError_t prepare_queries(const IndexView& ix, const float* q, uint32_t q_stride, uint32_t nq, uint32_t kpad, float* qd,
                                  float* q_aux, hipStream_t s) {
    const int ar = arith_of(ix.scalar, ix.metric);
    VS_AR_SWITCH(ar, hipLaunchKernelGGL((prepare_queries_kernel<A>), dim3((nq + 3) / 4), dim3(256), 0, s, ix, q, q_stride, nq, kpad,
                                        qd, q_aux))
    return hipGetLastError();
}

hipError_t launch_exact(const ExactArgs& a, void* scratch, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (a.k == 0 || a.k > 256) return hipErrorInvalidValue;
    const uint32_t kpad = exact_kpad(a.ix);
    char* p = (char*)scratch;
    float* D = (float*)p;
    p += (size_t)kExactQB * kExactCH * 4;
    const uint32_t S = exact_segments(a.nq);
    float* st_d = (float*)p;
    p += (size_t)a.nq * S * a.k * 4;
    uint32_t* st_s = (uint32_t*)p;
    p += (size_t)a.nq * S * a.k * 4;
    uint32_t* st_n = (uint32_t*)p;
    p += (size_t)a.nq * S * 4;
    float* q_aux = (float*)p;
    p += (size_t)a.nq * 4;
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    float* qd = (float*)p;
    hipError_t e = prepare_queries(a.ix, a.queries, a.q_stride, a.nq, kpad, qd, q_aux, s);
    if (e != hipSuccess) return e;
    if (a.slots == 0) {  // empty index: found = 0 everywhere
        hipLaunchKernelGGL(fill_u32_kernel, dim3(64), dim3(256), 0, s, a.out_found, 0u, (size_t)a.nq);
        hipLaunchKernelGGL(fill_u32_kernel, dim3(256), dim3(256), 0, s, (uint32_t*)a.out_keys, 0xFFFFFFFFu,
                           (size_t)a.nq * a.k * 2);
        hipLaunchKernelGGL(fill_u32_kernel, dim3(256), dim3(256), 0, s, (uint32_t*)a.out_dist, 0x7F800000u,
                           (size_t)a.nq * a.k);
        return hipGetLastError();
    }
    const bool l2 = exact_uses_l2(a.ix);
    for (uint32_t q0 = 0; q0 < a.nq; q0 += kExactQB) {
        uint32_t nqb = a.nq - q0 < kExactQB ? a.nq - q0 : kExactQB;
        for (uint32_t n0 = 0; n0 < a.slots; n0 += kExactCH) {
            uint32_t nb = a.slots - n0 < kExactCH ? a.slots - n0 : kExactCH;
            dim3 grid((nb + 63) / 64, (nqb + 63) / 64);
            if (l2)
                hipLaunchKernelGGL((exact_dist_kernel<KL2>), grid, dim3(256), 0, s, a.ix, qd, kpad, q_aux, q0, nqb, n0, nb, D);
            else if (a.use_valu)
                hipLaunchKernelGGL((exact_dist_kernel<KDOT>), grid, dim3(256), 0, s, a.ix, qd, kpad, q_aux, q0, nqb, n0, nb, D);
            else
            {
                const dim3 mg((nqb + 127) / 128, (nb + 127) / 128);
                switch (a.ix.scalar) {
                    case SC_F32: hipLaunchKernelGGL((exact_dist_mfma_kernel<SC_F32>), mg, dim3(256), 0, s, a.ix, qd, kpad, q_aux, q0, nqb, n0, nb, D); break;
                    case SC_F16: hipLaunchKernelGGL((exact_dist_mfma_kernel<SC_F16>), mg, dim3(256), 0, s, a.ix, qd, kpad, q_aux, q0, nqb, n0, nb, D); break;
                    case SC_BF16: hipLaunchKernelGGL((exact_dist_mfma_kernel<SC_BF16>), mg, dim3(256), 0, s, a.ix, qd, kpad, q_aux, q0, nqb, n0, nb, D); break;
                    default: hipLaunchKernelGGL((exact_dist_mfma_kernel<SC_I8>), mg, dim3(256), 0, s, a.ix, qd, kpad, q_aux, q0, nqb, n0, nb, D); break;
                }
            }
            int first = n0 == 0, last = n0 + kExactCH >= a.slots;
            hipLaunchKernelGGL(exact_select_kernel, dim3(nqb, S), dim3(64), 0, s, a.ix, D, q0, n0, nb, a.k, first, last, st_d,
                               st_s, st_n, a.out_keys, a.out_dist, a.out_found);
        }
        if (S > 1)
            hipLaunchKernelGGL(exact_finish_kernel, dim3(nqb), dim3(64), 0, s, a.ix, q0, S, a.k, st_d, st_s, st_n, a.out_keys,
                               a.out_dist, a.out_found);
    }
    return hipGetLastError();
}

// One query against every row (exhaustive path of filtered search / k beyond the LDS beam).
// qd / q_aux: the prepared query (prepare_queries_kernel).  One wave per row.
__global__ void distance_row_kernel(IndexView ix, const float* qd, const float* q_aux, uint32_t kpad, uint32_t n, float* out) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= n) return;
    const bool l2 = ix.scalar == SC_B1 || (ix.scalar != SC_I8 && ix.metric == L2SQ);
    float acc = 0.f;
    for (uint32_t k = (uint32_t)lane * 4; k < kpad; k += kWave * 4) {
        float v[4];
        load4_dequant(ix, (size_t)w, k, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = qd[k + j];
            if (l2) {
                float d = a - v[j];
                acc = fmaf(d, d, acc);
            } else {
                acc = fmaf(a, v[j], acc);
            }
        }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const bool aux = exact_needs_aux(ix);
        out[w] = finalize_exact(ix, acc, aux ? q_aux[0] : 0.f, aux ? ix.aux[w] : 0.f);
    }
}

hipError_t launch_distance_row(const IndexView& ix, const float* d_query, uint32_t n, float* d_scratch, hipStream_t s,
                               float* host_out) {
    if (!n) return hipSuccess;
    const uint32_t kpad = exact_kpad(ix);
    float* qd = d_scratch + n;  // scratch: n distances, then the prepared query and its aux
    float* q_aux = qd + kpad;
    hipError_t e = prepare_queries(ix, d_query, ix.dim, 1, kpad, qd, q_aux, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(distance_row_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ix, qd, q_aux, kpad, n, d_scratch);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!host_out) return hipSuccess;  // the caller keeps working on the device
    return hipMemcpyAsync(host_out, d_scratch, (size_t)n * 4, hipMemcpyDeviceToHost, s);
}

// ---------------------------------------------------------------- exhaustive ranking on the device
__global__ void rank_keys_kernel(IndexView ix, const float* d, uint32_t n, uint64_t* rank) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (ix.keys[s] == kFreeKey) {
        rank[s] = ~0ull;
        return;
    }
    const uint32_t b = __float_as_uint(d[s]);
    const uint32_t ord = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // unsigned order == float order (ip distances go negative)
    rank[s] = ((uint64_t)ord << 32) | s;
}
__global__ void rank_emit_kernel(IndexView ix, const uint64_t* sorted, uint32_t n, uint64_t* out_keys, float* out_dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = sorted[i];
    if (r == ~0ull) {
        out_keys[i] = kFreeKey;
        out_dist[i] = __builtin_inff();
        return;
    }
    const uint32_t ord = (uint32_t)(r >> 32);
    const uint32_t b = (ord & 0x80000000u) ? (ord & 0x7FFFFFFFu) : ~ord;
    out_keys[i] = ix.keys[(uint32_t)r];
    out_dist[i] = __uint_as_float(b);
}
hipError_t launch_rank_keys(const IndexView& ix, const float* d, uint32_t n, uint64_t* rank, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(rank_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ix, d, n, rank);
    return hipGetLastError();
}
hipError_t launch_rank_emit(const IndexView& ix, const uint64_t* sorted, uint32_t n, uint64_t* out_keys, float* out_dist, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(rank_emit_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ix, sorted, n, out_keys, out_dist);
    return hipGetLastError();
}

// ---------------------------------------------------------------- block search: split-bf16 MFMA candidates + exact re-score
// The q x candidate-block contraction (BASELINE configs[4], ground truth for recall) at the bf16 matrix rate WITHOUT giving up
// exactness.  Every f32 value is split x = hi + lo (two bf16, round-to-nearest: |x - hi - lo| <= 2^-18 |x|) and
// q . c ~= qh.ch + qh.cl + ql.ch: three v_mfma_f32_32x32x16_bf16 (f32 accumulate) per 16 k -- 96 matrix cycles against 512 for
// the f32-input MFMA.  The dropped ql.cl term and the residuals bound the error of a score by eps = 3e-5 |q| |c|.  The
// approximate scores only NOMINATE: the C = 256 best rows per query are re-scored with exact f32 arithmetic, and the answer
// is certified -- if the k-th exact score is below (C-th approximate score - eps), no row outside the nominees can beat it.
// A query that fails the certificate (dense ties at the cut) is counted, and the host re-runs the f32 path.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr uint32_t kBlockC = 256;    // nominees per query
constexpr uint32_t kP8CandCap = 8192, kP8CandParts = 8;  // the 8-bit plane's buffer: eight pieces of 1,024 (see CandParts in block_merge_body)
constexpr uint32_t kBlockCandCap = 4096;  // scores per query a row block may pass on to the nominee list (after the first block)
constexpr int kBlockKC = 64;         // k staged per LDS buffer.  Measured at 10M x 768, q = 256 (whole search): 64 deep, one buffer, two
                                     // barriers per stage 21.8 ms; 32 deep through two buffers (the f32 kernel's structure) 31.2 ms
constexpr int kBlockLd = kBlockKC + 8;  // halfwords per LDS row: 144 B, 16-byte reads of 16 consecutive rows hit 64 distinct dwords

// two f32 -> their (hi, lo) bf16 pairs, packed [x0 | x1 << 16]; the casts compile to v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ void split_bf16x2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
    const __bf16 l0 = (__bf16)(x0 - (float)h0), l1 = (__bf16)(x1 - (float)h1);
    hi = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    lo = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
}
__device__ __forceinline__ void split_bf16(float x, uint32_t& hi, uint32_t& lo) {
    uint32_t h, l;
    split_bf16x2(x, 0.f, h, l);
    hi = h & 0xFFFFu;
    lo = l & 0xFFFFu;
}

// qd (nq x kpad f32, the form the metric sees) -> qh / ql (nq x kpad bf16 bits) and |q|
__global__ void split_queries_kernel(const float* qd, uint32_t nq, uint32_t kpad, uint16_t* qh, uint16_t* ql, float* qnorm) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= nq) return;
    float sq = 0.f;
    for (uint32_t e = lane; e < kpad; e += kWave) {
        const float x = qd[(size_t)w * kpad + e];
        uint32_t h, l;
        split_bf16(x, h, l);
        qh[(size_t)w * kpad + e] = (uint16_t)h;
        ql[(size_t)w * kpad + e] = (uint16_t)l;
        sq = fmaf(x, x, sq);
    }
    for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) qnorm[w] = sqrtf(sq);
}

// 128 queries x 128 base rows per workgroup of EIGHT waves, each a 64 x 32 part (2 MFMA tiles: 32 accumulator registers), K
// staged 64 deep through one LDS buffer with the next stage's global loads in flight in registers.  Eight waves at ~110
// registers put 4 waves on every SIMD (two workgroups per CU by LDS): twice the latency cover of the 4-wave form, whose
// 64 accumulator + 64 prefetch registers allowed only 2 per SIMD (configs[4]: 20.9 ms per batch with that form).
template <int SC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void block_dist_bf16x3_kernel(IndexView ix, const uint16_t* qh, const uint16_t* ql, uint32_t kpad,
                                                                const float* q_aux, uint32_t q0, uint32_t nq_blk, uint32_t n0,
                                                                uint32_t n_blk, float* D, const float* thr, uint2* cand,
                                                                uint32_t* cand_cnt, uint32_t cand_cap) {
    __shared__ __attribute__((aligned(16))) uint16_t Ah[128][kBlockLd], Al[128][kBlockLd], Bh[128][kBlockLd], Bl[128][kBlockLd];
    __shared__ float thr_s[128];  // the tile's queries' thresholds (read in the epilogue; the K loop's barriers publish them)
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t wy = w >> 2, wx = w & 3;  // wave (wy, wx): query rows 64 wy .. +63, base rows 32 wx .. +31
    // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (each with its own L2), so the query tiles of ONE row
    // block are made consecutive workgroups of ONE XCD -- the second one finds the rows in that L2 instead of fetching
    // them from HBM again (measured with 2 query tiles: 2.36 x the algorithmic bytes per launch before).
    const uint32_t q_tiles = (nq_blk + 127u) / 128u;
    const uint32_t in_xcd = blockIdx.x >> 3;
    const uint32_t qt = (in_xcd % q_tiles) * 128, nt = ((in_xcd / q_tiles) * 8u + (blockIdx.x & 7u)) * 128;
    if (nt >= n_blk) return;
    if (thr && t < 128) thr_s[t] = qt + t < nq_blk ? thr[qt + t] : -__builtin_inff();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // Staging, coalesced along the rows: 16 lanes read the 256 contiguous bytes a base row contributes to a stage (a wave-load
    // = 4 rows, whole 128-byte lines), 8 lanes the 128 bytes of a pre-split query row.
    const uint32_t brow = t >> 4, bk = (t & 15) * 4;  // base: rows brow + 32 f, k bk..bk+3
    const uint32_t arow = t >> 3, ak = (t & 7) * 8;   // queries: rows arow + 64 f, k ak..ak+7
    uint4 ah[2], al[2], braw[4];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const uint32_t r = arow + 64u * f;
            ah[f] = al[f] = make_uint4(0u, 0u, 0u, 0u);
            if (qt + r < nq_blk && k0 + ak < kpad) {
                ah[f] = *reinterpret_cast<const uint4*>(qh + (size_t)(q0 + qt + r) * kpad + k0 + ak);
                al[f] = *reinterpret_cast<const uint4*>(ql + (size_t)(q0 + qt + r) * kpad + k0 + ak);
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint32_t r = brow + 32u * f;
            braw[f] = load4_raw<SC>(ix, (size_t)(n0 + nt + r), k0 + bk, nt + r < n_blk && k0 + bk < kpad);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            *reinterpret_cast<uint4*>(&Ah[arow + 64 * f][ak]) = ah[f];
            *reinterpret_cast<uint4*>(&Al[arow + 64 * f][ak]) = al[f];
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float bv[4];
            dequant4<SC>(braw[f], bv);
            uint32_t h0, l0, h1, l1;
            split_bf16x2(bv[0], bv[1], h0, l0);
            split_bf16x2(bv[2], bv[3], h1, l1);
            *reinterpret_cast<uint2*>(&Bh[brow + 32 * f][bk]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&Bl[brow + 32 * f][bk]) = make_uint2(l0, l1);
        }
    };
    fetch(0);
    for (uint32_t k0 = 0; k0 < kpad; k0 += kBlockKC) {
        __syncthreads();  // the previous stage has been multiplied
        stage();
        __syncthreads();
        if (k0 + kBlockKC < kpad) fetch(k0 + kBlockKC);
#pragma unroll
        for (int ks = 0; ks < kBlockKC / 16; ++ks) {
            const uint32_t kk = (uint32_t)ks * 16u + 8u * (lane >> 5), c = lane & 31;
            const bf16x8 fb_h = *reinterpret_cast<const bf16x8*>(&Bh[wx * 32 + c][kk]);
            const bf16x8 fb_l = *reinterpret_cast<const bf16x8*>(&Bl[wx * 32 + c][kk]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 fa_h = *reinterpret_cast<const bf16x8*>(&Ah[wy * 64 + i * 32 + c][kk]);
                const bf16x8 fa_l = *reinterpret_cast<const bf16x8*>(&Al[wy * 64 + i * 32 + c][kk]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_l, fb_h, acc[i], 0, 0, 0);  // small terms first
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_h, fb_l, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_h, fb_h, acc[i], 0, 0, 0);
            }
        }
    }
    const bool aux = exact_needs_aux(ix);
    const uint32_t ni = nt + wx * 32 + (lane & 31);  // C/D layout: column on the lane, rows in the registers
    const float ra = (aux && ni < n_blk) ? ix.aux[n0 + ni] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t qi = qt + wy * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (qi < nq_blk && ni < n_blk) {
                const float d = finalize_exact(ix, acc[i][r], aux ? q_aux[q0 + qi] : 0.f, ra);
                if (thr) {  // every row block after the first: only scores that can still enter the query's nominee list leave the tile
                    if (d <= thr_s[qi - qt]) {
                        const uint32_t at = atomicAdd(&cand_cnt[qi], 1u);
                        if (at < cand_cap) cand[(size_t)qi * cand_cap + at] = make_uint2(__float_as_uint(d), n0 + ni);
                    }
                } else {
                    D[(size_t)qi * kExactCH + ni] = d;
                }
            }
        }
}

// Row blocks after the first: the tile kernel appended the few scores at or below each query's threshold (the worst score on
// its full nominee list) to a per-query buffer; one wave per query merges them into the list -- same (score, slot) order as
// exact_select_kernel, so the nominees are the same -- and publishes the new threshold.  A buffer that overflowed (rows stored
// in an order that keeps improving on everything seen before) raises `uncertified`: the host re-runs the batch on the f32 path.
// order-preserving key of a distance's bits, and back
__device__ __forceinline__ uint32_t bm_key(uint32_t bits) { return bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ uint32_t bm_unkey(uint32_t key) { return (key & 0x80000000u) ? key ^ 0x80000000u : ~key; }
// PREFILTER (round 6, the 8-bit plane's first block: every score of 2,048 rows arrives, a few dozen belong on the list): the band_k-th
// smallest distance of the buffer by a radix select (four passes over the buffer, a histogram in LDS), and only scores within the band
// of it go through the merges -- 32 list merges became one or two.  Removed rows take part in the select, so the threshold can come out
// too tight when one of them is among the best: the caller checks it against the list it produced and runs again without it.
// (The later blocks too, since the last step of round 6: the select runs over the buffer AND the list the earlier blocks left -- the band_k-th
// smallest of the two together is where the list's band will end -- and the thousand or so scores a large block passes become one or two
// merges instead of one per 64.)
// skey: the buffer's keys staged in LDS by the first pass (four loads in flight per lane), so that the other three passes and the
// filter of the merge loop below read LDS, not one dependent L2 round trip per 64 scores -- that chain was most of a merge's 0.09 ms.
// where: place i of the (concatenated) buffer -> its element in `mine` (see CandParts)
template <class SH, class W>
__device__ __forceinline__ float block_merge_prefilter(SH& sh, const uint2* mine, uint32_t n, uint32_t sz, uint32_t band_k, float eps2, int lane,
                                                       uint32_t* skey, W&& where) {
    uint32_t* hist = reinterpret_cast<uint32_t*>(sh.vis_tag);  // 256 words (the visited table is not used by these kernels)
    uint32_t prefix = 0, want = band_k;
    for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hist[lane * 4 + i] = 0u;
        __syncthreads();
        if (shift == 24) {
            for (uint32_t i0 = 0; i0 < n; i0 += 4u * kWave) {
                uint32_t k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * kWave + (uint32_t)lane;
                    k4[u] = i < n ? bm_key(mine[where(i)].x) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = i0 + (uint32_t)u * kWave + (uint32_t)lane;
                    if (i < n) {
                        skey[i] = k4[u];
                        atomicAdd(&hist[k4[u] >> 24], 1u);
                    }
                }
            }
            for (uint32_t i = (uint32_t)lane; i < sz; i += kWave) atomicAdd(&hist[bm_key(__float_as_uint(sh.lst_d[0][i])) >> 24], 1u);
        } else {
        for (uint32_t i = (uint32_t)lane; i < n + sz; i += kWave) {
            const uint32_t key = i < n ? skey[i] : bm_key(__float_as_uint(sh.lst_d[0][i - n]));
            if ((key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        }
        __syncthreads();
        const uint32_t h0 = hist[lane * 4], h1 = hist[lane * 4 + 1], h2 = hist[lane * 4 + 2], h3 = hist[lane * 4 + 3];
        uint32_t incl = h0 + h1 + h2 + h3;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += up;
        }
        const uint64_t reach = __ballot(incl >= want);
        if (!reach) return __builtin_inff();  // fewer than band_k scores: no filter
        const int ol = __builtin_ctzll(reach);
        const uint32_t before = (uint32_t)__shfl((int)(incl - (h0 + h1 + h2 + h3)), ol);
        const uint32_t a0 = (uint32_t)__shfl((int)h0, ol), a1 = (uint32_t)__shfl((int)h1, ol), a2 = (uint32_t)__shfl((int)h2, ol);
        uint32_t c = before, b = (uint32_t)ol * 4u;
        if (c + a0 < want) { c += a0; ++b; if (c + a1 < want) { c += a1; ++b; if (c + a2 < want) { c += a2; ++b; } } }
        want -= c;
        prefix = shift == 24 ? b : ((prefix << 8) | b);
        __syncthreads();
    }
    return __uint_as_float(bm_unkey(prefix)) + eps2;
}
template <class SH>
__device__ __forceinline__ void block_merge_body(SH& sh, IndexView ix, uint32_t q0, uint32_t C, uint32_t cand_cap, uint32_t* cand_cnt, const uint2* cand,
                                                 uint64_t* list_slot, float* list_d, uint32_t* list_n, float* thr, uint32_t* uncertified,
                                                 const float* band_eps, uint32_t band_k, uint32_t* skey = nullptr, uint32_t parts = 1) {
    const bool prefilter = skey != nullptr;
    const int lane = lane_id();
    const uint32_t ql = blockIdx.x, qg = q0 + ql;
    // CandParts (round 6, the 8-bit plane): a query's buffer in `parts` equal pieces with a counter each (counter p of query q: cand_cnt[p * 256 + q]) --
    // the tile kernel's workgroups append to piece blockIdx & (parts - 1).  Behind the early blocks' loose thresholds hundreds of scores per query
    // pass, and the returning atomics of every workgroup queued on 256 counters: a hundred microseconds per launch (profiles/r06_c5_trace.txt).
    // The merge reads the pieces as one buffer: place i -> piece and offset.
    const uint32_t sub = cand_cap / parts;
    uint32_t off[9];
    off[0] = 0;
    bool over = false;
#pragma unroll
    for (uint32_t pp = 0; pp < 8u; ++pp) {
        uint32_t c = pp < parts ? cand_cnt[pp * 256u + ql] : 0u;
        if (c > sub) {
            over = true;
            c = sub;
        }
        off[pp + 1] = off[pp] + c;
    }
    auto where = [&](uint32_t i) -> uint32_t {
        uint32_t pp = 0;
#pragma unroll
        for (uint32_t t = 1; t < 8u; ++t) pp += i >= off[t] ? 1u : 0u;
        return pp * sub + (i - off[pp]);
    };
    uint32_t n = off[8];
    const uint32_t sz0 = list_n[qg];
    uint32_t sz = sz0;
    for (uint32_t i = lane; i < sz; i += kWave) {
        sh.lst_d[0][i] = list_d[(size_t)qg * C + i];
        sh.lst_s[0][i] = (uint32_t)list_slot[(size_t)qg * C + i];
    }
    __syncthreads();
    if ((uint32_t)lane < parts) cand_cnt[(uint32_t)lane * 256u + ql] = 0;
    if (over && lane == 0) atomicAdd(uncertified, 1u);
    float pre = __builtin_inff();
    if (prefilter && n >= 256u && band_eps && band_k)
        pre = block_merge_prefilter(sh, cand + (size_t)ql * cand_cap, n, sz, band_k, 2.0f * band_eps[ql], lane, skey, where);
    const bool staged = pre < __builtin_inff();  // skey holds the buffer's keys
    // ... and then the places of the scores inside the band, compacted (in place: a 64's places land in front of where its keys were read):
    // the few dozen survivors of a thousand scores lie scattered over nearly as many 64s, and each 64 with one of them was a list merge of its own
    uint32_t n_in = 0;
    if (staged) {
        const uint32_t pre_key = bm_key(__float_as_uint(pre));
        for (uint32_t i0 = 0; i0 < n; i0 += kWave) {
            const uint32_t i = i0 + (uint32_t)lane;
            const bool in = i < n && skey[i] <= pre_key;
            const uint64_t im = __ballot(in);
            __syncthreads();
            if (in) skey[n_in + mbcnt(im)] = i;
            n_in += (uint32_t)__popcll(im);
        }
        __syncthreads();
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
    const bool listed = staged && attempt == 0;  // this pass takes the compacted places
    const uint32_t count = listed ? n_in : n;
    for (uint32_t i0 = 0; i0 < count; i0 += kWave) {
        const uint32_t j = i0 + (uint32_t)lane;
        const uint32_t i = j < count ? (listed ? skey[j] : j) : 0u;
        const uint2 e = j < count ? cand[(size_t)ql * cand_cap + where(i)] : make_uint2(0u, 0u);
        const float d = __uint_as_float(e.x);
        const uint32_t slot = e.y;
        bool ok = j < count && d <= pre;
        if (ok && sz == C) ok = key_less(d, slot, sh.lst_d[0][C - 1], sh.lst_s[0][C - 1]);
        // one-product pass: a score more than 2 eps behind the k-th best so far cannot belong to a true top-k row (see the
        // threshold below), so it need not be listed either -- the lists stay short and the merges cheap
        if (ok && band_eps && band_k && sz >= band_k) ok = d <= sh.lst_d[0][band_k - 1] + 2.0f * band_eps[ql];
        ok = ok && ix.keys[ok ? slot : 0] != kFreeKey;
        const uint64_t mask = __ballot(ok);
        if (!mask) continue;
        const uint32_t ma = (uint32_t)__popcll(mask);
        __syncthreads();
        if (ok) {
            const uint32_t r = mbcnt(mask);
            sh.u_dist[r] = d;
            sh.u_slot[r] = slot;
        }
        __syncthreads();
        const float nd = (uint32_t)lane < ma ? sh.u_dist[lane] : __builtin_inff();
        const uint32_t ns = (uint32_t)lane < ma ? sh.u_slot[lane] : kInvalid;
        sz = list_merge(sh, 0, sz, C, nd, ns, ma, lane);
        __syncthreads();
    }
    // the prefilter was right iff the list's own band ends where it did (a removed row among the best makes it too tight: once more, without)
    if (!(pre < __builtin_inff()) || (sz >= band_k && sh.lst_d[0][band_k - 1] + 2.0f * band_eps[ql] <= pre)) break;
    pre = __builtin_inff();
    sz = sz0;  // once more from the list the earlier blocks left
    __syncthreads();
    for (uint32_t i = lane; i < sz; i += kWave) {
        sh.lst_d[0][i] = list_d[(size_t)qg * C + i];
        sh.lst_s[0][i] = (uint32_t)list_slot[(size_t)qg * C + i];
    }
    __syncthreads();
    }
    for (uint32_t i = lane; i < sz; i += kWave) {
        list_d[(size_t)qg * C + i] = sh.lst_d[0][i];
        list_slot[(size_t)qg * C + i] = (uint64_t)(sh.lst_s[0][i] & kSlotMask);
    }
    if (lane == 0) {
        list_n[qg] = sz;
        float t = sz == C ? sh.lst_d[0][C - 1] : __builtin_inff();
        // One-product pass: a row whose approximate score is more than 2 eps behind the k-th best so far cannot be among the true
        // k best (the k rows in front of it are exactly within eps of their scores, it is exactly within eps of its own), so the
        // tiles need not pass it on, whatever the nominee list still has room for.
        if (band_eps && band_k && sz >= band_k) t = fminf(t, sh.lst_d[0][band_k - 1] + 2.0f * band_eps[ql]);
        thr[ql] = t;
    }
}
__global__ __launch_bounds__(64) void block_merge_kernel(IndexView ix, uint32_t q0, uint32_t C, uint32_t cand_cap, uint32_t* cand_cnt,
                                                         const uint2* cand, uint64_t* list_slot, float* list_d, uint32_t* list_n, float* thr,
                                                         uint32_t* uncertified, const float* band_eps = nullptr, uint32_t band_k = 0) {
    __shared__ SelectShared sh;
    block_merge_body(sh, ix, q0, C, cand_cap, cand_cnt, cand, list_slot, list_d, list_n, thr, uncertified, band_eps, band_k);
}
// (round 6: the 8-bit plane's band is six times the bf16 plane's -- lists of up to 512 nominees)
using SelectShared512 = BeamShared<512, 256>;
__global__ __launch_bounds__(64) void block_merge512_kernel(IndexView ix, uint32_t q0, uint32_t C, uint32_t cand_cap, uint32_t* cand_cnt,
                                                            const uint2* cand, uint64_t* list_slot, float* list_d, uint32_t* list_n, float* thr,
                                                            uint32_t* uncertified, const float* band_eps, uint32_t band_k, uint32_t parts) {
    __shared__ SelectShared512 sh;
    __shared__ uint32_t skey[kP8CandCap];
    block_merge_body(sh, ix, q0, C, cand_cap, cand_cnt, cand, list_slot, list_d, list_n, thr, uncertified, band_eps, band_k, skey, parts);
}

// exact f32 score of nominee c of query q: one wave per (query, nominee)
__global__ __launch_bounds__(256) void block_rescore_kernel(IndexView ix, const float* qd, uint32_t kpad, const float* q_aux,
                                                            uint32_t C, const uint64_t* cand_slot, const uint32_t* cand_found,
                                                            float* exact_d) {
    const uint32_t q = blockIdx.x, c = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (c >= cand_found[q]) return;
    const uint32_t slot = (uint32_t)cand_slot[(size_t)q * C + c];
    float acc = 0.f;
    for (uint32_t k = (uint32_t)lane * 4u; k < kpad; k += kWave * 4u) {
        float v[4];
        load4_dequant(ix, (size_t)slot, k, v);
        const float4 a = *reinterpret_cast<const float4*>(qd + (size_t)q * kpad + k);
        acc = fmaf(a.x, v[0], acc);
        acc = fmaf(a.y, v[1], acc);
        acc = fmaf(a.z, v[2], acc);
        acc = fmaf(a.w, v[3], acc);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    const bool aux = exact_needs_aux(ix);
    if (lane == 0) exact_d[(size_t)q * C + c] = finalize_exact(ix, acc, aux ? q_aux[q] : 0.f, aux ? ix.aux[slot] : 0.f);
}

// per query: the nominees ordered by their exact scores, the k best out, and the certificate
template <uint32_t CMAX>
__global__ __launch_bounds__(256) void block_final_kernel_t(IndexView ix, uint32_t k, uint32_t C, const uint64_t* cand_slot, const float* cand_approx,
                                                           const uint32_t* cand_found, const float* exact_d, const float* qnorm,
                                                           float max_row_norm, uint64_t* out_keys, float* out_dist, uint32_t* out_found,
                                                           uint32_t* uncertified, const float* eps_q = nullptr) {
    __shared__ float sd[CMAX];
    __shared__ uint32_t ss[CMAX];
    __shared__ float kth_s;
    const uint32_t q = blockIdx.x;
    const uint32_t lane = threadIdx.x, step = blockDim.x;  // (one wave for lists of 256; four for the 8-bit plane's 512: the ranking is n^2)
    const uint32_t n = cand_found[q];
    if (lane == 0) kth_s = -__builtin_inff();
    for (uint32_t i = lane; i < n; i += step) {
        const float d = exact_d[(size_t)q * C + i];
        sd[i] = d == d ? d : __builtin_inff();
        ss[i] = (uint32_t)cand_slot[(size_t)q * C + i];
    }
    __syncthreads();
    const uint32_t found = n < k ? n : k;
    for (uint32_t i = lane; i < n; i += step) {  // rank by (score, slot): a strict total order
        const float d = sd[i];
        const uint32_t sl = ss[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += key_less(sd[j], ss[j], d, sl) ? 1u : 0u;
        if (rank < k) {
            out_keys[(size_t)q * k + rank] = ix.keys[sl];
            out_dist[(size_t)q * k + rank] = d;
        }
        if (rank + 1 == found) kth_s = d;  // (ranks are distinct: one writer)
    }
    for (uint32_t i = found + lane; i < k; i += step) {
        out_keys[(size_t)q * k + i] = kFreeKey;
        out_dist[(size_t)q * k + i] = __builtin_inff();
    }
    __syncthreads();
    const float kth = kth_s;
    if (lane == 0) {
        out_found[q] = found;
        if (n == C) {  // rows outside the nominees exist: they score at least (worst nominated approximate score - eps)
            // split residuals (3 x 2^-18) + worst-case f32 accumulation of both scores (2 x K x 2^-24), times |q| |c|
            const float scale = ix.metric == COS ? 1.f : qnorm[q] * max_row_norm;
            // eps_q != nullptr: the one-product nomination pass's own bound (p1_eps_kernel) + the exact score's f32 accumulation
            const float eps = eps_q ? eps_q[q] + 1.2e-7f * (float)((ix.dim + 31u) & ~31u) * scale
                                    : (1.15e-5f + 1.2e-7f * (float)((ix.dim + 31u) & ~31u)) * 1.05f * scale + 1e-6f;
            const float t = cand_approx[(size_t)q * C + C - 1];
            if (!(kth < t - eps)) atomicAdd(uncertified, 1u);
        }
    }
}

// max |row| over rows [first, first + n) (inner product only: scales the certificate's eps); atomicMax on the f32 bits
__global__ __launch_bounds__(256) void row_norm_max_kernel(IndexView ix, uint32_t first, uint32_t n, uint32_t kpad, uint32_t* max_bits) {
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (r >= n) return;
    float sq = 0.f;
    for (uint32_t k = (uint32_t)lane * 4u; k < kpad; k += kWave * 4u) {
        float v[4];
        load4_dequant(ix, (size_t)(first + r), k, v);
        sq = fmaf(v[0], v[0], sq);
        sq = fmaf(v[1], v[1], sq);
        sq = fmaf(v[2], v[2], sq);
        sq = fmaf(v[3], v[3], sq);
    }
    for (int o = 32; o; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0 && sq == sq) atomicMax(max_bits, __float_as_uint(sqrtf(sq)));
}

bool block_search_supported(const IndexView& ix, uint32_t k) {
    return (ix.scalar == SC_F32 || ix.scalar == SC_F16 || ix.scalar == SC_BF16) && (ix.metric == COS || ix.metric == IP) && k >= 1 &&
           k * 4 <= kBlockC;
}
static uint32_t block_nominees(uint32_t k) {  // 4 k, at least 64: the margin the certificate lives on
    uint32_t c = 64;
    while (c < 4 * k) c *= 2;
    return c;
}

size_t block_scratch_bytes(uint32_t nq, uint32_t dim) {
    const size_t S = exact_segments(nq), kpad = (dim + 31u) & ~31u;
    return (size_t)kExactQB * kExactCH * 4 + (size_t)nq * S * kBlockC * 8 + (size_t)nq * S * 4 + (size_t)nq * 8 +
           (size_t)nq * kpad * 8 + (size_t)nq * kBlockC * 16 + (size_t)nq * 4 + 4096;
}

// Filtered search with a lazily evaluated predicate (engine.hip filtered_lazy): the host's verdicts for the slots the last
// walk listed go into the query's `known` / `allow` bitmaps, which live on the device for the whole query.
__global__ void apply_verdicts_kernel(const uint32_t* __restrict__ list, const uint8_t* __restrict__ verdict, uint32_t m, uint32_t slots,
                                      uint32_t* __restrict__ allow, uint32_t* __restrict__ known) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t s = list[i];
    if (s >= slots) return;
    atomicOr(&known[s >> 5], 1u << (s & 31u));
    if (verdict[i]) atomicOr(&allow[s >> 5], 1u << (s & 31u));
}
__global__ void reset_round_kernel(uint32_t* unknown) {  // [listed, consulted]: zero for the next walk (after the apply kernel has read the list)
    if (threadIdx.x < 2) unknown[threadIdx.x] = 0u;
}

// One round's outcome -> the caller's pinned block, written by the device itself: counters, the answer, the listed slots.  (Five
// small copies through the copy engines instead: each waits for its stream's walk on a shared engine queue, and with many
// filtered calls in flight the copies of all of them queued up behind whichever walk was slowest.)
__global__ void export_round_kernel(const uint32_t* __restrict__ unknown, uint32_t cap, const uint64_t* __restrict__ d_k,
                                    const float* __restrict__ d_d, const uint32_t* __restrict__ d_f, uint32_t k, uint32_t* __restrict__ h_cnt,
                                    uint32_t* __restrict__ h_list, uint64_t* __restrict__ h_k, float* __restrict__ h_d) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, total = gridDim.x * blockDim.x;
    const uint32_t count = unknown[0] < cap ? unknown[0] : cap;
    for (uint32_t i = tid; i < count; i += total) h_list[i] = unknown[64 + i];
    for (uint32_t i = tid; i < k; i += total) {
        h_k[i] = d_k[i];
        h_d[i] = d_d[i];
    }
    if (tid == 0) {
        h_cnt[0] = unknown[0];
        h_cnt[1] = unknown[1];
        h_cnt[2] = d_f[0];
    }
}

hipError_t launch_export_round(const uint32_t* unknown, uint32_t cap, const uint64_t* d_k, const float* d_d, const uint32_t* d_f, uint32_t k,
                               uint32_t* h_cnt, uint32_t* h_list, uint64_t* h_k, float* h_d, hipStream_t s) {
    hipLaunchKernelGGL(export_round_kernel, dim3(32), dim3(256), 0, s, unknown, cap, d_k, d_d, d_f, k, h_cnt, h_list, h_k, h_d);
    return hipGetLastError();
}

hipError_t launch_apply_verdicts(uint32_t* unknown, const uint8_t* verdict, uint32_t m, uint32_t slots, uint32_t* allow, uint32_t* known,
                                 hipStream_t s) {
    if (m) hipLaunchKernelGGL(apply_verdicts_kernel, dim3((m + 255) / 256), dim3(256), 0, s, unknown + 64, verdict, m, slots, allow, known);
    hipLaunchKernelGGL(reset_round_kernel, dim3(1), dim3(64), 0, s, unknown);
    return hipGetLastError();
}

// A filter's remembered verdicts (PipeQuery::memo: [allow | known], `stride` words each) forget the slots whose member changed (removed,
// or re-used by another key): both bits, so that a later verdict of 0 finds no stale 1.  No search runs meanwhile (usearch.rs:590-612).
__global__ void memo_forget_kernel(uint32_t* __restrict__ memo, uint32_t stride, const uint32_t* __restrict__ slots, uint32_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t s = slots[i];
    if ((s >> 5) >= stride) return;
    atomicAnd(&memo[s >> 5], ~(1u << (s & 31u)));
    atomicAnd(&memo[stride + (s >> 5)], ~(1u << (s & 31u)));
}
hipError_t launch_memo_forget(uint32_t* memo, uint32_t stride, const uint32_t* slots, uint32_t m, hipStream_t s) {
    if (!m) return hipSuccess;
    hipLaunchKernelGGL(memo_forget_kernel, dim3((m + 255) / 256), dim3(256), 0, s, memo, stride, slots, m);
    return hipGetLastError();
}

hipError_t launch_row_norm_max(const IndexView& ix, uint32_t first, uint32_t n, uint32_t* d_max_bits, hipStream_t s) {
    if (!n) return hipSuccess;
    const uint32_t kpad = (ix.dim + 31u) & ~31u;
    hipLaunchKernelGGL(row_norm_max_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ix, first, n, kpad, d_max_bits);
    return hipGetLastError();
}

// d_uncertified: one zeroed word; after the stream has drained, non-zero means: re-run with launch_exact.
hipError_t launch_block_search(const ExactArgs& a, void* scratch, float max_row_norm, uint32_t* d_uncertified, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (!block_search_supported(a.ix, a.k) || a.slots == 0) return hipErrorInvalidValue;
    const uint32_t kpad = (a.ix.dim + 31u) & ~31u;  // the stage depth of the tile kernel
    const uint32_t S = exact_segments(a.nq), C = block_nominees(a.k);
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    float* D = (float*)take((size_t)kExactQB * kExactCH * 4);
    float* st_d = (float*)take((size_t)a.nq * S * kBlockC * 4);
    uint32_t* st_s = (uint32_t*)take((size_t)a.nq * S * kBlockC * 4);
    uint32_t* st_n = (uint32_t*)take((size_t)a.nq * S * 4);
    float* q_aux = (float*)take((size_t)a.nq * 4);
    float* qnorm = (float*)take((size_t)a.nq * 4);
    float* qd = (float*)take((size_t)a.nq * kpad * 4);
    uint16_t* qh = (uint16_t*)take((size_t)a.nq * kpad * 2);
    uint16_t* ql = (uint16_t*)take((size_t)a.nq * kpad * 2);
    uint64_t* cand_slot = (uint64_t*)take((size_t)a.nq * kBlockC * 8);
    float* cand_approx = (float*)take((size_t)a.nq * kBlockC * 4);
    float* exact_d = (float*)take((size_t)a.nq * kBlockC * 4);
    uint32_t* cand_found = (uint32_t*)take((size_t)a.nq * 4);
    hipError_t e = prepare_queries(a.ix, a.queries, a.q_stride, a.nq, kpad, qd, q_aux, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(split_queries_kernel, dim3((a.nq + 3) / 4), dim3(256), 0, s, qd, a.nq, kpad, qh, ql, qnorm);
    for (uint32_t q0 = 0; q0 < a.nq; q0 += kExactQB) {
        const uint32_t nqb = a.nq - q0 < kExactQB ? a.nq - q0 : kExactQB;
        // The first row block goes through the score block D and the segmented select pass, which leaves each query's nominee
        // list (cand_*); from then on that list's worst score is a threshold few scores pass (C per query over the second
        // block, fewer later), so the tiles append those to per-query buffers -- kept where D was: it is free by then -- and
        // one wave per query merges them: no 67 MB score block per launch, no pass over it.
        uint2* cand = reinterpret_cast<uint2*>(D);
        uint32_t* cand_cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(D) + (size_t)kExactQB * kBlockCandCap * 8);
        float* thr = reinterpret_cast<float*>(cand_cnt + kExactQB);
        for (uint32_t n0 = 0; n0 < a.slots; n0 += kExactCH) {
            const uint32_t nb = a.slots - n0 < kExactCH ? a.slots - n0 : kExactCH;
            const dim3 mg(((nqb + 127) / 128) * (((nb + 127) / 128 + 7) / 8 * 8));  // row blocks rounded up to the 8 XCDs
            const bool first = n0 == 0;
            const float* t = first ? nullptr : thr;
            switch (a.ix.scalar) {
                case SC_F32: hipLaunchKernelGGL((block_dist_bf16x3_kernel<SC_F32>), mg, dim3(512), 0, s, a.ix, qh, ql, kpad, q_aux, q0, nqb, n0, nb, D, t, cand, cand_cnt, kBlockCandCap); break;
                case SC_F16: hipLaunchKernelGGL((block_dist_bf16x3_kernel<SC_F16>), mg, dim3(512), 0, s, a.ix, qh, ql, kpad, q_aux, q0, nqb, n0, nb, D, t, cand, cand_cnt, kBlockCandCap); break;
                default: hipLaunchKernelGGL((block_dist_bf16x3_kernel<SC_BF16>), mg, dim3(512), 0, s, a.ix, qh, ql, kpad, q_aux, q0, nqb, n0, nb, D, t, cand, cand_cnt, kBlockCandCap); break;
            }
            if (first) {
                hipLaunchKernelGGL(exact_select_kernel, dim3(nqb, S), dim3(64), 0, s, a.ix, D, q0, n0, nb, C, 1, 1, st_d, st_s, st_n, cand_slot,
                                   cand_approx, cand_found, 1);
                if (S > 1)
                    hipLaunchKernelGGL(exact_finish_kernel, dim3(nqb), dim3(64), 0, s, a.ix, q0, S, C, st_d, st_s, st_n, cand_slot, cand_approx,
                                       cand_found, 1);
                if (n0 + kExactCH >= a.slots) break;
                e = hipMemsetAsync(cand_cnt, 0, (size_t)kExactQB * 4, s);  // D is free from here on
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(block_merge_kernel, dim3(nqb), dim3(64), 0, s, a.ix, q0, C, kBlockCandCap, cand_cnt, cand, cand_slot, cand_approx,
                               cand_found, thr, d_uncertified);
        }
    }
    hipLaunchKernelGGL(block_rescore_kernel, dim3(a.nq, C / 4), dim3(256), 0, s, a.ix, qd, kpad, q_aux, C, cand_slot, cand_found, exact_d);
    hipLaunchKernelGGL((block_final_kernel_t<kBlockC>), dim3(a.nq), dim3(64), 0, s, a.ix, a.k, C, cand_slot, cand_approx, cand_found, exact_d, qnorm,
                       max_row_norm, a.out_keys, a.out_dist, a.out_found, d_uncertified);
    return hipGetLastError();
}

// ---------------------------------------------------------------- block search, ONE bf16 product per score (round 3)
// The same exact answer with a third of the matrix work and half of the HBM bytes of the split-bf16 pass above.  The index
// keeps a bf16 PLANE of its rows (round to nearest, built lazily and incrementally next to the f32 / f16 rows: 288 GB of
// HBM are there to be used), the queries are rounded to bf16 once per batch (cosine: after scaling by 1 / |q|), and
//     s~(q, c) = sum_k qh[k] ch[k]            (v_mfma_f32_16x16x32_bf16, f32 accumulate)
// nominates.  Error of a score, rigorous: |q.c - qh.ch| <= |q - qh| |c| + |qh| |c - ch| + K 2^-24 |qh| |c| with
// |q - qh| measured per query (r_q) and |c - ch| <= rho |c|, rho = the largest relative rounding residual over the rows of
// the plane (measured when the plane is built, <= 2^-9).  The C = 256 best approximate scores per query are re-scored with
// exact f32 arithmetic and certified exactly as above (k-th exact score below the C-th approximate score - eps); a batch
// with an uncertified query falls back to the split-bf16 path, and from there to the f32 path.
//
// Tile kernel: 256 queries (all of a batch) x 256 rows per workgroup of 8 waves (2 x 4: 128 x 64 per wave, 128 accumulator
// registers), K in steps of 64 through LDS rings filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no
// ds_write), XOR-swizzled on the SOURCE side so that the fragment reads (ds_read_b128) are bank-conflict free
// (scripts/probe/lds_swizzle_check.py), one raw s_barrier per step with counted vmcnt waits, persistent over the row tiles
// of a launch (the ring runs on into the next tile during the epilogue).  Epilogue: common path branch-free (largest
// score - threshold per 16 x 16 tile); a tile that holds a nominee goes through a per-wave LDS scratch and appends.
// Both operands are stored TILE-MAJOR in LDS-image order: the 32 KiB a (256-row tile, 64-deep K step) stage occupies in LDS is
// one contiguous block of the plane (and of the rounded query block), already XOR-swizzled, so a stage is filled by 32
// wave-instructions that each copy 1 KiB of consecutive memory -- whole DRAM pages instead of 128 bytes out of every 1,536
// (row-major: 3.9 TB/s of plane reads at best, scripts/probe/tile1_probe.hip).  The plane is a derived structure, read by
// nothing else (the re-score reads the f32 rows), so its layout is free.
// Measured structure and variants: scripts/probe/tile1_probe.hip.
constexpr int kP1TN = 256, kP1BK = 64, kP1RA = 2, kP1RB = 2;
constexpr uint32_t kP1C = 256;            // nominees per query
constexpr uint32_t kP1FirstRows = 1024;   // rows whose scores are all kept (4 tiles), <= kBlockCandCap
using f32x4v = __attribute__((ext_vector_type(4))) float;
using i32x4v = __attribute__((ext_vector_type(4))) int;

__device__ __forceinline__ uint32_t p1_swz(uint32_t row) { return (row >> 1) & 7u; }
// byte offset of element k of row r in a tile-major operand of `ksteps` K steps per tile (k a multiple of 4: 8 bytes stay together)
__device__ __forceinline__ size_t p1_offset(uint32_t r, uint32_t k, uint32_t ksteps) {
    const uint32_t t = r >> 8, rt = r & 255u, ks = k >> 6, kw = k & 63u, chunk = kw >> 3;
    return ((size_t)(t * ksteps + ks) * 256u + rt) * 128u + ((chunk ^ p1_swz(rt)) << 4) + ((kw & 7u) << 1);
}
template <int AUX>  // AUX 2: non-temporal (the plane streams through once per query block and must not displace the query block from L2)
__device__ __forceinline__ void p1_glds16(const void* g, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, AUX);
}
template <int N>
__device__ __forceinline__ void p1_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// plane (tile-major, p1_offset) := bf16(row r) (zero beyond dim and for rows >= slots); *rho_bits = max over rows of |c - ch| / |c|
__global__ __launch_bounds__(256) void p1_plane_rows_kernel(IndexView ix, uint32_t first, uint32_t end, uint32_t slots, uint32_t kp,
                                                            uint16_t* plane, uint32_t* rho_bits) {
    const uint32_t r = first + blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (r >= end) return;
    float sq = 0.f, res = 0.f;
    for (uint32_t k = (uint32_t)lane * 4u; k < kp; k += kWave * 4u) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < slots) load4_dequant(ix, (size_t)r, k, v);
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = float_to_bf16_bits(v[j]);
            const float e = v[j] - bf16_bits_to_float(h[j]);
            sq = fmaf(v[j], v[j], sq);
            res = fmaf(e, e, res);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(plane) + p1_offset(r, k, kp >> 6)) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    }
    for (int o = 32; o; o >>= 1) {
        sq += __shfl_xor(sq, o);
        res += __shfl_xor(res, o);
    }
    if (lane == 0 && sq > 0.f && res == res) atomicMax(rho_bits, __float_as_uint(sqrtf(res / sq) * 1.0001f));
}

// qd (nq x kpad f32: the form the metric sees) -> A (rows_pad x kp bf16, tile-major; cosine: scaled by q_aux = 1 / |q| first), |A_q|, r_q
__global__ __launch_bounds__(256) void p1_round_queries_kernel(const float* qd, const float* q_aux, int cosine, uint32_t nq, uint32_t rows_pad,
                                                               uint32_t kpad, uint32_t kp, uint16_t* A, float* a_norm, float* r_q) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= rows_pad) return;
    const float sc = w < nq ? (cosine ? q_aux[w] : 1.f) : 0.f;
    float an = 0.f, rs = 0.f;
    for (uint32_t e = (uint32_t)lane * 4u; e < kp; e += kWave * 4u) {
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = (w < nq && e + j < kpad) ? qd[(size_t)w * kpad + e + j] * sc : 0.f;
            h[j] = float_to_bf16_bits(x);
            const float hv = bf16_bits_to_float(h[j]), d = x - hv;
            an = fmaf(hv, hv, an);
            rs = fmaf(d, d, rs);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(A) + p1_offset(w, e, kp >> 6)) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    }
    for (int o = 32; o; o >>= 1) {
        an += __shfl_xor(an, o);
        rs += __shfl_xor(rs, o);
    }
    if (lane == 0 && w < nq) {
        a_norm[w] = sqrtf(an) * 1.0001f;
        r_q[w] = sqrtf(rs) * 1.0001f;
    }
}

// eps_q = bound of |approximate score - exact score| for query q (see the header of this section); a zero-norm cosine query
// has no meaningful approximate scores (every row ties at 1 while SimSIMD's zero rules order them): the batch is left to the
// other paths.
__global__ void p1_eps_kernel(uint32_t nq, const float* q_aux, const float* a_norm, const float* r_q, float rho, float row_norm_max, float kdim, int cosine,
                              float* eps_q, uint32_t* uncertified) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (cosine && !(q_aux[q] > 0.f)) atomicAdd(uncertified, 1u);
    const float e = (r_q[q] + a_norm[q] * (rho + 1.2e-7f * kdim)) * row_norm_max;
    eps_q[q] = e * 1.05f + 2e-6f;
}

// Scores of queries [0, 256) of A against plane rows [n_begin, n_end) (n_begin a multiple of 256; the plane is padded to
// whole tiles).  FIRST (the first rows of a search, no thresholds yet): EVERY score goes to cand[q][n - n_begin] as (distance
// bits, slot) by plain stores -- the host sets cand_cnt[q] = n_end - n_begin; else: scores at or above 1 - thr[q] are appended
// to cand[q] (atomic counter).  row_scale: cosine: 1 / |row| (aux).
// I8 (round 6): the same kernel over an 8-BIT plane -- A and B hold int8 (per-query / per-row scale), a K step is 128 elements (the same
// 128 bytes per row, the same LDS image, the same fragment reads), the product is v_mfma_i32_16x16x64_i8 (exact integer accumulation,
// twice the bf16 rate) and a score is acc * q_scale[q] * row_scale[n] (row_scale: the row's quantisation step, times 1 / |row| for
// cosine).  Half the plane's bytes, half the LDS traffic, half the matrix time.  `kp` is the row length in 2-byte units either way
// (bf16: elements; int8: elements / 2).
template <bool WRITE_D, bool I8 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void p1_tile_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, uint32_t kp, uint32_t nq_blk, uint32_t n_begin, uint32_t n_end,
    const float* __restrict__ thr, const float* __restrict__ row_scale, float* __restrict__ D, uint2* __restrict__ cand,
    uint32_t* __restrict__ cand_cnt, uint32_t cand_cap, const float* __restrict__ q_scale = nullptr, uint32_t parts = 1) {
    constexpr int TM = 256, TN = kP1TN, BK = kP1BK, RA = kP1RA, RB = kP1RB, CH = BK / 8;
    constexpr int NW = 8;
    constexpr bool SPLIT = RA != RB;
    constexpr int A_BYTES = TM * BK * 2, B_BYTES = TN * BK * 2, RING_BYTES = RA * A_BYTES + RB * B_BYTES;
    constexpr int A_LW = SPLIT ? NW / 2 : NW, B_LW = SPLIT ? NW / 2 : NW;
    constexpr int A_PW = A_BYTES / 1024 / A_LW, B_PW = B_BYTES / 1024 / B_LW;  // 1-KiB pieces (one wave-instruction each) per loading wave per step
    constexpr int WN = TN / 4, FR = 16, WROWS = TM / 2, MT = WROWS / FR, NT = WN / FR, ACC = 4;
    extern __shared__ __attribute__((aligned(1024))) char p1_lds[];
    char* lds = p1_lds;
    float* thr_s = reinterpret_cast<float*>(lds + RING_BYTES);
    float* qs_s = reinterpret_cast<float*>(lds + RING_BYTES + 1024);  // I8: the queries' quantisation steps
    using acc_t = typename std::conditional<I8, i32x4v, f32x4v>::type;
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
    const bool loads_a = !SPLIT || w < (uint32_t)(NW / 2), loads_b = !SPLIT || w >= (uint32_t)(NW / 2);
    const uint32_t la = w, lb = SPLIT ? w - NW / 2 : w;
    const uint32_t ksteps = kp / BK;
    const uint32_t tile0 = n_begin / TN, n_tiles = (n_end - n_begin + TN - 1) / TN;
    // similarity thresholds: a score s is a nominee iff s >= 1 - thr; queries beyond the batch never nominate
    // (I8: the threshold is compared with acc * row_scale, so it is divided by the query's step here: one multiplication less per score)
    if (t < TM) {
        const float qs = I8 ? (t < nq_blk ? q_scale[t] : 0.f) : 1.f;
        if constexpr (I8) qs_s[t] = qs;
        thr_s[t] = (!WRITE_D && t < nq_blk && qs > 0.f) ? (1.0f - thr[t]) / qs : __builtin_inff();
    }

    uint32_t a_off[A_PW], b_off[B_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
        a_off[i] = (la * A_PW + i) * 1024 + lane * 16;  // tile-major operands: a stage is one contiguous block in LDS-image order
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
        b_off[i] = (lb * B_PW + i) * 1024 + lane * 16;
    }
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(B);
    const uint32_t my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const uint32_t total = my_tiles * ksteps;
    if (!total) return;
    // prefetch cursors (flattened step -> (tile, k step)) of the A and B loaders
    uint32_t pa = 0, pa_ks = 0, pb = 0, pb_ks = 0, pb_tile = blockIdx.x;
    auto stage_a = [&]() {
        char* base = lds + (pa % RA) * A_BYTES;
        const char* at = Ab + (size_t)pa_ks * A_BYTES;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) p1_glds16<0>(at + a_off[i], base + (la * A_PW + i) * 1024);
        ++pa;
        if (++pa_ks == ksteps) pa_ks = 0;
    };
    auto stage_b = [&]() {
        char* base = lds + RA * A_BYTES + (pb % RB) * B_BYTES;
        const char* bt = Bb + ((size_t)(tile0 + pb_tile) * ksteps + pb_ks) * B_BYTES;  // wave-uniform
#pragma unroll
        for (int i = 0; i < B_PW; ++i) p1_glds16<2>(bt + b_off[i], base + (lb * B_PW + i) * 1024);
        ++pb;
        if (++pb_ks == ksteps) {
            pb_ks = 0;
            pb_tile += gridDim.x;
        }
    };
    if (loads_a)
        for (int s = 0; s < RA - 1; ++s)
            if (pa < total) stage_a();
    if (loads_b)
        for (int s = 0; s < RB - 1; ++s)
            if (pb < total) stage_b();

    const uint32_t frow = lane & (FR - 1), fk = lane / FR;
    const uint32_t part = blockIdx.x & (parts - 1u), sub_cap = cand_cap / parts;
    uint32_t sg = 0;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        acc_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < ACC; ++r) acc[i][j][r] = 0;
        for (uint32_t ks = 0; ks < ksteps; ++ks, ++sg) {
            // stage sg has landed once at most (ring - 2) younger stages of this wave are in flight (the last steps drain)
            if constexpr (!SPLIT) {
                if (sg + RA - 2 < total) p1_wait_vm<(RA - 2) * (A_PW + B_PW)>();
                else p1_wait_vm<0>();
            } else if (loads_a) {
                if (sg + RA - 2 < total) p1_wait_vm<(RA - 2) * A_PW>();
                else p1_wait_vm<0>();
            } else {
                if (sg + RB - 2 < total) p1_wait_vm<(RB - 2) * B_PW>();
                else p1_wait_vm<0>();
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();  // everyone's pieces of stage sg are in LDS; everyone has finished reading stage sg - 1
            asm volatile("" ::: "memory");
            if (loads_a && pa < total) stage_a();
            if (loads_b && pb < total) stage_b();
            const char* abase = lds + (sg % RA) * A_BYTES;
            const char* bbase = lds + RA * A_BYTES + (sg % RB) * B_BYTES;
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                // (a fragment is 16 bytes per lane either way: eight bf16 of a 16 x 16 x 32 product, sixteen int8 of a 16 x 16 x 64 one)
                using frag_t = typename std::conditional<I8, i32x4v, bf16x8>::type;
                frag_t fa[MT], fb[NT];
                const uint32_t kc = kk * 4 + fk;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint32_t row = wm * WROWS + i * FR + frow;
                    fa[i] = *reinterpret_cast<const frag_t*>(abase + row * (BK * 2) + ((kc ^ p1_swz(row)) % CH) * 16);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const uint32_t row = wn * WN + j * FR + frow;
                    fb[j] = *reinterpret_cast<const frag_t*>(bbase + row * (BK * 2) + ((kc ^ p1_swz(row)) % CH) * 16);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (I8) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        // ---- epilogue: C/D layout -- column (row of the plane) on the lane, query rows in the registers
        const uint32_t nbase = n_begin + tile * TN + wn * WN + frow;
        float rs[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) rs[j] = !row_scale ? 1.f : (nbase + j * FR < n_end ? row_scale[nbase + j * FR] : 0.f);  // (the plane's padding rows have no aux)
        if constexpr (WRITE_D) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const uint32_t n = nbase + j * FR;
#pragma unroll
                    for (int r = 0; r < ACC; ++r) {
                        const uint32_t q = wm * WROWS + i * FR + 4 * fk + r;
                        if (n < n_end && q < nq_blk)
                            cand[(size_t)q * cand_cap + (n - n_begin)] = make_uint2(__float_as_uint(1.0f - (float)acc[i][j][r] * rs[j] * (I8 ? qs_s[q] : 1.f)), n);
                    }
                }
        } else {
            float tmax[MT][NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float4 th = *reinterpret_cast<const float4*>(&thr_s[wm * WROWS + i * FR + 4 * fk]);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float e0 = fmaf((float)acc[i][j][0], rs[j], -th.x), e1 = fmaf((float)acc[i][j][1], rs[j], -th.y);
                    const float e2 = fmaf((float)acc[i][j][2], rs[j], -th.z), e3 = fmaf((float)acc[i][j][3], rs[j], -th.w);
                    tmax[i][j] = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (__builtin_expect(__ballot(tmax[i][j] >= 0.f) != 0ull, 0)) {  // this 16 x 16 tile holds at least one nominee
                        // the tile's scores go through this wave's LDS scratch: a runtime index into acc[][] would put ALL the
                        // accumulators in scratch memory on every tile
                        float* sc = reinterpret_cast<float*>(lds + RING_BYTES + 2048) + w * (ACC * 64);
                        const uint32_t n = nbase + j * FR;
#pragma unroll
                        for (int r = 0; r < ACC; ++r) sc[r * 64 + lane] = (float)acc[i][j][r] * rs[j];
#pragma unroll 1
                        for (int r = 0; r < ACC; ++r) {
                            const uint32_t q = wm * WROWS + i * FR + 4 * fk + r;
                            const float v = sc[r * 64 + lane];
                            if (v >= thr_s[q] && n < n_end) {  // (piece `part` of the query's buffer: see CandParts in block_merge_body)
                                const uint32_t at = atomicAdd(&cand_cnt[part * 256u + q], 1u);
                                if (at < sub_cap) cand[(size_t)q * cand_cap + part * sub_cap + at] = make_uint2(__float_as_uint(1.0f - v * (I8 ? qs_s[q] : 1.f)), n);
                            }
                        }
                    }
                }
        }
    }
}
constexpr size_t kP1LdsBytes = (size_t)(kP1RA * 256 + kP1RB * kP1TN) * kP1BK * 2 + 2048 + 8 * 4 * 64 * 4;

bool block1_supported(const IndexView& ix, uint32_t k) {
    return (ix.scalar == SC_F32 || ix.scalar == SC_F16 || ix.scalar == SC_BF16) && (ix.metric == COS || ix.metric == IP) && k >= 1 && k <= 64;
}
uint32_t block1_plane_k(const IndexView& ix) { return (ix.dim + 63u) & ~63u; }
uint32_t block1_plane_rows(uint32_t slots) { return (slots + (uint32_t)kP1TN - 1u) / (uint32_t)kP1TN * (uint32_t)kP1TN; }

hipError_t launch_block1_plane_rows(const IndexView& ix, uint16_t* plane, uint32_t first, uint32_t end, uint32_t slots, uint32_t* d_rho_bits,
                                    hipStream_t s) {
    if (end <= first) return hipSuccess;
    hipLaunchKernelGGL(p1_plane_rows_kernel, dim3((end - first + 3) / 4), dim3(256), 0, s, ix, first, end, slots, block1_plane_k(ix), plane, d_rho_bits);
    return hipGetLastError();
}

size_t block1_scratch_bytes(uint32_t nq, uint32_t dim) {
    const size_t kpad = (dim + 31u) & ~31u, kp = (dim + 63u) & ~63u, rows = ((size_t)nq + 255) / 256 * 256;
    return (size_t)256 * kBlockCandCap * 8 + 256 * 8 + (size_t)nq * (4 * 5) + (size_t)nq * kpad * 4 + rows * kp * 2 + (size_t)nq * kP1C * 16 + 8192;
}

// d_uncertified: one zeroed word; non-zero after the stream has drained => run the split-bf16 / f32 path instead.
// plane: block1_plane_rows(slots) x block1_plane_k(ix) bf16, tile-major.  rho: see p1_plane_rows_kernel.  max_row_norm: inner product only.
hipError_t launch_block1_search(const ExactArgs& a, void* scratch, const uint16_t* plane, float rho, float max_row_norm,
                                uint32_t* d_uncertified, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (!block1_supported(a.ix, a.k) || a.slots < (1u << 16)) return hipErrorInvalidValue;
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    // the attribute belongs to the (kernel, device) pair: libvs_shards searches several devices from concurrent threads of one process
    static std::atomic<bool> attr_set[64];
    if (!attr_set[dev & 63].load(std::memory_order_acquire)) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(p1_tile_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP1LdsBytes);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(p1_tile_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP1LdsBytes);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_set[dev & 63].store(true, std::memory_order_release);  // (two threads setting it at once set the same value)
    }
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t kpad = (a.ix.dim + 31u) & ~31u, kp = block1_plane_k(a.ix), C = kP1C;
    const uint32_t rows_pad = (a.nq + 255u) / 256u * 256u;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    uint2* cand = (uint2*)take((size_t)256 * kBlockCandCap * 8);
    uint32_t* cand_cnt = (uint32_t*)take((size_t)256 * 4);
    float* thr = (float*)take((size_t)256 * 4);
    float* q_aux = (float*)take((size_t)a.nq * 4);
    float* a_norm = (float*)take((size_t)a.nq * 4);
    float* r_q = (float*)take((size_t)a.nq * 4);
    float* eps_q = (float*)take((size_t)a.nq * 4);
    uint32_t* cand_found = (uint32_t*)take((size_t)a.nq * 4);
    float* qd = (float*)take((size_t)a.nq * kpad * 4);
    uint16_t* A = (uint16_t*)take((size_t)rows_pad * kp * 2);
    uint64_t* cand_slot = (uint64_t*)take((size_t)a.nq * C * 8);
    float* cand_approx = (float*)take((size_t)a.nq * C * 4);
    float* exact_d = (float*)take((size_t)a.nq * C * 4);
    hipError_t e = prepare_queries(a.ix, a.queries, a.q_stride, a.nq, kpad, qd, q_aux, s);
    if (e != hipSuccess) return e;
    const int cosine = a.ix.metric == COS ? 1 : 0;
    hipLaunchKernelGGL(p1_round_queries_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, qd, q_aux, cosine, a.nq, rows_pad, kpad, kp, A, a_norm, r_q);
    // eps_q: bound of |approximate - exact| score: the band of the thresholds (block_merge_kernel) and the certificate of
    // block_final_kernel (instead of the split-bf16 bound)
    hipLaunchKernelGGL(p1_eps_kernel, dim3((a.nq + 255) / 256), dim3(256), 0, s, a.nq, q_aux, a_norm, r_q, rho, cosine ? 1.f : max_row_norm,
                       (float)((a.ix.dim + 63u) & ~63u), cosine, eps_q, d_uncertified);
    e = hipMemsetAsync(cand_found, 0, (size_t)a.nq * 4, s);  // the nominee lists start empty
    if (e != hipSuccess) return e;
    const float* row_scale = cosine ? a.ix.aux : nullptr;
    for (uint32_t q0 = 0; q0 < a.nq; q0 += 256) {
        const uint32_t nqb = a.nq - q0 < 256u ? a.nq - q0 : 256u;
        const uint16_t* Aq = A + (size_t)q0 * kp;  // (tile-major: query block q0 / 256 starts at row q0)
        // The first kP1FirstRows rows: no thresholds yet, every score is stored (plain stores, no atomics) and one merge per query
        // builds the first nominee list.  Then chunks that grow 32 x, each followed by a merge that refreshes the thresholds (the
        // worst score on a full nominee list, or -- far tighter -- the k-th best so far + 2 eps): about k (32 - 1) rows per query
        // and chunk beat the threshold when the rows come in no particular order; a buffer that overflows (rows stored best-last)
        // raises `uncertified` and the batch goes to the split-bf16 / f32 paths.
        const uint32_t n1 = a.slots < kP1FirstRows ? a.slots : kP1FirstRows;
        hipLaunchKernelGGL((p1_tile_kernel<true>), dim3((n1 + kP1TN - 1) / kP1TN), dim3(512), kP1LdsBytes, s, Aq, plane, kp, nqb, 0u, n1, (const float*)nullptr, row_scale,
                           (float*)nullptr, cand, cand_cnt, (uint32_t)kBlockCandCap);
        e = hipMemsetD32Async((hipDeviceptr_t)cand_cnt, (int)n1, 256, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(block_merge_kernel, dim3(nqb), dim3(64), 0, s, a.ix, 0u, C, (uint32_t)kBlockCandCap, cand_cnt, cand, cand_slot + (size_t)q0 * C,
                           cand_approx + (size_t)q0 * C, cand_found + q0, thr, d_uncertified, (const float*)(eps_q + q0), a.k);
        for (uint32_t n0 = n1; n0 < a.slots;) {
            const uint64_t want = (uint64_t)n0 * 32u;
            const uint32_t nend = want >= a.slots ? a.slots : (uint32_t)want;
            const uint32_t tiles = (nend - n0 + kP1TN - 1) / kP1TN;
            const uint32_t grid = tiles < (uint32_t)cus ? tiles : (uint32_t)cus;
            hipLaunchKernelGGL((p1_tile_kernel<false>), dim3(grid), dim3(512), kP1LdsBytes, s, Aq, plane, kp, nqb, n0, nend, thr, row_scale, (float*)nullptr, cand,
                               cand_cnt, (uint32_t)kBlockCandCap);
            hipLaunchKernelGGL(block_merge_kernel, dim3(nqb), dim3(64), 0, s, a.ix, 0u, C, (uint32_t)kBlockCandCap, cand_cnt, cand, cand_slot + (size_t)q0 * C,
                               cand_approx + (size_t)q0 * C, cand_found + q0, thr, d_uncertified, (const float*)(eps_q + q0), a.k);
            n0 = nend;
        }
    }
    hipLaunchKernelGGL(block_rescore_kernel, dim3(a.nq, C / 4), dim3(256), 0, s, a.ix, qd, kpad, q_aux, C, cand_slot, cand_found, exact_d);
    hipLaunchKernelGGL((block_final_kernel_t<kBlockC>), dim3(a.nq), dim3(64), 0, s, a.ix, a.k, C, cand_slot, cand_approx, cand_found, exact_d, a_norm, max_row_norm,
                       a.out_keys, a.out_dist, a.out_found, d_uncertified, (const float*)eps_q);
    return hipGetLastError();
}

// ---------------------------------------------------------------- block search over an 8-BIT plane (round 6)
// The one-product pass above streams a bf16 plane: 2 bytes per element, and at q = 256 the HBM stream, the LDS traffic and the matrix
// pipe each need about the same 2-2.5k clocks per K step of a tile -- the batch sits at 0.40 of the plane's HBM floor whatever the tile
// shape.  An int8 plane halves all three at once: row c is stored as ci = round(c / s_c), s_c = max|c_k| / 127
// (one f32 per row: `scale`, which also carries 1 / |c| for cosine), the queries likewise per batch, and
//     s~(q, c) = s_q s_c sum_k qi[k] ci[k]          (v_mfma_i32_16x16x64_i8: exact integer accumulation)
// nominates.  Error, rigorous: |q.c - s~| <= |q - q^| |c| + |q^| |c - c^|, with |q - q^| measured per query (r_q) and
// |c - c^| <= rho8 |c|, rho8 = the largest relative quantisation residual over the rows of the plane (measured when the plane is built:
// ~0.008-0.012 for rows whose components look Gaussian) -- the same eps formula as the bf16 plane's with rho8 for rho, about six times
// wider.  Nominee lists are 512 long for it; re-score and certificate are unchanged; an uncertified batch goes on to the bf16 plane.
constexpr uint32_t kP8C = 512;
constexpr uint32_t kP8FirstRows = 2048;          // rows whose scores are all kept (<= kBlockCandCap)
constexpr uint32_t kP8SlowGrowthBelow = 262144;  // chunks grow 8 x up to here, 32 x beyond
uint32_t block8_plane_k(const IndexView& ix) { return (ix.dim + 127u) & ~127u; }  // elements (= bytes) per row: whole 128-element K steps
size_t block8_scratch_bytes(uint32_t nq, uint32_t dim) {
    const size_t kpad = (dim + 31u) & ~31u, kp = (dim + 127u) & ~127u, rows = ((size_t)nq + 255) / 256 * 256;
    return (size_t)256 * kP8CandCap * 8 + (size_t)kP8CandParts * 256 * 4 + 256 * 4 + (size_t)nq * (4 * 6) + rows * 4 + (size_t)nq * kpad * 4 + rows * kp + (size_t)nq * kP8C * 16 + 8192;
}
// byte offset of element k of row r in a tile-major int8 operand (k a multiple of 4: 4 bytes stay together): the bf16 layout with
// 16 elements per 16-byte chunk
__device__ __forceinline__ size_t p8_offset(uint32_t r, uint32_t k, uint32_t ksteps) {
    const uint32_t t = r >> 8, rt = r & 255u, ks = k >> 7, kw = k & 127u, chunk = kw >> 4;
    return ((size_t)(t * ksteps + ks) * 256u + rt) * 128u + ((chunk ^ p1_swz(rt)) << 4) + (kw & 15u);
}
__device__ __forceinline__ uint32_t p8_pack(const float v[4], float inv_step, float step, float& res) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float qf = rintf(v[j] * inv_step);
        qf = fminf(fmaxf(qf, -127.f), 127.f);
        const float e = v[j] - qf * step;
        res = fmaf(e, e, res);
        w |= ((uint32_t)(int)qf & 255u) << (8 * j);
    }
    return w;
}
// plane8 (tile-major, p8_offset) := int8(row r / step_r), scale[r] = step_r (x 1 / |row| for cosine: aux), zero rows beyond `slots`;
// *rho_bits = max over rows of |c - c^| / |c|
__global__ __launch_bounds__(256) void p8_plane_rows_kernel(IndexView ix, uint32_t first, uint32_t end, uint32_t slots, uint32_t kp, uint8_t* plane,
                                                            float* scale, uint32_t* rho_bits) {
    const uint32_t r = first + blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = lane_id();
    if (r >= end) return;
    float mx = 0.f, sq = 0.f;
    if (r < slots)
        for (uint32_t k = (uint32_t)lane * 4u; k < kp; k += kWave * 4u) {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (k < ((ix.dim + 3u) & ~3u)) load4_dequant(ix, (size_t)r, k, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = (k + j < ix.dim) ? v[j] : 0.f;
                mx = fmaxf(mx, fabsf(x));
                sq = fmaf(x, x, sq);
            }
        }
    for (int o = 32; o; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        sq += __shfl_xor(sq, o);
    }
    const bool ok = r < slots && mx > 0.f && mx < __builtin_inff() && sq == sq;  // (a row with a NaN / an infinity scores nothing here: the exact paths rank it)
    const float step = ok ? mx / 127.f : 0.f, inv = ok ? 127.f / mx : 0.f;
    float res = 0.f;
    for (uint32_t k = (uint32_t)lane * 4u; k < kp; k += kWave * 4u) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (ok && k < ((ix.dim + 3u) & ~3u)) load4_dequant(ix, (size_t)r, k, v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j >= ix.dim) v[j] = 0.f;
        *reinterpret_cast<uint32_t*>(plane + p8_offset(r, k, kp >> 7)) = p8_pack(v, inv, step, res);
    }
    for (int o = 32; o; o >>= 1) res += __shfl_xor(res, o);
    if (lane == 0) {
        scale[r] = ok ? step * (ix.metric == COS ? ix.aux[r] : 1.f) : 0.f;
        if (ok && res == res) atomicMax(rho_bits, __float_as_uint(sqrtf(res / sq) * 1.0001f));
        if (r < slots && !ok && mx != 0.f) atomicMax(rho_bits, __float_as_uint(__builtin_inff()));  // (an unrepresentable row: no bound holds, the plane is not used)
    }
}
// qd (nq x kpad f32) -> A (rows_pad x kp int8, tile-major; cosine: scaled by q_aux = 1 / |q| first), q_scale, |A_q| (dequantised), r_q
__global__ __launch_bounds__(256) void p8_round_queries_kernel(const float* qd, const float* q_aux, int cosine, uint32_t nq, uint32_t rows_pad,
                                                               uint32_t kpad, uint32_t kp, uint8_t* A, float* q_scale, float* a_norm, float* r_q) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= rows_pad) return;
    const float sc = w < nq ? (cosine ? q_aux[w] : 1.f) : 0.f;
    float mx = 0.f;
    for (uint32_t e = (uint32_t)lane * 4u; e < kp; e += kWave * 4u)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf((w < nq && e + j < kpad) ? qd[(size_t)w * kpad + e + j] * sc : 0.f));
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const bool ok = mx > 0.f && mx < __builtin_inff();
    const float step = ok ? mx / 127.f : 0.f, inv = ok ? 127.f / mx : 0.f;
    float an = 0.f, rs = 0.f;
    for (uint32_t e = (uint32_t)lane * 4u; e < kp; e += kWave * 4u) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (ok && w < nq && e + j < kpad) ? qd[(size_t)w * kpad + e + j] * sc : 0.f;
        float res = 0.f;
        const uint32_t word = p8_pack(v, inv, step, res);
        rs += res;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float hv = (float)(int8_t)((word >> (8 * j)) & 255u) * step;
            an = fmaf(hv, hv, an);
        }
        *reinterpret_cast<uint32_t*>(A + p8_offset(w, e, kp >> 7)) = word;
    }
    for (int o = 32; o; o >>= 1) {
        an += __shfl_xor(an, o);
        rs += __shfl_xor(rs, o);
    }
    if (lane == 0) {
        q_scale[w] = step;
        if (w < nq) {
            a_norm[w] = sqrtf(an) * 1.0001f;
            r_q[w] = ok ? sqrtf(rs) * 1.0001f : __builtin_inff();  // (a query that cannot be represented: its eps is infinite, the batch goes on to the other paths)
        }
    }
}

hipError_t launch_block8_plane_rows(const IndexView& ix, uint8_t* plane, float* scale, uint32_t first, uint32_t end, uint32_t slots, uint32_t* d_rho_bits,
                                    hipStream_t s) {
    if (end <= first) return hipSuccess;
    hipLaunchKernelGGL(p8_plane_rows_kernel, dim3((end - first + 3) / 4), dim3(256), 0, s, ix, first, end, slots, block8_plane_k(ix), plane, scale, d_rho_bits);
    return hipGetLastError();
}

// As launch_block1_search, over the int8 plane (block1_plane_rows(slots) x block8_plane_k(ix) bytes, tile-major) and its per-row scales.
hipError_t launch_block8_search(const ExactArgs& a, void* scratch, const uint8_t* plane, const float* scale, float rho, float max_row_norm,
                                uint32_t* d_uncertified, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (!block1_supported(a.ix, a.k) || a.slots < (1u << 16) || !(rho < 0.25f)) return hipErrorInvalidValue;
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<bool> attr_set[64];
    if (!attr_set[dev & 63].load(std::memory_order_acquire)) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(p1_tile_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP1LdsBytes);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(p1_tile_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP1LdsBytes);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_set[dev & 63].store(true, std::memory_order_release);
    }
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t kpad = (a.ix.dim + 31u) & ~31u, kp = block8_plane_k(a.ix), C = kP8C;
    const uint32_t kp2 = kp / 2;  // the tile kernel's row length, in 2-byte units
    const uint32_t rows_pad = (a.nq + 255u) / 256u * 256u;
    char* p = (char*)scratch;
    auto take = [&](size_t bytes) {
        char* r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    uint2* cand = (uint2*)take((size_t)256 * kP8CandCap * 8);
    uint32_t* cand_cnt = (uint32_t*)take((size_t)kP8CandParts * 256 * 4);
    float* thr = (float*)take((size_t)256 * 4);
    float* q_aux = (float*)take((size_t)a.nq * 4);
    float* a_norm = (float*)take((size_t)a.nq * 4);
    float* r_q = (float*)take((size_t)a.nq * 4);
    float* eps_q = (float*)take((size_t)a.nq * 4);
    uint32_t* cand_found = (uint32_t*)take((size_t)a.nq * 4);
    float* q_scale = (float*)take((size_t)rows_pad * 4);
    float* qd = (float*)take((size_t)a.nq * kpad * 4);
    uint8_t* A = (uint8_t*)take((size_t)rows_pad * kp);
    uint64_t* cand_slot = (uint64_t*)take((size_t)a.nq * C * 8);
    float* cand_approx = (float*)take((size_t)a.nq * C * 4);
    float* exact_d = (float*)take((size_t)a.nq * C * 4);
    hipError_t e = prepare_queries(a.ix, a.queries, a.q_stride, a.nq, kpad, qd, q_aux, s);
    if (e != hipSuccess) return e;
    const int cosine = a.ix.metric == COS ? 1 : 0;
    hipLaunchKernelGGL(p8_round_queries_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, s, qd, q_aux, cosine, a.nq, rows_pad, kpad, kp, A, q_scale, a_norm, r_q);
    hipLaunchKernelGGL(p1_eps_kernel, dim3((a.nq + 255) / 256), dim3(256), 0, s, a.nq, q_aux, a_norm, r_q, rho, cosine ? 1.f : max_row_norm, 0.f, cosine, eps_q,
                       d_uncertified);
    e = hipMemsetAsync(cand_found, 0, (size_t)a.nq * 4, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(cand_cnt, 0, (size_t)kP8CandParts * 256 * 4, s);  // (every merge leaves the counters it read at zero)
    if (e != hipSuccess) return e;
    const uint16_t* plane16 = reinterpret_cast<const uint16_t*>(plane);
    for (uint32_t q0 = 0; q0 < a.nq; q0 += 256) {
        const uint32_t nqb = a.nq - q0 < 256u ? a.nq - q0 : 256u;
        const uint16_t* Aq = reinterpret_cast<const uint16_t*>(A + (size_t)q0 * kp);
        const float* qs = q_scale + q0;
        // (the 8-bit band is six times the bf16 plane's: the first rows whose scores are all kept are 4,096 -- the candidate buffer's size
        // -- and the chunks grow 8 x while the thresholds are young, so that each merge sees hundreds of candidates, not thousands)
        // (VS_P8_FIRST / VS_P8_GROWTH / VS_P8_SLOW_BELOW: the schedule, for measurements)
        static const uint32_t first_rows = std::getenv("VS_P8_FIRST") ? (uint32_t)std::atoi(std::getenv("VS_P8_FIRST")) : kP8FirstRows;
        static const uint32_t slow_growth = std::getenv("VS_P8_GROWTH") ? (uint32_t)std::atoi(std::getenv("VS_P8_GROWTH")) : 8u;
        static const uint32_t slow_below = std::getenv("VS_P8_SLOW_BELOW") ? (uint32_t)std::atoi(std::getenv("VS_P8_SLOW_BELOW")) : kP8SlowGrowthBelow;
        const uint32_t n1 = a.slots < first_rows ? a.slots : first_rows;
        hipLaunchKernelGGL((p1_tile_kernel<true, true>), dim3((n1 + kP1TN - 1) / kP1TN), dim3(512), kP1LdsBytes, s, Aq, plane16, kp2, nqb, 0u, n1, (const float*)nullptr, scale,
                           (float*)nullptr, cand, cand_cnt, (uint32_t)kP8CandCap, qs);
        e = hipMemsetD32Async((hipDeviceptr_t)cand_cnt, (int)n1, 256, s);  // the first block's scores: one piece, all of them
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(block_merge512_kernel, dim3(nqb), dim3(64), 0, s, a.ix, 0u, C, (uint32_t)kP8CandCap, cand_cnt, cand, cand_slot + (size_t)q0 * C,
                           cand_approx + (size_t)q0 * C, cand_found + q0, thr, d_uncertified, (const float*)(eps_q + q0), a.k, 1u);
        for (uint32_t n0 = n1; n0 < a.slots;) {
            const uint64_t want = (uint64_t)n0 * (n0 < slow_below ? slow_growth : 32u);
            const uint32_t nend = want >= a.slots ? a.slots : (uint32_t)want;
            const uint32_t tiles = (nend - n0 + kP1TN - 1) / kP1TN;
            const uint32_t grid = tiles < (uint32_t)cus ? tiles : (uint32_t)cus;
            hipLaunchKernelGGL((p1_tile_kernel<false, true>), dim3(grid), dim3(512), kP1LdsBytes, s, Aq, plane16, kp2, nqb, n0, nend, thr, scale, (float*)nullptr, cand, cand_cnt,
                               (uint32_t)kP8CandCap, qs, (uint32_t)kP8CandParts);
            hipLaunchKernelGGL(block_merge512_kernel, dim3(nqb), dim3(64), 0, s, a.ix, 0u, C, (uint32_t)kP8CandCap, cand_cnt, cand, cand_slot + (size_t)q0 * C,
                               cand_approx + (size_t)q0 * C, cand_found + q0, thr, d_uncertified, (const float*)(eps_q + q0), a.k, (uint32_t)kP8CandParts);
            n0 = nend;
        }
    }
    hipLaunchKernelGGL(block_rescore_kernel, dim3(a.nq, C / 4), dim3(256), 0, s, a.ix, qd, kpad, q_aux, C, cand_slot, cand_found, exact_d);
    hipLaunchKernelGGL((block_final_kernel_t<kP8C>), dim3(a.nq), dim3(256), 0, s, a.ix, a.k, C, cand_slot, cand_approx, cand_found, exact_d, a_norm, max_row_norm,
                       a.out_keys, a.out_dist, a.out_found, d_uncertified, (const float*)eps_q);
    return hipGetLastError();
}

// ---------------------------------------------------------------- multi-GPU top-k merge (after the RCCL all-gather)
// key_stride / dist_stride: elements between the lists of consecutive parts (nq * k for two plain arrays; for the packed
// all-gather blocks [keys | distances] of include/vs_ranks.h both arrays advance by one block per part).
__global__ __launch_bounds__(64) void topk_merge_kernel(const uint64_t* part_keys, const float* part_dist, uint32_t parts,
                                                        uint32_t nq, uint32_t k, size_t key_stride, size_t dist_stride,
                                                        uint64_t* out_keys, float* out_dist, uint32_t* out_found) {
    __shared__ SelectShared sh;
    const int lane = lane_id();
    const uint32_t q = blockIdx.x;
    int cur = 0;
    uint32_t sz = 0;
    const uint32_t total = parts * k;
    for (uint32_t c0 = 0; c0 < total; c0 += kWave) {
        uint32_t c = c0 + lane;
        bool ok = c < total;
        uint32_t p = ok ? c / k : 0, j = ok ? c % k : 0;
        const size_t in_part = (size_t)q * k + j;
        float d = ok ? part_dist[(size_t)p * dist_stride + in_part] : __builtin_inff();
        ok = ok && part_keys[(size_t)p * key_stride + in_part] != kFreeKey;
        if (ok && sz == k) ok = key_less(d, c, sh.lst_d[cur][k - 1], sh.lst_s[cur][k - 1]);
        uint64_t mask = __ballot(ok);
        if (!mask) continue;
        uint32_t ma = (uint32_t)__popcll(mask);
        __syncthreads();
        if (ok) {
            uint32_t r = mbcnt(mask);
            sh.u_dist[r] = d;
            sh.u_slot[r] = c;
        }
        __syncthreads();
        float nd = (uint32_t)lane < ma ? sh.u_dist[lane] : __builtin_inff();
        uint32_t ns = (uint32_t)lane < ma ? sh.u_slot[lane] : kInvalid;
        sz = list_merge(sh, cur, sz, k, nd, ns, ma, lane);
        __syncthreads();
    }
    for (uint32_t i = lane; i < k; i += kWave) {
        uint64_t key = kFreeKey;
        float d = __builtin_inff();
        if (i < sz) {
            uint32_t c = sh.lst_s[cur][i];
            key = part_keys[(size_t)(c / k) * key_stride + (size_t)q * k + (c % k)];
            d = sh.lst_d[cur][i];
        }
        out_keys[(size_t)q * k + i] = key;
        out_dist[(size_t)q * k + i] = d;
    }
    if (lane == 0 && out_found) out_found[q] = sz;
}

hipError_t launch_topk_merge(const uint64_t* part_keys, const float* part_dist, uint32_t parts, uint32_t nq, uint32_t k,
                             uint64_t* out_keys, float* out_dist, uint32_t* out_found, hipStream_t s, size_t key_stride,
                             size_t dist_stride) {
    if (!nq) return hipSuccess;
    if (k == 0 || k > 256 || (size_t)parts * k >= 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (!key_stride) key_stride = (size_t)nq * k;
    if (!dist_stride) dist_stride = (size_t)nq * k;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(nq), dim3(64), 0, s, part_keys, part_dist, parts, nq, k, key_stride, dist_stride,
                       out_keys, out_dist, out_found);
    return hipGetLastError();
}

}  // namespace vs
