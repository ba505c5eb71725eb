// kernels_misc.hip -- row staging, norms, exact (brute-force) top-k and the multi-GPU top-k merge.
#include "kernels.hpp"

namespace vs {

// ---------------------------------------------------------------- row staging / small utilities
__global__ void inv_norms_kernel(IndexView ix, float* inv_norm, const uint32_t* slots, uint32_t first, uint32_t n) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= n) return;
    const uint32_t slot = slots ? slots[w] : first + w;
    const float4* row = ix.vectors + (size_t)slot * ix.stride4;
    float s = 0.f;
    for (uint32_t i = lane; i < ix.stride4; i += kWave) s = accumulate<KDOT>(s, row[i], row[i]);
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) inv_norm[slot] = s > 0.f ? 1.0f / sqrtf(s) : 0.f;
}

hipError_t launch_inv_norms(const IndexView& ix, float* inv_norm, const uint32_t* slots, uint32_t first, uint32_t n,
                            hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(inv_norms_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ix, inv_norm, slots, first, n);
    return hipGetLastError();
}

__global__ void scatter_rows_kernel(float* vectors, uint32_t stride_f, const float* src, uint32_t src_stride, uint32_t dim,
                                    const uint32_t* slots, uint32_t first) {
    const uint32_t r = blockIdx.x;
    const uint32_t slot = slots ? slots[r] : first + r;
    float* dst = vectors + (size_t)slot * stride_f;
    const float* s = src + (size_t)r * src_stride;
    for (uint32_t c = threadIdx.x; c < stride_f; c += blockDim.x) dst[c] = c < dim ? s[c] : 0.f;
}

hipError_t launch_scatter_rows(float* vectors, uint32_t stride_f, const float* src, uint32_t src_stride, uint32_t dim,
                               const uint32_t* slots, uint32_t first, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(stride_f >= 256 ? 256 : 64), 0, s, vectors, stride_f, src,
                       src_stride, dim, slots, first);
    return hipGetLastError();
}

__global__ void gather_rows_kernel(const float* vectors, uint32_t stride_f, float* dst, uint32_t dim) {
    const uint32_t r = blockIdx.x;
    for (uint32_t c = threadIdx.x; c < dim; c += blockDim.x) dst[(size_t)r * dim + c] = vectors[(size_t)r * stride_f + c];
}

hipError_t launch_gather_rows(const float* vectors, uint32_t stride_f, float* dst, uint32_t dim, uint32_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(dim >= 256 ? 256 : 64), 0, s, vectors, stride_f, dst, dim);
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

__global__ void query_inv_norms_kernel(const float* q, uint32_t q_stride, uint32_t dim, uint32_t nq, float* out) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= nq) return;
    float s = 0.f;
    for (uint32_t i = lane; i < dim; i += kWave) {
        float x = q[(size_t)w * q_stride + i];
        s = fmaf(x, x, s);
    }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[w] = s > 0.f ? 1.0f / sqrtf(s) : 0.f;
}

template <int KIND>
__global__ __launch_bounds__(256) void exact_dist_kernel(IndexView ix, const float* queries, uint32_t q_stride,
                                                         const float* q_inv, uint32_t q0, uint32_t nq_blk, uint32_t n0,
                                                         uint32_t n_blk, float* D) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const uint32_t qt = blockIdx.y * 64, nt = blockIdx.x * 64;
    const uint32_t lrow = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
    const uint32_t stride_f = ix.stride4 * 4;
    const float* vec = reinterpret_cast<const float*>(ix.vectors);
    float acc[4][4] = {};
    for (uint32_t k0 = 0; k0 < stride_f; k0 += 16) {
        {
            uint32_t qi = qt + lrow;
            const float* src = queries + (size_t)(q0 + qi) * q_stride;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t k = k0 + lk + j;
                As[lk + j][lrow] = (qi < nq_blk && k < ix.dim) ? src[k] : 0.f;
            }
            uint32_t ni = nt + lrow;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ni < n_blk && k0 + lk < stride_f)
                v = *reinterpret_cast<const float4*>(vec + (size_t)(n0 + ni) * stride_f + k0 + lk);
            Bs[lk + 0][lrow] = v.x;
            Bs[lk + 1][lrow] = v.y;
            Bs[lk + 2][lrow] = v.z;
            Bs[lk + 3][lrow] = v.w;
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t qi = qt + ty * 4 + i;
        if (qi >= nq_blk) continue;
        float qinv = ix.metric == COS ? q_inv[q0 + qi] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t ni = nt + tx * 4 + j;
            if (ni >= n_blk) continue;
            float binv = ix.metric == COS ? ix.inv_norm[n0 + ni] : 0.f;
            D[(size_t)qi * kExactCH + ni] = finalize(ix.metric, acc[i][j], qinv, binv);
        }
    }
}

using SelectShared = BeamShared<256, 256>;

__global__ __launch_bounds__(64) void exact_select_kernel(IndexView ix, const float* D, uint32_t q0, uint32_t n0,
                                                          uint32_t n_blk, uint32_t k, int first, int last, float* st_d,
                                                          uint32_t* st_s, uint32_t* st_n, uint64_t* out_keys,
                                                          float* out_dist, uint32_t* out_found) {
    __shared__ SelectShared sh;
    const int lane = lane_id();
    const uint32_t ql = blockIdx.x, qg = q0 + ql;
    int cur = 0;
    uint32_t sz = 0;
    if (!first) {
        sz = st_n[qg];
        for (uint32_t i = lane; i < sz; i += kWave) {
            sh.lst_d[0][i] = st_d[(size_t)qg * k + i];
            sh.lst_s[0][i] = st_s[(size_t)qg * k + i];
        }
    }
    __syncthreads();
    const float* row = D + (size_t)ql * kExactCH;
    for (uint32_t j0 = 0; j0 < n_blk; j0 += kWave) {
        uint32_t j = j0 + lane;
        bool ok = j < n_blk;
        float d = ok ? row[j] : __builtin_inff();
        uint32_t slot = n0 + j;
        ok = ok && ix.keys[ok ? slot : 0] != kFreeKey;
        if (ok && sz == k) ok = key_less(d, slot, sh.lst_d[cur][k - 1], sh.lst_s[cur][k - 1]);
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
    if (!last) {
        for (uint32_t i = lane; i < sz; i += kWave) {
            st_d[(size_t)qg * k + i] = sh.lst_d[cur][i];
            st_s[(size_t)qg * k + i] = sh.lst_s[cur][i];
        }
        if (lane == 0) st_n[qg] = sz;
        return;
    }
    for (uint32_t i = lane; i < k; i += kWave) {
        out_keys[(size_t)qg * k + i] = i < sz ? ix.keys[sh.lst_s[cur][i] & kSlotMask] : kFreeKey;
        out_dist[(size_t)qg * k + i] = i < sz ? sh.lst_d[cur][i] : __builtin_inff();
    }
    if (lane == 0) out_found[qg] = sz;
}

size_t exact_scratch_bytes(uint32_t nq, uint32_t k) {
    return (size_t)kExactQB * kExactCH * 4 + (size_t)nq * k * 8 + (size_t)nq * 8 + 256;
}

hipError_t launch_exact(const ExactArgs& a, void* scratch, hipStream_t s) {
    if (a.nq == 0) return hipSuccess;
    if (a.k == 0 || a.k > 256) return hipErrorInvalidValue;
    char* p = (char*)scratch;
    float* D = (float*)p;
    p += (size_t)kExactQB * kExactCH * 4;
    float* st_d = (float*)p;
    p += (size_t)a.nq * a.k * 4;
    uint32_t* st_s = (uint32_t*)p;
    p += (size_t)a.nq * a.k * 4;
    uint32_t* st_n = (uint32_t*)p;
    p += (size_t)a.nq * 4;
    float* q_inv = (float*)p;
    hipLaunchKernelGGL(query_inv_norms_kernel, dim3((a.nq + 3) / 4), dim3(256), 0, s, a.queries, a.q_stride, a.ix.dim,
                       a.nq, q_inv);
    if (a.slots == 0) {  // empty index: found = 0 everywhere
        hipLaunchKernelGGL(fill_u32_kernel, dim3(64), dim3(256), 0, s, a.out_found, 0u, (size_t)a.nq);
        hipLaunchKernelGGL(fill_u32_kernel, dim3(256), dim3(256), 0, s, (uint32_t*)a.out_keys, 0xFFFFFFFFu,
                           (size_t)a.nq * a.k * 2);
        hipLaunchKernelGGL(fill_u32_kernel, dim3(256), dim3(256), 0, s, (uint32_t*)a.out_dist, 0x7F800000u,
                           (size_t)a.nq * a.k);
        return hipGetLastError();
    }
    for (uint32_t q0 = 0; q0 < a.nq; q0 += kExactQB) {
        uint32_t nqb = a.nq - q0 < kExactQB ? a.nq - q0 : kExactQB;
        for (uint32_t n0 = 0; n0 < a.slots; n0 += kExactCH) {
            uint32_t nb = a.slots - n0 < kExactCH ? a.slots - n0 : kExactCH;
            dim3 grid((nb + 63) / 64, (nqb + 63) / 64);
            if (a.ix.metric == L2SQ)
                hipLaunchKernelGGL((exact_dist_kernel<KL2>), grid, dim3(256), 0, s, a.ix, a.queries, a.q_stride, q_inv, q0,
                                   nqb, n0, nb, D);
            else
                hipLaunchKernelGGL((exact_dist_kernel<KDOT>), grid, dim3(256), 0, s, a.ix, a.queries, a.q_stride, q_inv, q0,
                                   nqb, n0, nb, D);
            int first = n0 == 0, last = n0 + kExactCH >= a.slots;
            hipLaunchKernelGGL(exact_select_kernel, dim3(nqb), dim3(64), 0, s, a.ix, D, q0, n0, nb, a.k, first, last, st_d,
                               st_s, st_n, a.out_keys, a.out_dist, a.out_found);
        }
    }
    return hipGetLastError();
}

// One query against every row (exhaustive path of filtered search / k beyond the LDS beam).
__global__ void distance_row_kernel(IndexView ix, const float* q, uint32_t n, float* out) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = lane_id();
    if (w >= n) return;
    const float4* row = ix.vectors + (size_t)w * ix.stride4;
    float acc = 0.f, q2 = 0.f;
    for (uint32_t i = lane; i < ix.stride4; i += kWave) {
        float4 v = row[i];
        uint32_t e = i * 4;
        float4 qv;
        qv.x = e + 0 < ix.dim ? q[e + 0] : 0.f;
        qv.y = e + 1 < ix.dim ? q[e + 1] : 0.f;
        qv.z = e + 2 < ix.dim ? q[e + 2] : 0.f;
        qv.w = e + 3 < ix.dim ? q[e + 3] : 0.f;
        acc = ix.metric == L2SQ ? accumulate<KL2>(acc, qv, v) : accumulate<KDOT>(acc, qv, v);
        q2 = accumulate<KDOT>(q2, qv, qv);
    }
    for (int o = 32; o; o >>= 1) {
        acc += __shfl_xor(acc, o);
        q2 += __shfl_xor(q2, o);
    }
    if (lane == 0) {
        float q_inv = q2 > 0.f ? 1.0f / sqrtf(q2) : 0.f;
        out[w] = finalize(ix.metric, acc, q_inv, ix.metric == COS ? ix.inv_norm[w] : 0.f);
    }
}

hipError_t launch_distance_row(const IndexView& ix, const float* d_query, uint32_t n, float* d_scratch, hipStream_t s,
                               float* host_out) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(distance_row_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ix, d_query, n, d_scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(host_out, d_scratch, (size_t)n * 4, hipMemcpyDeviceToHost, s);
}

// ---------------------------------------------------------------- multi-GPU top-k merge (after the RCCL all-gather)
__global__ __launch_bounds__(64) void topk_merge_kernel(const uint64_t* part_keys, const float* part_dist, uint32_t parts,
                                                        uint32_t nq, uint32_t k, uint64_t* out_keys, float* out_dist,
                                                        uint32_t* out_found) {
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
        size_t src = ((size_t)p * nq + q) * k + j;
        float d = ok ? part_dist[src] : __builtin_inff();
        ok = ok && part_keys[src] != kFreeKey;
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
            size_t src = ((size_t)(c / k) * nq + q) * k + (c % k);
            key = part_keys[src];
            d = sh.lst_d[cur][i];
        }
        out_keys[(size_t)q * k + i] = key;
        out_dist[(size_t)q * k + i] = d;
    }
    if (lane == 0 && out_found) out_found[q] = sz;
}

hipError_t launch_topk_merge(const uint64_t* part_keys, const float* part_dist, uint32_t parts, uint32_t nq, uint32_t k,
                             uint64_t* out_keys, float* out_dist, uint32_t* out_found, hipStream_t s) {
    if (!nq) return hipSuccess;
    if (k == 0 || k > 256 || (size_t)parts * k >= 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(nq), dim3(64), 0, s, part_keys, part_dist, parts, nq, k, out_keys, out_dist,
                       out_found);
    return hipGetLastError();
}

}  // namespace vs
