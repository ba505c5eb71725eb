// kernels_arith.hip -- hnsw_search / hnsw_insert / hnsw_link for ONE arithmetic (-DVS_AR=0..7, see the AR_*
// enum in hnsw_device.hpp).  Compiled once per arithmetic so the eight variants build in parallel.
//
// hnsw_search_kernel : one wavefront per query (K5 of SURVEY.md section 2.3); replaces usearch::Index::search
//                      as called at reference vs_index/usearch.rs:210-212.
// hnsw_insert_kernel / hnsw_link_kernel : batched insertion (K7); replaces usearch::Index::add (usearch.rs:194-196).
//   A sub-batch of new nodes is inserted against the graph frozen at the start of the sub-batch (the host
//   keeps sub-batches small relative to the index, see engine.hip):
//     1. hnsw_insert_kernel : one wave per new node -- greedy descent, per-level beam search (expansion_add),
//        neighbour-selection heuristic, forward links, reverse-link requests;
//     2. radix sort of the requests by (level, target)   (kernels_sort.hip);
//     3. hnsw_link_kernel   : one wave per (level, target) group -- append or re-run the heuristic on the
//        target's list (usearch reconnect_neighbor_nodes_).
//   No locks: every adjacency row is written by exactly one wave per kernel.
#include "kernels.hpp"

#ifndef VS_AR
#error "compile with -DVS_AR=<arithmetic>"
#endif

namespace vs {

// TEAM == 1: one wavefront per query (full batches).  TEAM > 1: one workgroup of TEAM waves per query for small
// batches -- wave 0 walks exactly as before, every wave evaluates its share of each hop's neighbours, so a lone
// query has TEAM times the loads in flight; results are identical (same distances, same order of decisions).
// NT: vector rows loaded non-temporally (tables far larger than the caches; chosen by the host, IndexView::nt_rows).
// WT: wide visited tags, for indexes of more than 2^25 / 2^26 slots (one handle over everything 288 GB hold).
template <int AR, int I, int EFCAP, int NB, int CH, int TEAM = 1, bool NT = false, bool WT = false>
__global__ __launch_bounds__(64 * TEAM) void hnsw_search_kernel(SearchArgs a) {
    using Sh = BeamShared<EFCAP, NB, false, CH, TEAM, NT, WT>;
    __shared__ Sh sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    uint32_t qi = blockIdx.x;
    if (a.qlist) {  // only the queries the pipelined walk handed over
        if (qi >= *a.qcount) return;
        qi = a.qlist[qi];
    }
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
    Query<AR, I> q;
    query_from_f32<AR, I>(ix, a.queries + (size_t)qi * a.q_stride, q, lane);
    if constexpr (TEAM > 1) {
        const uint32_t w = threadIdx.x >> 6;
        if (w != 0) {
            team_helper_loop<AR, I>(ix, q, sh, lane, w);
            return;
        }
    }
    Counters cnt = {0, 0, 0};
    // usearch index_gt::search: search_for_one_ down to level 1, then the base-level beam.
    uint32_t start = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, 0, cnt, lane);
    int cur = 0;
    uint32_t sz;
    {
        sz = beam_search<AR, I>(ix, sh, q, start, 0, a.ef, kInvalid, cnt, lane, cur, a.has_removed != 0);
    }
    team_release(sh, lane);
    wsync<Sh>();
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


// TEAM > 1: small sub-batches (the first geometric steps of a build, streaming adds between searches) get a
// workgroup of TEAM waves per new node, as the search kernel does for small query batches; same decisions, same graph.
template <int AR, int I, int EFCAP, int NB, int CH, int TEAM = 1, bool NT = false, bool WT = false>
__global__ __launch_bounds__(64 * TEAM) void hnsw_insert_kernel(InsertArgs a) {
    using Sh = BeamShared<EFCAP, NB, true, CH, TEAM, NT, WT>;
    __shared__ Sh sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t b = blockIdx.x;
    const uint32_t slot = a.slots[b];
    const int level = a.levels[b];
    Query<AR, I> q;
    query_from_row<AR, I>(ix, slot, q, lane);
    if (threadIdx.x == 0) {  // read by wave 0 only, after its first wsync
        sh.tie_salt = slot * 0x9E3779B1u;
        sh.tie_newest = a.tie_newest;
    }
    if constexpr (TEAM > 1) {
        const uint32_t w = threadIdx.x >> 6;
        if (w != 0) {
            team_helper_loop<AR, I>(ix, q, sh, lane, w);
            return;
        }
    }
    Counters cnt = {0, 0, 0};
    uint32_t closest = ix.entry_slot;
    if (ix.max_level > level)
        closest = greedy_descent<AR, I>(ix, sh, q, ix.entry_slot, ix.max_level, level, cnt, lane);
    const int top = level < ix.max_level ? level : ix.max_level;
    uint32_t req = a.req_off[b] - a.req_base;
    for (int l = top; l >= 0; --l) {
        int cur = 0;
        uint32_t sz = beam_search<AR, I>(ix, sh, q, closest, l, a.ef_add, slot, cnt, lane, cur);
        // usearch connect_new_node_: forward links are refined to `connectivity` on every level
        uint32_t nsel = refine<AR, I>(ix, sh, cur, sz, ix.M, cnt, lane);
        uint32_t cap;
        uint32_t* row = const_cast<uint32_t*>(adjacency(ix, slot, l, cap));
        if ((uint32_t)lane < cap) row[lane] = (uint32_t)lane < nsel ? sh.sel_s[lane] : kInvalid;
        if ((uint32_t)kWave + (uint32_t)lane < cap) row[kWave + lane] = kInvalid;  // (connectivity above 32: level-0 rows of up to 128 ids; nsel <= M <= 64)
        if ((uint32_t)lane < ix.M) {
            bool on = (uint32_t)lane < nsel;
            a.req_key[req + lane] = on ? (((uint64_t)(uint32_t)l << 32) | sh.sel_s[lane]) : ~0ull;
            a.req_val[req + lane] = on ? (((uint64_t)__float_as_uint(sh.sel_d[lane]) << 32) | slot) : 0ull;
        }
        req += ix.M;
        if (nsel) closest = sh.sel_s[0];
        wsync<Sh>();
    }
    team_release(sh, lane);
    if (lane == 0) {
        atomicAdd(&a.stats[ST_ADD_EVALS], cnt.evals);
        atomicAdd(&a.stats[ST_ADD_HOPS], cnt.hops);
        atomicAdd(&a.stats[ST_ADDED], 1ull);
        if (cnt.overflow) atomicAdd(&a.stats[ST_OVERFLOW], cnt.overflow);
    }
}

struct LinkShared {  // sized for a level-0 row of 128 ids (connectivity 64) + 64 new requests
    static constexpr int kTeam = 1;
    float lst_d[1][192];
    uint32_t lst_s[1][192];
    float t_d[192];
    uint32_t t_s[192];
    uint32_t u_slot[128];
    float u_dist[128];
    uint32_t sel_s[128];
    float sel_d[128];
};

constexpr uint32_t kMaxNewPerTarget = 64;

// refine_ for the link kernel with the ACCEPTED rows kept in LDS.  The heuristic measures every candidate against all
// neighbours accepted so far: read from HBM that is (accepted so far) rows per candidate, ~165 row reads for a
// 33-candidate list.  A candidate's row is in registers when it is accepted, so it is written to LDS once and every
// later candidate is measured against LDS copies: one HBM row read per candidate.  Same accumulate / group_sum /
// finalize as eval_batch, so the distances -- and the selected links -- are bit-identical to refine().
// rowbuf: cache_rows x stride4 chunks (row layout as in HBM), then cache_rows aux values.
template <int AR, int I>
__device__ uint32_t refine_cached(const IndexView& ix, LinkShared& sh, uint4* rowbuf, uint32_t cache_rows, uint32_t sz,
                                  uint32_t needed, Counters& cnt, int lane) {
    float* aux_lds = reinterpret_cast<float*>(rowbuf + (size_t)cache_rows * ix.stride4);
    const uint32_t lg = ix.lanes_log2, V = 64u >> lg;
    const uint32_t grp = (uint32_t)lane >> lg, li = (uint32_t)lane & (ix.lanes - 1);
    auto keep = [&](uint32_t pos, const Query<AR, I>& row, uint32_t slot) {  // accepted neighbour `pos` := row
        if (pos < cache_rows && grp == 0) {
#pragma unroll
            for (int i = 0; i < I; ++i) rowbuf[(size_t)pos * ix.stride4 + li + (uint32_t)i * ix.lanes] = row.c[i];
            if (li == 0) aux_lds[pos] = needs_aux<AR>(ix.metric) ? ix.aux[slot] : 0.f;
        }
    };
    if (sz == 0) return 0;
    __syncthreads();
    if (sz < needed) {  // usearch: fewer candidates than slots -> all of them
        for (uint32_t e = (uint32_t)lane; e < sz; e += kWave) {
            sh.sel_s[e] = sh.lst_s[0][e] & kSlotMask;
            sh.sel_d[e] = sh.lst_d[0][e];
        }
        __syncthreads();
        return sz;
    }
    {
        const uint32_t s0 = sh.lst_s[0][0] & kSlotMask;
        if (lane == 0) {
            sh.sel_s[0] = s0;
            sh.sel_d[0] = sh.lst_d[0][0];
        }
        Query<AR, I> first;
        query_from_row<AR, I>(ix, s0, first, lane);
        keep(0, first, s0);
    }
    __syncthreads();
    uint32_t nsel = 1;
    Query<AR, I> nextv;  // the next candidate's row is already on its way while this one is measured
    if (sz > 1) query_from_row<AR, I>(ix, sh.lst_s[0][1] & kSlotMask, nextv, lane);
    for (uint32_t c = 1; c < sz && nsel < needed; ++c) {
        const uint32_t cs = sh.lst_s[0][c] & kSlotMask;
        const float cd = sh.lst_d[0][c];
        Query<AR, I> cv = nextv;
        if (c + 1 < sz) query_from_row<AR, I>(ix, sh.lst_s[0][c + 1] & kSlotMask, nextv, lane);
        const uint32_t in_lds = nsel < cache_rows ? nsel : cache_rows;
        bool bad = false;
        for (uint32_t a0 = 0; a0 < in_lds; a0 += V) {
            const uint32_t a = a0 + grp;
            const bool valid = a < in_lds;
            typename Arith<AR>::acc_t acc = 0;
#pragma unroll
            for (int i = 0; i < I; ++i) {
                const uint4 r = valid ? rowbuf[(size_t)a * ix.stride4 + li + (uint32_t)i * ix.lanes] : make_uint4(0u, 0u, 0u, 0u);
                acc = accumulate<AR>(acc, cv.c[i], r);
            }
            acc = group_sum(acc, ix.lanes);
            if (valid && li == 0) bad = bad || finalize<AR>(ix.metric, acc, cv.aux, aux_lds[a]) < cd;
        }
        bool reject = __ballot(bad) != 0ull;
        if (!reject && nsel > in_lds) {  // accepted beyond the cache: measured from HBM as before
            eval_batch<AR, I>(ix, cv, sh.sel_s + in_lds, sh.u_dist, nsel - in_lds, lane);
            __syncthreads();
            bool bad2 = false;
            for (uint32_t b0 = 0; b0 < nsel - in_lds; b0 += kWave) bad2 = bad2 || (b0 + (uint32_t)lane < nsel - in_lds && sh.u_dist[b0 + lane] < cd);
            reject = __ballot(bad2) != 0ull;
            __syncthreads();
        }
        cnt.evals += nsel;
        if (!reject) {
            if (lane == 0) {
                sh.sel_s[nsel] = cs;
                sh.sel_d[nsel] = cd;
            }
            keep(nsel, cv, cs);
            ++nsel;
            __syncthreads();
        }
    }
    return nsel;
}

template <int AR, int I>
__global__ __launch_bounds__(64) void hnsw_link_kernel(LinkArgs a) {
    __shared__ LinkShared sh;
    const IndexView& ix = a.ix;
    const int lane = lane_id();
    const uint32_t r = blockIdx.x;
    const uint64_t key = a.req_key[r];
    if (key == ~0ull) return;
    if (r > 0 && a.req_key[r - 1] == key) return;  // not the head of its (level, target) group
    const uint32_t target = (uint32_t)key;
    const int level = (int)(key >> 32);
    uint32_t cap;
    uint32_t* row = const_cast<uint32_t*>(adjacency(ix, target, level, cap));
    Counters c = {0, 0, 0};
    // A group of more than 64 requests (hub targets: clustered or duplicate-heavy data) is taken 64 at a time, each round
    // against the row the round before left -- as if the sources had arrived in that order (usearch's reconnect runs once
    // per arriving source); before, requests 65.. of a group were dropped (advisor finding, round 1).
    for (uint32_t base = r;; base += kMaxNewPerTarget) {
        uint32_t n_new;
        {
            uint32_t idx = base + (uint32_t)lane;
            bool same = idx < a.total && a.req_key[idx] == key;
            uint64_t mask = __ballot(same);
            n_new = ~mask ? (uint32_t)__builtin_ctzll(~mask) : kMaxNewPerTarget;
        }
        if (n_new == 0) break;
        uint32_t src = kInvalid;
        float src_d = 0.f;
        if ((uint32_t)lane < n_new) {
            uint64_t v = a.req_val[base + lane];
            src = (uint32_t)v;
            src_d = __uint_as_float((uint32_t)(v >> 32));
        }
        // existing links (up to 128 at connectivity 64: two per lane); a link to a source being (re)inserted is superseded by the new request
        uint32_t ex = (uint32_t)lane < cap ? row[lane] : kInvalid;
        uint32_t ex2 = (uint32_t)kWave + (uint32_t)lane < cap ? row[kWave + lane] : kInvalid;
        for (uint32_t j = 0; j < n_new; ++j) {
            uint32_t sj = (uint32_t)__shfl((int)src, (int)j);
            if (ex == sj) ex = kInvalid;
            if (ex2 == sj) ex2 = kInvalid;
        }
        const uint64_t emask = __ballot(ex != kInvalid), emask2 = __ballot(ex2 != kInvalid);
        const uint32_t cnt1 = (uint32_t)__popcll(emask), cnt = cnt1 + (uint32_t)__popcll(emask2);
        __syncthreads();
        if (ex != kInvalid) sh.u_slot[mbcnt(emask)] = ex;
        if (ex2 != kInvalid) sh.u_slot[cnt1 + mbcnt(emask2)] = ex2;
        __syncthreads();
        if (cnt + n_new <= cap) {  // usearch: close_header.push_back(new_slot)
            if (lane < (int)n_new) sh.u_slot[cnt + lane] = src;  // (cnt + n_new <= cap <= 128)
            __syncthreads();
            for (uint32_t e = (uint32_t)lane; e < cap; e += kWave) row[e] = e < cnt + n_new ? sh.u_slot[e] : kInvalid;
        } else {
            // usearch: top = {new} U existing, all measured from `close_slot`; refine_(connectivity_max)
            Query<AR, I> q;
            query_from_row<AR, I>(ix, target, q, lane);
            eval_batch<AR, I>(ix, q, sh.u_slot, sh.u_dist, cnt, lane);
            __syncthreads();
            c.evals += cnt;
            const uint32_t total = cnt + n_new;  // <= 128 + 64
            for (uint32_t e = (uint32_t)lane; e < cnt; e += kWave) {
                sh.t_d[e] = sh.u_dist[e];
                sh.t_s[e] = sh.u_slot[e];
            }
            if ((uint32_t)lane < n_new) {
                sh.t_d[cnt + lane] = src_d;
                sh.t_s[cnt + lane] = src;
            }
            __syncthreads();
            for (uint32_t e = (uint32_t)lane; e < total; e += kWave) {  // rank sort, ascending
                float ed = sh.t_d[e];
                uint32_t es = sh.t_s[e];
                uint32_t rank = 0;
                if (a.tie_newest) {
                    // usearch builds this list with top.insert(new), then top.insert(existing...) in adjacency order, each
                    // going IN FRONT of equal ones: among equal distances the later members of the row come first, the
                    // new link last
                    const uint32_t te = e < cnt ? cnt - 1u - e : e;
                    for (uint32_t f = 0; f < total; ++f) {
                        const uint32_t tf = f < cnt ? cnt - 1u - f : f;
                        rank += (sh.t_d[f] < ed || (sh.t_d[f] == ed && tf < te)) ? 1u : 0u;
                    }
                } else {
                    // A/B only (VS_HNSW_TIE=random): a pseudo-random order per target among equal distances
                    const uint32_t salt = target * 0x9E3779B1u;
                    for (uint32_t f = 0; f < total; ++f)
                        rank += key_less(sh.t_d[f], (sh.t_s[f] ^ salt) * 0x85EBCA6Bu, ed, (es ^ salt) * 0x85EBCA6Bu) ? 1u : 0u;
                }
                sh.lst_d[0][rank] = ed;
                sh.lst_s[0][rank] = es;
            }
            __syncthreads();
            extern __shared__ uint4 link_rowbuf[];
            uint32_t nsel = a.cache_rows ? refine_cached<AR, I>(ix, sh, link_rowbuf, a.cache_rows, total, cap, c, lane)
                                         : refine<AR, I>(ix, sh, 0, total, cap, c, lane);
            for (uint32_t e = (uint32_t)lane; e < cap; e += kWave) row[e] = e < nsel ? sh.sel_s[e] : kInvalid;
        }
        if (n_new < kMaxNewPerTarget) break;
        __syncthreads();
    }
    if (lane == 0 && c.evals) {
        atomicAdd(&a.stats[ST_ADD_EVALS], c.evals);
        atomicAdd(&a.stats[ST_LINK_EVALS], c.evals);
    }
}


// Rows of 12 / 16 KiB (I = 12 / 16: 3072-d / 4096-d f32) get a reduced set of instances -- one wave per query, rows always
// loaded non-temporally (a table of such rows is far beyond the caches long before it matters), no team forms --, which
// keeps the build time of the eight arithmetics in check; same code, same results.
template <int AR, int I>
static hipError_t search_ef(const SearchArgs& a, hipStream_t s) {
    dim3 grid(a.nq), block(64);
    if constexpr (I >= 12) {
        if (a.wide_tags) {
            if (a.ef <= 256) hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2, 1, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 512, 2048, 2, 1, true, true>), grid, block, 0, s, a);
        } else if (a.ef <= 128) {
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 128, 1024, 1, 1, true>), grid, block, 0, s, a);
        } else if (a.ef <= 256) {
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2, 1, true>), grid, block, 0, s, a);
        } else {
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 512, 2048, 2, 1, true>), grid, block, 0, s, a);
        }
        return hipGetLastError();
    }
    if (a.wide_tags) {  // huge index: always far beyond the caches (non-temporal rows), one wave per query
        if (a.ef <= 128)
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 128, 1024, 1, 1, true, true>), grid, block, 0, s, a);
        else if (a.ef <= 256)
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2, 1, true, true>), grid, block, 0, s, a);
        else
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 512, 2048, 2, 1, true, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    const uint32_t team = a.team & 0xFFu;  // (bit 8: speculation off)
    if (team == kSearchTeamMid && a.ef <= 256 && !a.stress_small_table) {
        dim3 tblock(64 * kSearchTeamMid);
        if (a.ef <= 128)
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 128, 1024, 1, kSearchTeamMid>), grid, tblock, 0, s, a);
        else
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2, kSearchTeamMid>), grid, tblock, 0, s, a);
        return hipGetLastError();
    }
    if (team == kSearchTeam && !a.stress_small_table) {
        dim3 tblock(64 * kSearchTeam);
        if (a.ef <= 128)
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 128, 1024, 1, kSearchTeam>), grid, tblock, 0, s, a);
        else if (a.ef <= 256)
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2, kSearchTeam>), grid, tblock, 0, s, a);
        else  // beams of 257..512 (configs[2] needs 304): lone callers get the team there too
            hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 512, 2048, 2, kSearchTeam>), grid, tblock, 0, s, a);
        return hipGetLastError();
    }
    const bool nt = a.ix.nt_rows != 0;
    if (I == 1 && a.stress_small_table && a.ef <= 128)
        hipLaunchKernelGGL((hnsw_search_kernel<AR, 1, 128, 256, 1>), grid, block, 0, s, a);
    else if (a.ef <= 128 && nt)
        hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 128, 1024, 1, 1, true>), grid, block, 0, s, a);
    else if (a.ef <= 128)
        hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 128, 1024, 1>), grid, block, 0, s, a);
    else if (a.ef <= 256 && nt)
        hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2, 1, true>), grid, block, 0, s, a);
    else if (a.ef <= 256)
        hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 256, 1024, 2>), grid, block, 0, s, a);
    else if (nt)  // wide beams (k up to 512: CQL LIMIT x oversampling): 39 KB LDS, 4 waves per CU
        hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 512, 2048, 2, 1, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((hnsw_search_kernel<AR, I, 512, 2048, 2>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <int AR, int I>
static hipError_t insert_ef(const InsertArgs& a, hipStream_t s) {
    dim3 grid(a.n), block(64);
    if (a.ef_add > 256) {  // construction beams of 257..512: ONE instance (512-entry list, wide two-choice tags, non-temporal rows)
        hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 512, 2048, 2, 1, true, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if constexpr (I >= 12) {
        if (a.wide_tags || a.ef_add > 128) hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 256, 1024, 2, 1, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 128, 1024, 1, 1, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.wide_tags) {
        if (a.ef_add <= 128)
            hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 128, 1024, 1, 1, true, true>), grid, block, 0, s, a);
        else
            hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 256, 1024, 2, 1, true, true>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.team == kSearchTeam) {
        dim3 tblock(64 * kSearchTeam);
        if (a.ef_add <= 128)
            hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 128, 1024, 1, kSearchTeam>), grid, tblock, 0, s, a);
        else
            hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 256, 1024, 2, kSearchTeam>), grid, tblock, 0, s, a);
        return hipGetLastError();
    }
    const bool nt = a.ix.nt_rows != 0;
    if (a.ef_add <= 128 && nt)
        hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 128, 1024, 1, 1, true>), grid, block, 0, s, a);
    else if (a.ef_add <= 128)
        hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 128, 1024, 1>), grid, block, 0, s, a);
    else if (nt)
        hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 256, 1024, 2, 1, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((hnsw_insert_kernel<AR, I, 256, 1024, 2>), grid, block, 0, s, a);
    return hipGetLastError();
}

template <>
hipError_t launch_search_ar<VS_AR>(const SearchArgs& a, uint32_t iters, hipStream_t s) {
    switch (iters) {
        case 1: return search_ef<VS_AR, 1>(a, s);
        case 2: return search_ef<VS_AR, 2>(a, s);
        case 3: return search_ef<VS_AR, 3>(a, s);
        case 4: return search_ef<VS_AR, 4>(a, s);
        case 6: return search_ef<VS_AR, 6>(a, s);
        case 8: return search_ef<VS_AR, 8>(a, s);
        case 12: return search_ef<VS_AR, 12>(a, s);
        case 16: return search_ef<VS_AR, 16>(a, s);
        default: return hipErrorInvalidValue;
    }
}

template <>
hipError_t launch_insert_ar<VS_AR>(const InsertArgs& a, uint32_t iters, hipStream_t s) {
    switch (iters) {
        case 1: return insert_ef<VS_AR, 1>(a, s);
        case 2: return insert_ef<VS_AR, 2>(a, s);
        case 3: return insert_ef<VS_AR, 3>(a, s);
        case 4: return insert_ef<VS_AR, 4>(a, s);
        case 6: return insert_ef<VS_AR, 6>(a, s);
        case 8: return insert_ef<VS_AR, 8>(a, s);
        case 12: return insert_ef<VS_AR, 12>(a, s);
        case 16: return insert_ef<VS_AR, 16>(a, s);
        default: return hipErrorInvalidValue;
    }
}

template <>
hipError_t launch_link_ar<VS_AR>(const LinkArgs& a, uint32_t iters, hipStream_t s) {
    dim3 grid(a.total), block(64);
    const size_t dyn = a.cache_rows ? (size_t)a.cache_rows * ((size_t)a.ix.stride4 * 16 + 4) : 0;
    switch (iters) {
        case 1: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 1>), grid, block, dyn, s, a); break;
        case 2: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 2>), grid, block, dyn, s, a); break;
        case 3: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 3>), grid, block, dyn, s, a); break;
        case 4: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 4>), grid, block, dyn, s, a); break;
        case 6: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 6>), grid, block, dyn, s, a); break;
        case 8: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 8>), grid, block, dyn, s, a); break;
        case 12: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 12>), grid, block, dyn, s, a); break;
        case 16: hipLaunchKernelGGL((hnsw_link_kernel<VS_AR, 16>), grid, block, dyn, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vs
