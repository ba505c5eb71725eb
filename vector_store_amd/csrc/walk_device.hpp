// walk_device.hpp -- the usearch-ORDER walk: search_to_find_in_base_ / search_to_insert_ with usearch's two
// structures kept apart, step for step as oracle/cpu_hnsw.cpp restates them (reference call sites
// crates/vector-store/src/vs_index/usearch.rs:210-212 search, :233-236 filtered_search):
//
//   top   sorted_buffer_gt: ascending by distance, a new entry goes IN FRONT of equal ones (lower_bound), bounded by
//         ef, predicate-gated (only admitted members live here);
//   next  max_heap_gt on negated distances: the array heap, emulated swap for swap (shift_up / shift_down), because
//         which of several EQUAL distances pops first is a property of that exact heap, not of (distance, slot);
//   the radius is re-read after every single admission, in adjacency order, as the CPU loop does.
//
// hnsw_device.hpp's fused list (one sorted list with "expanded" bits) is equivalent to this only while no two
// distances tie; this walk is what serves the tie-heavy metrics (Hamming, i8), filtered search (rejected nodes are
// expanded but never enter `top`, so `next` is unbounded), beams beyond the fused kernel's 512 entries, and any index
// when the caller asks for usearch's order.  Ids are then bit-identical to the oracle's on the same graph.
//
// One wavefront walks; the data-parallel parts (visited test-and-set of the <= 64 neighbours, the distances, the list
// merge) use all lanes, the sequential parts (admission against the moving radius, the heap) are wave-uniform scalar
// code over LDS.
#pragma once
#include "hnsw_device.hpp"

// -DVS_WALK_PROFILE: shader-clock stamps around the phases of a hop, summed per query into WalkArgs::debug[4..11]
// (measurement builds only; scripts/probe/walk_profile.sh).
#ifdef VS_WALK_PROFILE
#define WALK_STAMP(slot)                                         \
    do {                                                         \
        const uint64_t now__ = __builtin_amdgcn_s_memtime();     \
        prof[slot] += now__ - prof_t;                            \
        prof_t = now__;                                          \
    } while (0)
#else
#define WALK_STAMP(slot) \
    do {                 \
    } while (0)
#endif

namespace vs {

// Global workspace of ONE workgroup (one query at a time).
struct WalkSpace {
    uint32_t* bitmap;  // visited bits for the whole slot range (only kernels with a global visited set); all zero between queries
    uint32_t* vlog;    // slots marked during the current query, so that only their words are cleared afterwards
    uint2* heap;       // `next` beyond the part that lives in LDS: (distance bits, slot)
    uint32_t bitmap_words, vlog_cap, heap_cap;
};

constexpr int kWalkOvf = 62;  // overflow list of the walk's visited table (the 128-entry `top` instance then fits 7 per CU)
template <bool ON, int NB, int BS = 8>
struct VisitedLds {
    alignas(16) uint16_t vis_tag[NB * BS];
    uint32_t vis_cnt[NB / 4];
    uint32_t vis_ovf[kWalkOvf];
    uint32_t ovf_cnt;
    uint32_t overflowed;
};
template <int NB, int BS>
struct VisitedLds<false, NB, BS> {};

// EFCAP: capacity of `top`; LCAP: entries of `next` kept in LDS (the first levels of the heap -- every pop walks from the
// root, the deep levels are touched once per pop); VISG: visited set = bitmap in global memory (no limit on the index
// size or on the number of visited nodes) instead of the LDS tag table of hnsw_device.hpp.
// BS: tags per bucket of the visited table (8; 12 = the DENSE table: 512 buckets x 12 hold the 6,144 nodes a beam of up to 256 visits
// in 12.5 KB instead of 17: 7 walks per CU instead of 6).
// HWV (round 6, the walk pods' two-wave form): wave 1 owns `next` -- usearch's array heap, swap for swap -- and repairs it while the walker
// (wave 0) goes on with the hop: pop_heap's sift-down runs under the visited test and the row loads, the hop's push_heaps under the
// merge into `top`.  The two meet at workgroup barriers; a command (hw_cmd, two slots used in turn) goes with each.
// (The hop's pushes travel in u_dist / u_slot, which the walker is done with by then: 20 bytes more than a one-wave walk, so the batch
// instances keep their walks per CU.)
template <bool ON>
struct HeapWaveBox {
    uint2 hw_cmd[2];
    uint32_t hw_hn;
};
template <>
struct HeapWaveBox<false> {};

template <int EFCAP, int LCAP, int NB, int CH, bool VISG, int TM = 1, bool NT = false, bool SEL = false, int BS = 8, bool HWV = false>
struct WalkShared : SelArrays<SEL>, TeamBox<TM>, VisitedLds<!VISG, NB, BS>, HeapWaveBox<HWV> {
    static_assert(!HWV || (TM == 2 && !VISG && EFCAP <= 512), "the heap wave is wave 1 of a two-wave team over an LDS heap; `top` is merged in registers");
    static constexpr bool kHeapWave = HWV;
    static constexpr bool kNT = NT;
    static constexpr int kChoices = CH;
    static constexpr int kEfCap = EFCAP;
    static constexpr int kNB = NB;
    static constexpr int kBucket = BS;
    static constexpr int kTeam = TM;
    static constexpr bool kSel = SEL;
    static constexpr int kHeapLds = LCAP;
    static constexpr bool kVisGlobal = VISG;
    static constexpr bool kWideTags = false;  // indexes beyond the plain tags' reach take the global-bitmap instance
    static constexpr bool kHeapSpill = VISG;
    static constexpr uint32_t kOvfCap = (uint32_t)kWalkOvf;  // `next` may outgrow LDS into WalkSpace::heap (LDS instances hand the query over instead)
    float lst_d[EFCAP];
    uint32_t lst_s[EFCAP];
    uint2 hp[LCAP + 1];  // + 1: the right child of the last parent is read unconditionally
    uint32_t u_slot[64];
    float u_dist[64];
};

// ---- `next`: usearch max_heap_gt, entries (distance, slot), less(a, b) = a.d > b.d ----------------------------------
// Every lane runs the same scalar code on wave-uniform values; LDS reads broadcast, lane 0 writes.  LDS (and the
// vector-memory path) keep one wave's accesses in order, so no barrier is needed between the steps.
// Heap indices, sizes and entries are wave-uniform; entries read from LDS are passed through v_readfirstlane so that
// the compiler keeps the whole sift loop on the scalar unit (s_cmp / s_cbranch, no exec-mask bookkeeping): the loops
// are latency chains of LDS reads, and executed as vector code each level cost ~4x as many cycles.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class Sh>
__device__ __forceinline__ uint2 heap_get(const Sh& sh, const WalkSpace& ws, uint32_t i) {
    uint2 e;
    if constexpr (Sh::kHeapSpill) {
        i = uni(i);  // a scalar branch between two loads, not one flat load that waits for every outstanding row load
        if (i < (uint32_t)Sh::kHeapLds) e = sh.hp[i];
        else e = ws.heap[i - (uint32_t)Sh::kHeapLds];
    } else {
        e = sh.hp[i];
    }
    return make_uint2(uni(e.x), uni(e.y));
}
template <class Sh>
__device__ __forceinline__ void heap_set(Sh& sh, const WalkSpace& ws, uint32_t i, uint2 e, int lane) {
    if constexpr (Sh::kHeapSpill) {
        i = uni(i);
        if (i < (uint32_t)Sh::kHeapLds) {
            if (lane == 0) sh.hp[i] = e;
        } else if (lane == 0) {
            ws.heap[i - (uint32_t)Sh::kHeapLds] = e;
        }
    } else {
        if (lane == 0) sh.hp[i] = e;  // one lane: 64 lanes storing to one address would be a 32-way bank conflict
    }
}

// per-lane access to entry i of a heap whose deep levels live in global memory (lanes of one wave may hit both)
template <class Sh>
__device__ __forceinline__ uint2 heap_lane_load(const Sh& sh, const WalkSpace& ws, uint32_t i, bool valid) {
    uint2 e = make_uint2(0u, 0u);
    if (valid) {
        if (i < (uint32_t)Sh::kHeapLds) e = sh.hp[i];
        else e = ws.heap[i - (uint32_t)Sh::kHeapLds];
    }
    return e;
}
template <class Sh>
__device__ __forceinline__ void heap_lane_store(Sh& sh, const WalkSpace& ws, uint32_t i, uint2 e) {
    if (i < (uint32_t)Sh::kHeapLds) sh.hp[i] = e;
    else ws.heap[i - (uint32_t)Sh::kHeapLds] = e;
}

// emplace + shift_up: the new entry climbs while its parent is strictly farther.
// LDS-only heaps (LDS instances): the ancestors of the insertion point are known up front -- ((pos + 1) >> k) - 1 --, so
// lane k loads ancestor k, one ballot says how far the entry climbs, and the entries on that stretch move down one
// level each in ONE parallel store: one LDS round trip per push instead of one per level.
template <class Sh>
__device__ __forceinline__ void heap_push(Sh& sh, const WalkSpace& ws, uint32_t& hn, float d, uint32_t slot, int lane) {
    if constexpr (!Sh::kHeapSpill) {
        const uint32_t pos = uni(hn);
        hn = pos + 1u;
        const uint32_t depth = 32u - (uint32_t)__builtin_clz(pos + 1u) - 1u;  // ancestors 1..depth (ancestor `depth` is the root)
        const uint32_t k = (uint32_t)lane;
        const bool on = k >= 1u && k <= depth;
        const uint32_t anc = ((pos + 1u) >> (on ? k : 0u)) - 1u;
        const uint2 e = on ? sh.hp[anc] : make_uint2(0u, 0u);
        // climbs past ancestor k iff every ancestor 1..k is strictly farther: leading ones of the ballot from bit 1
        const uint64_t far = __ballot(on && __uint_as_float(e.x) > d) >> 1;
        const uint32_t climb = (uint32_t)__builtin_ctzll(~far);  // <= depth
        if (on && k <= climb) sh.hp[((pos + 1u) >> (k - 1u)) - 1u] = e;  // ancestor k moves down to where ancestor k - 1 was
        if (lane == 0) sh.hp[((pos + 1u) >> climb) - 1u] = make_uint2(__float_as_uint(d), slot);
        return;
    } else {
        // the same with the deep levels in global memory: every lane reads ITS ancestor from wherever it lives, so a push is
        // one LDS and (for a deep heap) one global round trip, whatever the depth (a filtered walk's `next` holds thousands of
        // entries: one dependent global read per level made the heap the larger part of a hop)
        const uint32_t pos = uni(hn);
        hn = pos + 1u;
        const uint32_t depth = 32u - (uint32_t)__builtin_clz(pos + 1u) - 1u;
        const uint32_t k = (uint32_t)lane;
        const bool on = k >= 1u && k <= depth;
        const uint32_t anc = ((pos + 1u) >> (on ? k : 0u)) - 1u;
        const uint2 e = heap_lane_load(sh, ws, anc, on);
        const uint64_t far = __ballot(on && __uint_as_float(e.x) > d) >> 1;
        const uint32_t climb = (uint32_t)__builtin_ctzll(~far);  // <= depth
        if (on && k <= climb) heap_lane_store(sh, ws, ((pos + 1u) >> (k - 1u)) - 1u, e);
        if (lane == 0) heap_lane_store(sh, ws, ((pos + 1u) >> climb) - 1u, make_uint2(__float_as_uint(d), slot));
    }
}

// shift_down of a heap whose deep levels may live in global memory: where the entry with distance `ld` lands when it sinks from the
// root of a heap of n entries; the entries it passes move up on the way (see heap_pop).
template <class Sh>
__device__ __forceinline__ uint32_t heap_sink_windows(Sh& sh, const WalkSpace& ws, uint32_t n, float ld, int lane) {
    const uint32_t L = (uint32_t)lane;
    const uint32_t j = 31u - (uint32_t)__builtin_clz(L + 1u);  // depth of lane L below the window's root (lane 0: the root)
    const uint32_t t = L + 1u - (1u << j);
    uint32_t i = 0;  // the node the last entry is sinking from (absolute index, wave-uniform)
    for (;;) {
        if (2u * i + 1u >= n) break;
        const uint64_t idx64 = (((uint64_t)i + 1ull) << j) - 1ull + (uint64_t)t;
        const bool valid = L <= 62u && idx64 < (uint64_t)n;
        const uint2 e = heap_lane_load(sh, ws, (uint32_t)idx64, valid && L >= 1u);
        const uint64_t vmask = __ballot(valid);
        // preference of every internal lane (0..30): the right child only when it exists and the left one is strictly farther
        const uint32_t cl = 2u * L + 1u;
        const float dl = __uint_as_float((uint32_t)__shfl((int)e.x, (int)(cl & 63u)));
        const float dr = __uint_as_float((uint32_t)__shfl((int)e.x, (int)((cl + 1u) & 63u)));
        const bool has_l = L <= 30u && ((vmask >> cl) & 1ull) != 0ull, has_r = L <= 30u && ((vmask >> (cl + 1u)) & 1ull) != 0ull;
        const uint32_t pref = (has_r && dl > dr) ? cl + 1u : cl;
        const uint64_t hc = __ballot(has_l);
        // the preferred path through the window: path[0] = lane 0 (node i), up to five more
        uint32_t path[6];
        path[0] = 0;
        uint32_t len = 0;
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            const uint32_t p = path[s5];
            const bool go = len == (uint32_t)s5 && ((hc >> p) & 1ull) != 0ull;
            path[s5 + 1] = go ? (uint32_t)__builtin_amdgcn_readlane((int)pref, (int)p) : 0u;
            len += go ? 1u : 0u;
        }
        // lane k (1..len) takes the k-th node of the path, and the node above it
        uint32_t mine = 0, above = 0;
#pragma unroll
        for (int k5 = 1; k5 <= 5; ++k5) {
            mine = L == (uint32_t)k5 ? path[k5] : mine;
            above = L == (uint32_t)k5 ? path[k5 - 1] : above;
        }
        const bool on = L >= 1u && L <= len;
        const uint32_t px = (uint32_t)__shfl((int)e.x, (int)mine), py = (uint32_t)__shfl((int)e.y, (int)mine);
        const uint64_t closer = __ballot(on && ld > __uint_as_float(px)) >> 1;  // bit k - 1: the last entry sinks past path node k
        const uint32_t sink = (uint32_t)__builtin_ctzll(~closer);                // <= len
        // absolute index of a window lane
        auto abs_of = [&](uint32_t wl) -> uint32_t {
            const uint32_t dj = 31u - (uint32_t)__builtin_clz(wl + 1u);
            return (uint32_t)((((uint64_t)i + 1ull) << dj) - 1ull + (uint64_t)(wl + 1u - (1u << dj)));
        };
        if (on && L <= sink) heap_lane_store(sh, ws, abs_of(above), make_uint2(px, py));  // path node k moves up to node k - 1
        uint32_t land = 0;  // the entry lands on path node `sink` (the window's root when it does not sink at all)
#pragma unroll
        for (int k5 = 1; k5 <= 5; ++k5) land = sink == (uint32_t)k5 ? path[k5] : land;
        i = uni(abs_of(land));
        if (sink < 5u) break;  // stopped inside the window: at a closer node, or at a leaf (sink <= len <= 5); else on with the next five levels
    }
    return i;
}

// pop: swap(first, last), shrink, shift_down(0): the larger child is the right one only when the left one is strictly
// farther ("less(left, right)"), and the entry sinks only while it is strictly farther than that child.
// LDS-only heaps: which child a node prefers does not depend on the sinking entry, so (1) every internal node's
// preference is computed in parallel (lane i handles nodes i, i + 64, ...: ballots -> bit masks in scalar registers),
// (2) the path of preferred children from the root to a leaf is followed on those bits with scalar code, (3) lane k loads
// the k-th node of the path, one ballot says where the last entry stops, one parallel store moves the stretch up.
// Three LDS round trips per pop instead of two per level.
template <class Sh>
__device__ __forceinline__ void heap_pop(Sh& sh, const WalkSpace& ws, uint32_t& hn, int lane) {
    const uint32_t n = uni(hn) - 1u;
    hn = n;
    if (n == 0) return;
    if constexpr (!Sh::kHeapSpill) {
        constexpr int RMAX = (Sh::kHeapLds + 127) / 128;  // 64 internal nodes per ballot
        const uint2 last = sh.hp[n];
        const float ld = __uint_as_float(uni(last.x));
        uint64_t pref[RMAX];  // bit i of pref[r]: node 64 r + i prefers its RIGHT child
        const uint32_t internal = n >> 1;  // nodes with a left child below n: 2 i + 1 < n  <=>  i < n / 2 (n >= 1)
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            pref[r] = 0;
            if ((uint32_t)r * 64u < internal) {
                const uint32_t i = (uint32_t)r * 64u + (uint32_t)lane, l = 2u * i + 1u;
                const bool two = l + 1u < n;  // both children exist
                const float dl = two ? __uint_as_float(sh.hp[l].x) : 0.f, dr = two ? __uint_as_float(sh.hp[l + 1u].x) : 0.f;
                pref[r] = __ballot(two && dl > dr);
            }
        }
        // the path root -> leaf; lane k keeps node k of it (and node k - 1, where its entry may move to)
        uint32_t p = 0, len = 0, mine = 0, above = 0;
        for (;;) {
            const uint32_t l = 2u * p + 1u;
            if (l >= n) break;
            uint64_t w = pref[0];
#pragma unroll
            for (int r = 1; r < RMAX; ++r) w = (p >> 6) == (uint32_t)r ? pref[r] : w;
            const uint32_t c = l + (uint32_t)((w >> (p & 63u)) & 1ull);
            ++len;
            if ((uint32_t)lane == len) {
                mine = c;
                above = p;
            }
            p = c;
        }
        const bool on = (uint32_t)lane >= 1u && (uint32_t)lane <= len;
        const uint2 e = on ? sh.hp[mine] : make_uint2(0u, 0u);
        const uint64_t closer = __ballot(on && ld > __uint_as_float(e.x)) >> 1;  // bit k - 1: the last entry sinks past path node k
        const uint32_t sink = (uint32_t)__builtin_ctzll(~closer);                // <= len
        if (on && (uint32_t)lane <= sink) sh.hp[above] = e;                       // path node k moves up to node k - 1
        // the last entry lands on path node `sink` (the root when it does not sink at all)
        const uint32_t land = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)sink);
        if (lane == 0) sh.hp[sink ? land : 0u] = make_uint2(last.x, last.y);
        return;
    } else {
        // Heaps with their deep levels in global memory: the descendants of node i down to five levels are the 62 entries
        // (i + 1) 2^j - 1 + t (j = 1..5, t < 2^j): lane L = 2^j - 1 + t loads its entry from LDS or global memory, whichever
        // holds it -- one round trip for five levels.  In lane space the children of L are 2 L + 1 and 2 L + 2 again, and, as
        // in the LDS-only form, which child a node prefers does not depend on the sinking entry: every lane computes its
        // preference, five scalar steps follow the preferred path through the window, one ballot says how far the last entry
        // sinks along it, one parallel store moves that stretch up.  Same comparisons, same order as the level-by-level loop.
        const uint2 last = heap_get(sh, ws, n);
        const float ld = __uint_as_float(last.x);
        const uint32_t i = heap_sink_windows(sh, ws, n, ld, lane);
        heap_set(sh, ws, i, last, lane);
    }
}

// ---- visited ---------------------------------------------------------------------------------------------------------
// Per-lane test-and-set.  LDS table: hnsw_device.hpp; when that table is exhausted the walk cannot stay exact
// (a node evaluated twice would be pushed twice), so `exhausted` is raised and the kernel hands the query to the
// global-bitmap instance.
// group_sum for short rows (groups of 2 / 4 / 8 lanes: bit rows, low dimensions): quad permutes and row shifts are DPP
// operands, a few cycles each, where group_sum's generic loop takes one ds_bpermute round trip per halving -- 1.5k of a lone b1
// walk's 14.7k clocks per hop (four row groups, three halvings each, one after the other).
// (the halvings in group_sum's own order -- lane ^ 4, ^ 2, ^ 1 -- so that a float sum has the bits every other kernel gives it)
__device__ __forceinline__ int walk_dpp_xor4(int v) {  // lane i of each group of eight reads lane i ^ 4: a shift by four either way, chosen per bank of four lanes
    int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0 and 2
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);   // row_shr:4 into banks 1 and 3
}
__device__ __forceinline__ int walk_group_sum(int v, uint32_t lanes) {
    if (lanes >= 16) return group_sum(v, lanes);
    if (lanes >= 8) v += walk_dpp_xor4(v);
    if (lanes >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);  // quad_perm:[2,3,0,1]
    if (lanes >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
    return v;
}
__device__ __forceinline__ float walk_group_sum(float v, uint32_t lanes) {
    if (lanes >= 16) return group_sum(v, lanes);
    if (lanes >= 8) v += __int_as_float(walk_dpp_xor4(__float_as_int(v)));
    if (lanes >= 4) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    if (lanes >= 2) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    return v;
}
// group_reduce (hnsw_device.hpp) over walk_group_sum
template <int AR, int I, int U, int TEAM>
__device__ __forceinline__ void walk_group_reduce(const IndexView& ix, const RowGroup<I, U>& g, const Query<AR, I>& q, float* u_dist, uint32_t L,
                                                  uint32_t w, uint32_t vshift, uint32_t grp, uint32_t li) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        typename Arith<AR>::acc_t acc = 0;
#pragma unroll
        for (int i = 0; i < I; ++i) acc = accumulate<AR>(acc, q.c[i], g.buf[u][i]);
        acc = walk_group_sum(acc, ix.lanes);
        if (g.slot[u] != kInvalid && li == 0) {
            const float d = finalize<AR>(ix.metric, acc, q.aux, g.aux[u]);
            u_dist[(((L + (uint32_t)u) * (uint32_t)TEAM + w) << vshift) + grp] = d == d ? d : __builtin_inff();  // (NaN ranks as +inf, as there)
        }
    }
}

// One wave's share of a batch: eval_batch (hnsw_device.hpp) over walk_group_reduce.  TEAM / w: the layout of the shares, as there.
template <int AR, int I, int TEAM, class Sh>
__device__ __forceinline__ void walk_eval_part(const IndexView& ix, const Query<AR, I>& q, Sh& sh, uint32_t m, int lane, uint32_t w) {
    constexpr bool NT = Sh::kNT;
    constexpr int U = I >= 12 ? 1 : (I >= 6 || (AR == AR_I8 && I >= 3)) ? 2 : 4;  // (eval_batch's)
    const uint32_t lg = ix.lanes_log2;
    const uint32_t vshift = 6u - lg;
    const uint32_t grp = (uint32_t)lane >> lg, li = (uint32_t)lane & (ix.lanes - 1);
    const uint32_t nl_all = (m + (1u << vshift) - 1u) >> vshift;
    const uint32_t nl = TEAM == 1 ? nl_all : (nl_all > w ? (nl_all - w + (uint32_t)TEAM - 1u) / (uint32_t)TEAM : 0u);
    if (nl == 0) return;
    RowGroup<I, U> a, b;
    group_issue<AR, I, U, TEAM, NT>(ix, a, sh.u_slot, m, 0, w, vshift, grp, li);
    for (uint32_t L = 0;;) {
        if (L + U < nl) group_issue<AR, I, U, TEAM, NT>(ix, b, sh.u_slot, m, L + U, w, vshift, grp, li);
        walk_group_reduce<AR, I, U, TEAM>(ix, a, q, sh.u_dist, L, w, vshift, grp, li);
        L += U;
        if (L >= nl) break;
        if (L + U < nl) group_issue<AR, I, U, TEAM, NT>(ix, a, sh.u_slot, m, L + U, w, vshift, grp, li);
        walk_group_reduce<AR, I, U, TEAM>(ix, b, q, sh.u_dist, L, w, vshift, grp, li);
        L += U;
        if (L >= nl) break;
    }
}

// ---- the heap wave's protocol (WalkShared<.., HWV = true>) ----
enum : uint32_t { HW_NOP = 0, HW_EVAL = 1, HW_POP = 2, HW_PUSH = 3, HW_EXIT = 4 };
// LDS is all the two waves share: the barrier waits for the wave's LDS operations only -- a __syncthreads() would also wait for the walker's
// adjacency prefetches, which are meant to stay in flight across it.
__device__ __forceinline__ void hw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// walker: command number `gen` goes into slot gen & 1 -- the heap wave reads slot g right behind barrier g and the walker writes it again only
// behind barrier g + 1, which the heap wave reaches after that read
template <class Sh>
__device__ __forceinline__ void hw_send(Sh& sh, uint32_t& gen, uint32_t op, uint32_t arg, int lane) {
    if (lane == 0) sh.hw_cmd[gen & 1u] = make_uint2(op, arg);
    hw_barrier();
    ++gen;
}
// wave 1 of the team, from the moment the descent's helper loop ends (team_release) until the walk says HW_EXIT
template <int AR, int I, class Sh>
__device__ __forceinline__ void walk_heap_wave_loop(const IndexView& ix, const Query<AR, I>& q, Sh& sh, int lane) {
    const WalkSpace ws = {nullptr, nullptr, nullptr, 0u, 0u, 0u};
    uint32_t gen = 0;
    for (;;) {
        hw_barrier();
        const uint2 c = sh.hw_cmd[gen & 1u];
        ++gen;
        const uint32_t op = uni(c.x), arg = uni(c.y);
        if (op == HW_EXIT) return;
        if (op == HW_POP) {
            uint32_t hn = arg;
            heap_pop(sh, ws, hn, lane);
        } else if (op == HW_PUSH) {
            uint32_t hn = uni(sh.hw_hn);
            for (uint32_t j = 0; j < arg; ++j) heap_push(sh, ws, hn, __uint_as_float(uni(__float_as_uint(sh.u_dist[j]))), uni(sh.u_slot[j]), lane);
        } else if (op == HW_EVAL) {  // (not sent at present: see eval_walker_alone)
            walk_eval_part<AR, I, 2>(ix, q, sh, arg, lane, 1u);
        }
    }
}

// eval_shared (hnsw_device.hpp) for the walker of a heap-wave team: it measures alone -- a hop's rows are one or two wave-loads, and sharing
// them would make the walker wait for the heap wave to finish the pop (2.5k clocks against the visited test's 1.3k) before either could
// start: measured, 17 callers 14.3k queries/s with the rows shared, 16.3k with the walker alone.
template <int AR, int I, class Sh>
__device__ __forceinline__ void eval_walker_alone(const IndexView& ix, const Query<AR, I>& q, Sh& sh, uint32_t m, int lane) {
    walk_eval_part<AR, I, 1>(ix, q, sh, m, lane, 0u);
    wsync<Sh>();
}

// the smallest value of a wave, in every lane: four DPP rotations inside the rows of 16, four scalar reads across them
__device__ __forceinline__ float walk_wave_min(float v) {
    v = fminf(v, VS_DPP_ROR(v, 8));
    v = fminf(v, VS_DPP_ROR(v, 4));
    v = fminf(v, VS_DPP_ROR(v, 2));
    v = fminf(v, VS_DPP_ROR(v, 1));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}

template <class Sh>
__device__ __forceinline__ bool walk_visit(Sh& sh, const WalkSpace& ws, uint32_t slot) {
    if constexpr (Sh::kVisGlobal) {
        const uint32_t bit = 1u << (slot & 31u);
        return (atomicOr(&ws.bitmap[slot >> 5], bit) & bit) != 0u;
    } else {
        return visited_test_and_set(sh, slot);
    }
}

// The walk.  On return `top` is sh.lst_d / sh.lst_s [0, size) -- ascending, exactly the oracle's order -- and the
// size is returned.  self: slot that is never a result (search_to_insert_ of an update), or kInvalid.
// tomb: some members carry the free key (never results); allow: optional bitmap over slots (filtered_search).
template <int AR, int I, class Sh>
__device__ uint32_t walk_usearch(const IndexView& ix, Sh& sh, const WalkSpace& ws, const Query<AR, I>& q, uint32_t start, int level,
                                 uint32_t ef, uint32_t self, bool tomb, const uint32_t* allow, Counters& cnt, int lane,
                                 bool& exhausted, uint32_t* debug = nullptr, const uint32_t* known = nullptr,
                                 uint32_t* unknown_list = nullptr, uint32_t* unknown_count = nullptr, uint32_t unknown_cap = 0,
                                 uint32_t unknown_budget = 0, uint32_t* consulted_out = nullptr) {
    uint32_t dbg_max_hn = 0, dbg_pushed = 0;
#ifdef VS_WALK_PROFILE
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t prof_t = __builtin_amdgcn_s_memtime();
#endif
    start = uni(start);   // wave-uniform by construction; said so, the loop control below stays on the scalar unit
    uint32_t vcount = 0;  // entries of ws.vlog (global visited set only)
    bool vlog_lost = false;
    exhausted = false;
    if constexpr (!Sh::kVisGlobal) {
        visited_clear(sh, lane);
        wsync<Sh>();
    }
    // May slot s be a result?  Wave-level (every lane calls it, `valid` says whether the lane holds a slot): with a lazy
    // predicate the slots whose verdict is not known yet are listed for the host and count as rejected for this launch.
    bool over_budget = false;
    uint32_t consulted = 0;
    // unknown_budget: low 24 bits = the number of unknown slots one round may list; high 8 bits = t: in an exploratory round a slot
    // without a verdict counts as admitted with probability t / 256 (a hash of the slot) instead of as rejected.  With every unknown
    // rejected `top` fills late and the radius stays wide (the middle round of a 10 % filter ran 4.3k hops against the exact walk's
    // 2.05k); guessing at HALF the filter's observed selectivity keeps the radius wider than the exact walk's -- the round still
    // explores a superset of what the exact walk will consult -- at a fraction of the detour.  A round that met no unknown slot made no
    // guess and is exact, as before.
    const uint32_t guess_t = unknown_budget >> 24;
    unknown_budget &= 0xFFFFFFu;
    auto allowed = [&](uint32_t s, bool valid) -> bool {
        bool ok = valid;
        if (tomb) ok = ok && ix.keys[valid ? s : 0u] != kFreeKey;
        if (allow) {
            consulted += (uint32_t)__popcll(__ballot(ok));
            if (known) {
                const bool kn = ok && ((known[s >> 5] >> (s & 31u)) & 1u) != 0u;
                const bool unk = ok && !kn;  // (removed members need no verdict)
                const uint64_t um = __ballot(unk);
                if (um) {
                    const uint32_t c = (uint32_t)__popcll(um);
                    uint32_t base = 0;
                    if (lane == (int)__builtin_ctzll(um)) base = atomicAdd(unknown_count, c);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(um));
                    if (unk && base + mbcnt(um) < unknown_cap) unknown_list[base + mbcnt(um)] = s;
                    if (base + c >= unknown_budget) over_budget = true;
                }
                const bool guess = unk && guess_t != 0u && ((s * 2654435761u) >> 24) < guess_t;
                ok = kn;
                ok = ok && ((allow[(ok ? s : 0u) >> 5] >> (s & 31u)) & 1u) != 0u;
                return ok || guess;
            }
            ok = ok && ((allow[(ok ? s : 0u) >> 5] >> (s & 31u)) & 1u) != 0u;
        }
        return ok;
    };
    auto mark = [&](uint32_t n) -> bool {  // true: n is new to the visited set (and, for the bitmap, logged)
        const bool fresh = n != kInvalid && !walk_visit(sh, ws, n);
        if constexpr (Sh::kVisGlobal) {
            const uint64_t fm = __ballot(fresh);
            const uint32_t c = (uint32_t)__popcll(fm);
            if (vcount + c <= ws.vlog_cap) {
                if (fresh) ws.vlog[vcount + mbcnt(fm)] = n;
            } else {
                vlog_lost = true;  // too many to log: the whole bitmap is cleared at the end
            }
            vcount += c;
        }
        return fresh;
    };
    // visits.set(start); visits.set(self)
    (void)mark(lane == 0 ? start : (lane == 1 && self != start) ? self : kInvalid);
    if (lane == 0) sh.u_slot[0] = start;
    wsync<Sh>();
    // the heap wave's command counter, and whether it may still be busy with a pop or the pushes (kHeapWave only)
    uint32_t hw_gen = 0;
    bool hw_busy = false;
    auto hw_sync = [&]() {
        if constexpr (Sh::kHeapWave) {
            if (hw_busy) hw_send(sh, hw_gen, HW_NOP, 0u, lane);
            hw_busy = false;
        }
    };
    if constexpr (Sh::kHeapWave) eval_walker_alone<AR, I>(ix, q, sh, 1, lane);
    else eval_shared<AR, I>(ix, q, sh, 1, lane);
    cnt.evals += 1;
    const float d0 = sh.u_dist[0];
    uint32_t hn = 0, sz = 0;
    heap_push(sh, ws, hn, d0, start, lane);  // (the heap wave waits at its barrier: the first entry is the walker's)
    if (start != self && (__ballot(allowed(start, lane == 0)) & 1ull)) {
        if (lane == 0) {
            sh.lst_d[0] = d0;
            sh.lst_s[0] = start;
        }
        sz = 1;
    }
    wsync<Sh>();
    // Adjacency prefetch: once the candidate is popped, the new root of `next` is the runner-up; its row is loaded while
    // this hop's vectors stream in and is used when that node is indeed expanded next (nothing closer was pushed).
    // Heap-wave walks (round 6) ask for a second row: more often than not the next candidate is the closest neighbour the hop has just
    // measured (the walk is greedy).  Both rows are asked for when the distances are out.
    // (Measured on the one-wave walks and NOT kept there: the second row, and the repair of `next` moved under the hop's row loads,
    // removed no clocks from a lone b1 walk -- a hop is a chain of LDS round trips, not a wait for memory,
    // profiles/r06_b1_walk_phases.txt -- and the i8 batch walk, seven walks to a CU, fell from 837k to 342k queries/s with them.)
    uint32_t pf_slot = kInvalid, pf_n = kInvalid, pg_slot = kInvalid, pg_n = kInvalid;
    WALK_STAMP(0);  // start-up: clear, first evaluation
    while (hn) {
        sz = uni(sz);  // wave-uniform by construction (see `start`)
        hw_sync();     // the last hop's pushes are in
        const uint2 ce = heap_get(sh, ws, 0);
        const float cd = __uint_as_float(ce.x);
        const uint32_t cs = ce.y;
        if (sz == ef && cd > __uint_as_float(uni(__float_as_uint(sh.lst_d[sz - 1])))) break;  // `candidate.distance > radius && top.size() == top_limit`
        uint32_t n;
        if (cs == pf_slot) {
            n = pf_n;
        } else if (Sh::kHeapWave && cs == pg_slot) {
            n = pg_n;
        } else {  // on its way while the heap is repaired
            uint32_t cap;
            const uint32_t* row = adjacency(ix, cs, level, cap);
            n = (uint32_t)lane < cap ? row[lane] : kInvalid;
        }
        cnt.hops += 1;
        bool asked = false;
        auto ask_runner_up = [&]() {
            if (asked) return;
            asked = true;
            hw_sync();  // (the pop has ended)
            pf_slot = hn ? heap_get(sh, ws, 0).y : kInvalid;
            if (pf_slot != kInvalid) {
                uint32_t cap2;
                const uint32_t* row2 = adjacency(ix, pf_slot, level, cap2);
                pf_n = (uint32_t)lane < cap2 ? row2[lane] : kInvalid;
            }
        };
        if constexpr (Sh::kHeapWave) {  // the heap wave starts on pop_heap's sift-down now; the walker goes on
            hw_send(sh, hw_gen, HW_POP, hn, lane);
            hn -= 1u;
            hw_busy = true;
        } else {
            heap_pop(sh, ws, hn, lane);
            ask_runner_up();
        }
        if (cs == self) {
            ask_runner_up();
            continue;
        }
        WALK_STAMP(1);  // candidate, pop, prefetch issue (heap-wave walks: the candidate and the command)
        // connectivity above 32: a level-0 row holds up to 128 ids, taken 64 at a time in adjacency order (as the CPU loop would)
        uint32_t n_hi = kInvalid;
        {
            uint32_t capc;
            const uint32_t* rowc = adjacency(ix, cs, level, capc);
            if (capc > (uint32_t)kWave) n_hi = (uint32_t)kWave + (uint32_t)lane < capc ? rowc[kWave + lane] : kInvalid;
        }
        const uint32_t halves = (level == 0 ? ix.M0 : ix.M) > (uint32_t)kWave ? 2u : 1u;
        for (uint32_t half = 0; half < halves && !exhausted; ++half) {
        if (half) {
            n = n_hi;
            hw_sync();  // (the first half's pushes are read from u_slot / u_dist)
        }
        const bool fresh = mark(n);
        const uint64_t fmask = __ballot(fresh);
        const uint32_t m = (uint32_t)__popcll(fmask);
        if constexpr (!Sh::kVisGlobal) {
            if (uni(sh.overflowed)) {  // wave-uniform (LDS flag set by any lane of this hop)
                exhausted = true;
                break;
            }
        }
        if (fresh) sh.u_slot[mbcnt(fmask)] = n;
        wsync<Sh>();
        if (m == 0) continue;
        WALK_STAMP(2);  // visited test-and-set, compaction
        if constexpr (Sh::kHeapWave) eval_walker_alone<AR, I>(ix, q, sh, m, lane);
        else eval_shared<AR, I>(ix, q, sh, m, lane);
        cnt.evals += m;
        WALK_STAMP(3);  // distances
        float nd = (uint32_t)lane < m ? sh.u_dist[lane] : __builtin_inff();
        uint32_t ns = (uint32_t)lane < m ? sh.u_slot[lane] : kInvalid;
        if constexpr (Sh::kHeapWave) {  // the runner-up's row, and the row of the closest of this hop's neighbours when it beats the runner-up
            ask_runner_up();
            const float mn = walk_wave_min(nd);
            const uint64_t bm = __ballot((uint32_t)lane < m && nd == mn);
            const float root_d = hn ? __uint_as_float(heap_get(sh, ws, 0).x) : __builtin_inff();
            if (bm && mn <= root_d) {
                const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)ns, (int)__builtin_ctzll(bm));
                if (s2 != pf_slot) {
                    uint32_t cap3;
                    const uint32_t* row3 = adjacency(ix, s2, level, cap3);
                    pg_n = (uint32_t)lane < cap3 ? row3[lane] : kInvalid;
                    pg_slot = s2;
                }
            }
        }
        // the verdict is only needed for neighbours that can still be admitted (usearch asks the predicate inside
        // `if (top.size() < top_limit || d < radius)`): once `top` is full, those below the radius of the hop's start
        const float radius0 = sz == ef ? __uint_as_float(uni(__float_as_uint(sh.lst_d[sz - 1]))) : __builtin_inff();
        const uint64_t okmask = __ballot(allowed(ns, (uint32_t)lane < m && (sz < ef || nd < radius0)));
        if (over_budget) {  // enough unknown slots listed for one round: the host evaluates them and launches again
            exhausted = true;
            break;
        }
        // ---- admission, one neighbour at a time in adjacency order (wave-uniform scalar code) ----
        // T: old entries of `top` that survive; alive: new entries that are in `top` at the end; pushed: new entries of `next`
        uint64_t pushed = 0, alive = 0;
        uint32_t T = sz, cur = sz;
        const uint64_t all_m = m >= 64u ? ~0ull : ((1ull << m) - 1ull);
        if (sz + m <= ef) {  // `top` cannot fill up during this hop: everything is admitted
            pushed = all_m;
            alive = okmask;
            cur = sz + (uint32_t)__popcll(alive);
        } else {
            // once full the radius only shrinks: what is not below it now never will be
            uint64_t cand = sz == ef ? __ballot((uint32_t)lane < m && nd < sh.lst_d[sz - 1]) : all_m;
            float tail_d = T ? __uint_as_float(uni(__float_as_uint(sh.lst_d[T - 1]))) : -__builtin_inff();
            float max_d = -__builtin_inff();  // worst of the alive new entries: largest distance, the OLDEST among equals
            uint32_t max_j = 0, a_cnt = 0;
            for (; cand; cand &= cand - 1ull) {
                const uint32_t j = (uint32_t)__builtin_ctzll(cand);
                const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)j));
                // the worst entry of `top` right now: old entries are older than new ones, so they lose ties
                const bool worst_is_old = T && (a_cnt == 0 || tail_d >= max_d);
                if (cur == ef && !(dj < (worst_is_old ? tail_d : max_d))) continue;  // `top.size() < top_limit || d < radius`
                pushed |= 1ull << j;
                if (!((okmask >> j) & 1ull)) continue;
                if (cur == ef) {  // top.insert at the limit drops the last (worst) entry
                    if (worst_is_old) {
                        --T;
                        tail_d = T ? __uint_as_float(uni(__float_as_uint(sh.lst_d[T - 1]))) : -__builtin_inff();
                    } else {
                        alive &= ~(1ull << max_j);
                        --a_cnt;
                        max_d = -__builtin_inff();
                        for (uint64_t r = alive; r; r &= r - 1ull) {
                            const uint32_t i = (uint32_t)__builtin_ctzll(r);
                            const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)i));
                            if (di > max_d) {
                                max_d = di;
                                max_j = i;
                            }
                        }
                    }
                } else {
                    ++cur;
                }
                alive |= 1ull << j;
                ++a_cnt;
                if (dj > max_d) {
                    max_d = dj;
                    max_j = j;
                }
            }
        }
        WALK_STAMP(4);  // admission
        // ---- next.insert for every admitted neighbour, in order ----
        if constexpr (Sh::kHeapWave) {  // the list goes to the heap wave, which pushes it in this order while `top` is merged here
            const uint32_t np = (uint32_t)__popcll(pushed);
            if (np) {
                if (hn + np > (uint32_t)Sh::kHeapLds) {
                    exhausted = true;
                    break;
                }
                if ((pushed >> lane) & 1ull) {  // (every lane holds its own in nd / ns: the list is compacted in place)
                    sh.u_dist[mbcnt(pushed)] = nd;
                    sh.u_slot[mbcnt(pushed)] = ns;
                }
                if (lane == 0) sh.hw_hn = hn;
                hw_send(sh, hw_gen, HW_PUSH, np, lane);
                hn += np;
                hw_busy = true;
            }
        } else {
        for (uint64_t r = pushed; r; r &= r - 1ull) {
            const uint32_t j = (uint32_t)__builtin_ctzll(r);
            if (hn >= (uint32_t)Sh::kHeapLds + (Sh::kHeapSpill ? ws.heap_cap : 0u)) {  // `next` outgrew its workspace
                exhausted = true;
                break;
            }
            heap_push(sh, ws, hn, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)j)),
                      (uint32_t)__builtin_amdgcn_readlane((int)ns, (int)j), lane);
        }
        }
        if (exhausted) break;
        WALK_STAMP(5);  // pushes
        dbg_max_hn = hn > dbg_max_hn ? hn : dbg_max_hn;
        dbg_pushed += (uint32_t)__popcll(pushed);
        // ---- top: merge the alive new entries into the surviving T old ones, in place ----
        const uint32_t a = (uint32_t)__popcll(alive);
        if (a == 0) continue;
        if constexpr (Sh::kEfCap <= 512) {
            // small `top`: every lane holds its R = EFCAP / 64 old entries in registers; ranks by ballot + popcount, no
            // dependent LDS chain (a binary search costs log2(ef) LDS latencies), one scatter at the end
            constexpr int R = Sh::kEfCap / kWave;
            float keep_d[R];
            uint32_t keep_s[R], shift[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
                keep_d[r] = p < T ? sh.lst_d[p] : __builtin_inff();
                keep_s[r] = p < T ? sh.lst_s[p] : 0u;
                shift[r] = 0;
            }
            const bool mine = ((alive >> lane) & 1ull) != 0ull;
            uint32_t lo = 0, rn = 0;
            for (uint64_t rem = alive; rem; rem &= rem - 1ull) {
                const uint32_t i = (uint32_t)__builtin_ctzll(rem);
                const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)i));
                uint32_t below = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool valid = (uint32_t)lane + (uint32_t)r * kWave < T;
                    below += (uint32_t)__popcll(__ballot(valid && keep_d[r] < di));  // a new entry precedes equal old ones
                    shift[r] += (valid && di <= keep_d[r]) ? 1u : 0u;
                }
                if ((uint32_t)lane == i) lo = below;
                rn += (mine && (di < nd || (di == nd && i > (uint32_t)lane))) ? 1u : 0u;  // closer first, the NEWER first among equals
            }
            __builtin_amdgcn_wave_barrier();  // every read of the old list is in registers
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t p = (uint32_t)lane + (uint32_t)r * kWave;
                if (p < T && shift[r]) {
                    sh.lst_d[p + shift[r]] = keep_d[r];
                    sh.lst_s[p + shift[r]] = keep_s[r];
                }
            }
            if (mine) {
                sh.lst_d[lo + rn] = nd;
                sh.lst_s[lo + rn] = ns;
            }
            sz = T + a;
            wsync<Sh>();
            WALK_STAMP(6);  // merge into top
            continue;
        }
        wsync<Sh>();
        if ((alive >> lane) & 1ull) {
            const uint32_t r = mbcnt(alive);
            sh.u_dist[r] = nd;
            sh.u_slot[r] = ns;
        }
        wsync<Sh>();
        nd = (uint32_t)lane < a ? sh.u_dist[lane] : __builtin_inff();
        ns = (uint32_t)lane < a ? sh.u_slot[lane] : kInvalid;
        // rank among the old entries: the first one that is not strictly closer (a new entry precedes equal old ones)
        uint32_t lo = 0, hi = (uint32_t)lane < a ? T : 0u;
        while (__ballot(lo < hi)) {
            if (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sh.lst_d[mid] < nd) lo = mid + 1u; else hi = mid;
            }
        }
        // rank among the new ones: closer first, the NEWER first among equals
        uint32_t rn = 0, pmin = T;
        for (uint32_t i = 0; i < a; ++i) {
            const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)i));
            const uint32_t li = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)i);
            rn += (di < nd || (di == nd && i > (uint32_t)lane)) ? 1u : 0u;
            pmin = li < pmin ? li : pmin;
        }
        // old entries from the first insertion point on move up by the number of new entries that precede them;
        // highest chunk first, so nothing is overwritten before it has been read
        if (T > pmin) {
            for (int c = (int)((T - 1u) >> 6); c >= (int)(pmin >> 6); --c) {
                const uint32_t p = (uint32_t)c * 64u + (uint32_t)lane;
                const bool act = p >= pmin && p < T;
                const float od = act ? sh.lst_d[p] : 0.f;
                const uint32_t os = act ? sh.lst_s[p] : 0u;
                uint32_t shift = 0;
                for (uint32_t i = 0; i < a; ++i)
                    shift += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), (int)i)) <= od ? 1u : 0u;
                if (act && shift) {
                    sh.lst_d[p + shift] = od;
                    sh.lst_s[p + shift] = os;
                }
            }
        }
        if ((uint32_t)lane < a) {
            sh.lst_d[lo + rn] = nd;
            sh.lst_s[lo + rn] = ns;
        }
        sz = T + a;
        wsync<Sh>();
        }  // half
        if (exhausted) break;
        ask_runner_up();  // (heap-wave walks: a hop whose halves all turned back early)
    }
    if constexpr (Sh::kHeapWave) hw_send(sh, hw_gen, HW_EXIT, 0u, lane);  // (whatever the heap wave was told last, it gets here behind it)
    if (consulted_out && lane == 0) *consulted_out = consulted;
    if (debug && lane == 0) {
        debug[0] = dbg_max_hn;
        debug[1] = (uint32_t)cnt.evals;
        debug[2] = (uint32_t)cnt.hops;
        debug[3] = dbg_pushed;
#ifdef VS_WALK_PROFILE
        for (int i = 0; i < 8; ++i) debug[4 + i] = (uint32_t)(prof[i] >> 4);
#endif
    }
    if constexpr (Sh::kVisGlobal) {  // leave the bitmap all zero for the next query of this workgroup
        if (vlog_lost) {
            for (uint32_t w = (uint32_t)lane; w < ws.bitmap_words; w += kWave) ws.bitmap[w] = 0u;
        } else {
            for (uint32_t i = (uint32_t)lane; i < vcount; i += kWave) ws.bitmap[ws.vlog[i] >> 5] = 0u;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached L2 before the next query's atomics
    }
    return sz;
}

}  // namespace vs
