// pipe_device.hpp -- the PIPELINED walk: usearch's search_to_find_in_base_ (reference call sites
// crates/vector-store/src/vs_index/usearch.rs:210-212 search, :233-236 filtered_search) for a LONE query, served by one
// workgroup of waves that no longer meet at barriers.
//
// Why: a lone walk is a chain of dependent hops.  The team kernels (hnsw_device.hpp / walk_device.hpp) spread a hop's
// distances over eight waves, but every hop still pays, one after the other: heap pop, adjacency read, visited
// test-and-set, the HBM round trip of the rows, verdict words, admission, heap pushes, list merge -- 14k shader clocks
// (profiles/r03_walk_phase_clocks.txt), of which the row round trip is a quarter and the rest a chain of LDS / global
// round trips on wave 0.  A filtered walk at 10 % selectivity is 3,300 such hops.
//
// Here wave 0 (the WALKER) only takes decisions, on structures that live in its registers, and waves 1.. (the HELPERS)
// evaluate candidates AHEAD of it:
//
//   next   = [front: the 64 closest unexpanded candidates, one per lane of wave 0, sorted] + [pool: the rest, unsorted, in
//            LDS].  pop = a lane shift, push = a ballot rank + a lane shift (or an append to the pool); the front is refilled
//            from the pool by a radix select when it runs empty (rare: every push below the front's worst goes to it).
//   top    = usearch's sorted buffer, <= 512 entries in registers (consecutive positions per lane).  Round 4: an insertion was
//            R ballots + a lane shift, admission the CPU loop as written -- one neighbour at a time, in adjacency order,
//            against the moving radius.  Round 5 (TopOps): who passes is decided for the whole row in closed form, and the
//            admitted ones are merged with one scatter; in plain walks a wave of its own (pipe_top_loop) keeps the buffer and
//            merges while the walker pushes.  The one-by-one loop remains for rows with equal distances or inside a tie window.
//   helpers: the walker publishes jobs "evaluate candidate X into cache entry e" for the first kPipeAhead entries of the
//            front that have no entry yet.  A helper loads X's adjacency row, drops the neighbours the visited bitmap
//            already holds, fetches their verdict bits, measures the rest (eval_batch: the same code as every other
//            kernel, so the same distance bits) and stores (neighbour, distance, flags) in adjacency order.  Speculation
//            only MEASURES: the visited set is marked by the walker alone, when X is really popped, and the bitmap only
//            grows during a query, so a neighbour a helper skipped is visited then too and what it measured is a superset
//            of what the pop needs.  A candidate that falls back in the queue keeps its entry; one that is never popped
//            cost bandwidth, which a lone walk has to spare.
//
// Per hop the walker is left with: shift the front, find the entry (a ballot over tags in registers), one returning
// atomicOr on the visited bitmap, the admission loop, the pushes -- about a fifth of the old chain.
//
// Order among EQUAL distances is the one thing these structures do not reproduce (usearch's array heap and lower_bound
// insertion decide it, walk_device.hpp emulates them swap for swap).  Every insertion therefore checks for an equal
// distance among the entries it is ranked against; the first tie ends the walk with status kPipeRedo and the query is
// answered by the usearch-order walk instead.  Float metrics on real data never tie; lattice data and duplicates go to the old
// kernels, and so does b1 (a few hundred distinct distances); i8 takes the exact instance (filtered queries since round 4, lone plain
// ones since round 5: its tie windows hand over only where two orders could differ).  Tie-free, the decisions -- and with the
// shared distance code the ids and distance bits -- equal walk_usearch's.
#pragma once
#include "walk_device.hpp"

namespace vs {

#ifndef VS_PIPE_TEAM
#define VS_PIPE_TEAM 12
#endif
constexpr int kPipeTeam = VS_PIPE_TEAM;  // waves per query: the walker + the helpers (165 registers: three waves per SIMD, twelve per CU)
constexpr int kPipeCache = 16;  // evaluated-candidate entries (LDS)
#ifndef VS_PIPE_AHEAD
#define VS_PIPE_AHEAD 8
#endif
constexpr int kPipeAhead = VS_PIPE_AHEAD;  // front positions the helpers keep evaluated
#ifndef VS_PIPE_PARTS
#define VS_PIPE_PARTS 3
#endif
constexpr uint32_t kPipeParts = VS_PIPE_PARTS;  // helpers that share one candidate measured ahead
#ifndef VS_PIPE_URGENT
#define VS_PIPE_URGENT 4
#endif
constexpr uint32_t kPipeUrgentParts = VS_PIPE_URGENT;  // ... and the one the walker needs at once
#ifndef VS_PIPE_URGENT_FLAGS
#define VS_PIPE_URGENT_FLAGS 2
#endif
constexpr uint32_t kPipeUrgentFlags = VS_PIPE_URGENT_FLAGS;  // 2: an urgent job measures every neighbour without waiting for the visited word
constexpr uint32_t kPipeRedo = 0xFFFFFFFEu;  // out_found: not answered here (a tie, or a structure outgrown): the usearch-order walk must answer

// flags of a cache entry's neighbour
constexpr uint32_t kPfEvaluated = 1u;  // c_dist holds its distance
constexpr uint32_t kPfLive = 2u;       // not a removed member (asks for a verdict when a filter is on)
constexpr uint32_t kPfKnown = 4u;      // the filter's verdict is known ...
constexpr uint32_t kPfAllowed = 8u;    // ... and admits it (no filter: every live member)
constexpr uint32_t kPfSeen = 16u;      // measured although the visited set held it already (urgent jobs measure every neighbour)

// VISG: the visited set is a bitmap over all slots in global memory (filtered walks visit tens of thousands of nodes; any index size);
// else the exact LDS tag table of hnsw_device.hpp (unfiltered lone walks: a beam of 200 visits ~5,000 nodes) -- the walker's
// test-and-set is then an LDS round trip instead of a returning global atomic, and nothing has to be wiped afterwards.
template <int EFCAP, int TM, bool NT, bool VISG>
struct PipeShared : TeamBox<TM>, VisitedLds<!VISG, (EFCAP <= 256 ? 1024 : 2048), 8> {
    static constexpr bool kNT = NT;
    static constexpr int kEfCap = EFCAP;
    static constexpr int kTeam = TM;
    static constexpr bool kSel = false;
    static constexpr bool kVisGlobal = VISG;
    static constexpr int kNB = EFCAP <= 256 ? 1024 : 2048;
    static constexpr int kChoices = 2;
    static constexpr int kBucket = 8;
    static constexpr bool kWideTags = false;
    static constexpr uint32_t kOvfCap = (uint32_t)kWalkOvf;
    // the team phase (greedy descent through the upper levels: eval_shared / team_helper_loop)
    uint32_t u_slot[64];
    float u_dist[64];
    // jobs: walker -> helper w
    uint32_t job_state[TM];  // 0 idle, 1 posted
    uint32_t job_slot[TM];
    uint32_t job_entry[TM];
    uint32_t job_part[TM];   // part | parts << 8: the helper takes the neighbours whose adjacency position % parts == part
    uint32_t stop;
    // evaluated candidates
    uint32_t c_ready[kPipeCache];  // parts still out (0: complete, or not in use)
    uint32_t c_slot[kPipeCache][64];
    float c_dist[kPipeCache][64];
    uint32_t c_flag[kPipeCache][64];
    // helper scratch: the compacted list eval_batch works on
    uint32_t h_slot[TM][64];
    float h_dist[TM][64];
    // refill
    uint32_t hist[256];
    uint2 stage[64];
    alignas(16) uint2 merge[kEfCap + 64];  // hop_batch: one hop's admissions merged into `top` (a scatter by destination)
    // the top wave's mailbox (plain walks, pipe_top_loop): a hop's admitted neighbours from the walker; `top`'s size and radius back
    float tw_nd[64];
    uint32_t tw_n[64];
    uint32_t tw_cand[2], tw_ok[2], tw_pass[2];
    uint32_t tw_flags;  // 1: inside a tie window; 4: a row with an infinity or a NaN (the literal loop: who passed comes back in tw_pass)
    float tw_tie_v;
    uint32_t tw_sz, tw_redo;
    float tw_radius;
    uint32_t tw_req, tw_done;  // sequence numbers: posted by the walker / merged into `top`
    // asks (kPipeAsk, round 6): the walker's questions to the host's predicate and the answers, carried by the courier wave
    uint32_t aq_slot[256];  // ring: the slot asked about, by ask number & 255
    uint32_t av[256];       // ring: its answer (0 none yet, 1 rejected, 2 admitted) -- zeroed by the walker when it asks
    uint32_t aq_tail;       // asks posted by the walker so far
    uint32_t courier_done;  // the courier has left, its last store to the caller's memory is out (the query's flag must come after it)
    uint32_t prof_jobs[8];  // profile builds: job parts done, their clocks; [2..7] (VS_WALK_PROFILE == 3) urgent parts: count, clocks until the row is here / the list is written / the distances are out / the part is reported, neighbours measured
};

__device__ __forceinline__ uint32_t lds_load_acquire(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_load_relaxed(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The WALKER's flag traffic.  LDS executes one wave's instructions in order, so a flag stored after its payload is seen after it, and
// payload read after a flag is read after it -- no s_waitcnt needed; what must not happen is the compiler moving them across each
// other (wavefront-scope fences: ordering only).  A workgroup-scope release / acquire here would also wait for every global store the
// walker has in flight (visited log, unknown list) and for the visited atomics it issued on purpose ahead of the bookkeeping.
__device__ __forceinline__ void lds_flag_store(uint32_t* p, uint32_t v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_flag_load(const uint32_t* p) {
    const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return v;
}

// lane l <- lane l - 1 (lane 0 keeps `fill`); lane l <- lane l + 1 (lane 63 keeps `fill`)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_shl1(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ float rl_f(float v, uint32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)l)); }
__device__ __forceinline__ uint32_t rl_u(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// lane l of v <- x (x, l wave-uniform, both from scalar instructions where this is used: no VALU-written-SGPR hazard to pad).  A VALU
// instruction of gfx9 reads ONE scalar register, so the lane goes through m0.  m0 is a reserved register: the compiler keeps no value
// in it, but it does write it right before instructions of its own that read it (readlane / writelane lane selects, LDS-DMA, movrel),
// so the asm DECLARES the clobber -- the scheduler must not slip it between such a write and its use (round-5 advisor); clang's warning
// about a reserved register on the clobber list is what is silenced here.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ uint32_t wl_u(uint32_t v, uint32_t x, uint32_t l) {
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(l) : "m0");
    return v;
}
#pragma clang diagnostic pop
// order-preserving key of a distance (NaN never gets here: group_reduce ranks it as +inf)
__device__ __forceinline__ uint32_t dist_key(uint32_t bits) { return bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
// minimum over the wave, in every lane: four DPP rotations inside the rows of 16, then the four row results on the scalar side
// (the walker calls this two or three times per hop: written out, v_min_f32 with a DPP source is the move and the minimum in ONE
// instruction; through fminf() every step was a v_mov_dpp, two canonicalising v_max and the v_min.  No NaN gets here -- see dist_key.
// The s_nop pads are the VALU-write -> DPP-read wait states the compiler cannot see inside an asm.)
__device__ __forceinline__ float wave_min(float v) {
    asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0"
        : "+v"(v));
    const float a = rl_f(v, 0), b = rl_f(v, 16), c = rl_f(v, 32), d = rl_f(v, 48);
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "s"(a), "v"(b), "v"(c));
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(d), "v"(r));
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(r)));
}

// ---- helper waves --------------------------------------------------------------------------------------------------
template <int AR, int I, class Sh>
__device__ __forceinline__ void pipe_helper_loop(const IndexView& ix, const Query<AR, I>& q, Sh& sh, const WalkSpace& ws, bool tomb,
                                                 const uint32_t* allow, const uint32_t* known, int lane, uint32_t w) {
    for (;;) {
        while (lds_load_acquire(&sh.job_state[w]) == 0u) {
            if (lds_load_relaxed(&sh.stop)) return;
            __builtin_amdgcn_s_sleep(1);
        }
#ifdef VS_WALK_PROFILE
        const uint64_t job_t0 = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t slot = uni(sh.job_slot[w]), e = uni(sh.job_entry[w]), pp = uni(sh.job_part[w]);
        const uint32_t part = pp & 255u, parts = (pp >> 8) & 255u;
        const bool claim = ((pp >> 16) & 1u) != 0u;  // exploring rounds: the helper itself marks what it measures (no order to keep)
        const bool nofilter = ((pp >> 17) & 1u) != 0u;  // the candidate is needed at once: every neighbour is measured, visited or not
                                                      // (the visited word's round trip would come before the first row load)
        const uint32_t cap = ix.M0;  // <= 64 (host-checked)
        const uint32_t* row = ix.adj0 + (size_t)slot * ix.M0;
        const bool mine = (uint32_t)lane % parts == part;  // the positions of the adjacency row this helper answers for
        const uint32_t n = (mine && (uint32_t)lane < cap) ? row[lane] : kInvalid;
        const bool valid = n != kInvalid;
#if defined(VS_WALK_PROFILE) && VS_WALK_PROFILE == 3
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t job_t1 = __builtin_amdgcn_s_memtime();
#endif
        // Everything that depends on the neighbour ids alone leaves together: the visited word (read past L1: the walker's atomics live
        // in L2), the verdict words, the key (removed members), and one dword of the neighbour's own adjacency row -- should it become
        // the closest candidate at once, that row is two dependent loads away from ITS neighbours' distances; the touch pulls it into
        // L2 meanwhile (the value is not used).  Only the visited word is waited for before the rows are asked for.
        uint32_t vw = 0, kw = ~0u, aw = ~0u, touch = 0;
        uint64_t key = 0;
        if (valid) {
            if (claim) vw = atomicOr(&ws.bitmap[n >> 5], 1u << (n & 31u));
            else vw = __hip_atomic_load(&ws.bitmap[n >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            touch = ix.adj0[(size_t)n * ix.M0];
            if (tomb) key = ix.keys[n];
            if (allow) {
                aw = allow[n >> 5];
                if (known) kw = known[n >> 5];
            }
        }
        // The bitmap only grows during a query, so "seen" stays true; "not seen" is re-tested by the walker when the candidate is popped.
        // (an urgent job does not wait for the visited word: it is looked at after the rows, as a hint for the walker)
        // (a real branch: as one expression this is a select, and a select waits for the visited word's round trip whichever side is
        // taken -- the urgent job's rows would leave a memory latency late)
        bool need = valid;
        if (!nofilter) {
            asm volatile("" ::: "memory");
            need = valid && ((vw >> (n & 31u)) & 1u) == 0u;
        }
        const uint64_t nm = __builtin_amdgcn_ballot_w64(need);
        const uint32_t m = (uint32_t)__popcll(nm);
        if (need) sh.h_slot[w][mbcnt(nm)] = n;
        // (the list is this wave's own: LDS keeps one wave's accesses in order, only the compiler must be held -- a workgroup-scope fence
        // would also wait for the touch loads, a full HBM round trip, before the first row load is issued)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#if defined(VS_WALK_PROFILE) && VS_WALK_PROFILE == 3
        const uint64_t job_t2 = __builtin_amdgcn_s_memtime();
#endif
        eval_batch<AR, I, 1, Sh::kNT>(ix, q, sh.h_slot[w], sh.h_dist[w], m, lane);
#if defined(VS_WALK_PROFILE) && VS_WALK_PROFILE == 3
        const uint64_t job_t3 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t fl = 0;
        if (need) {
            const bool live = !tomb || key != kFreeKey;
            const bool kn = ((kw >> (n & 31u)) & 1u) != 0u, al = ((aw >> (n & 31u)) & 1u) != 0u;
            fl = kPfEvaluated | (live ? kPfLive : 0u);
            if (live) fl |= (kn ? kPfKnown : 0u) | ((kn && al) ? kPfAllowed : 0u);
            if (((vw >> (n & 31u)) & 1u) != 0u) fl |= kPfSeen;
        }
        if (mine) {
            sh.c_slot[e][lane] = n;
            sh.c_flag[e][lane] = fl;
        }
        if (need) sh.c_dist[e][lane] = sh.h_dist[w][mbcnt(nm)];
#ifdef VS_WALK_PROFILE
        if (lane == 0) {
            atomicAdd(&sh.prof_jobs[0], 1u);
            atomicAdd(&sh.prof_jobs[1], (uint32_t)(__builtin_amdgcn_s_memtime() - job_t0));
#if VS_WALK_PROFILE == 3
            if (nofilter) {
                atomicAdd(&sh.prof_jobs[2], 1u);
                atomicAdd(&sh.prof_jobs[3], (uint32_t)(job_t1 - job_t0));
                atomicAdd(&sh.prof_jobs[4], (uint32_t)(job_t2 - job_t0));
                atomicAdd(&sh.prof_jobs[5], (uint32_t)(job_t3 - job_t0));
                atomicAdd(&sh.prof_jobs[6], (uint32_t)(__builtin_amdgcn_s_memtime() - job_t0));
                atomicAdd(&sh.prof_jobs[7], m);
            }
#endif
        }
#endif
        if (lane == 0) {
            __hip_atomic_fetch_sub(&sh.c_ready[e], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            lds_store_release(&sh.job_state[w], 0u);
        }
        asm volatile("" ::"v"(touch));
    }
}

// ---- the walker ----------------------------------------------------------------------------------------------------
// EFCAP / 64 consecutive positions of `top` per lane.
template <int R>
struct PipeTop {
    float d[R];
    uint32_t s[R];
};

constexpr uint32_t kPipeTopWave = 1u;  // plain walks: the wave that keeps `top` (it measures nothing)

// `top` -- usearch's sorted result buffer, EFCAP / 64 consecutive positions per lane -- and what a hop does to it.  The walker's own
// object in filtered walks; the top wave's in plain ones (pipe_top_loop).
template <int R, bool kFilter, class Sh>
struct TopOps {
    PipeTop<R>& top;
    Sh& sh;
    const uint32_t L, ef;
    const bool fused_order;
    uint32_t sz = 0;
    float radius = __builtin_inff();  // top's last distance once it is full
    bool redo = false, tie_active = false;
    float tie_v = 0.f;

    __device__ __forceinline__ TopOps(PipeTop<R>& t, Sh& s, uint32_t lane, uint32_t limit, bool fused) : top(t), sh(s), L(lane), ef(limit), fused_order(fused) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            top.d[j] = __builtin_inff();
            top.s[j] = kInvalid;
        }
    }
    // distance at position p of `top` (wave-uniform p): one readlane per register row, chosen on the scalar side (a select chain over
    // the rows themselves is turned into an indexed load from a scratch copy of the array)
    __device__ __forceinline__ float at(uint32_t p) const {
        const uint32_t pl = p / (uint32_t)R, pr = p % (uint32_t)R;
        float v = rl_f(top.d[0], pl);
#pragma unroll
        for (int j = 1; j < R; ++j) {
            const float x = rl_f(top.d[j], pl);
            v = pr == (uint32_t)j ? x : v;
        }
        return v;
    }
    // top.insert({d, s}, ef) with d below the radius when full: in front of equal entries
    // lazy (kPipeAsk): the entry arrives LATER than usearch would have inserted it (its verdict was on its way) -- the set `top` holds
    // does not depend on the order of arrival, the order among equal distances does: such an insertion that meets an equal distance
    // hands the walk over
    __device__ __forceinline__ void insert(float d, uint32_t s, bool lazy = false) {
        uint32_t rank = 0;
        bool eq = false;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool in = L * (uint32_t)R + (uint32_t)j < sz;
            rank += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && top.d[j] < d));  // usearch: a new entry goes in front of equal ones
            eq = eq || (in && top.d[j] == d);
        }
        const bool eq_any = __builtin_amdgcn_ballot_w64(eq) != 0ull;
        if (eq_any && fused_order) {
            // (the caller's other kernels keep ONE list ordered by (distance, slot): among equal distances the lower slot first, as they
            // do -- a second pass for the rare insertion that meets an equal distance, not a second comparison in every one)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const bool in = L * (uint32_t)R + (uint32_t)j < sz;
                rank += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && top.d[j] == d && top.s[j] < s));
            }
        }
        if (tie_active && (eq_any || ((!kFilter || fused_order) && sz + 1u >= ef))) redo = true;
        if (lazy && eq_any) redo = true;
        const float cd = __uint_as_float(wave_shr1(__float_as_uint(top.d[R - 1]), 0u));
        const uint32_t cs = wave_shr1(top.s[R - 1], 0u);
#pragma unroll
        for (int j = R - 1; j >= 0; --j) {
            const uint32_t pos = L * (uint32_t)R + (uint32_t)j;
            const float pd = j ? top.d[j > 0 ? j - 1 : 0] : cd;
            const uint32_t ps = j ? top.s[j > 0 ? j - 1 : 0] : cs;
            top.d[j] = pos > rank ? pd : pos == rank ? d : top.d[j];
            top.s[j] = pos > rank ? ps : pos == rank ? s : top.s[j];
        }
        sz = sz < ef ? sz + 1u : ef;
        if (sz == ef) radius = at(ef - 1u);
        if constexpr (kFilter) {
            if (tie_active && !fused_order && sz == ef && radius < tie_v) redo = true;  // (the window's other candidates would end the walk, not be expanded)
        }
    }
    // One hop's admissions AT ONCE (round 5).  The CPU loop takes a hop's fresh neighbours one at a time against a radius that moves with
    // every admission; done literally that is a rank by ballots over every register row, a lane shift of the whole buffer and an insertion
    // into the front PER NEIGHBOUR -- half of a lone walk's clocks (scripts/probe/pipe_phase_probe.sh).  The same decisions in closed form:
    //   * neighbour j passes `top.size() < limit || d < radius` exactly when fewer than `limit` members of
    //     top  U  {admissible neighbours before j in the row}  are not farther than it (the radius IS the limit-th smallest of that
    //     multiset; neighbours turned away earlier lie at or beyond it and cannot change it): one count over `top`, one over the row --
    //     accept(), exact whatever distances are equal (finite ones: the caller sends a row with an infinity or a NaN through the
    //     literal loop);
    //   * `top` afterwards is the `limit` closest of top U {passed and admissible}: every old member moves up by the number of new ones
    //     closer than it, every new one lands at (old members not farther) + (new ones in front of it) -- ONE scatter by destination
    //     through LDS (sh.merge), whatever lands at `limit` or beyond is gone: merge().  The closed form is exact for the SET; the ORDER
    //     among equal distances is what insert() defines, so the merged list is checked for equal neighbours (up to the first entry that
    //     left) BEFORE anything is committed: false = nothing done, the caller inserts them one by one.
    // lanes of `cand`: their own nd.  Most neighbours need no count at all: with B admissible neighbours in the row, one that is closer than
    // the member at position limit - 1 - B has at most limit - 1 - B members not farther than it, so it passes whatever the B others do
    // (`sure`: one readlane chain and one comparison for the whole row).  In a filtered walk -- the radius is the limit-th best ADMITTED
    // member, far beyond most neighbours -- that is nearly every lane; in a plain one about two in three.  The others: the exact count.
    __device__ __forceinline__ uint64_t accept(uint64_t cand, uint64_t okmask, float nd) const {
        // (written for the instruction count -- the walker issues one instruction per four clocks: masks straight from the comparisons,
        // the per-lane result by v_writelane, the verdict once after the loop; the first version -- a select per result, the verdict
        // inside the loop -- was 42 instructions per neighbour, this one is 27, for the lanes that need it)
        const uint64_t okm = cand & okmask;
        const uint32_t admissible = (uint32_t)__popcll(okm);
        uint64_t sure = 0ull;
        if (admissible < ef) sure = __builtin_amdgcn_ballot_w64(nd < at(ef - 1u - admissible)) & cand;  // (unused positions hold +inf)
        uint32_t my_tot = 0;  // lane j: members of top U {admissible neighbours before j} not farther than neighbour j
        for (uint64_t r = cand & ~sure; r;) {
            const uint32_t j = (uint32_t)__builtin_ctzll(r);
            r &= ~(1ull << j);
            const float dj = rl_f(nd, j);
            uint32_t gt = 0;  // (the rows' unused positions hold +inf: counted here, taken off below)
#pragma unroll
            for (int i = 0; i < R; ++i) gt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(top.d[i] > dj));
            const uint32_t before = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(nd <= dj) & okm & ((1ull << j) - 1ull));
            my_tot = wl_u(my_tot, 64u * (uint32_t)R - gt + before, j);
        }
        return sure | (__builtin_amdgcn_ballot_w64(my_tot < ef) & cand & ~sure);
    }
    // lane j of `tm` (neighbours about to enter `top`): members of `top` not farther than it -- where merge() puts it among them
    __device__ __forceinline__ uint32_t ranks(uint64_t tm, float nd) const {
        uint32_t my_le = 0;
        for (uint64_t r = tm; r;) {
            const uint32_t j = (uint32_t)__builtin_ctzll(r);
            r &= ~(1ull << j);
            const float dj = rl_f(nd, j);
            uint32_t gt = 0;
#pragma unroll
            for (int i = 0; i < R; ++i) gt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(top.d[i] > dj));
            my_le = wl_u(my_le, 64u * (uint32_t)R - gt, j);
        }
        return my_le;
    }
    __device__ __forceinline__ bool merge(uint64_t tm, float nd, uint32_t n, uint32_t my_le) {
        const float INF = __builtin_inff();
        const bool tmine = ((tm >> L) & 1ull) != 0ull;
        uint32_t in_front = 0;  // my NEW entry: new ones in front of it (closer, or as close and earlier in the row)
        uint32_t shift[R];      // my OLD entries: new ones closer than each
#pragma unroll
        for (int i = 0; i < R; ++i) shift[i] = 0u;
        for (uint64_t r = tm; r; r &= r - 1ull) {
            const uint32_t j = (uint32_t)__builtin_ctzll(r);
            const float dj = rl_f(nd, j);
#pragma unroll
            for (int i = 0; i < R; ++i) shift[i] += dj < top.d[i] ? 1u : 0u;
            in_front += (tmine && (dj < nd || (dj == nd && j < L))) ? 1u : 0u;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint32_t p = L * (uint32_t)R + (uint32_t)i;
            if (p < sz) sh.merge[p + shift[i]] = make_uint2(__float_as_uint(top.d[i]), top.s[i]);
        }
        if (tmine) sh.merge[my_le + in_front] = make_uint2(__float_as_uint(nd), n);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint32_t total = sz + (uint32_t)__popcll(tm);
        uint2 t[R + 1];
#pragma unroll
        for (int i = 0; i <= R; ++i) {
            const uint32_t p = L * (uint32_t)R + (uint32_t)i;
            t[i] = p < total ? sh.merge[p] : make_uint2(__float_as_uint(INF), kInvalid);
        }
        bool tie = false;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint32_t p = L * (uint32_t)R + (uint32_t)i;
            tie = tie || (p + 1u < total && p < ef && __uint_as_float(t[i].x) == __uint_as_float(t[i + 1].x));
        }
        if (__builtin_amdgcn_ballot_w64(tie)) return false;  // (nothing committed: `top` is as it was)
        const uint32_t kept = total < ef ? total : ef;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const bool keep = L * (uint32_t)R + (uint32_t)i < kept;
            top.d[i] = keep ? __uint_as_float(t[i].x) : INF;
            top.s[i] = keep ? t[i].y : kInvalid;
        }
        sz = kept;
        if (sz == ef) radius = at(ef - 1u);
        return true;
    }
};

// ---- the top wave (plain walks, round 5) ------------------------------------------------------------------------------
// What a hop does to `top` is a third of a plain hop's instructions, and the walker -- one wave, one instruction per four clocks -- is
// what a lone walk waits for.  The walker needs `top` for two things only: who passes (TopOps::accept: the distances, which it keeps a
// copy of) and the radius at the next pop.  So a wave of its own keeps the buffer: the walker decides who passes, posts the admitted
// ones with their ranks (LDS) and goes on to the pushes while this wave merges them in; at the next pop the merge has finished, and the
// walker reads the new size, the radius and the distances (sh.merge holds the merged list).  The sets are what one wave would have
// produced -- the same closed form, the same tie rules (TopOps) --; the wave takes the place of one helper (the walker posts it no jobs).
template <int R, class Sh>
__device__ __forceinline__ void pipe_top_loop(Sh& sh, uint32_t ef, bool fused_order, int lane) {
    const uint32_t L = (uint32_t)lane;
    PipeTop<R> top;
    TopOps<R, false, Sh> T(top, sh, L, ef, fused_order);
    uint32_t seen = 0;
    auto mirror = [&]() {  // `top` as it is -> sh.merge (after insertions one by one: merge() leaves it there itself)
#pragma unroll
        for (int i = 0; i < R; ++i) sh.merge[L * (uint32_t)R + (uint32_t)i] = make_uint2(__float_as_uint(top.d[i]), top.s[i]);
    };
    for (;;) {
        uint32_t req;
        while ((req = uni(lds_flag_load(&sh.tw_req))) == seen) {
            if (lds_load_relaxed(&sh.stop)) return;
            __builtin_amdgcn_s_sleep(1);  // (64 clocks: the helpers on this SIMD and the LDS port are not for polling)
        }
        seen = req;
        const uint32_t flags = uni(sh.tw_flags);
        {
            const uint64_t cand = ((uint64_t)uni(sh.tw_cand[1]) << 32) | uni(sh.tw_cand[0]);
            const uint64_t okmask = ((uint64_t)uni(sh.tw_ok[1]) << 32) | uni(sh.tw_ok[0]);
            const float nd = sh.tw_nd[L];
            const uint32_t n = sh.tw_n[L];
            T.tie_active = (flags & 1u) != 0u;
            T.tie_v = __uint_as_float(uni(__float_as_uint(sh.tw_tie_v)));
            if (!(flags & 4u)) {  // `cand`: the lanes that passed and may be results (their ranks among the members: counted here, off the walker's path)
                if (T.tie_active || !T.merge(cand, nd, n, T.ranks(cand, nd))) {
                    for (uint64_t r = cand; r; r &= r - 1ull) {  // equal distances, or a tie window: one by one, as the CPU does
                        const uint32_t j = (uint32_t)__builtin_ctzll(r);
                        T.insert(rl_f(nd, j), rl_u(n, j));
                    }
                    mirror();
                }
            } else {  // an infinity or a NaN in the row: the CPU loop as written
                uint64_t pass = 0ull;
                for (uint64_t r = cand; r; r &= r - 1ull) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(r);
                    const float dj = rl_f(nd, j);
                    if (T.sz == ef && !(dj < T.radius)) continue;
                    pass |= 1ull << j;
                    if ((okmask >> j) & 1ull) T.insert(dj, rl_u(n, j));
                }
                mirror();
                if (L == 0u) {
                    sh.tw_pass[0] = (uint32_t)pass;
                    sh.tw_pass[1] = (uint32_t)(pass >> 32);
                }
            }
        }
        if (L == 0u) {
            sh.tw_sz = T.sz;
            sh.tw_radius = T.radius;
            sh.tw_redo = T.redo ? 1u : 0u;
            lds_flag_store(&sh.tw_done, req);
        }
    }
}

// ---- the courier wave (kPipeAsk, round 6) -------------------------------------------------------------------------------
// An opaque predicate lives on the host (usearch.rs:224-248 binds a Rust closure: a table read-lock + restriction evaluation per
// candidate, :1118-1124).  Rounds 3-5 answered such a query in ROUNDS -- an exploring walk that lists the verdicts an exact walk will
// need, the host's answers, the exact walk from scratch: 2.2 walks and twice the CPU's predicate calls.  Here the walk ASKS while it
// runs: the walker posts the slots whose verdict it needs into an LDS ring, this wave (it takes one helper's place, as the top wave
// does in plain walks) writes them to the caller's pinned list and publishes the count (cnt[5]); the caller -- it is waiting for this
// very query -- evaluates the predicate and writes one byte per ask (1 rejected, 2 admitted) into the pinned verdict array; this wave
// polls those bytes (one 64-byte read past the caches per look) and hands them to the walker through LDS.  The walker never touches
// host memory: a store or a load that crosses PCIe would sit in its vmcnt queue in front of the visited atomics it waits for.
constexpr uint32_t kPipeCourierWave = 1u;
template <class Sh>
__device__ __forceinline__ void pipe_courier_loop(Sh& sh, uint32_t* list, const uint8_t* verdict, uint32_t* cnt, int lane) {
    const uint32_t L = (uint32_t)lane;
    uint32_t pub = 0, ans = 0;  // asks published to the host; the answered prefix
    for (uint32_t idle = 0;;) {
        const uint32_t tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_flag_load(&sh.aq_tail));
        if (tail != pub) {
            // (the list and the counter live in the caller's pinned block: stores to it go past the caches, in order -- waiting for their
            // acknowledgement is all the ordering the counter needs.  A system-scope RELEASE here is a write-back of the whole L2, and
            // this happens two thousand times per walk: measured, every phase of every hop of the walker ran twice as slow)
            for (uint32_t i = pub + L; i < tail; i += 64u) __hip_atomic_store(list + i, sh.aq_slot[i & 255u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (L == 0u) __hip_atomic_store(cnt + 5, tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            pub = tail;
            idle = 0;
        }
        if (ans != pub) {
            const uint32_t i = ans + L;
            uint32_t v = 0;
            if (i < pub) v = (uint32_t)__hip_atomic_load(verdict + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v) sh.av[i & 255u] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const uint64_t got = __builtin_amdgcn_ballot_w64(v != 0u);
            ans += got == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~got);  // the answered prefix (the host answers in order)
            if (!got) __builtin_amdgcn_s_sleep(8);
        } else {
            if (lds_load_relaxed(&sh.stop)) break;
            if (++idle > 64u) __builtin_amdgcn_s_sleep(16);
            else __builtin_amdgcn_s_sleep(2);
        }
        if (lds_load_relaxed(&sh.stop)) break;
    }
    // nothing of this query reaches the caller's block after its flag: the walker waits for this before it reports
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (L == 0u) lds_store_release(&sh.courier_done, 1u);
}

struct PipeOut {
    uint32_t status;  // 0 answered; 1 redo (an order-relevant tie / structure outgrown); 2 the round's budget of unknown verdicts is spent (lazy filter)
    uint32_t sz;
};

// MODE: what the instance is for -- each carries only the state it needs (the walker's loop lives on scalar registers, and every
// wave-uniform variable it does not need is one it does not have to spill):
//   kPipePlain    plain lone queries (no filter): no verdict bookkeeping, `next` never outgrows LDS (no spilling), fused-list tie order
//   kPipeFiltered the exact walk of a filtered query
//   kPipeExplore  an exploring round of a lazily filtered query
//   kPipeAsk      the exact walk of a filtered query whose verdicts are ASKED FOR while it runs (round 6: opaque predicates)
enum : int { kPipePlain = 0, kPipeFiltered = 1, kPipeExplore = 2, kPipeAsk = 4 };
template <int AR, int I, int MODE, class Sh>
__device__ __forceinline__ PipeOut pipe_walk(const IndexView& ix, Sh& sh, uint2* pool, uint32_t pool_cap, const WalkSpace& ws, uint32_t start,
                                             float start_d, uint32_t ef, bool tomb, const uint32_t* allow, const uint32_t* known,
                                             uint32_t* unknown_list, uint32_t* unknown_count, uint32_t unknown_cap, uint32_t unknown_budget,
                                             uint32_t* consulted_out, Counters& cnt, int lane, PipeTop<Sh::kEfCap / 64>& top, uint32_t* debug,
                                             bool fused_order) {
    constexpr bool explore = MODE == kPipeExplore;
    constexpr bool kFilter = MODE != kPipePlain;
    constexpr bool ask = MODE == kPipeAsk;
    if constexpr (!kFilter) {
        allow = nullptr;
        known = nullptr;
    }
    constexpr int R = Sh::kEfCap / 64;
    constexpr uint32_t TM = (uint32_t)Sh::kTeam;
    constexpr uint32_t K = (uint32_t)kPipeCache;
    const uint32_t L = (uint32_t)lane;
    const float INF = __builtin_inff();
    // `top`: the walker's own in filtered walks; plain walks keep it in the top wave (pipe_top_loop) and hold its size, its radius and a
    // copy of its distances here
    constexpr bool kTopWave = MODE == kPipePlain;
    // (plain walks, ten helpers: the candidate needed at once split five ways measured 806-808 -> 799-801 us; six ways 1,072 -- the
    // candidates measured ahead starve)
    constexpr uint32_t kUrgentParts = kTopWave ? kPipeUrgentParts + 1u : kPipeUrgentParts;
    TopOps<R, kFilter, Sh> T(top, sh, L, ef, fused_order);
    bool& redo = T.redo;
    uint32_t& sz = T.sz;
    float& radius = T.radius;
    bool& tie_active = T.tie_active;
    float& tie_v = T.tie_v;
#ifdef VS_WALK_PROFILE
    uint64_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t prof_t = __builtin_amdgcn_s_memtime();
#endif
    [[maybe_unused]] uint32_t dbg_max_next = 0, dbg_pushed = 0, dbg_miss = 0, dbg_refill = 0, dbg_early = 0, dbg_windows = 0, dbg_waits = 0, dbg_spills = 0;
    bool over_budget = false;
    const uint32_t guess_t = unknown_budget >> 24;
    unknown_budget &= 0xFFFFFFu;
    uint32_t ucount = 0, consulted = 0, vcount = 0;
    bool vlog_lost = false;
    // front / pool / top
    float f_d = INF;
    uint32_t f_s = kInvalid, f_c = 0, nf = 0, np = 0;  // f_c: 1 = a cache entry (complete or being measured) belongs to this candidate
    float pool_lb = INF;  // the smallest distance in the pool
    // (`next` beyond the pool: slots in global memory, see spill() below)
    const uint32_t slot_cap = kFilter ? pool_cap / 2u : 0u;
    const uint32_t n_slots = (kFilter && slot_cap && ws.heap) ? (ws.heap_cap / slot_cap < 64u ? ws.heap_cap / slot_cap : 64u) : 0u;
    uint32_t g_cnt = 0;    // lane j: entries of slot j
    float g_min = INF;     // lane j: their smallest distance
    float spill_lb = INF;  // the smallest distance in any slot (plain queries: never anything)
    constexpr uint32_t kPoolSlack = 192u;  // room a hop in progress may still need (64 pushes and as many entries displaced from the front)
    bool spill_now = false;
    uint32_t tag = kInvalid;  // lane e < kPipeCache: the candidate cache entry e belongs to
    // Order among EQUAL distances: the one thing these structures do not reproduce.  While two candidates with one distance v wait in
    // `next` together ("window": from the pop of the first of them until the head of `next` lies beyond v) usearch's heap decides who goes
    // first.  The sets the walk works on -- visited, `next`, `top` -- come out the same either way as long as no comparison against the
    // radius can tell the orders apart and no two equal distances are inserted into `top` in an order-dependent sequence.  Precisely
    // (filtered walks; plain ones keep the stricter round-4 rule: inside a window `top` is not full and does not fill up):
    //   * every candidate of the window is expanded whichever goes first iff the radius never falls below v while the window lasts (the
    //     walk ends at the first candidate BEYOND the radius): then the same nodes are evaluated, and `top` -- the ef best admitted
    //     members of what was evaluated -- is the same set;
    //   * `next` may differ by entries at or beyond the radius of their evaluation, which are never expanded -- unless one EQUALS the
    //     radius: a neighbour rejected at `d == radius` inside a window, a candidate popped at `d == radius` after one, an insertion
    //     into `top` that meets an equal distance there inside one.
    // Any of these hands the round to the usearch-order walk (status redo).  One exact walk in 13 did under the stricter rule at 10 %
    // selectivity (10M x 768), seven in ten at 1 %.
    bool any_window = false;

    auto top_insert = [&](float d, uint32_t s) { T.insert(d, s); };
    auto pool_append = [&](bool mine, float d, uint32_t s) {  // every lane with `mine` appends its entry
        const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
        if (!mk) return;
        const uint32_t c = (uint32_t)__popcll(mk);
        if (np + c + (kFilter ? kPoolSlack : 0u) > pool_cap) {
            if constexpr (kFilter) spill_now = true;  // (the hop in progress ends first; then the pool's farther half moves to global memory: spill())
            if (!kFilter || np + c > pool_cap) {
                redo = true;
                return;
            }
        }
        if (mine) pool[np + mbcnt(mk)] = make_uint2(__float_as_uint(d), s);
        np += c;
        pool_lb = fminf(pool_lb, wave_min(mine ? d : INF));
    };
    auto pool_append_one = [&](float d, uint32_t s) {  // one wave-uniform entry
        if (np + 1u + (kFilter ? kPoolSlack : 0u) > pool_cap) {
            if constexpr (kFilter) spill_now = true;
            if (!kFilter || np + 1u > pool_cap) {
                redo = true;
                return;
            }
        }
        if (L == 0u) pool[np] = make_uint2(__float_as_uint(d), s);
        np += 1u;
        pool_lb = fminf(pool_lb, d);
    };
    auto free_entry_of = [&](uint32_t s) {  // the candidate lost its place among the cached ones
        if (L < K && tag == s) tag = kInvalid;
    };
    auto front_insert = [&](float d, uint32_t s, uint32_t c) {  // one entry (wave-uniform) into the sorted front; the displaced worst goes to the pool
        const bool in = L < nf;
        const uint32_t rank = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && f_d <= d));  // (behind equal ones: any order among them is as good)
        const bool full = nf == 64u;
        if (full && rank >= 64u) {  // (an earlier insertion of the same hop moved the front's reach below it)
            pool_append_one(d, s);
            if (c) free_entry_of(s);
            return;
        }
        const float ev_d = rl_f(f_d, 63);
        const uint32_t ev_s = rl_u(f_s, 63), ev_c = rl_u(f_c, 63);
        const float sd = __uint_as_float(wave_shr1(__float_as_uint(f_d), 0u));
        const uint32_t ss = wave_shr1(f_s, 0u), sc = wave_shr1(f_c, 0u);
        f_d = L > rank ? sd : L == rank ? d : f_d;
        f_s = L > rank ? ss : L == rank ? s : f_s;
        f_c = L > rank ? sc : L == rank ? c : f_c;
        if (full) {
            pool_append_one(ev_d, ev_s);
            if (ev_c) free_entry_of(ev_s);
        } else {
            nf += 1u;
        }
    };
    // Several entries (the lanes of `fm`, their own nd / ns, none with a cache entry) into the sorted front AT ONCE (round 5: a plain hop
    // pushes ~8 neighbours below the front's reach, and one front_insert each -- a ballot rank, three readlanes, three lane shifts, six
    // selects, the displaced entry's append -- was half of the lone walk's clocks, scripts/probe/pipe_phase_probe.sh).  One pass over the
    // new entries gives every OLD entry the number of new ones in front of it and every NEW entry its rank among the old (behind equal
    // ones, as front_insert places it) and among the new; then ONE scatter through LDS (sh.stage) puts the 64 closest in place, and
    // whatever lands beyond position 63 is appended to the pool.  The multiset the front and the pool hold is what the insertions one by
    // one leave; the order among EQUAL distances may differ, which nothing reads (windows look at distances only).
    auto front_merge = [&](uint64_t fm, float nd, uint32_t ns) {
        // (a lane may hold an old entry -- position L of the front -- AND a new one -- neighbour L of the hop: two destinations)
        const bool in = L < nf, mine = ((fm >> L) & 1ull) != 0ull;
        uint32_t new_before_old = 0;  // my OLD entry: new entries strictly closer than it
        uint32_t new_before_new = 0;  // my NEW entry: new entries in front of it (closer, or as close and earlier in the row)
        uint32_t old_le_new = 0;      // my NEW entry: old entries not farther than it (it goes behind equal ones, as front_insert places it)
        // (branch-free, for the instruction count: the lanes beyond the front hold +inf, so no `in` is needed in the comparisons; what a
        // lane without an old / a new entry computes is not looked at.  As `a || (b && c)` the rank among the new ones was three nested
        // exec-mask branches per neighbour)
        for (uint64_t r = fm; r;) {
            const uint32_t i = (uint32_t)__builtin_ctzll(r);
            r &= ~(1ull << i);
            const float di = rl_f(nd, i);
            const uint32_t c = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(f_d <= di));
            old_le_new = wl_u(old_le_new, c, i);
            new_before_old += di < f_d ? 1u : 0u;
            new_before_new += (uint32_t)((int)(di < nd) | ((int)(di == nd) & (int)(i < L)));
        }
        const uint32_t old_dest = L + new_before_old, new_dest = old_le_new + new_before_new;
        // beyond position 63: to the pool (old ones may own a cache entry, which they lose)
        const bool old_out = in && old_dest >= 64u, new_out = mine && new_dest >= 64u;
        for (uint64_t r = __builtin_amdgcn_ballot_w64(old_out && f_c != 0u); r; r &= r - 1ull) free_entry_of(rl_u(f_s, (uint32_t)__builtin_ctzll(r)));
        pool_append(old_out, f_d, f_s);
        pool_append(new_out, nd, ns);
        // the scatter (slot | cache flag << 31: slots are below 2^30)
        __builtin_amdgcn_wave_barrier();
        if (in && !old_out) sh.stage[old_dest] = make_uint2(__float_as_uint(f_d), f_s | (f_c << 31));
        if (mine && !new_out) sh.stage[new_dest] = make_uint2(__float_as_uint(nd), ns);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const uint32_t total = nf + (uint32_t)__popcll(fm);
        nf = total < 64u ? total : 64u;
        const uint2 me = L < nf ? sh.stage[L] : make_uint2(__float_as_uint(INF), kInvalid);
        f_d = __uint_as_float(me.x);
        f_s = L < nf ? (me.y & 0x7FFFFFFFu) : kInvalid;
        f_c = L < nf ? (me.y >> 31) : 0u;
    };
    // next.insert for the lanes in `mask` (their own nd / ns): below the front's reach -> the front (one: front_insert; several: one merge);
    // the rest -> the pool
    auto push_lanes = [&](uint64_t mask, float nd, uint32_t ns) {
        const float reach = nf == 64u ? rl_f(f_d, 63) : (kFilter ? fminf(pool_lb, spill_lb) : pool_lb);  // closer than this: belongs to the front
        const bool mine = ((mask >> L) & 1ull) != 0ull;
        const uint64_t fm = __builtin_amdgcn_ballot_w64(mine && nd < reach);
        pool_append(mine && !(nd < reach), nd, ns);
        if (__popcll(fm) > 1) {
            front_merge(fm, nd, ns);
        } else if (fm) {
            const uint32_t j = (uint32_t)__builtin_ctzll(fm);
            front_insert(rl_f(nd, j), rl_u(ns, j), 0u);
        }
        dbg_pushed += (uint32_t)__popcll(mask);
    };
    auto push_one = [&](float d, uint32_t s) {  // next.insert of one wave-uniform entry
        const float reach = nf == 64u ? rl_f(f_d, 63) : (kFilter ? fminf(pool_lb, spill_lb) : pool_lb);
        if (d < reach) front_insert(d, s, 0u);
        else pool_append_one(d, s);
        dbg_pushed += 1u;
    };
    // One hop's admissions at once (TopOps::accept / merge), then the pushes; false: the literal loop must do the hop (a tie window,
    // equal distances in `top`, a distance that is not a finite number).  Plain walks: the top wave does the first two (hop_posted).
    auto hop_batch = [&](uint64_t cand, uint64_t okmask, float nd, uint32_t n) -> bool {
        if (tie_active) return false;
        if (__builtin_amdgcn_ballot_w64((((cand >> L) & 1ull) != 0ull) && !(nd < INF))) return false;
        const uint64_t pass = T.accept(cand, okmask, nd);
        WALK_STAMP(8);  // (inside "pushes, top": who passes)
        const uint64_t tm = pass & okmask;
        if (tm && !T.merge(tm, nd, n, T.ranks(tm, nd))) return false;
        WALK_STAMP(9);  // (the merge into `top`)
        push_lanes(pass, nd, n);
        WALK_STAMP(10);  // (the pushes)
        return true;
    };
    // ---- the top wave's mailbox (plain walks) ----
    uint32_t tw_seq = 0;
    bool tw_pending = false, tw_stale = false;
    auto tw_sync = [&]() {  // the last hop posted is in `top`: its size, its radius
        if (!tw_pending) return;
        for (uint32_t spins = 0; uni(lds_flag_load(&sh.tw_done)) != tw_seq; ++spins) {
            if (spins > (1u << 24)) {
                redo = true;
                break;
            }
        }
        sz = uni(sh.tw_sz);
        radius = __uint_as_float(uni(__float_as_uint(sh.tw_radius)));
        if (uni(sh.tw_redo)) redo = true;
        tw_pending = false;
        tw_stale = true;
    };
    auto tw_distances = [&]() {  // the copy of `top`'s distances accept() works on (asked for while the visited atomics are on their way)
        if (!tw_stale) return;
        // (every position is inside the buffer: read them all, then choose -- as `p < sz ? read : inf` each was a branch around its load)
        uint32_t raw[R];
#pragma unroll
        for (int j = 0; j < R; ++j) raw[j] = sh.merge[L * (uint32_t)R + (uint32_t)j].x;
#pragma unroll
        for (int j = 0; j < R; ++j) top.d[j] = L * (uint32_t)R + (uint32_t)j < sz ? __uint_as_float(raw[j]) : INF;
        tw_stale = false;
    };
    auto tw_post = [&](uint64_t cand, uint64_t okmask, float nd, uint32_t n, uint32_t flags) {
        sh.tw_nd[L] = nd;
        sh.tw_n[L] = n;
        ++tw_seq;
        if (L == 0u) {
            sh.tw_cand[0] = (uint32_t)cand;
            sh.tw_cand[1] = (uint32_t)(cand >> 32);
            sh.tw_ok[0] = (uint32_t)okmask;
            sh.tw_ok[1] = (uint32_t)(okmask >> 32);
            sh.tw_flags = flags | (tie_active ? 1u : 0u);
            sh.tw_tie_v = tie_v;
            lds_flag_store(&sh.tw_req, tw_seq);
        }
        tw_pending = true;
    };
    // one hop (plain walks): who passes is decided here, what it does to `top` is the top wave's; returns the lanes that passed
    auto hop_posted = [&](uint64_t cand, uint64_t okmask, float nd, uint32_t n) -> uint64_t {
        if (__builtin_amdgcn_ballot_w64((((cand >> L) & 1ull) != 0ull) && !(nd < INF))) {  // an infinity or a NaN: the literal loop, over there
            tw_post(cand, okmask, nd, n, 4u);
            tw_sync();
            return ((uint64_t)uni(sh.tw_pass[1]) << 32) | uni(sh.tw_pass[0]);
        }
        const uint64_t pass = T.accept(cand, okmask, nd);
        const uint64_t tm = pass & okmask;
        if (tm) tw_post(tm, tm, nd, n, 0u);
        return pass;
    };
    // Radix select on the order-preserving distance bits of pool[0 .. np): a threshold with between limit / 4 and limit keys below it
    // (all: the whole pool is at most `limit` entries).  ok = false: more than `limit` entries share the smallest distance (count says how many).
    struct Sel {
        uint32_t thr, count;
        bool all, ok;
    };
    auto select_thr = [&](uint32_t limit) -> Sel {
        Sel r{0xFFFFFFFFu, np, true, true};
        if (np <= limit) return r;
        r.all = false;
        uint32_t prefix = 0, below = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) sh.hist[L * 4u + (uint32_t)i] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i = L; i < np; i += 64u) {
                const uint32_t k = dist_key(pool[i].x);
                if (shift == 24 || (k >> (shift + 8)) == prefix) atomicAdd(&sh.hist[(k >> shift) & 255u], 1u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t h0 = sh.hist[L * 4u], h1 = sh.hist[L * 4u + 1u], h2 = sh.hist[L * 4u + 2u], h3 = sh.hist[L * 4u + 3u];
            uint32_t incl = h0 + h1 + h2 + h3;  // inclusive scan over lanes
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if (L >= (uint32_t)o) incl += up;
            }
            const uint32_t excl = incl - (h0 + h1 + h2 + h3);
            const uint32_t room = limit - below;  // entries that may still move
            const uint64_t over = __builtin_amdgcn_ballot_w64(incl > room);  // first bucket whose inclusive count exceeds the room
            if (!over) {  // (cannot happen: the first pass sees np > limit entries, a later one a bucket that exceeded the room)
                r.ok = false;
                r.count = 0;
                return r;
            }
            const uint32_t ol = (uint32_t)__builtin_ctzll(over);
            const uint32_t e0 = rl_u(excl, ol), a0 = rl_u(h0, ol), a1 = rl_u(h1, ol), a2 = rl_u(h2, ol);
            uint32_t c = e0, b = ol * 4u;
            if (c + a0 <= room) { c += a0; ++b; if (c + a1 <= room) { c += a1; ++b; if (c + a2 <= room) { c += a2; ++b; } } }
            below += c;
            r.thr = shift == 24 ? (b << 24) : ((prefix << (shift + 8)) | (b << shift));
            if (below >= limit / 4u) break;
            if (shift == 0) {
                // thr is a full 32-bit key now.  Nothing below it: more than `limit` entries share the smallest distance; that run is
                // the selection (the caller decides whether it can take it whole)
                if (below == 0u) {
                    const uint32_t a3 = rl_u(h3, ol);
                    below = (b & 3u) == 0u ? a0 : (b & 3u) == 1u ? a1 : (b & 3u) == 2u ? a2 : a3;
                    r.thr = r.thr + 1u;  // (0xFFFFFFFF is the key of no distance: +inf maps below it)
                    r.ok = false;
                }
                break;
            }
            prefix = shift == 24 ? b : ((prefix << 8) | b);
        }
        r.count = below;
        return r;
    };
    // ---- `next` beyond LDS: slots of pool_cap / 2 entries in global memory (ws.heap).  A filter that admits 1 % keeps `top` short of its
    // limit for thousands of hops, and `next` grows to tens of thousands of entries.  When the pool is full its farther half moves to one or
    // two free slots (lane j keeps slot j's count and smallest distance); when the front needs entries that may lie beyond the pool, the
    // slot with the smallest distance comes back whole.  Order is untouched: the front still pops the global minimum.
    auto spill = [&]() {
        if constexpr (!kFilter) {
            redo = true;  // (a plain walk's `next` stays below four times the beam: a pool that fills up is handed over)
            return;
        } else {
        ++dbg_spills;
        const Sel sel = select_thr(slot_cap);
        if (sel.all || (!sel.ok && sel.count == 0u)) {
            redo = true;
            return;
        }
        const uint64_t freem_s = __builtin_amdgcn_ballot_w64(L < n_slots && g_cnt == 0u);
        if ((uint32_t)__popcll(freem_s) < 2u) {  // global memory for `next` is used up too: the other walk (or exhaustive ranking) serves
            redo = true;
            return;
        }
        const uint32_t sa = (uint32_t)__builtin_ctzll(freem_s), sb = (uint32_t)__builtin_ctzll(freem_s & (freem_s - 1ull));
        uint32_t kept = 0, moved = 0;
        float lb = INF, mina = INF, minb = INF;
        for (uint32_t base = 0; base < np; base += 64u) {
            const bool valid = base + L < np;
            const uint2 e = valid ? pool[base + L] : make_uint2(0u, 0u);
            const float d = __uint_as_float(e.x);
            const bool keep = valid && dist_key(e.x) < sel.thr;
            const bool move = valid && !keep && !(sz == ef && d > radius);  // (beyond the radius of a full `top`: never expanded, dropped)
            const uint64_t km = __builtin_amdgcn_ballot_w64(keep), mm = __builtin_amdgcn_ballot_w64(move);
            if (keep) {
                pool[kept + mbcnt(km)] = e;
                lb = fminf(lb, d);
            }
            if (move) {
                const uint32_t pos = moved + mbcnt(mm);
                const bool in_a = pos < slot_cap;
                ws.heap[(size_t)(in_a ? sa : sb) * slot_cap + (in_a ? pos : pos - slot_cap)] = e;
                mina = in_a ? fminf(mina, d) : mina;
                minb = in_a ? minb : fminf(minb, d);
            }
            kept += (uint32_t)__popcll(km);
            moved += (uint32_t)__popcll(mm);
        }
        np = kept;
        pool_lb = wave_min(lb);
        mina = wave_min(mina);
        minb = wave_min(minb);
        if (L == sa) {
            g_cnt = moved < slot_cap ? moved : slot_cap;
            g_min = mina;
        }
        if (L == sb && moved > slot_cap) {
            g_cnt = moved - slot_cap;
            g_min = minb;
        }
        spill_lb = wave_min(g_cnt ? g_min : INF);
        }
    };
    auto unspill = [&]() {  // the slot that holds the smallest spilled distance comes back into the pool
        if constexpr (!kFilter) {
            redo = true;
            return;
        } else {
        ++dbg_spills;
        if (np > slot_cap) {
            spill();
            if (redo) return;
        }
        const uint64_t hm = __builtin_amdgcn_ballot_w64(L < n_slots && g_cnt != 0u && g_min == spill_lb);
        if (!hm) {
            spill_lb = INF;
            return;
        }
        const uint32_t sj = (uint32_t)__builtin_ctzll(hm);
        const uint32_t cnt = rl_u(g_cnt, sj);
        if (np + cnt > pool_cap) {  // (only when a run of equal distances kept the pool above half after a spill)
            redo = true;
            return;
        }
        float lb = INF;
        uint32_t added = 0;
        for (uint32_t base = 0; base < cnt; base += 64u) {
            const bool valid = base + L < cnt;
            uint2 e = make_uint2(0u, 0u);
            if (valid) e = ws.heap[(size_t)sj * slot_cap + base + L];
            const bool keep = valid && !(sz == ef && __uint_as_float(e.x) > radius);
            const uint64_t km = __builtin_amdgcn_ballot_w64(keep);
            if (keep) {
                pool[np + added + mbcnt(km)] = e;
                lb = fminf(lb, __uint_as_float(e.x));
            }
            added += (uint32_t)__popcll(km);
        }
        np += added;
        pool_lb = fminf(pool_lb, wave_min(lb));
        if (L == sj) {
            g_cnt = 0u;
            g_min = INF;
        }
        spill_lb = wave_min(g_cnt ? g_min : INF);
        }
    };
    // the front ran empty: the closest entries of the pool move up
    auto refill = [&](uint32_t limit = 64u) {  // limit: at most this many move up
        ++dbg_refill;
        Sel sel;
        for (uint32_t rounds = 0;; ++rounds) {
            sel = select_thr(limit);
            if (!sel.ok) {
                if (sel.count == 0u || sel.count > 64u) {  // a run of more than 64 equal distances (or the impossible): the other walk serves
                    redo = true;
                    return;
                }
            }
            // may a spilled entry belong among the ones selected?  (all: the pool's last entries move, whatever lies in the slots comes next)
            const bool spilled_first = kFilter && spill_lb != INF && (sel.all || dist_key(__float_as_uint(spill_lb)) < sel.thr);
            if (!spilled_first) break;
            if (rounds > 128u) {
                redo = true;
                return;
            }
            unspill();
            if (redo) return;
        }
        const uint32_t thr = sel.thr;
        const bool all = sel.all;
        uint32_t taken = 0, kept = 0;
        float lb = INF;
        for (uint32_t base = 0; base < np; base += 64u) {
            const bool valid = base + L < np;
            const uint2 e = valid ? pool[base + L] : make_uint2(0u, 0u);
            const bool take = valid && (all || dist_key(e.x) < thr);
            // (once `top` is full the radius only shrinks: a candidate beyond it can never be expanded -- `candidate.distance > radius` ends
            // the walk when it is the closest -- so it is dropped here instead of being carried along)
            const bool keep = valid && !take && !(sz == ef && __uint_as_float(e.x) > radius);
            const uint64_t tm = __builtin_amdgcn_ballot_w64(take), km = __builtin_amdgcn_ballot_w64(keep);
            if (taken + (uint32_t)__popcll(tm) > 64u) {  // (cannot happen: the select counted them)
                redo = true;
                return;
            }
            if (take) sh.stage[taken + mbcnt(tm)] = e;
            if (keep) {
                pool[kept + mbcnt(km)] = e;
                lb = fminf(lb, __uint_as_float(e.x));
            }
            taken += (uint32_t)__popcll(tm);
            kept += (uint32_t)__popcll(km);
        }
        np = kept;
        pool_lb = wave_min(lb);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint2 me = L < taken ? sh.stage[L] : make_uint2(__float_as_uint(INF), kInvalid);
        const float md = __uint_as_float(me.x);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < taken; ++j) {
            const float dj = rl_f(md, j);
            rank += (dj < md || (dj == md && j < L)) ? 1u : 0u;
        }
        // lane l sends its entry to lane rank (a permutation of 0 .. taken - 1)
        f_d = __int_as_float(__builtin_amdgcn_ds_permute((int)(rank << 2), __float_as_int(md)));
        f_s = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)me.y);
        if (L >= taken) {
            f_d = INF;
            f_s = kInvalid;
        }
        f_c = 0u;
        nf = taken;
    };

    // ---- jobs ----
    uint64_t idle = 0, freem = 0;  // helpers without a job / cache entries that can take one, as of the hop's start minus what the hop used
    auto refresh_jobs = [&]() {
        uint32_t js = 1u, cr = 1u;
        if (L >= 1u && L < TM) js = lds_load_relaxed(&sh.job_state[L]);
        if (L < K) cr = lds_load_relaxed(&sh.c_ready[L]);
        idle = __builtin_amdgcn_ballot_w64(L >= 1u && L < TM && js == 0u && !(kTopWave && L == kPipeTopWave)  && !(ask && L == kPipeCourierWave));  // (the top wave / the courier take no jobs)
        freem = __builtin_amdgcn_ballot_w64(L < K && tag == kInvalid && cr == 0u);
    };
    // "measure candidate s", split over up to `want` helpers; the entry, or kInvalid when no helper or no entry is to be had
    auto post_job = [&](uint32_t s, uint32_t want, uint32_t flags = 0u) -> uint32_t {  // flags: 1 claim (exploring), 2 measure every neighbour (urgent)
        const uint32_t avail = (uint32_t)__popcll(idle);
        if (!avail || !freem) return kInvalid;
        const uint32_t parts = want < avail ? want : avail;
        const uint32_t e = (uint32_t)__builtin_ctzll(freem);
        freem &= freem - 1ull;
        if (L == e) tag = s;
        if (L == 0u) sh.c_ready[e] = parts;
        // lane h of the first `parts` idle helpers writes helper h's job: four stores for the whole post, the flag last
        const uint32_t rank = mbcnt(idle);
        const bool pick = ((idle >> L) & 1ull) != 0ull && rank < parts;
        if (pick) {
            sh.job_slot[L] = s;
            sh.job_entry[L] = e;
            sh.job_part[L] = rank | (parts << 8) | (flags << 16);
            lds_flag_store(&sh.job_state[L], 1u);
        }
        idle &= ~__builtin_amdgcn_ballot_w64(pick);
        return e;
    };
    // keep the first kPipeAhead entries of the front measured (two helpers stay in reserve for the candidate a hop needs at once)
    auto schedule = [&]() {
        const uint32_t want = nf < (uint32_t)kPipeAhead ? nf : (uint32_t)kPipeAhead;
        uint64_t missing = __builtin_amdgcn_ballot_w64(L < want && f_c == 0u);
        for (; missing; missing &= missing - 1ull) {
            const uint32_t i = (uint32_t)__builtin_ctzll(missing);
            const uint32_t avail = (uint32_t)__popcll(idle);
            if (i == 0u ? avail == 0u : avail < kPipeParts + 2u) break;
            const uint32_t e = post_job(rl_u(f_s, i), i == 0u ? kUrgentParts : kPipeParts, i == 0u ? kPipeUrgentFlags : 0u);
            if (e == kInvalid) break;
            if (L == i) f_c = 1u;
        }
    };

    // ---- start: visits.set(start); next.insert(start); top.insert(start) if it may be a result ----
    auto mark = [&](uint32_t n, bool valid) -> bool {  // true: n is new to the visited set
        bool fresh = false;
        if constexpr (!Sh::kVisGlobal) {
            return valid && !visited_test_and_set(sh, n);
        }
        if (valid) {
            const uint32_t bit = 1u << (n & 31u);
            fresh = (atomicOr(&ws.bitmap[n >> 5], bit) & bit) == 0u;
        }
        const uint64_t fm = __builtin_amdgcn_ballot_w64(fresh);
        const uint32_t c = (uint32_t)__popcll(fm);
        if (vcount + c <= ws.vlog_cap) {
            if (fresh) ws.vlog[vcount + mbcnt(fm)] = n;
        } else {
            vlog_lost = true;
        }
        vcount += c;
        return fresh;
    };
    // verdict bookkeeping of the lanes in `ask` (flags fl): consulted / unknown listing / guess; returns the admitted lanes
    auto verdicts = [&](uint64_t ask, uint32_t n, uint32_t fl) -> uint64_t {
        const bool mine = ((ask >> L) & 1ull) != 0ull;
        const bool live = mine && (fl & kPfLive) != 0u;
        if (!allow) return __builtin_amdgcn_ballot_w64(live);
        consulted += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
        const bool unk = live && !(fl & kPfKnown);
        const uint64_t um = __builtin_amdgcn_ballot_w64(unk);
        if (um) {
            const uint32_t c = (uint32_t)__popcll(um);
            if (unk && ucount + mbcnt(um) < unknown_cap) unknown_list[ucount + mbcnt(um)] = n;
            ucount += c;
            if (ucount >= unknown_budget) over_budget = true;
        }
        const bool guess = unk && guess_t != 0u && ((n * 2654435761u) >> 24) < guess_t;
        return __builtin_amdgcn_ballot_w64((live && (fl & kPfAllowed) != 0u) || guess);
    };
    // ---- asks (kPipeAsk): the walk goes on while its questions are on their way --------------------------------------------------
    // A neighbour that passes the radius test and whose verdict the host has not given yet is ASKED ABOUT (the courier wave carries the
    // question and the answer) and waits in a PENDING lane of the walker: (distance, slot, ask number).  It is pushed to `next` at once
    // -- usearch pushes whatever passes the radius test, admitted or not -- and enters `top` when its answer arrives.  Until then the
    // walker's `top` is the PESSIMISTIC one (pending neighbours left out): its radius is at or beyond the true one, so
    //   * who passes the radius test, who is pushed, what refill / spill keep: supersets of usearch's, by entries at or beyond the
    //     true radius of their hop.  The radius only shrinks once `top` is full, so such an entry can never be expanded: when it is the
    //     closest candidate, everything usearch's own `next` holds lies as far or farther, and usearch ends its walk there too.  (At
    //     EQUALITY with the radius the orders could differ: a candidate popped at the radius that is not top's last member hands over.)
    //   * `top` as a SET is the `ef` closest admitted neighbours of everything that was offered, whatever the order of arrival; the
    //     order among EQUAL distances is arrival order, so a late insertion that meets an equal distance hands over (insert(lazy)).
    //   * the one decision that needs the TRUE radius is the pop: `candidate.distance > radius && top.size() == limit` ends the walk.
    //     With P answers out, the walk certainly goes on while fewer than ef - P members of the pessimistic `top` lie at or below the
    //     candidate (even if all P were admitted and closer, the ef-th best would still lie beyond it); it certainly ends when the
    //     pessimistic `top` is full and the candidate lies beyond its radius (the true one is not larger) -- after the last answers are
    //     in.  In between the walker waits for answers: the last stretch of a walk, where the candidates approach the radius.
    //   * inside a tie window (equal distances waiting in `next` together) every rule about the radius is exact: the window starts with
    //     all answers in, and its hops ask and WAIT (ask_now) before they admit anybody.
    // Every member is asked about at most once per walk (it is asked when it is NEW to the visited set), as usearch does.
    // (State in VECTOR registers on purpose: the walker's loop lives on scalar registers, and the first version of this -- a 64-bit mask of
    // the lanes in use, the ask counter and two more wave-uniform words -- doubled the scalar spills of the kernel (161 against 77) and
    // made every phase of every hop slower: 8.4 ms a walk against 6.7 with the same decisions and no asking, scripts/probe/ask_probe.py.)
    float p_d = INF;
    uint32_t p_s = kInvalid, p_i = 0u;
    bool p_on = false;        // this lane holds a pending neighbour
    uint32_t asked_v = 0u;    // asks posted so far (the same in every lane)
    auto pending = [&]() -> uint32_t { return (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(p_on)); };
    // what the courier has brought: admitted neighbours enter `top`, their lanes are free again; returns the lanes still pending
    auto settle = [&](uint32_t target) -> uint32_t {  // ... and waits until at most `target` answers are out (64: no wait)
        uint32_t left = 0;
        for (uint32_t spins = 0;; ++spins) {
            uint32_t v = 0u;
            if (p_on) v = lds_load_relaxed(&sh.av[p_i & 255u]);
            for (uint64_t r = __builtin_amdgcn_ballot_w64(v == 2u); r; r &= r - 1ull) {
                const uint32_t j = (uint32_t)__builtin_ctzll(r);
                const float dj = rl_f(p_d, j);
                if (sz < ef || dj < radius) T.insert(dj, rl_u(p_s, j), true);
            }
            p_on = p_on && v == 0u;
            left = pending();
            if (left <= target || redo) break;
            if (spins > (1u << 21)) {  // (a host that does not answer for a fifth of a second: the rounds of the old path serve the query)
                redo = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        return left;
    };
    auto ask_post = [&](uint64_t um, uint32_t n) -> uint32_t {  // the lanes of `um` ask about their n; this lane's ask number
        const uint32_t c = (uint32_t)__popcll(um);
        const uint32_t mine_i = asked_v + mbcnt(um);
        if ((um >> L) & 1ull) {
            sh.av[mine_i & 255u] = 0u;
            sh.aq_slot[mine_i & 255u] = n;
        }
        asked_v += c;
        if (__builtin_amdgcn_ballot_w64(asked_v > unknown_cap)) redo = true;  // (the caller's list is full: the rounds serve the query)
        else if (L == 0u) lds_flag_store(&sh.aq_tail, asked_v);
        return mine_i;
    };
    auto ask_now = [&](uint64_t um, uint32_t n) -> uint64_t {  // the answers before anything else happens (a tie window; no pending lane free)
        const uint32_t mine_i = ask_post(um, n);
        if (redo) return 0ull;
        const bool mine = ((um >> L) & 1ull) != 0ull;
        uint32_t v = 0u;
        for (uint32_t spins = 0;; ++spins) {
            if (mine && v == 0u) v = lds_load_relaxed(&sh.av[mine_i & 255u]);
            if (!__builtin_amdgcn_ballot_w64(mine && v == 0u)) break;
            if (spins > (1u << 21)) {
                redo = true;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        return __builtin_amdgcn_ballot_w64(mine && v == 2u);
    };
    // lazily: the asked neighbours wait in pending lanes -- the r-th asking lane's entry goes to the r-th free lane, a broadcast per ask (a
    // hop asks about two or three neighbours).  false: not enough lanes free (the caller asks with ask_now instead)
    auto ask_lanes = [&](uint64_t um, float nd, uint32_t n) -> bool {
        const uint32_t c = (uint32_t)__popcll(um);
        const uint64_t fm = __builtin_amdgcn_ballot_w64(!p_on);
        if ((uint32_t)__popcll(fm) < c) return false;
        const uint32_t mine_i = ask_post(um, n);
        if (redo) return true;
        const uint32_t fr = mbcnt(fm);
        uint32_t r = 0;
        for (uint64_t m = um; m; m &= m - 1ull, ++r) {
            const uint32_t j = (uint32_t)__builtin_ctzll(m);
            const float dj = rl_f(nd, j);
            const uint32_t sj = rl_u(n, j), ij = rl_u(mine_i, j);
            if (!p_on && fr == r) {
                p_d = dj;
                p_s = sj;
                p_i = ij;
            }
        }
        p_on = p_on || fr < c;
        return true;
    };
    if constexpr (!Sh::kVisGlobal) {
        visited_clear(sh, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    (void)mark(start, L == 0u);
    cnt.evals += 1;  // (the walk measures its start, as walk_usearch does; here the descent's value is reused -- same code, same bits)
    {
        uint32_t fl0 = 0;
        if (L == 0u) {
            const bool live = !tomb || ix.keys[start] != kFreeKey;
            fl0 = live ? kPfLive : 0u;
            if (live) {
                if (!allow) fl0 |= kPfKnown | kPfAllowed;
                else {
                    const bool kn = !known || ((known[start >> 5] >> (start & 31u)) & 1u) != 0u;
                    const bool al = ((allow[start >> 5] >> (start & 31u)) & 1u) != 0u;
                    fl0 |= (kn ? kPfKnown : 0u) | ((kn && al) ? kPfAllowed : 0u);
                }
            }
        }
        if constexpr (ask) {
            front_insert(start_d, start, 0u);
            const uint32_t f0 = rl_u(fl0, 0);
            if (f0 & kPfLive) {
                consulted += 1u;
                if (allow && (f0 & kPfKnown)) {
                    if (f0 & kPfAllowed) top_insert(start_d, start);
                } else {
                    (void)ask_lanes(1ull, start_d, start);
                }
            }
        } else {
        const uint64_t ok0 = verdicts(1ull, start, fl0);
        front_insert(start_d, start, 0u);
        if constexpr (kTopWave) {
            if (ok0 & 1ull) tw_post(1ull, 1ull, start_d, start, 0u);
        } else {
            if (ok0 & 1ull) top_insert(start_d, start);
        }
        }
    }
    // ---- exploring round (lazy filter): no answer is taken from it, so no order has to be kept.  The closest candidates are expanded
    // several at a time, each by one helper that also marks what it measures; the walker only merges: verdict bookkeeping (which lists
    // the slots whose verdict is missing -- what the round is for), pushes, `top` (with guessed verdicts, so that the radius behaves
    // as the exact walk's will).  Two batches are kept in flight: the next one is posted before the last one is merged.
    if constexpr (explore) {
        constexpr uint32_t kBatch = (TM - 1u) / 2u;
        auto post_batch = [&]() -> uint64_t {
            refresh_jobs();
            uint64_t ents = 0;
            for (uint32_t c = 0; c < kBatch && nf && idle && freem; ++c) {
                const float cd = rl_f(f_d, 0);
                const uint32_t cs = rl_u(f_s, 0);
                if (sz == ef && cd > radius) break;
                f_d = __uint_as_float(wave_shl1(__float_as_uint(f_d), __float_as_uint(INF)));
                f_s = wave_shl1(f_s, kInvalid);
                f_c = wave_shl1(f_c, 0u);
                nf -= 1u;
                cnt.hops += 1;
                const uint32_t e = post_job(cs, 1u, 1u);
                ents |= 1ull << e;
            }
            return ents;
        };
        auto merge = [&](uint64_t ents) {
            for (; ents && !redo; ents &= ents - 1ull) {
                const uint32_t e = (uint32_t)__builtin_ctzll(ents);
                for (uint32_t spins = 0; lds_flag_load(&sh.c_ready[e]) != 0u; ++spins) {
                    if (spins > (1u << 22)) {
                        redo = true;
                        return;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                WALK_STAMP(7);  // exploring: waiting for an entry
                const uint32_t n = sh.c_slot[e][lane];
                const uint32_t fl = sh.c_flag[e][lane];
                float nd = sh.c_dist[e][lane];
                if (L == e) tag = kInvalid;
                const bool evd = n != kInvalid && (fl & kPfEvaluated) != 0u;  // claimed by the helper: new to the visited set
                if (!evd) nd = INF;
                cnt.evals += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(evd));
                if (over_budget) continue;  // (the entries of a batch in flight are still drained)
                const uint64_t cand = __builtin_amdgcn_ballot_w64(evd && (sz < ef || nd < radius));
                const uint64_t okmask = verdicts(cand, n, fl);
                WALK_STAMP(4);  // exploring: verdicts
                push_lanes(cand, nd, n);
                WALK_STAMP(5);  // exploring: pushes
                for (uint64_t r = okmask; r; r &= r - 1ull) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(r);
                    const float dj = rl_f(nd, j);
                    if (sz < ef || dj < radius) top_insert(dj, rl_u(n, j));
                }
                WALK_STAMP(6);  // exploring: top
                if (spill_now && !redo) {
                    spill();
                    spill_now = false;
                }
            }
        };
        uint64_t flying = 0;
        for (;;) {
            WALK_STAMP(1);
            if (spill_now && !redo) {
                spill();
                spill_now = false;
            }
            if (nf == 0u && (np != 0u || spill_lb != INF) && !over_budget && !redo) refill();
            WALK_STAMP(2);  // exploring: refill
            const uint64_t posted = (redo || over_budget) ? 0ull : post_batch();
            WALK_STAMP(3);  // exploring: posting a batch
            if (flying) merge(flying);
            flying = posted;
            dbg_max_next = nf + np > dbg_max_next ? nf + np : dbg_max_next;
            if (redo) break;
            if (!flying) {
                if (over_budget) break;
                if (nf && sz == ef && rl_f(f_d, 0) > radius) nf = 0u;  // (the batch's rest lies beyond the radius: dropped)
                if (nf == 0u) {
                    if ((np == 0u && spill_lb == INF) || (sz == ef && fminf(pool_lb, spill_lb) > radius)) break;
                    continue;  // (refill next)
                }
                __builtin_amdgcn_s_sleep(1);  // (no helper or entry free although nothing is in flight: cannot last)
            }
        }
        // every helper must have finished marking before the bitmap is wiped (the visited log was not kept: the helpers marked)
        for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
            refresh_jobs();
            if ((uint32_t)__popcll(idle) == TM - 1u) break;
            __builtin_amdgcn_s_sleep(1);
        }
        vlog_lost = true;
    }
    refresh_jobs();
    if constexpr (!explore) schedule();
    WALK_STAMP(0);
    if constexpr (!explore)
    while (!redo && !over_budget) {
        if constexpr (kFilter) {
            if (spill_now) {
                spill();
                spill_now = false;
                if (redo) break;
            }
        }
        if (nf == 0u) {
            if (np == 0u && (!kFilter || spill_lb == INF)) break;
            refill();
            if (redo) break;
            if (nf == 0u) continue;  // (what was selected lay beyond the radius and was dropped: look again)
        }
        const float cd = rl_f(f_d, 0);
        const uint32_t cs = rl_u(f_s, 0);
        if constexpr (kTopWave) {
            tw_sync();
            if (redo) break;
        }
        if constexpr (kFilter && !ask) WALK_STAMP(8);
        if constexpr (ask) {  // (see "asks" above: the pop is the one decision that needs the true radius)
            WALK_STAMP(8);  // (profile builds, asking walks: spill / refill)
            // (the answers are looked at when the lanes fill up or the decision needs them -- not every hop: the look is an LDS round trip
            // on the walker's path, and nothing waits for an admitted neighbour but the radius)
            uint32_t out_n = pending();
            if (out_n >= 24u) out_n = settle(64u);
            for (bool looked = false; out_n && !redo;) {
                const bool ending = sz == ef && cd > radius;  // ends here, once the last answers are in (the radius only shrinks)
                if (!ending) {
                    uint32_t le = 0;  // members of the pessimistic `top` at or below the candidate (unused positions hold +inf)
#pragma unroll
                    for (int j = 0; j < R; ++j) le += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(top.d[j] <= cd));
                    // (only the pending neighbours at or below the candidate could pull the ef-th best below it)
                    if (le + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(p_on && p_d <= cd)) < ef) break;
                }
                // undecided: first whatever has arrived meanwhile, then the wait -- for every answer when the walk is about to end, else for one more
                out_n = settle(!looked ? 64u : ending ? 0u : out_n - 1u);
                looked = true;
            }
            if (redo) break;
            WALK_STAMP(9);  // (asking walks: the answers taken in, the pop decided)
        }
        if (sz == ef && cd > radius) break;  // `candidate.distance > radius && top.size() == top_limit`
        if (kFilter && (any_window || ask) && !fused_order && sz == ef && cd == radius) {
            // at the radius: the last member of `top` itself, as a rule (every admitted member waits in `next` too) -- or another node at
            // the same distance, which usearch's `next` may not hold: the one distance where that matters
            const uint32_t lp = (ef - 1u) / (uint32_t)R, lr = (ef - 1u) % (uint32_t)R;
            uint32_t last_s = rl_u(top.s[0], lp);
#pragma unroll
            for (int j = 1; j < R; ++j) {
                const uint32_t x = rl_u(top.s[j], lp);
                last_s = lr == (uint32_t)j ? x : last_s;
            }
            if (last_s != cs) {
                redo = true;
                break;
            }
        }
        // pop
        f_d = __uint_as_float(wave_shl1(__float_as_uint(f_d), __float_as_uint(INF)));
        f_s = wave_shl1(f_s, kInvalid);
        f_c = wave_shl1(f_c, 0u);
        nf -= 1u;
        cnt.hops += 1;
        const float next_d = nf ? rl_f(f_d, 0) : (kFilter ? fminf(pool_lb, spill_lb) : pool_lb);  // (+inf when nothing waits beyond the front)
        if (tie_active && cd > tie_v) tie_active = false;
        if (!tie_active && next_d == cd) {
            tie_active = true;
            if constexpr (kFilter) any_window = true;
            tie_v = cd;
            ++dbg_windows;
            if constexpr (ask) {  // the window's rules are about the true radius: it starts with every answer in
                if (pending()) (void)settle(0u);
                if (redo) break;
            }
        }
        // its evaluated neighbours
        uint64_t hitm = __builtin_amdgcn_ballot_w64(L < K && tag == cs);
        if (!hitm) {  // not even posted (every helper was busy, or it arrived with this very hop): post it now, first in line
            ++dbg_miss;
            for (uint32_t spins = 0;; ++spins) {
                if (spins > (1u << 22)) {
                    redo = true;
                    break;
                }
                refresh_jobs();
                const uint32_t e = post_job(cs, kUrgentParts, kPipeUrgentFlags);
                if (e != kInvalid) {
                    hitm = 1ull << e;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (redo) break;
        const uint32_t e = (uint32_t)__builtin_ctzll(hitm);
        WALK_STAMP(1);  // pop, lookup
        for (uint32_t spins = 0; lds_flag_load(&sh.c_ready[e]) != 0u; ++spins) {
            if (spins > (1u << 22)) {  // (a helper that never answers: give the query to the other walk rather than hang the device)
                redo = true;
                break;
            }
            if (spins == 0) ++dbg_waits;
            __builtin_amdgcn_s_sleep(1);
        }
        if (redo) break;
        WALK_STAMP(7);  // wait for the entry
        const uint32_t n = sh.c_slot[e][lane];
        const uint32_t fl = sh.c_flag[e][lane];
        float nd = sh.c_dist[e][lane];
        if (L == e) tag = kInvalid;  // the entry is free again
        refresh_jobs();
        // visited test-and-set: the atomics are on their way while the next candidate is looked for
        // (only the atomic itself sits in the divergent block: its value is looked at after the early post, so the round trip runs under it)
        const uint32_t vbit = 1u << (n & 31u);
        uint32_t vold = vbit;
        if constexpr (Sh::kVisGlobal) {
            if (n != kInvalid) vold = atomicOr(&ws.bitmap[n >> 5], vbit);
        } else {
            if (n != kInvalid && !visited_test_and_set(sh, n)) vold = 0u;
        }
        // The closest neighbour measured for this candidate, when it is closer than everything that waits in `next`, is the very next
        // candidate (if it is new, which the atomics will tell): its own measurement starts NOW, not after this hop's bookkeeping.
        if constexpr (kTopWave) tw_distances();
        const bool evd = n != kInvalid && (fl & kPfEvaluated) != 0u;
        if (!evd) nd = INF;
        uint32_t early_slot = kInvalid, early_e = kInvalid;
        {
            const float best = wave_min((fl & kPfSeen) ? INF : nd);
            if (best < next_d && (sz < ef || best < radius)) {
                const uint64_t bm = __builtin_amdgcn_ballot_w64(evd && !(fl & kPfSeen) && nd == best);
                early_slot = rl_u(n, (uint32_t)__builtin_ctzll(bm));
                early_e = post_job(early_slot, kUrgentParts, kPipeUrgentFlags);
                if (early_e == kInvalid) early_slot = kInvalid;
                else ++dbg_early;
            }
        }
        WALK_STAMP(2);  // entry read, atomics issued, early post
        const bool fresh = (vold & vbit) == 0u;
        const uint64_t fmask = __builtin_amdgcn_ballot_w64(fresh);
        WALK_STAMP(3);  // the atomics' round trip
        if constexpr (!Sh::kVisGlobal) {
            if (uni(sh.overflowed)) {  // the table is full: the other walk serves the query
                redo = true;
                break;
            }
        } else {
            const uint32_t c = (uint32_t)__popcll(fmask);
            if (vcount + c <= ws.vlog_cap) {
                if (fresh) ws.vlog[vcount + mbcnt(fmask)] = n;
            } else {
                vlog_lost = true;
            }
            vcount += c;
        }
        const uint32_t m = (uint32_t)__popcll(fmask);
        if (__builtin_amdgcn_ballot_w64(fresh && !(fl & kPfEvaluated))) {  // (cannot happen: the bitmap only grows) -- never trust a distance that is not there
            redo = true;
            break;
        }
        cnt.evals += m;
        if (m) {
            if (!fresh) nd = INF;
            // verdicts are needed only for neighbours that can still be admitted: once `top` is full, those below the hop's first radius
            const float radius0 = sz == ef ? radius : INF;
            const uint64_t cand = __builtin_amdgcn_ballot_w64(fresh && (sz < ef || nd < radius0));
            if (kFilter && tie_active && sz == ef && __builtin_amdgcn_ballot_w64(fresh && nd == radius0)) {  // rejected AT the radius inside a window
                redo = true;
                break;
            }
            uint64_t okmask;
            bool done;
            if constexpr (ask) {
                const bool cm = ((cand >> L) & 1ull) != 0ull, live = cm && (fl & kPfLive) != 0u;
                const bool kn = live && allow != nullptr && (fl & kPfKnown) != 0u;
                okmask = __builtin_amdgcn_ballot_w64(kn && (fl & kPfAllowed) != 0u);
                const uint64_t um = __builtin_amdgcn_ballot_w64(live && !kn);
                consulted += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(live));
                if (tie_active) {  // inside a window: the hop as usearch does it, every verdict in hand first
                    if (um) okmask |= ask_now(um, n);
                    if (redo) break;
                    done = hop_batch(cand, okmask, nd, n);
                } else {
                    // (no pending lane free: this hop's answers are waited for -- the lanes fill up only when the caller is slow)
                    if (um && !ask_lanes(um, nd, n)) okmask |= ask_now(um, n);
                    if (redo) break;
                    for (uint64_t r = okmask; r; r &= r - 1ull) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(r);
                        const float dj = rl_f(nd, j);
                        if (sz < ef || dj < radius) T.insert(dj, rl_u(n, j), true);
                    }
                    WALK_STAMP(4);  // verdicts: asked
                    push_lanes(cand, nd, n);
                    done = true;
                }
            } else {
            okmask = verdicts(cand, n, fl);
            if (over_budget) break;  // enough unknown slots listed for one round: the host evaluates them and launches again
            WALK_STAMP(4);  // verdicts
            if constexpr (kTopWave) {
                const uint64_t pass = hop_posted(cand, okmask, nd, n);
                WALK_STAMP(8);  // (inside "pushes, top": who passes; the admitted ones are on their way to the top wave)
                push_lanes(pass, nd, n);
                WALK_STAMP(10);  // (the pushes)
                done = true;
            } else {
                done = hop_batch(cand, okmask, nd, n);
            }
            }
            // the CPU loop as written: one neighbour at a time, in adjacency order, against the moving radius
            for (uint64_t r = done ? 0ull : cand; r; r &= r - 1ull) {
                const uint32_t j = (uint32_t)__builtin_ctzll(r);
                const float dj = rl_f(nd, j);
                if (sz == ef && !(dj < radius)) {  // `top.size() < top_limit || d < radius`
                    if (kFilter && tie_active && dj == radius) redo = true;
                    continue;
                }
                const uint32_t sj = rl_u(n, j);
                push_one(dj, sj);
                if ((okmask >> j) & 1ull) top_insert(dj, sj);
            }
        }
        if (early_slot != kInvalid) {  // where the candidate measured ahead went: the front (it keeps its entry), or nowhere (visited already / beyond the radius)
            const bool here = L < nf && f_s == early_slot;
            if (here) f_c = 1u;
            if (!__builtin_amdgcn_ballot_w64(here)) free_entry_of(early_slot);
        }
        WALK_STAMP(5);  // pushes, top
        dbg_max_next = nf + np > dbg_max_next ? nf + np : dbg_max_next;
        schedule();
        WALK_STAMP(6);  // scheduling
    }
    if constexpr (ask) {  // the last answers: their admitted neighbours belong to the result
        if (!redo && pending()) (void)settle(0u);
        if (unknown_count && L == 0u) unknown_count[9] = (uint32_t)cnt.hops;  // (vs_hnsw_filter_ask_stats)
    }
    if constexpr (kTopWave) {  // `top` comes home: the answer is read from the walker's registers
        tw_sync();
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint32_t p = L * (uint32_t)R + (uint32_t)j;
            top.s[j] = p < sz ? sh.merge[p].y : kInvalid;
        }
    }
    if (L == 0u) lds_flag_store(&sh.stop, 1u);
    if (unknown_count && L == 0u) *unknown_count = ucount;
    if (consulted_out && L == 0u) *consulted_out = consulted;
    if (debug && L == 0u) {
        debug[0] = dbg_max_next;
        debug[1] = (uint32_t)cnt.evals;
        debug[2] = (uint32_t)cnt.hops;
        debug[3] = dbg_early | (dbg_windows << 16);
#ifdef VS_WALK_PROFILE
        for (int i = 0; i < 6; ++i) debug[4 + i] = (uint32_t)(prof[1 + i] >> 4);
        debug[0] = (uint32_t)(prof[7] >> 4);                      // (profile builds: the wait for the entry instead of the largest `next`)
        debug[1] = dbg_waits;                                     // (hops that waited)
        debug[10] = sh.prof_jobs[0] ? sh.prof_jobs[1] / sh.prof_jobs[0] : 0u;  // (helpers: clocks per job part)
#if VS_WALK_PROFILE == 3  // (urgent job parts: clocks until the row / the list / the distances / the report, in the places of pop .. push+top; neighbours per part in `schedule`)
        for (int i = 0; i < 4; ++i) debug[4 + i] = sh.prof_jobs[2] ? sh.prof_jobs[3 + i] / sh.prof_jobs[2] : 0u;
        debug[8] = sh.prof_jobs[2];
        debug[9] = sh.prof_jobs[2] ? sh.prof_jobs[7] * 100u / sh.prof_jobs[2] : 0u;
#endif
#if VS_WALK_PROFILE == 2  // (inside "pushes, top": who passes / the merge / the pushes, in the places of atomics / verdicts / schedule)
        debug[6] = (uint32_t)(prof[8] >> 4);
        debug[7] = (uint32_t)(prof[9] >> 4);
        debug[9] = (uint32_t)(prof[10] >> 4);
#endif
#endif
#ifndef VS_WALK_PROFILE
        debug[10] = dbg_miss;
#endif
        debug[11] = dbg_refill | (dbg_spills << 16);
    }
#ifdef VS_WALK_PROFILE  // (profile builds, posted walks: the phases of a hop in the caller's counter block, shader clocks / 16)
    if constexpr (kFilter && !explore) {
        if (unknown_count && L == 0u) {
            unknown_count[9] = (uint32_t)cnt.hops;
            for (int i = 0; i < 6; ++i) unknown_count[10 + i] = (uint32_t)(prof[1 + i] >> 4);  // pop, entry, atomics, verdicts, pushes, schedule
            unknown_count[6] = (uint32_t)(prof[7] >> 4);  // the wait for the entry
            unknown_count[7] = (uint32_t)(prof[8] >> 4);  // the loop's head: spill / refill
            unknown_count[1] = (uint32_t)(prof[9] >> 4);  // asking walks: answers taken in, the pop decided
        }
    }
#endif
    // leave the bitmap all zero for the next query that gets this workspace
    if constexpr (!Sh::kVisGlobal) {
    } else if (vlog_lost) {
        uint4* b4 = reinterpret_cast<uint4*>(ws.bitmap);  // (the workspace is 256-byte aligned)
        const uint32_t quads = ws.bitmap_words / 4u;
        for (uint32_t i = L; i < quads; i += 64u) b4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t wd = quads * 4u + L; wd < ws.bitmap_words; wd += 64u) ws.bitmap[wd] = 0u;
    } else {
        for (uint32_t i = L; i < vcount; i += 64u) ws.bitmap[ws.vlog[i] >> 5] = 0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PipeOut out;
    out.status = redo ? 1u : explore ? 3u : over_budget ? 2u : 0u;
    out.sz = sz;
    return out;
}

}  // namespace vs
