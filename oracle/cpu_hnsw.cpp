// oracle/cpu_hnsw.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the algorithm that the reference delegates to the
// third-party crate `usearch = "2.22.0"` (reference Cargo.toml:93,
// Cargo.lock:5920-5927; C++17 headers index.hpp / index_dense.hpp /
// index_plugins.hpp, NOT vendored under /root/reference and not available in
// this image).  It is written from the published behaviour of that library
//  and anchored on the reference's own call sites:
//
//   crates/vector-store/src/vs_index/usearch.rs:74-82    IndexOptions mapping
//   crates/vector-store/src/vs_index/usearch.rs:181-185  reserve
//   crates/vector-store/src/vs_index/usearch.rs:191-197  add
//   crates/vector-store/src/vs_index/usearch.rs:199-201  remove
//   crates/vector-store/src/vs_index/usearch.rs:203-222  search
//   crates/vector-store/src/vs_index/usearch.rs:224-248  filtered_search
//   crates/vector-store/src/vs_index/usearch.rs:450-513  metric / scalar mapping
//   crates/vector-store/src/vs_index/usearch.rs:1179-1205 f32_to_b1x8
//   crates/vector-store/src/distance.rs:58-105           Distance::try_from ranges
//   crates/vector-store/src/similarity.rs:12-38          SimilarityScore::from
//
// PARITY STATUS: pinned for metric definitions, packing, range/score rules and
// the exhaustive-regime (ef >= N) known-answer tests of the reference
// (tests/golden/*.json).  UNPINNED for large-graph
// traversal order (tie order, level RNG stream of the real usearch binary):
// the usearch sources/binary cannot be built or run here.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  The product (vector_store_amd/) never links or calls it.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

constexpr uint32_t kInvalidSlot = 0xFFFFFFFFu;
constexpr uint64_t kFreeKey = ~0ull;  // usearch default_free_value<u64>() [UPSTREAM]

enum Metric : int { kCos = 0, kL2sq = 1, kIP = 2, kHamming = 3 };

thread_local std::string g_error;

// ---------------------------------------------------------------------------
// Distances (usearch index_plugins.hpp metric_cos_gt / metric_l2sq_gt /
// metric_ip_gt, with the SimSIMD zero rules and >=0 clamp for cosine
// [UPSTREAM]).  f32 accumulation.
// ---------------------------------------------------------------------------
// AVX2+FMA kernels (8-wide f32, 4 independent accumulators); scalar tail.
#include <immintrin.h>  // F16C conversions (and the AVX2 kernels below)
#if defined(__AVX2__) && defined(__FMA__)
static inline float hsum8(__m256 v) {
    __m128 lo = _mm256_castps256_ps128(v), hi = _mm256_extractf128_ps(v, 1);
    lo = _mm_add_ps(lo, hi);
    lo = _mm_hadd_ps(lo, lo);
    lo = _mm_hadd_ps(lo, lo);
    return _mm_cvtss_f32(lo);
}
static float dist_l2sq(const float* a, const float* b, size_t d) {
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    size_t i = 0;
    for (; i + 32 <= d; i += 32) {
        __m256 t0 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        __m256 t1 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8));
        __m256 t2 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16));
        __m256 t3 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24));
        s0 = _mm256_fmadd_ps(t0, t0, s0);
        s1 = _mm256_fmadd_ps(t1, t1, s1);
        s2 = _mm256_fmadd_ps(t2, t2, s2);
        s3 = _mm256_fmadd_ps(t3, t3, s3);
    }
    for (; i + 8 <= d; i += 8) {
        __m256 t0 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        s0 = _mm256_fmadd_ps(t0, t0, s0);
    }
    float s = hsum8(_mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3)));
    for (; i < d; ++i) {
        float t = a[i] - b[i];
        s += t * t;
    }
    return s;
}
static float dot(const float* a, const float* b, size_t d) {
    __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
    size_t i = 0;
    for (; i + 32 <= d; i += 32) {
        s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24), s3);
    }
    for (; i + 8 <= d; i += 8) s0 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s0);
    float s = hsum8(_mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3)));
    for (; i < d; ++i) s += a[i] * b[i];
    return s;
}
static void dot3(const float* a, const float* b, size_t d, float& ab, float& a2, float& b2) {
    __m256 x0 = _mm256_setzero_ps(), x1 = x0, y0 = x0, y1 = x0, z0 = x0, z1 = x0;
    size_t i = 0;
    for (; i + 16 <= d; i += 16) {
        __m256 a0 = _mm256_loadu_ps(a + i), b0 = _mm256_loadu_ps(b + i);
        __m256 a1 = _mm256_loadu_ps(a + i + 8), b1 = _mm256_loadu_ps(b + i + 8);
        x0 = _mm256_fmadd_ps(a0, b0, x0);
        x1 = _mm256_fmadd_ps(a1, b1, x1);
        y0 = _mm256_fmadd_ps(a0, a0, y0);
        y1 = _mm256_fmadd_ps(a1, a1, y1);
        z0 = _mm256_fmadd_ps(b0, b0, z0);
        z1 = _mm256_fmadd_ps(b1, b1, z1);
    }
    for (; i + 8 <= d; i += 8) {
        __m256 a0 = _mm256_loadu_ps(a + i), b0 = _mm256_loadu_ps(b + i);
        x0 = _mm256_fmadd_ps(a0, b0, x0);
        y0 = _mm256_fmadd_ps(a0, a0, y0);
        z0 = _mm256_fmadd_ps(b0, b0, z0);
    }
    ab = hsum8(_mm256_add_ps(x0, x1));
    a2 = hsum8(_mm256_add_ps(y0, y1));
    b2 = hsum8(_mm256_add_ps(z0, z1));
    for (; i < d; ++i) {
        ab += a[i] * b[i];
        a2 += a[i] * a[i];
        b2 += b[i] * b[i];
    }
}
#else
static float dist_l2sq(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) {
        float t = a[i] - b[i];
        s += t * t;
    }
    return s;
}
static float dot(const float* a, const float* b, size_t d) {
    float s = 0.f;
    for (size_t i = 0; i < d; ++i) s += a[i] * b[i];
    return s;
}
static void dot3(const float* a, const float* b, size_t d, float& ab, float& a2, float& b2) {
    ab = a2 = b2 = 0.f;
    for (size_t i = 0; i < d; ++i) {
        ab += a[i] * b[i];
        a2 += a[i] * a[i];
        b2 += b[i] * b[i];
    }
}
#endif

static float dist_ip(const float* a, const float* b, size_t d) { return 1.0f - dot(a, b, d); }

static float dist_cos(const float* a, const float* b, size_t d) {
    float sab, sa2, sb2;
    dot3(a, b, d, sab, sa2, sb2);
    if (sa2 == 0.f && sb2 == 0.f) return 0.f;  // both zero  -> 0
    if (sa2 == 0.f || sb2 == 0.f) return 1.f;  // exactly one zero -> 1
    if (sab == 0.f) return 1.f;                // SimSIMD: ab == 0 -> 1
    float r = 1.0f - sab / (std::sqrt(sa2) * std::sqrt(sb2));
    return r > 0.f ? r : 0.f;                  // SimSIMD clamps to >= 0
}

static float dist_hamming_b1(const uint8_t* a, const uint8_t* b, size_t bytes) {
    size_t c = 0;
    for (size_t i = 0; i < bytes; ++i) c += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return (float)c;
}

// ---------------------------------------------------------------------------
// Scalar casts applied on add/search when quantization != F32, and the metrics on the
// stored types (usearch index_plugins.hpp cast_gt / metric_*_gt [UPSTREAM, from memory]):
//   f16  : IEEE round-to-nearest-even; bf16 : round-to-nearest-even on the high 16 bits;
//   i8   : trunc(x * 127 / |x|) clamped to [-127, 127] (|x| accumulated in f64);
//   b1   : bit i of byte j = v[8j+i] > 0 (reference usearch.rs:1179-1205).
// f16/bf16 metrics widen every element to f32 and run the f32 formulas; i8 metrics run on
// the integers: cos = 1 - ab/sqrt(a2*b2) (zero rules as above), l2sq = sum (a-b)^2, ip = 1 - ab.
// ---------------------------------------------------------------------------
enum Scalar : int { kF32 = 0, kF16 = 1, kBF16 = 2, kI8 = 3, kB1 = 4 };

static inline uint16_t f32_to_f16(float f) { return (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
static inline float f16_to_f32(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

static size_t bytes_per_vector_of(int scalar, size_t dim) {
    switch (scalar) {
        case kF32: return dim * 4;
        case kF16:
        case kBF16: return dim * 2;
        case kI8: return dim;
        default: return (dim + 7) / 8;
    }
}

static void cast_from_f32(int scalar, const float* v, size_t dim, uint8_t* out) {
    switch (scalar) {
        case kF32: std::memcpy(out, v, dim * 4); break;
        case kF16:
            for (size_t i = 0; i < dim; ++i) ((uint16_t*)out)[i] = f32_to_f16(v[i]);
            break;
        case kBF16:
            for (size_t i = 0; i < dim; ++i) ((uint16_t*)out)[i] = f32_to_bf16(v[i]);
            break;
        case kI8: {
            double s = 0.0;
            for (size_t i = 0; i < dim; ++i) s += (double)v[i] * (double)v[i];
            float mag = (float)std::sqrt(s);
            for (size_t i = 0; i < dim; ++i) {
                float t = mag > 0.f ? (v[i] * 127.0f) / mag : 0.f;
                t = std::min(std::max(t, -127.f), 127.f);
                ((int8_t*)out)[i] = (int8_t)(int)t;
            }
        } break;
        default: {
            size_t nb = (dim + 7) / 8;
            for (size_t j = 0; j < nb; ++j) {
                uint8_t byte = 0;
                for (size_t i = 0; i < 8 && j * 8 + i < dim; ++i)
                    if (v[j * 8 + i] > 0.0f) byte |= (uint8_t)(1u << i);
                out[j] = byte;
            }
        }
    }
}

static void widen(int scalar, const uint16_t* src, size_t d, float* dst) {
    size_t i = 0;
#if defined(__AVX2__) && defined(__F16C__)
    if (scalar == kF16) {
        for (; i + 8 <= d; i += 8) _mm256_storeu_ps(dst + i, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(src + i))));
    } else {
        for (; i + 8 <= d; i += 8) {
            __m256i w = _mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(src + i)));
            _mm256_storeu_ps(dst + i, _mm256_castsi256_ps(_mm256_slli_epi32(w, 16)));
        }
    }
#endif
    for (; i < d; ++i) dst[i] = scalar == kF16 ? f16_to_f32(src[i]) : bf16_to_f32(src[i]);
}

// f16 / bf16: every element widened to f32, then the f32 formulas (products of two half values are exact in f32).
static float dist_widened(int metric, int scalar, const uint16_t* a, const uint16_t* b, size_t d) {
    thread_local std::vector<float> fa, fb;
    fa.resize(d);
    fb.resize(d);
    widen(scalar, a, d, fa.data());
    widen(scalar, b, d, fb.data());
    if (metric == kL2sq) return dist_l2sq(fa.data(), fb.data(), d);
    if (metric == kIP) return dist_ip(fa.data(), fb.data(), d);
    return dist_cos(fa.data(), fb.data(), d);
}

static float dist_i8(int metric, const int8_t* a, const int8_t* b, size_t d) {
    int32_t ab = 0, a2 = 0, b2 = 0;
    for (size_t i = 0; i < d; ++i) {
        ab += (int32_t)a[i] * b[i];
        a2 += (int32_t)a[i] * a[i];
        b2 += (int32_t)b[i] * b[i];
    }
    if (metric == kL2sq) return (float)a2 + (float)b2 - 2.0f * (float)ab;
    if (metric == kIP) return 1.0f - (float)ab;
    if (a2 == 0 && b2 == 0) return 0.f;
    if (a2 == 0 || b2 == 0 || ab == 0) return 1.f;
    float r = 1.0f - (float)ab / (std::sqrt((float)a2) * std::sqrt((float)b2));
    return r > 0.f ? r : 0.f;
}

// ---------------------------------------------------------------------------
// Per-thread search context (usearch index.hpp context_t [UPSTREAM]).
// ---------------------------------------------------------------------------
struct Cand {
    float d;
    uint32_t slot;
};

// sorted_buffer_gt: ascending by distance, bounded insert.
struct TopBuffer {
    std::vector<Cand> v;
    void clear() { v.clear(); }
    size_t size() const { return v.size(); }
    const Cand& worst() const { return v.back(); }
    void insert(Cand c, size_t limit) {
        // lower_bound on distance: a new element goes in front of equal ones.
        auto it = std::lower_bound(v.begin(), v.end(), c, [](const Cand& x, const Cand& y) { return x.d < y.d; });
        if (v.size() < limit) {
            v.insert(it, c);
        } else if (it != v.end()) {
            v.insert(it, c);
            v.pop_back();
        }
    }
};

// usearch max_heap_gt<candidate_t> holding NEGATED distances == a min-heap on distance [UPSTREAM, from memory].
// usearch does not use std::push_heap / std::pop_heap: it carries its own array heap
//   emplace: append, shift_up(last)        shift_up(i):   while i && less(parent, i): swap, i = parent
//   pop:     swap(first, last), shrink,    shift_down(i): child = right iff less(left, right) else left;
//            shift_down(0)                                swap and descend iff less(i, child), else stop
// with less(a, b) = (-a.d) < (-b.d) = a.d > b.d.  Which of several EQUAL distances is popped first is decided by this
// exact sequence of swaps (the order is not a function of (distance, slot)), so the GPU engine emulates the same
// array heap step for step (vector_store_amd/csrc/walk_device.hpp) and tie-heavy metrics (Hamming, i8) return
// identical ids.  (Round 1 used std::push_heap / pop_heap here, libstdc++'s bottom-up variant: same set of results on
// tie-free data, a different order among equal distances.)
struct NextHeap {
    std::vector<Cand> v;
    static bool less(const Cand& a, const Cand& b) { return a.d > b.d; }
    void clear() { v.clear(); }
    bool empty() const { return v.empty(); }
    const Cand& top() const { return v.front(); }
    void push(Cand c) {
        v.push_back(c);
        size_t i = v.size() - 1;
        for (; i && less(v[(i - 1) / 2], v[i]); i = (i - 1) / 2) std::swap(v[(i - 1) / 2], v[i]);
    }
    void pop() {
        std::swap(v.front(), v.back());
        v.pop_back();
        const size_t n = v.size();
        for (size_t i = 0; 2 * i + 1 < n;) {
            const size_t l = 2 * i + 1, r = 2 * i + 2;
            const size_t c = (r < n && less(v[l], v[r])) ? r : l;
            if (!less(v[i], v[c])) break;
            std::swap(v[i], v[c]);
            i = c;
        }
    }
};

// growing_hash_set_gt<slot>: open addressing, power-of-two capacity.
struct VisitSet {
    std::vector<uint32_t> t;
    size_t n = 0, mask = 0;
    void clear() {
        if (t.empty()) {
            t.assign(1024, kInvalidSlot);
            mask = 1023;
        } else if (n) {
            std::fill(t.begin(), t.end(), kInvalidSlot);
        }
        n = 0;
    }
    static uint32_t h(uint32_t x) {
        x ^= x >> 16;
        x *= 0x7feb352dU;
        x ^= x >> 15;
        x *= 0x846ca68bU;
        x ^= x >> 16;
        return x;
    }
    void grow() {
        std::vector<uint32_t> old;
        old.swap(t);
        t.assign(old.size() * 2, kInvalidSlot);
        mask = t.size() - 1;
        n = 0;
        for (uint32_t s : old)
            if (s != kInvalidSlot) set(s);
    }
    // returns true when the slot was already present (usearch visits.set()).
    bool set(uint32_t s) {
        if ((n + 1) * 2 > t.size()) grow();
        size_t i = h(s) & mask;
        while (t[i] != kInvalidSlot) {
            if (t[i] == s) return true;
            i = (i + 1) & mask;
        }
        t[i] = s;
        ++n;
        return false;
    }
    bool test(uint32_t s) const {
        if (t.empty()) return false;
        size_t i = h(s) & mask;
        while (t[i] != kInvalidSlot) {
            if (t[i] == s) return true;
            i = (i + 1) & mask;
        }
        return false;
    }
};

struct Context {
    TopBuffer top;
    NextHeap next;
    VisitSet visits;
    std::default_random_engine level_generator;  // default seed, as usearch's context_t
    uint64_t computed_distances = 0;
    uint64_t iteration_cycles = 0;  // node expansions
};

typedef int (*pred_fn)(uint64_t key, void* ctx);

struct Index {
    // config (reference usearch.rs:74-82; zero => usearch defaults)
    size_t dim = 0, bytes_per_vector = 0;
    int metric = kCos;
    int scalar = kF32;
    size_t M = 16, M0 = 32, ef_add = 128, ef_search = 64;
    double inverse_log_connectivity = 0;
    bool b1 = false;

    // storage
    size_t capacity = 0;
    std::atomic<size_t> nodes_count{0};
    size_t live = 0;  // members with a key (size())
    std::vector<uint8_t> vectors;
    std::vector<uint64_t> keys;
    std::vector<int16_t> levels;
    std::vector<uint32_t> adj0;                 // capacity * (1 + M0): count then slots
    std::vector<std::unique_ptr<uint32_t[]>> upper;  // level * (1 + M) words per node
    std::unique_ptr<std::atomic<uint8_t>[]> locks;

    std::mutex global_mutex;
    int32_t max_level = -1;
    uint32_t entry_slot = 0;

    std::mutex lookup_mutex;
    std::unordered_map<uint64_t, uint32_t> slot_lookup;
    std::deque<uint32_t> free_slots;  // ring_gt: FIFO

    std::vector<std::unique_ptr<Context>> contexts;
    std::mutex ctx_mutex;
    std::atomic<uint64_t> batch_distances{0}, batch_cycles{0};  // from per-call contexts of orc_search_batch

    Context& ctx(size_t thread) {
        std::lock_guard<std::mutex> g(ctx_mutex);
        if (thread >= contexts.size()) contexts.resize(thread + 1);
        if (!contexts[thread]) contexts[thread].reset(new Context());
        return *contexts[thread];
    }

    const uint8_t* vec(uint32_t s) const { return vectors.data() + (size_t)s * bytes_per_vector; }

    float measure(const void* a, const void* b, Context& c) const {
        ++c.computed_distances;
        switch (scalar) {
            case kF16:
            case kBF16: return dist_widened(metric, scalar, (const uint16_t*)a, (const uint16_t*)b, dim);
            case kI8: return dist_i8(metric, (const int8_t*)a, (const int8_t*)b, dim);
            case kB1: return dist_hamming_b1((const uint8_t*)a, (const uint8_t*)b, bytes_per_vector);
            default: break;
        }
        switch (metric) {
            case kCos: return dist_cos((const float*)a, (const float*)b, dim);
            case kL2sq: return dist_l2sq((const float*)a, (const float*)b, dim);
            default: return dist_ip((const float*)a, (const float*)b, dim);
        }
    }

    uint32_t* nbrs(uint32_t s, int level) {
        if (level == 0) return adj0.data() + (size_t)s * (1 + M0);
        return upper[s].get() + (size_t)(level - 1) * (1 + M);
    }

    struct NodeLock {
        std::atomic<uint8_t>* f;
        explicit NodeLock(std::atomic<uint8_t>* p) : f(p) {
            uint8_t e = 0;
            while (!f->compare_exchange_weak(e, 1, std::memory_order_acquire)) {
                e = 0;
                std::this_thread::yield();
            }
        }
        ~NodeLock() { f->store(0, std::memory_order_release); }
    };
    NodeLock lock(uint32_t s) { return NodeLock(&locks[s]); }

    // --- usearch index_gt::choose_random_level_ [UPSTREAM] ---
    int16_t choose_level(Context& c) const {
        std::uniform_real_distribution<double> distribution(0.0, 1.0);
        double r = -std::log(distribution(c.level_generator)) * inverse_log_connectivity;
        return (int16_t)r;
    }

    // --- usearch index_gt::search_for_one_ [UPSTREAM] ---
    uint32_t search_for_one(const void* q, uint32_t closest, int32_t begin_level, int32_t end_level, Context& c) {
        float closest_dist = measure(q, vec(closest), c);
        for (int32_t level = begin_level; level > end_level; --level) {
            bool changed;
            do {
                changed = false;
                auto g = lock(closest);
                const uint32_t* nb = nbrs(closest, level);
                uint32_t cnt = nb[0];
                ++c.iteration_cycles;
                for (uint32_t i = 0; i < cnt; ++i) {
                    uint32_t cand = nb[1 + i];
                    float d = measure(q, vec(cand), c);
                    if (d < closest_dist) {
                        closest_dist = d;
                        closest = cand;
                        changed = true;
                    }
                }
            } while (changed);
        }
        return closest;
    }

    // --- usearch index_gt::search_to_insert_ / search_to_update_ [UPSTREAM] ---
    // `self` = slot being (re)inserted: never expanded, never a result.
    void search_to_insert(const void* q, uint32_t start, uint32_t self, int level, size_t top_limit, Context& c) {
        c.visits.clear();
        c.next.clear();
        c.top.clear();
        float radius = measure(q, vec(start), c);
        c.next.push({radius, start});
        if (start != self) c.top.insert({radius, start}, top_limit);
        c.visits.set(start);
        c.visits.set(self);
        while (!c.next.empty()) {
            Cand cand = c.next.top();
            if (cand.d > radius && c.top.size() == top_limit) break;
            c.next.pop();
            ++c.iteration_cycles;
            if (cand.slot == self) continue;
            auto g = lock(cand.slot);
            const uint32_t* nb = nbrs(cand.slot, level);
            uint32_t cnt = nb[0];
            for (uint32_t i = 0; i < cnt; ++i) {
                uint32_t succ = nb[1 + i];
                if (c.visits.set(succ)) continue;
                float d = measure(q, vec(succ), c);
                if (c.top.size() < top_limit || d < radius) {
                    c.next.push({d, succ});
                    c.top.insert({d, succ}, top_limit);
                    radius = c.top.worst().d;
                }
            }
        }
    }

    // --- usearch index_gt::search_to_find_in_base_ [UPSTREAM] ---
    // Predicate gates admission to `top` only; rejected nodes are still expanded.
    // The loop stops when the best candidate is farther than the worst result
    // AND `top` is full (so fewer-than-k filtered matches are all found, as the
    // reference tests require: tests/integration/vs_index.rs:1119-1158).
    void search_to_find_in_base(const void* q, uint32_t start, size_t top_limit, pred_fn pred, void* pctx, Context& c) {
        c.visits.clear();
        c.next.clear();
        c.top.clear();
        auto allowed = [&](uint32_t s) {
            uint64_t k = keys[s];
            if (k == kFreeKey) return false;  // index_dense: removed entries are never results
            return pred ? pred(k, pctx) != 0 : true;
        };
        float radius = measure(q, vec(start), c);
        c.next.push({radius, start});
        c.visits.set(start);
        if (allowed(start)) c.top.insert({radius, start}, top_limit);
        while (!c.next.empty()) {
            Cand cand = c.next.top();
            if (cand.d > radius && c.top.size() == top_limit) break;
            c.next.pop();
            ++c.iteration_cycles;
            const uint32_t* nb = nbrs(cand.slot, 0);
            uint32_t cnt = nb[0];
            for (uint32_t i = 0; i < cnt; ++i) {
                uint32_t succ = nb[1 + i];
                if (c.visits.set(succ)) continue;
                float d = measure(q, vec(succ), c);
                if (c.top.size() < top_limit || d < radius) {
                    c.next.push({d, succ});
                    if (allowed(succ)) c.top.insert({d, succ}, top_limit);
                    if (c.top.size()) radius = c.top.worst().d;
                }
            }
        }
    }

    // --- usearch index_gt::refine_ (neighbour-selection heuristic) [UPSTREAM] ---
    // `top` ascending. Returns number kept (in place, prefix of top.v).
    size_t refine(size_t needed, Context& c) {
        auto& t = c.top.v;
        size_t top_count = t.size();
        if (top_count < needed) return top_count;
        size_t submitted = 1, consumed = 1;
        while (submitted < needed && consumed < top_count) {
            Cand cand = t[consumed];
            bool good = true;
            for (size_t i = 0; i < submitted; ++i) {
                float inter = measure(vec(cand.slot), vec(t[i].slot), c);
                if (inter < cand.d) {
                    good = false;
                    break;
                }
            }
            if (good) t[submitted++] = t[consumed];
            ++consumed;
        }
        t.resize(submitted);
        return submitted;
    }

    // --- usearch index_gt::connect_new_node_ [UPSTREAM]: forward links, <= M on every level ---
    uint32_t connect_new_node(uint32_t new_slot, int level, Context& c) {
        uint32_t* nb = nbrs(new_slot, level);
        size_t kept = refine(M, c);
        nb[0] = 0;
        for (size_t i = 0; i < kept; ++i) nb[1 + nb[0]++] = c.top.v[i].slot;
        return kept ? nb[1] : kInvalidSlot;
    }

    // --- usearch index_gt::reconnect_neighbor_nodes_ [UPSTREAM]: reverse links ---
    void reconnect_neighbor_nodes(uint32_t new_slot, const void* value, int level, Context& c) {
        const size_t connectivity_max = level ? M : M0;
        uint32_t* mine = nbrs(new_slot, level);
        uint32_t mine_cnt = mine[0];
        std::vector<uint32_t> my(mine + 1, mine + 1 + mine_cnt);
        for (uint32_t close_slot : my) {
            if (close_slot == new_slot) continue;
            auto g = lock(close_slot);
            uint32_t* ch = nbrs(close_slot, level);
            if (ch[0] < connectivity_max) {
                ch[1 + ch[0]++] = new_slot;
                continue;
            }
            c.top.clear();
            c.top.insert({measure(value, vec(close_slot), c), new_slot}, connectivity_max + 1);
            for (uint32_t i = 0; i < ch[0]; ++i) {
                uint32_t succ = ch[1 + i];
                c.top.insert({measure(vec(close_slot), vec(succ), c), succ}, connectivity_max + 1);
            }
            size_t kept = refine(connectivity_max, c);
            ch[0] = 0;
            for (size_t i = 0; i < kept; ++i) ch[1 + ch[0]++] = c.top.v[i].slot;
        }
    }

    // --- usearch index_dense_gt::add_ -> index_gt::add / update [UPSTREAM] ---
    // `value_f32`: dim floats, cast to the storage type here (usearch casts inside add()).
    int add(uint64_t key, const void* value_f32, size_t thread, int forced_level) {
        Context& c = ctx(thread);
        std::vector<uint8_t> casted(bytes_per_vector);
        cast_from_f32(scalar, (const float*)value_f32, dim, casted.data());
        const void* value = casted.data();
        if (key == kFreeKey) return fail("Key is reserved for internal use");
        uint32_t free_slot = kInvalidSlot;
        {
            std::lock_guard<std::mutex> g(lookup_mutex);
            if (slot_lookup.count(key)) return fail("Duplicate keys not allowed in high-level wrappers");
            if (!free_slots.empty()) {
                free_slot = free_slots.front();
                free_slots.pop_front();
            }
        }
        bool reuse = free_slot != kInvalidSlot;
        uint32_t slot;
        int32_t level;
        int32_t max_level_copy;
        uint32_t entry_copy;
        std::unique_lock<std::mutex> new_level_lock(global_mutex);
        max_level_copy = max_level;
        entry_copy = entry_slot;
        if (reuse) {
            new_level_lock.unlock();
            slot = free_slot;
            level = levels[slot];
            auto g = lock(slot);
            std::memset(nbrs(slot, 0), 0, sizeof(uint32_t) * (1 + M0));
            if (level) std::memset(upper[slot].get(), 0, sizeof(uint32_t) * (size_t)level * (1 + M));
        } else {
            level = forced_level >= 0 ? forced_level : choose_level(c);
            size_t cur = nodes_count.load();
            if (cur >= capacity) return fail("Reserve capacity ahead of insertions!");
            slot = (uint32_t)nodes_count.fetch_add(1);
            if (level <= max_level_copy) new_level_lock.unlock();
            levels[slot] = (int16_t)level;
            std::memset(nbrs(slot, 0), 0, sizeof(uint32_t) * (1 + M0));
            if (level) {
                upper[slot].reset(new uint32_t[(size_t)level * (1 + M)]);
                std::memset(upper[slot].get(), 0, sizeof(uint32_t) * (size_t)level * (1 + M));
            }
        }
        std::memcpy(vectors.data() + (size_t)slot * bytes_per_vector, value, bytes_per_vector);
        const void* v = vec(slot);

        if (!reuse && slot == 0) {  // first element becomes the entry point
            entry_slot = 0;
            max_level = level;
        } else if (max_level_copy >= 0) {
            uint32_t closest = search_for_one(v, entry_copy, max_level_copy, level, c);
            for (int32_t l = std::min(level, max_level_copy); l >= 0; --l) {
                search_to_insert(v, closest, slot, l, ef_add, c);
                uint32_t first;
                {
                    auto g = lock(slot);
                    first = connect_new_node(slot, l, c);
                }
                if (first != kInvalidSlot) closest = first;
                reconnect_neighbor_nodes(slot, v, l, c);
            }
            if (!reuse && level > max_level_copy) {
                entry_slot = slot;
                max_level = level;
            }
        }
        {
            std::lock_guard<std::mutex> g(lookup_mutex);
            keys[slot] = key;
            slot_lookup.emplace(key, slot);
            ++live;
        }
        return 0;
    }

    // --- usearch index_dense_gt::remove [UPSTREAM] ---
    int remove(uint64_t key) {
        std::lock_guard<std::mutex> g(lookup_mutex);
        auto it = slot_lookup.find(key);
        if (it == slot_lookup.end()) return 0;
        uint32_t slot = it->second;
        slot_lookup.erase(it);
        keys[slot] = kFreeKey;
        free_slots.push_back(slot);
        --live;
        return 1;
    }

    // --- usearch index_gt::search [UPSTREAM] ---
    size_t search(const void* q, size_t wanted, pred_fn pred, void* pctx, uint64_t* out_keys, float* out_d,
                  size_t thread, uint32_t* out_slots) {
        return search(q, wanted, pred, pctx, out_keys, out_d, ctx(thread), out_slots);
    }
    size_t search(const void* q_f32, size_t wanted, pred_fn pred, void* pctx, uint64_t* out_keys, float* out_d,
                  Context& c, uint32_t* out_slots) {
        std::vector<uint8_t> casted;
        const void* q = q_f32;
        if (scalar != kF32) {
            casted.resize(bytes_per_vector);
            cast_from_f32(scalar, (const float*)q_f32, dim, casted.data());
            q = casted.data();
        }
        if (nodes_count.load() == 0 || max_level < 0) return 0;
        size_t expansion = std::max(ef_search, wanted);
        uint32_t closest = search_for_one(q, entry_slot, max_level, 0, c);
        search_to_find_in_base(q, closest, expansion, pred, pctx, c);
        size_t n = std::min(wanted, c.top.size());
        for (size_t i = 0; i < n; ++i) {
            out_keys[i] = keys[c.top.v[i].slot];
            out_d[i] = c.top.v[i].d;
            if (out_slots) out_slots[i] = c.top.v[i].slot;
        }
        return n;
    }

    int reserve(size_t cap) {
        if (cap < nodes_count.load()) return fail("Can't reserve less than the current size");
        if (cap > 0xFFFFFFF0ull) return fail("capacity exceeds 32-bit slots");
        vectors.resize(cap * bytes_per_vector);
        keys.resize(cap, kFreeKey);
        levels.resize(cap, 0);
        adj0.resize(cap * (1 + M0), 0);
        upper.resize(cap);
        std::unique_ptr<std::atomic<uint8_t>[]> nl(new std::atomic<uint8_t>[cap ? cap : 1]);
        for (size_t i = 0; i < cap; ++i) nl[i].store(0);
        locks.swap(nl);
        capacity = cap;
        return 0;
    }

    static int fail(const char* m) {
        g_error = m;
        return -1;
    }
};

}  // namespace

extern "C" {

const char* orc_last_error() { return g_error.c_str(); }

// metric: 0 cos, 1 l2sq, 2 ip, 3 hamming; scalar: 0 f32, 1 f16, 2 bf16, 3 i8, 4 b1.
// Vectors enter as f32 (add / search) and are cast to the storage type, as usearch does; b1 forces
// hamming (reference usearch.rs:450-457).  Raw storage rows: orc_add_raw / orc_vectors / import / export.
void* orc_create_ex(size_t dim, int metric, int scalar, size_t connectivity, size_t expansion_add, size_t expansion_search) {
    if (!dim) {
        g_error = "dimensions must be > 0";
        return nullptr;
    }
    Index* ix = new Index();
    ix->dim = dim;
    ix->scalar = (metric == kHamming) ? (int)kB1 : scalar;
    ix->metric = ix->scalar == kB1 ? (int)kHamming : metric;
    ix->b1 = ix->scalar == kB1;
    ix->bytes_per_vector = bytes_per_vector_of(ix->scalar, dim);
    ix->M = connectivity ? connectivity : 16;       // usearch default_connectivity()
    ix->M0 = ix->M * 2;                             // connectivity_base
    ix->ef_add = expansion_add ? expansion_add : 128;       // default_expansion_add()
    ix->ef_search = expansion_search ? expansion_search : 64;  // default_expansion_search()
    ix->inverse_log_connectivity = 1.0 / std::log((double)ix->M);
    return ix;
}
void* orc_create(size_t dim, int metric, size_t connectivity, size_t expansion_add, size_t expansion_search) {
    return orc_create_ex(dim, metric, kF32, connectivity, expansion_add, expansion_search);
}
size_t orc_bytes_per_vector(void* h) { return ((Index*)h)->bytes_per_vector; }
void orc_free(void* h) { delete (Index*)h; }
int orc_reserve(void* h, size_t cap) { return ((Index*)h)->reserve(cap); }
size_t orc_capacity(void* h) { return ((Index*)h)->capacity; }
size_t orc_size(void* h) { return ((Index*)h)->live; }
size_t orc_slots(void* h) { return ((Index*)h)->nodes_count.load(); }
int orc_max_level(void* h) { return ((Index*)h)->max_level; }
uint32_t orc_entry_slot(void* h) { return ((Index*)h)->entry_slot; }
void orc_set_expansion_search(void* h, size_t ef) { ((Index*)h)->ef_search = ef; }

int orc_add(void* h, uint64_t key, const void* v, size_t thread) { return ((Index*)h)->add(key, v, thread, -1); }
int orc_add_with_level(void* h, uint64_t key, const void* v, int level) { return ((Index*)h)->add(key, v, 0, level); }
int orc_remove(void* h, uint64_t key) { return ((Index*)h)->remove(key); }

int orc_search(void* h, const void* q, size_t k, uint64_t* keys, float* d, size_t* found) {
    *found = ((Index*)h)->search(q, k, nullptr, nullptr, keys, d, 0, nullptr);
    return 0;
}
int orc_search_slots(void* h, const void* q, size_t k, uint64_t* keys, float* d, uint32_t* slots, size_t* found) {
    *found = ((Index*)h)->search(q, k, nullptr, nullptr, keys, d, 0, slots);
    return 0;
}
int orc_filtered_search(void* h, const void* q, size_t k, pred_fn pred, void* pctx, uint64_t* keys, float* d,
                        size_t* found) {
    *found = ((Index*)h)->search(q, k, pred, pctx, keys, d, 0, nullptr);
    return 0;
}

// Threaded drivers: one vector / one query per call from T threads, mirroring
// ThreadedUsearchIndex::{add,search} under the worker pool (reference worker.rs:44-118).
int orc_add_batch(void* h, const uint64_t* keys, const void* vecs, size_t n, size_t threads) {
    Index* ix = (Index*)h;
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    std::atomic<int> errors{0};
    size_t first = 0;
    // The very first member is inserted alone (it becomes the entry point).
    if (ix->nodes_count.load() == 0 && n) {
        if (ix->add(keys[0], vecs, 0, -1)) return -1;
        first = 1;
    }
    next = first;
    auto work = [&](size_t t) {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n) break;
            if (ix->add(keys[i], (const float*)vecs + i * ix->dim, t, -1)) errors.fetch_add(1);
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return errors.load() ? -1 : 0;
}

int orc_search_batch(void* h, const void* Q, size_t nq, size_t k, uint64_t* keys, float* d, size_t* found,
                     size_t threads) {
    Index* ix = (Index*)h;
    if (threads < 1) threads = 1;
    std::atomic<size_t> next{0};
    // Every call owns its contexts: concurrent batch calls (search || search) never share scratch state.
    std::vector<Context> local(threads);
    auto work = [&](size_t t) {
        Context& c = local[t];
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nq) break;
            found[i] = ix->search((const float*)Q + i * ix->dim, k, nullptr, nullptr, keys + i * k,
                                  d + i * k, c, nullptr);
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (auto& c : local) {
        ix->batch_distances += c.computed_distances;
        ix->batch_cycles += c.iteration_cycles;
    }
    return 0;
}

// The same driver for filtered_search (reference filtered_ann, usearch.rs:1107-1154: every filtered query runs on a blocking
// thread, :937-948): predicate key % modulus == 0 -- what libvs_callers' filtered run asks the GPU engine -- for `seconds`
// of wall time or until `nq` queries are answered, whichever comes first.  out3: [0] queries answered, [1] predicate calls,
// [2] nanoseconds of wall time.
int orc_filtered_search_timed(void* h, const void* Q, size_t nq, size_t k, uint64_t modulus, uint64_t* keys, float* d,
                              size_t* found, size_t threads, double seconds, uint64_t* out3) {
    Index* ix = (Index*)h;
    if (threads < 1) threads = 1;
    if (!modulus) return -1;
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> calls{0}, done{0};
    std::atomic<bool> stop{false};
    std::vector<Context> local(threads);
    struct Local {
        uint64_t modulus;
        uint64_t calls;
    };
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&](size_t t) {
        Context& c = local[t];
        Local l{modulus, 0};
        auto pred = [](uint64_t key, void* p) -> int {
            Local* x = (Local*)p;
            ++x->calls;
            return key % x->modulus == 0 ? 1 : 0;
        };
        while (!stop.load(std::memory_order_relaxed)) {
            size_t i = next.fetch_add(1);
            if (i >= nq) break;
            found[i] = ix->search((const float*)Q + i * ix->dim, k, pred, &l, keys + i * k, d + i * k, c, nullptr);
            done.fetch_add(1);
            if (t == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) stop = true;
        }
        calls += l.calls;
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    stop = true;
    for (auto& x : th) x.join();
    for (auto& c : local) {
        ix->batch_distances += c.computed_distances;
        ix->batch_cycles += c.iteration_cycles;
    }
    out3[0] = done.load();
    out3[1] = calls.load();
    out3[2] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// ---- `trait UsearchIndex` (reference usearch.rs:142-160) over the oracle, with the C signatures of include/vs_actor.h's
// vs_actor_index_vtable: lets bench.py's cpu_baseline leg and the tests run the SAME dispatch actor (libvs_actor) and the same
// mixed add / search driver over the CPU restatement.  Every calling thread gets a context of its own, as usearch's
// thread-slot contexts (usearch.rs:184: reserve_capacity_and_threads(capacity, num_workers())).
struct orc_trait_options {  // = vs_hnsw_options (include/vs_hnsw.h)
    size_t dimensions, connectivity, expansion_add, expansion_search;
    int metric, quantization, device, reserved;
};
static size_t trait_thread() {
    static std::atomic<size_t> next{0};
    thread_local size_t mine = next.fetch_add(1) % 1024;
    return mine;
}
int orc_trait_create(const orc_trait_options* o, void** out) {
    if (!o || !out) return -1;
    void* h = orc_create_ex(o->dimensions, o->metric, o->quantization, o->connectivity, o->expansion_add, o->expansion_search);
    if (!h) return -1;
    *out = h;
    return 0;
}
void orc_trait_stop(void* h) { orc_free(h); }
int orc_trait_reserve(void* h, size_t cap, size_t /*threads*/) { return orc_reserve(h, cap) ? -5 : 0; }
size_t orc_trait_capacity(void* h) { return orc_capacity(h); }
int orc_trait_add(void* h, uint64_t key, const float* v, size_t dim) {
    Index* ix = (Index*)h;
    if (dim != ix->dim) return -2;
    return ix->add(key, v, trait_thread(), -1) ? -3 : 0;
}
int orc_trait_remove(void* h, uint64_t key, int* removed) {
    const int r = ((Index*)h)->remove(key);
    if (removed) *removed = r;
    return 0;
}
int orc_trait_search(void* h, const float* q, size_t dim, size_t k, uint64_t* keys, float* d, size_t* found) {
    Index* ix = (Index*)h;
    if (dim != ix->dim) return -2;
    thread_local Context c;
    *found = ix->search(q, k, nullptr, nullptr, keys, d, c, nullptr);
    return 0;
}
int orc_trait_filtered_search(void* h, const float* q, size_t dim, size_t k, pred_fn pred, void* pctx, uint64_t* keys, float* d, size_t* found) {
    Index* ix = (Index*)h;
    if (dim != ix->dim) return -2;
    thread_local Context c;
    *found = ix->search(q, k, pred, pctx, keys, d, c, nullptr);
    return 0;
}
// the table itself (nine function pointers, the layout of vs_actor_index_vtable)
void orc_trait_vtable(void** out9) {
    out9[0] = (void*)orc_trait_create;
    out9[1] = (void*)orc_trait_stop;
    out9[2] = (void*)orc_trait_reserve;
    out9[3] = (void*)orc_trait_capacity;
    out9[4] = (void*)orc_trait_add;
    out9[5] = (void*)orc_trait_remove;
    out9[6] = (void*)orc_trait_search;
    out9[7] = (void*)orc_trait_filtered_search;
    out9[8] = (void*)orc_last_error;
}

// stats summed over all thread contexts: [0] computed distances, [1] node expansions
void orc_stats(void* h, uint64_t* out2, int reset) {
    Index* ix = (Index*)h;
    out2[0] = ix->batch_distances.load();
    out2[1] = ix->batch_cycles.load();
    if (reset) ix->batch_distances = ix->batch_cycles = 0;
    for (auto& c : ix->contexts)
        if (c) {
            out2[0] += c->computed_distances;
            out2[1] += c->iteration_cycles;
            if (reset) c->computed_distances = c->iteration_cycles = 0;
        }
}

// Exact brute-force top-k (usearch `exact` search): ascending (distance, slot).
int orc_exact_search(void* h, const void* q, size_t k, uint64_t* keys, float* d, size_t* found) {
    Index* ix = (Index*)h;
    Context& c = ix->ctx(0);
    size_t n = ix->nodes_count.load();
    std::vector<uint8_t> casted(ix->bytes_per_vector);
    cast_from_f32(ix->scalar, (const float*)q, ix->dim, casted.data());
    q = casted.data();
    std::vector<Cand> all;
    all.reserve(n);
    for (uint32_t s = 0; s < n; ++s)
        if (ix->keys[s] != kFreeKey) all.push_back({ix->measure(q, ix->vec(s), c), s});
    size_t m = std::min(k, all.size());
    std::partial_sort(all.begin(), all.begin() + m, all.end(), [](const Cand& a, const Cand& b) {
        return a.d < b.d || (a.d == b.d && a.slot < b.slot);
    });
    for (size_t i = 0; i < m; ++i) {
        keys[i] = ix->keys[all[i].slot];
        d[i] = all[i].d;
    }
    *found = m;
    return 0;
}

// The index's own distance from an f32 query (cast as a search casts it) to the stored row of `slot`: what the parity
// checks use to decide whether two ids that swapped places are an f32 near-tie.
float orc_distance_to_slot(void* h, const float* q, uint32_t slot) {
    Index* ix = (Index*)h;
    Context c;
    std::vector<uint8_t> casted(ix->bytes_per_vector);
    cast_from_f32(ix->scalar, q, ix->dim, casted.data());
    return ix->measure(casted.data(), ix->vec(slot), c);
}

// Graph export / import in the flat layout shared with the HIP engine's
// vs_hnsw_export_graph / vs_hnsw_import_graph (include/vs_hnsw.h):
//   levels[n] (i32), keys[n] (u64, ~0 = removed), adj0[n*M0] (0xFFFFFFFF padded),
//   upper_off[n] (u32 index of the node's first upper block, 0xFFFFFFFF = none),
//   upper[blocks*M] (0xFFFFFFFF padded; block b of node s, level l = upper_off[s] + l - 1).
size_t orc_upper_blocks(void* h) {
    Index* ix = (Index*)h;
    size_t n = ix->nodes_count.load(), b = 0;
    for (size_t s = 0; s < n; ++s) b += (size_t)ix->levels[s];
    return b;
}
int orc_export_graph(void* h, int32_t* levels, uint64_t* keys, uint32_t* adj0, uint32_t* upper_off, uint32_t* upper) {
    Index* ix = (Index*)h;
    size_t n = ix->nodes_count.load(), b = 0;
    for (size_t s = 0; s < n; ++s) {
        levels[s] = ix->levels[s];
        keys[s] = ix->keys[s];
        const uint32_t* nb = ix->nbrs((uint32_t)s, 0);
        for (size_t i = 0; i < ix->M0; ++i) adj0[s * ix->M0 + i] = i < nb[0] ? nb[1 + i] : kInvalidSlot;
        if (ix->levels[s] == 0) {
            upper_off[s] = kInvalidSlot;
            continue;
        }
        upper_off[s] = (uint32_t)b;
        for (int l = 1; l <= ix->levels[s]; ++l, ++b) {
            const uint32_t* ub = ix->nbrs((uint32_t)s, l);
            for (size_t i = 0; i < ix->M; ++i) upper[b * ix->M + i] = i < ub[0] ? ub[1 + i] : kInvalidSlot;
        }
    }
    return 0;
}
int orc_import_graph(void* h, size_t n, const void* vectors, const int32_t* levels, const uint64_t* keys,
                     const uint32_t* adj0, const uint32_t* upper_off, const uint32_t* upper, int32_t max_level,
                     uint32_t entry_slot) {
    Index* ix = (Index*)h;
    if (n > ix->capacity && ix->reserve(n)) return -1;
    // vectors == arena: the caller exported straight into orc_vectors() (bench.py at 10M: no second 30 GB copy)
    if (vectors && vectors != (const void*)ix->vectors.data()) std::memcpy(ix->vectors.data(), vectors, n * ix->bytes_per_vector);
    ix->slot_lookup.clear();
    ix->free_slots.clear();
    ix->live = 0;
    for (size_t s = 0; s < n; ++s) {
        ix->levels[s] = (int16_t)levels[s];
        ix->keys[s] = keys[s];
        if (keys[s] != kFreeKey) {
            ix->slot_lookup.emplace(keys[s], (uint32_t)s);
            ++ix->live;
        } else {
            ix->free_slots.push_back((uint32_t)s);
        }
        uint32_t* nb = ix->nbrs((uint32_t)s, 0);
        nb[0] = 0;
        for (size_t i = 0; i < ix->M0; ++i) {
            uint32_t x = adj0[s * ix->M0 + i];
            if (x != kInvalidSlot) nb[1 + nb[0]++] = x;
        }
        if (levels[s] > 0) {
            ix->upper[s].reset(new uint32_t[(size_t)levels[s] * (1 + ix->M)]);
            for (int l = 1; l <= levels[s]; ++l) {
                uint32_t* ub = ix->nbrs((uint32_t)s, l);
                ub[0] = 0;
                size_t b = (size_t)upper_off[s] + l - 1;
                for (size_t i = 0; i < ix->M; ++i) {
                    uint32_t x = upper[b * ix->M + i];
                    if (x != kInvalidSlot) ub[1 + ub[0]++] = x;
                }
            }
        }
    }
    ix->nodes_count = n;
    ix->max_level = max_level;
    ix->entry_slot = entry_slot;
    return 0;
}
const void* orc_vectors(void* h) { return ((Index*)h)->vectors.data(); }

// Level stream of a single-threaded usearch context (std::default_random_engine,
// default seed): the HIP engine's host scheduler must draw the same levels.
void orc_level_stream(size_t connectivity, size_t n, int32_t* out) {
    std::default_random_engine gen;
    double inv = 1.0 / std::log((double)(connectivity ? connectivity : 16));
    for (size_t i = 0; i < n; ++i) {
        std::uniform_real_distribution<double> distribution(0.0, 1.0);
        out[i] = (int32_t)(int16_t)(-std::log(distribution(gen)) * inv);
    }
}

float orc_distance(int metric, const void* a, const void* b, size_t dim) {
    switch (metric) {
        case kCos: return dist_cos((const float*)a, (const float*)b, dim);
        case kL2sq: return dist_l2sq((const float*)a, (const float*)b, dim);
        case kIP: return dist_ip((const float*)a, (const float*)b, dim);
        default: return dist_hamming_b1((const uint8_t*)a, (const uint8_t*)b, (dim + 7) / 8);
    }
}

// distance between two f32 vectors as an index of the given storage type would see them
float orc_distance_as(int metric, int scalar, const float* a, const float* b, size_t dim) {
    Index ix;
    ix.dim = dim;
    ix.scalar = metric == kHamming ? (int)kB1 : scalar;
    ix.metric = ix.scalar == kB1 ? (int)kHamming : metric;
    ix.bytes_per_vector = bytes_per_vector_of(ix.scalar, dim);
    std::vector<uint8_t> ca(ix.bytes_per_vector), cb(ix.bytes_per_vector);
    cast_from_f32(ix.scalar, a, dim, ca.data());
    cast_from_f32(ix.scalar, b, dim, cb.data());
    Context c;
    return ix.measure(ca.data(), cb.data(), c);
}

// reference vs_index/usearch.rs:1179-1205: bit i of byte j set iff v[8j+i] > 0.0; tail zero padded.
void orc_f32_to_b1x8(const float* v, size_t n, uint8_t* out) {
    size_t nb = (n + 7) / 8;
    for (size_t j = 0; j < nb; ++j) {
        uint8_t byte = 0;
        for (size_t i = 0; i < 8 && j * 8 + i < n; ++i)
            if (v[j * 8 + i] > 0.0f) byte |= (uint8_t)(1u << i);
        out[j] = byte;
    }
}

// reference distance.rs:58-105: returns 1 when Distance::try_from accepts the value.
int orc_distance_valid(float v, int metric, size_t dim) {
    switch (metric) {
        case kCos: return v >= 0.0f && v <= 2.0f;
        case kL2sq: return v >= 0.0f;  // NaN fails, +inf passes
        case kIP: return !std::isnan(v);
        default:
            return v >= 0.0f && std::isfinite(v) && v == std::trunc(v) && v <= (float)dim;
    }
}

// reference similarity.rs:28-35
float orc_similarity(float d, int metric, size_t dim) {
    switch (metric) {
        case kCos:
        case kIP: return (2.0f - d) / 2.0f;
        case kL2sq: return 1.0f / (1.0f + d);
        default: return 1.0f - d / (float)dim;
    }
}

}  // extern "C"
