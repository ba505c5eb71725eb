// shards.cpp -- libvs_shards.so: one handle over several GPUs in one process (include/vs_shards.h).
#include "../../include/vs_shards.h"
#include "../../include/vs_hnsw_debug.h"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
thread_local std::string g_err;
constexpr uint64_t kIdxMask = (1ull << 48) - 1;
constexpr uint64_t kStripe = 4096;
constexpr uint64_t kFreeKey = ~0ull;

struct Cand {
    float d;
    uint64_t key;
    bool operator<(const Cand& o) const { return d < o.d || (d == o.d && key < o.key); }
};

// n_lists ascending lists (len[i] entries at keys + i*stride) -> the k best, ascending.
size_t merge_lists(size_t n_lists, size_t stride, const uint64_t* keys, const float* dist, const size_t* len, size_t k,
                   uint64_t* out_keys, float* out_dist) {
    std::vector<Cand> all;
    for (size_t i = 0; i < n_lists; ++i)
        for (size_t j = 0; j < len[i]; ++j)
            if (keys[i * stride + j] != kFreeKey) all.push_back({dist[i * stride + j], keys[i * stride + j]});
    size_t m = std::min(k, all.size());
    std::partial_sort(all.begin(), all.begin() + m, all.end());
    for (size_t j = 0; j < m; ++j) {
        out_keys[j] = all[j].key;
        out_dist[j] = all[j].d;
    }
    return m;
}
}  // namespace

struct vs_shards {
    std::vector<vs_hnsw*> shard;
    size_t dim = 0;
    ~vs_shards() {
        for (vs_hnsw* h : shard) vs_hnsw_free(h);
    }
    size_t owner(uint64_t key) const { return (size_t)(((key & kIdxMask) / kStripe) % shard.size()); }
};

static int fail_from_engine(int rc) {
    if (rc != VS_OK) g_err = vs_hnsw_last_error();
    return rc;
}

extern "C" {

const char* vs_shards_last_error(void) { return g_err.c_str(); }

int vs_shards_create(const vs_hnsw_options* o, const int* devices, size_t n, vs_shards** out) {
    if (!o || !devices || !n || !out) {
        g_err = "null / empty argument";
        return VS_ERR_INVALID_ARGUMENT;
    }
    vs_shards* s = new vs_shards();
    s->dim = o->dimensions;
    for (size_t i = 0; i < n; ++i) {
        vs_hnsw_options oi = *o;
        oi.device = devices[i];
        vs_hnsw* h = nullptr;
        int rc = vs_hnsw_create(&oi, &h);
        if (rc != VS_OK) {
            g_err = vs_hnsw_last_error();
            delete s;
            return rc;
        }
        s->shard.push_back(h);
    }
    *out = s;
    return VS_OK;
}
void vs_shards_free(vs_shards* s) { delete s; }
size_t vs_shards_count(const vs_shards* s) { return s ? s->shard.size() : 0; }
size_t vs_shards_owner(const vs_shards* s, uint64_t key) { return s ? s->owner(key) : 0; }

int vs_shards_reserve(vs_shards* s, size_t capacity, size_t threads) {
    if (!s) return VS_ERR_INVALID_ARGUMENT;
    const size_t g = s->shard.size();
    // even split, rounded up to whole key stripes so a dense key space never overflows one shard early
    size_t per = (capacity + g - 1) / g;
    per = (per + kStripe - 1) / kStripe * kStripe;
    for (vs_hnsw* h : s->shard) {
        if (vs_hnsw_capacity(h) >= per) continue;
        int rc = vs_hnsw_reserve(h, per, threads);
        if (rc != VS_OK) return fail_from_engine(rc);
    }
    return VS_OK;
}
size_t vs_shards_capacity(const vs_shards* s) {
    size_t c = 0;
    if (s)
        for (vs_hnsw* h : s->shard) c += vs_hnsw_capacity(h);
    return c;
}
size_t vs_shards_size(const vs_shards* s) {
    size_t c = 0;
    if (s)
        for (vs_hnsw* h : s->shard) c += vs_hnsw_size(h);
    return c;
}

int vs_shards_add(vs_shards* s, uint64_t key, const float* v, size_t dim) {
    if (!s || !v) return VS_ERR_INVALID_ARGUMENT;
    return fail_from_engine(vs_hnsw_add(s->shard[s->owner(key)], key, v, dim));
}

int vs_shards_add_batch(vs_shards* s, const uint64_t* keys, const float* vecs, size_t n, size_t dim) {
    if (!s || (n && (!keys || !vecs))) return VS_ERR_INVALID_ARGUMENT;
    const size_t g = s->shard.size();
    std::vector<std::vector<uint64_t>> pk(g);
    std::vector<std::vector<float>> pv(g);
    for (size_t i = 0; i < n; ++i) {
        size_t o = s->owner(keys[i]);
        pk[o].push_back(keys[i]);
        pv[o].insert(pv[o].end(), vecs + i * dim, vecs + (i + 1) * dim);
    }
    std::vector<int> rc(g, VS_OK);
    std::vector<std::string> err(g);
    std::vector<std::thread> th;
    for (size_t o = 0; o < g; ++o)
        th.emplace_back([&, o] {  // every device builds its own graph concurrently
            if (pk[o].empty()) return;
            rc[o] = vs_hnsw_add_batch(s->shard[o], pk[o].data(), pv[o].data(), pk[o].size(), dim);
            if (rc[o] != VS_OK) err[o] = vs_hnsw_last_error();
        });
    for (auto& t : th) t.join();
    for (size_t o = 0; o < g; ++o)
        if (rc[o] != VS_OK) {
            g_err = err[o];
            return rc[o];
        }
    return VS_OK;
}

int vs_shards_remove(vs_shards* s, uint64_t key, int* removed) {
    if (!s) return VS_ERR_INVALID_ARGUMENT;
    return fail_from_engine(vs_hnsw_remove(s->shard[s->owner(key)], key, removed));
}

namespace {
struct Gather {  // completion of the per-shard async searches of one query
    std::mutex m;
    std::condition_variable c;
    size_t pending;
    int status = VS_OK;
};
void on_done(void* ctx, int status) {
    Gather* g = (Gather*)ctx;
    std::lock_guard<std::mutex> lk(g->m);
    if (status != VS_OK) g->status = status;
    if (--g->pending == 0) g->c.notify_one();
}
}  // namespace

int vs_shards_search(vs_shards* s, const float* q, size_t dim, size_t k, uint64_t* keys, float* dist, size_t* found) {
    if (!s || !q || !keys || !dist || !found || !k) return VS_ERR_INVALID_ARGUMENT;
    const size_t g = s->shard.size();
    std::vector<uint64_t> pk(g * k);
    std::vector<float> pd(g * k);
    std::vector<size_t> pf(g, 0);
    Gather ga;
    ga.pending = g;
    for (size_t o = 0; o < g; ++o) {
        int rc = vs_hnsw_search_async(s->shard[o], q, dim, k, &pk[o * k], &pd[o * k], &pf[o], on_done, &ga);
        if (rc != VS_OK) {  // not submitted: synchronous path (k beyond the beam, wrong dimension, ...)
            rc = vs_hnsw_search(s->shard[o], q, dim, k, &pk[o * k], &pd[o * k], &pf[o]);
            on_done(&ga, rc);
            if (rc != VS_OK) g_err = vs_hnsw_last_error();
        }
    }
    {
        std::unique_lock<std::mutex> lk(ga.m);
        ga.c.wait(lk, [&] { return ga.pending == 0; });
    }
    if (ga.status != VS_OK) {
        *found = 0;
        if (g_err.empty()) g_err = "shard search failed";
        return ga.status;
    }
    *found = merge_lists(g, k, pk.data(), pd.data(), pf.data(), k, keys, dist);
    return VS_OK;
}

int vs_shards_filtered_search(vs_shards* s, const float* q, size_t dim, size_t k, vs_hnsw_predicate pred, void* ctx,
                              uint64_t* keys, float* dist, size_t* found) {
    if (!s || !q || !keys || !dist || !found || !k || !pred) return VS_ERR_INVALID_ARGUMENT;
    const size_t g = s->shard.size();
    std::vector<uint64_t> pk(g * k);
    std::vector<float> pd(g * k);
    std::vector<size_t> pf(g, 0);
    for (size_t o = 0; o < g; ++o) {  // the predicate is host state of the caller: evaluated on this thread only
        int rc = vs_hnsw_filtered_search(s->shard[o], q, dim, k, pred, ctx, &pk[o * k], &pd[o * k], &pf[o]);
        if (rc != VS_OK) return fail_from_engine(rc);
    }
    *found = merge_lists(g, k, pk.data(), pd.data(), pf.data(), k, keys, dist);
    return VS_OK;
}

int vs_shards_search_batch(vs_shards* s, const float* q, size_t nq, size_t dim, size_t k, uint64_t* keys, float* dist,
                           size_t* found) {
    if (!s || (nq && (!q || !keys || !dist || !found)) || !k) return VS_ERR_INVALID_ARGUMENT;
    const size_t g = s->shard.size();
    std::vector<std::vector<uint64_t>> pk(g, std::vector<uint64_t>(nq * k));
    std::vector<std::vector<float>> pd(g, std::vector<float>(nq * k));
    std::vector<std::vector<size_t>> pf(g, std::vector<size_t>(nq));
    std::vector<int> rc(g, VS_OK);
    std::vector<std::string> err(g);
    std::vector<std::thread> th;
    for (size_t o = 0; o < g; ++o)
        th.emplace_back([&, o] {
            rc[o] = vs_hnsw_search_batch(s->shard[o], q, nq, dim, k, pk[o].data(), pd[o].data(), pf[o].data());
            if (rc[o] != VS_OK) err[o] = vs_hnsw_last_error();
        });
    for (auto& t : th) t.join();
    for (size_t o = 0; o < g; ++o)
        if (rc[o] != VS_OK) {
            g_err = err[o];
            return rc[o];
        }
    std::vector<uint64_t> lk(g * k);
    std::vector<float> ld(g * k);
    std::vector<size_t> lf(g);
    for (size_t i = 0; i < nq; ++i) {
        for (size_t o = 0; o < g; ++o) {
            std::copy(pk[o].begin() + i * k, pk[o].begin() + (i + 1) * k, lk.begin() + o * k);
            std::copy(pd[o].begin() + i * k, pd[o].begin() + (i + 1) * k, ld.begin() + o * k);
            lf[o] = pf[o][i];
        }
        found[i] = merge_lists(g, k, lk.data(), ld.data(), lf.data(), k, keys + i * k, dist + i * k);
        for (size_t j = found[i]; j < k; ++j) {
            keys[i * k + j] = kFreeKey;
            dist[i * k + j] = __builtin_inff();
        }
    }
    return VS_OK;
}

int vs_shards_set_expansion_search(vs_shards* s, size_t ef) {
    if (!s) return VS_ERR_INVALID_ARGUMENT;
    for (vs_hnsw* h : s->shard) {
        int rc = vs_hnsw_set_expansion_search(h, ef);
        if (rc != VS_OK) return fail_from_engine(rc);
    }
    return VS_OK;
}

int vs_shards_stats(vs_shards* s, uint64_t out[8], int reset) {
    if (!s || !out) return VS_ERR_INVALID_ARGUMENT;
    std::fill(out, out + 8, 0ull);
    for (vs_hnsw* h : s->shard) {
        uint64_t one[8];
        int rc = vs_hnsw_stats(h, one, reset);
        if (rc != VS_OK) return fail_from_engine(rc);
        for (int i = 0; i < 8; ++i) out[i] += one[i];
    }
    return VS_OK;
}

}  // extern "C"
