// engine_abi.hpp -- part of the engine's single translation unit (included by engine.hip, in this order: engine_base.hpp,
// engine_pods.hpp, the Engine itself in engine.hip, engine_service.hpp, engine_abi.hpp).  The C ABI of include/vs_hnsw.h over Engine.
#pragma once

// =============================================================================== C ABI
using vs::Engine;
using vs::Fail;

struct vs_hnsw {
    Engine e;
};

template <class F>
static int guarded(F&& f) {
    try {
        f();
        return VS_OK;
    } catch (const Fail& x) {
        vs::g_err = x.msg;
        return x.code;
    } catch (const std::bad_alloc&) {
        vs::g_err = "host out of memory";
        return VS_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& x) {
        vs::g_err = x.what();
        return VS_ERR_DEVICE;
    } catch (...) {
        vs::g_err = "unknown error";
        return VS_ERR_DEVICE;
    }
}

static void need(bool cond, const char* what) {
    if (!cond) vs::fail(VS_ERR_INVALID_ARGUMENT, what);
}
static void check_dim(const vs_hnsw* h, size_t dim) {
    if (dim != h->e.dim)
        vs::fail(VS_ERR_DIMENSION, "wrong embedding dimension: got " + std::to_string(dim) + ", index has " +
                                       std::to_string(h->e.dim));
}

extern "C" {

const char* vs_hnsw_version(void) { return VS_VERSION; }
const char* vs_hnsw_last_error(void) { return vs::g_err.c_str(); }

int vs_hnsw_create(const vs_hnsw_options* o, vs_hnsw** out) {
    return guarded([&] {
        need(o && out, "null argument");
        std::unique_ptr<vs_hnsw> h(new vs_hnsw());
        h->e.init(*o);
        *out = h.release();
    });
}
void vs_hnsw_free(vs_hnsw* h) {
    try {
        delete h;
    } catch (...) {
    }
}
int vs_hnsw_reserve(vs_hnsw* h, size_t capacity, size_t /*threads*/) {
    return guarded([&] {
        need(h, "null index");
        h->e.flush_pending();
        h->e.reserve(capacity);
    });
}
size_t vs_hnsw_capacity(const vs_hnsw* h) { return h ? h->e.capacity : 0; }
size_t vs_hnsw_size(const vs_hnsw* h) {
    if (!h) return 0;
    try {
        const_cast<vs_hnsw*>(h)->e.flush_pending();  // staged adds count: they are indexed before anyone can look
    } catch (...) {
    }
    return h->e.live.load();
}
size_t vs_hnsw_bytes_per_vector(const vs_hnsw* h) { return h ? h->e.row_bytes : 0; }

int vs_hnsw_add(vs_hnsw* h, uint64_t key, const float* v, size_t dim) {
    int rc = VS_OK;
    int g = guarded([&] {
        need(h && v, "null argument");
        check_dim(h, dim);
        rc = h->e.add_one(key, v);
    });
    return g != VS_OK ? g : rc;
}

static int add_many(vs_hnsw* h, const uint64_t* keys, const float* vecs, size_t n, size_t dim, bool on_device) {
    return guarded([&] {
        need(h && (n == 0 || (keys && vecs)), "null argument");
        check_dim(h, dim);
        h->e.flush_pending();
        std::vector<int> status;
        std::string err;
        h->e.add_batch(keys, vecs, on_device, n, status, err);
        size_t bad = 0;
        int code = VS_OK;
        for (int s : status)
            if (s != VS_OK) {
                if (!bad) code = s;
                ++bad;
            }
        if (bad) vs::fail(code, err + " (" + std::to_string(bad) + " of " + std::to_string(n) + " vectors rejected)");
    });
}
int vs_hnsw_add_batch(vs_hnsw* h, const uint64_t* keys, const float* vecs, size_t n, size_t dim) {
    return add_many(h, keys, vecs, n, dim, false);
}
int vs_hnsw_add_batch_device(vs_hnsw* h, const uint64_t* keys, const float* d_vecs, size_t n, size_t dim) {
    return add_many(h, keys, d_vecs, n, dim, true);
}

int vs_hnsw_remove(vs_hnsw* h, uint64_t key, int* removed) {
    return guarded([&] {
        need(h, "null index");
        bool r = false;
        const int rc = h->e.remove_one(key, &r);
        if (rc != VS_OK) vs::fail(rc, vs::g_err);
        if (removed) *removed = r ? 1 : 0;
    });
}

int vs_hnsw_search(vs_hnsw* h, const float* q, size_t dim, size_t k, uint64_t* keys, float* dist, size_t* found) {
    int rc = VS_OK;
    int g = guarded([&] {
        need(h && q && keys && dist && found, "null argument");
        check_dim(h, dim);
        vs::Engine::Stopwatch sw(h->e.c_search_ns);
        h->e.c_searches.fetch_add(1, std::memory_order_relaxed);
        {
            vs::Engine::Stopwatch fw(h->e.c_flush_wait_ns);
            h->e.flush_pending();
        }
        uint32_t ef;
        *found = 0;
        need(k > 0, "k must be > 0");
        const size_t beam = std::max<size_t>(k, h->e.ef_search.load());
        if (beam > vs::kMaxWalkBeam) {  // beyond the widest walk: exhaustive ranking (exact, a superset of any beam's answer)
            *found = h->e.rank_all(q, k, keys, dist);
            return;
        }
        h->e.check_search(k, ef);
        if (h->e.needs_global_walk(ef)) {  // wide beams / huge indexes: the global-bitmap walk, own launch
            h->e.search_host(q, 1, k, keys, dist, found, false);
            if (*found == (size_t)-1) *found = h->e.rank_all(q, k, keys, dist);
            return;
        }
        rc = h->e.search_one(q, k, keys, dist, found);
    });
    return g != VS_OK ? g : rc;
}

int vs_hnsw_search_async(vs_hnsw* h, const float* q, size_t dim, size_t k, uint64_t* keys, float* dist, size_t* found,
                         vs_hnsw_completion done, void* ctx) {
    return guarded([&] {
        need(h && q && keys && dist && found && done, "null argument");
        check_dim(h, dim);
        h->e.flush_pending();
        uint32_t ef;
        h->e.check_search(k, ef);
        *found = 0;
        h->e.search_async(q, k, keys, dist, found, done, ctx);
    });
}

int vs_hnsw_filtered_search(vs_hnsw* h, const float* q, size_t dim, size_t k, vs_hnsw_predicate pred, void* ctx,
                            uint64_t* keys, float* dist, size_t* found) {
    return guarded([&] {
        need(h && q && keys && dist && found && pred, "null argument");
        need(k > 0, "k must be > 0");
        check_dim(h, dim);
        vs::Engine::Stopwatch sw(h->e.c_filtered_ns);
        h->e.c_filtered.fetch_add(1, std::memory_order_relaxed);
        {
            vs::Engine::Stopwatch fw(h->e.c_flush_wait_ns);
            h->e.flush_pending();
        }
        *found = h->e.filtered(q, k, pred, ctx, keys, dist);
    });
}

int vs_hnsw_filtered_search_keyed(vs_hnsw* h, const float* q, size_t dim, size_t k, vs_hnsw_predicate pred, void* ctx, uint64_t filter_key,
                                  uint64_t* keys, float* dist, size_t* found) {
    return guarded([&] {
        need(h && q && keys && dist && found && pred, "null argument");
        need(k > 0, "k must be > 0");
        check_dim(h, dim);
        vs::Engine::Stopwatch sw(h->e.c_filtered_ns);
        h->e.c_filtered.fetch_add(1, std::memory_order_relaxed);
        {
            vs::Engine::Stopwatch fw(h->e.c_flush_wait_ns);
            h->e.flush_pending();
        }
        *found = h->e.filtered(q, k, pred, ctx, keys, dist, false, filter_key);
    });
}

int vs_hnsw_filter_forget(vs_hnsw* h, uint64_t filter_key, size_t* dropped) {
    return guarded([&] {
        need(h, "null argument");
        const size_t d = h->e.memo_forget_filter(filter_key);
        if (dropped) *dropped = d;
    });
}

int vs_hnsw_filter_forget_keys(vs_hnsw* h, const uint64_t* keys, size_t n) {
    return guarded([&] {
        need(h && (n == 0 || keys), "null argument");
        h->e.memo_forget_keys(keys, n);
    });
}

int vs_hnsw_filter_ask_stats(vs_hnsw* h, uint64_t out[8]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.ask_queries.load();
    out[1] = h->e.ask_handed_over.load();
    out[2] = h->e.ask_no_pod.load();
    out[3] = h->e.ask_calls.load();
    out[4] = h->e.ask_waits.load();
    out[5] = h->e.ask_wait_ticks.load();
    out[6] = h->e.ask_hops.load();
    out[7] = h->e.ask_walk_ticks.load();
    return VS_OK;
}

int vs_hnsw_filter_memo_stats(vs_hnsw* h, uint64_t out[6]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.memo_queries.load();
    out[1] = h->e.memo_asked.load();
    out[2] = h->e.memo_created.load();
    {
        std::lock_guard<std::mutex> g(h->e.memo_mu);
        out[3] = h->e.memos.size();
    }
    out[4] = h->e.memo_forgets.load();
    out[5] = h->e.memo_forgotten_keys.load();
    return VS_OK;
}

int vs_hnsw_search_batch(vs_hnsw* h, const float* q, size_t nq, size_t dim, size_t k, uint64_t* keys, float* dist,
                         size_t* found) {
    return guarded([&] {
        need(h && (nq == 0 || (q && keys && dist && found)), "null argument");
        check_dim(h, dim);
        h->e.flush_pending();
        if (std::max<size_t>(k, h->e.ef_search.load()) > vs::kMaxWalkBeam) {
            for (size_t i = 0; i < nq; ++i) found[i] = h->e.rank_all(q + i * dim, k, keys + i * k, dist + i * k);
            return;
        }
        h->e.search_host(q, nq, k, keys, dist, found, false);
        for (size_t i = 0; i < nq; ++i)
            if (found[i] == (size_t)-1) found[i] = h->e.rank_all(q + i * dim, k, keys + i * k, dist + i * k);
    });
}
int vs_hnsw_exact_search_batch(vs_hnsw* h, const float* q, size_t nq, size_t dim, size_t k, uint64_t* keys, float* dist,
                               size_t* found) {
    return guarded([&] {
        need(h && (nq == 0 || (q && keys && dist && found)), "null argument");
        check_dim(h, dim);
        h->e.flush_pending();
        h->e.search_host(q, nq, k, keys, dist, found, true);
    });
}
int vs_hnsw_search_batch_device(vs_hnsw* h, const float* d_q, size_t nq, size_t dim, size_t k, uint64_t* d_keys,
                                float* d_dist, uint32_t* d_found, void* stream) {
    return guarded([&] {
        need(h && (nq == 0 || (d_q && d_keys && d_dist && d_found)), "null argument");
        check_dim(h, dim);
        h->e.flush_pending();
        h->e.use_device();
        h->e.search_device(d_q, nq, k, d_keys, d_dist, d_found, (hipStream_t)stream);
    });
}
int vs_hnsw_exact_search_batch_device(vs_hnsw* h, const float* d_q, size_t nq, size_t dim, size_t k, uint64_t* d_keys,
                                      float* d_dist, uint32_t* d_found, void* stream) {
    return guarded([&] {
        need(h && (nq == 0 || (d_q && d_keys && d_dist && d_found)), "null argument");
        check_dim(h, dim);
        h->e.flush_pending();
        h->e.use_device();
        vs::Lease w(h->e.device);
        h->e.exact_device(d_q, nq, k, d_keys, d_dist, d_found, (hipStream_t)stream, *w.ctx);
        HIP_OK(hipStreamSynchronize((hipStream_t)stream));  // scratch returns to the pool with the lease
    });
}

int vs_hnsw_set_expansion_search(vs_hnsw* h, size_t ef) {
    return guarded([&] {
        need(h && ef > 0, "invalid argument");
        h->e.ef_search = (uint32_t)ef;
    });
}

int vs_hnsw_stats(vs_hnsw* h, uint64_t out[8], int reset) {
    return guarded([&] {
        need(h && out, "null argument");
        h->e.flush_pending();
        h->e.use_device();
        vs::PodHold hold(vs::pod_pool(h->e.device));
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(out, h->e.d_stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (reset) HIP_OK(hipMemset(h->e.d_stats, 0, 8 * sizeof(uint64_t)));
    });
}

int vs_search_service_stats(uint64_t out[4]) {
    if (!out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = vs::SearchService::n_batches.load();
    out[1] = vs::SearchService::n_queries.load();
    out[2] = vs::SearchService::n_team_batches.load();
    out[3] = vs::SearchService::n_team_queries.load();
    return VS_OK;
}

int vs_hnsw_memory_info(vs_hnsw* h, uint64_t out[4]) {
    return guarded([&] {
        need(h && out, "null argument");
        std::lock_guard<std::mutex> g(h->e.mod_mu);
        Engine& e = h->e;
        out[0] = out[1] = out[2] = 0;
        for (const vs::Arena* a : {&e.ar_vectors, &e.ar_aux, &e.ar_adj0, &e.ar_upper, &e.ar_upper_off, &e.ar_keys, &e.ar_levels, &e.ar_plane, &e.ar_plane8, &e.ar_p8scale}) {
            out[0] += a->bytes;
            if (a->vmm) {
                out[1] += a->bytes;
                out[2] += a->chunks.size();
            }
        }
        out[3] = vs::Arena::copied_bytes.load();
    });
}

int vs_hnsw_filter_stats(vs_hnsw* h, uint64_t out[2]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.lazy_rounds.load();
    out[1] = h->e.lazy_predicate_calls.load();
    return VS_OK;
}

int vs_hnsw_filter_batch_stats(vs_hnsw* h, uint64_t out[2]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.batcher.launches.load();
    out[1] = h->e.batcher.rounds.load();
    return VS_OK;
}

uint64_t vs_hnsw_streams_created(void) { return vs::g_streams_created.load(); }

int vs_hnsw_pod_stats(vs_hnsw* h, uint64_t out[12]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    vs::PodPool& pp = vs::pod_pool(h->e.device);
    out[0] = h->e.pod_opens.load();
    out[1] = h->e.pod_rounds.load();
    out[2] = pp.n_opened.load() - pp.n_closed.load();
    out[3] = pp.enabled ? 1 : 0;
    out[4] = pp.plain_queries.load();
    out[5] = pp.plain_ns.load();
    out[6] = pp.plain_wait_ns.load();
    out[7] = pp.plain_gpu_ticks.load() * 10;
    out[8] = h->e.batched_done.load();
    out[9] = h->e.batched_handed_over.load();
    out[10] = h->e.batched_no_pod.load();
    out[11] = h->e.batched_second_chances.load();
    return VS_OK;
}

int vs_hnsw_modify_stats(vs_hnsw* h, uint64_t out[8]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.m_flushes.load();
    out[1] = h->e.m_flushed.load();
    out[2] = h->e.m_flush_ns.load();
    out[3] = h->e.m_quiesces.load();
    out[4] = h->e.m_quiesce_ns.load();
    out[5] = h->e.m_removes.load();
    out[6] = h->e.m_remove_ns.load();
    out[7] = h->e.pod_opens.load();
    return VS_OK;
}

int vs_hnsw_call_stats(vs_hnsw* h, uint64_t out[8]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.c_searches.load();
    out[1] = h->e.c_search_ns.load();
    out[2] = h->e.c_filtered.load();
    out[3] = h->e.c_filtered_ns.load();
    out[4] = h->e.c_filtered_wait_ns.load();
    out[5] = h->e.c_filtered_pred_ns.load();
    out[6] = h->e.c_flush_wait_ns.load();
    out[7] = 0;
    return VS_OK;
}

int vs_hnsw_pipe_stats(vs_hnsw* h, uint64_t out[2]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.pipe_launches.load();
    out[1] = vs::SearchService::n_pipe_redone.load();
    return VS_OK;
}

int vs_hnsw_walk_info(vs_hnsw* h, uint64_t out[2]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    const uint32_t inst = h->e.last_walk_instance.load();
    out[0] = inst == 0xFFFFFFFFu ? ~0ull : inst;
    out[1] = vs::SearchService::n_ranked_fallbacks.load();
    return VS_OK;
}

int vs_hnsw_exact_stats(vs_hnsw* h, uint64_t out[2]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.block_batches.load();
    out[1] = h->e.block_fallbacks.load();
    return VS_OK;
}

int vs_hnsw_exact_stats2(vs_hnsw* h, uint64_t out[4]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.plane_batches.load();
    out[1] = h->e.plane_fallbacks.load();
    out[2] = h->e.block_batches.load();
    out[3] = h->e.block_fallbacks.load();
    return VS_OK;
}

int vs_hnsw_exact_stats3(vs_hnsw* h, uint64_t out[4]) {
    if (!h || !out) return VS_ERR_INVALID_ARGUMENT;
    out[0] = h->e.plane8_batches.load();
    out[1] = h->e.plane8_fallbacks.load();
    uint32_t bits = 0;
    std::memcpy(&bits, &h->e.plane8_rho, 4);
    out[2] = bits;
    out[3] = h->e.plane8_done;
    return VS_OK;
}

int vs_hnsw_graph_info_get(vs_hnsw* h, vs_hnsw_graph_info* info) {
    return guarded([&] {
        need(h && info, "null argument");
        h->e.flush_pending();
        std::lock_guard<std::mutex> g(h->e.mod_mu);
        info->slots = h->e.slots;
        info->upper_blocks = h->e.upper_blocks;
        info->max_level = h->e.max_level.load();
        info->entry_slot = h->e.entry_slot.load();
        info->connectivity = h->e.M;
        info->connectivity_base = h->e.M0;
    });
}

int vs_hnsw_export_graph(vs_hnsw* h, void* vectors, int32_t* levels, uint64_t* keys, uint32_t* adj0, uint32_t* upper_off,
                         uint32_t* upper) {
    return guarded([&] {
        need(h, "null index");
        Engine& e = h->e;
        e.flush_pending();
        std::lock_guard<std::mutex> g(e.mod_mu);
        e.use_device();
        vs::PodHold hold(vs::pod_pool(e.device));
        HIP_OK(hipDeviceSynchronize());
        const size_t n = e.slots;
        if (!n) return;
        if (vectors) {  // storage format, unpadded: row_bytes per vector (f32 storage: the floats themselves)
            vs::Lease w(e.device);
            void* tmp = w->a.ensure(n * (size_t)e.row_bytes);
            HIP_OK(vs::launch_copy_rows(tmp, e.row_bytes, e.d_vectors, e.stride4 * 16, e.row_bytes, e.row_bytes, (uint32_t)n,
                                        w->stream));
            HIP_OK(hipMemcpyAsync(vectors, tmp, n * (size_t)e.row_bytes, hipMemcpyDeviceToHost, w->stream));
            HIP_OK(hipStreamSynchronize(w->stream));
        }
        if (levels) HIP_OK(hipMemcpy(levels, e.d_levels, n * 4, hipMemcpyDeviceToHost));
        if (keys) HIP_OK(hipMemcpy(keys, e.d_keys, n * 8, hipMemcpyDeviceToHost));
        if (adj0) HIP_OK(hipMemcpy(adj0, e.d_adj0, n * e.M0 * 4, hipMemcpyDeviceToHost));
        if (upper_off) HIP_OK(hipMemcpy(upper_off, e.d_upper_off, n * 4, hipMemcpyDeviceToHost));
        if (upper && e.upper_blocks) HIP_OK(hipMemcpy(upper, e.d_upper, e.upper_blocks * e.M * 4, hipMemcpyDeviceToHost));
    });
}

int vs_hnsw_import_graph(vs_hnsw* h, size_t n, const void* vectors, const int32_t* levels, const uint64_t* keys,
                         const uint32_t* adj0, const uint32_t* upper_off, const uint32_t* upper, size_t upper_blocks,
                         int32_t max_level, uint32_t entry_slot) {
    return guarded([&] {
        need(h && (n == 0 || (vectors && levels && keys && adj0 && upper_off)), "null argument");
        Engine& e = h->e;
        need(e.slots == 0, "import needs an empty index");
        e.pods_quiesce();
        if (n > e.capacity) e.reserve(n);
        std::lock_guard<std::mutex> g(e.mod_mu);
        e.use_device();
        if (!n) return;
        e.ensure_upper(upper_blocks);
        vs::Lease w(e.device);
        void* tmp = w->a.ensure(n * (size_t)e.row_bytes);
        HIP_OK(hipMemcpyAsync(tmp, vectors, n * (size_t)e.row_bytes, hipMemcpyHostToDevice, w->stream));
        HIP_OK(vs::launch_copy_rows(e.d_vectors, e.stride4 * 16, tmp, e.row_bytes, e.row_bytes, e.stride4 * 16, (uint32_t)n,
                                    w->stream));
        vs::IndexView ix = e.view();
        HIP_OK(vs::launch_aux_rows(ix, e.d_aux, (uint32_t)n, w->stream));
        HIP_OK(hipStreamSynchronize(w->stream));
        HIP_OK(hipMemcpy(e.d_levels, levels, n * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(e.d_keys, keys, n * 8, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(e.d_adj0, adj0, n * e.M0 * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(e.d_upper_off, upper_off, n * 4, hipMemcpyHostToDevice));
        if (upper_blocks) HIP_OK(hipMemcpy(e.d_upper, upper, upper_blocks * e.M * 4, hipMemcpyHostToDevice));
        e.slots = n;
        e.slots_atomic.store(n, std::memory_order_release);
        e.linked = n;
        e.upper_blocks = upper_blocks;
        size_t live = 0;
        for (size_t s = 0; s < n; ++s) {
            e.h_levels[s] = (uint8_t)levels[s];
            e.h_upper_off[s] = upper_off[s];
            e.h_keys[s] = keys[s];
            if (keys[s] != vs::kFreeKey) {
                e.lookup.emplace(keys[s], (uint32_t)s);
                ++live;
            } else {
                e.free_slots.push_back((uint32_t)s);
                ++e.removed;
            }
        }
        e.live = live;
        e.committed = live;
        {
            std::lock_guard<std::mutex> ng(e.norm_mu);
            e.max_norm_slots = 0;  // the contents were replaced
        }
        {
            std::lock_guard<std::mutex> pg(e.plane_mu);
            e.plane_done = 0;
        }
        e.max_level = max_level;
        e.entry_slot = entry_slot;
    });
}

int vs_topk_merge_device(const uint64_t* d_part_keys, const float* d_part_dists, size_t parts, size_t nq, size_t k,
                         uint64_t* d_keys, float* d_dists, uint32_t* d_found, void* stream) {
    return guarded([&] {
        need(d_part_keys && d_part_dists && d_keys && d_dists, "null argument");
        HIP_OK(vs::launch_topk_merge(d_part_keys, d_part_dists, (uint32_t)parts, (uint32_t)nq, (uint32_t)k, d_keys, d_dists,
                                     d_found, (hipStream_t)stream));
    });
}

int vs_topk_merge_packed_device(const void* d_blocks, size_t parts, size_t block_bytes, size_t nq, size_t k, uint64_t* d_keys,
                                float* d_dists, uint32_t* d_found, void* stream) {
    return guarded([&] {
        need(d_blocks && d_keys && d_dists, "null argument");
        need(block_bytes % 16 == 0 && block_bytes >= nq * k * 12, "blocks must be 16-byte multiples of at least nq * k * 12 bytes");
        const char* base = (const char*)d_blocks;
        HIP_OK(vs::launch_topk_merge((const uint64_t*)base, (const float*)(base + nq * k * 8), (uint32_t)parts, (uint32_t)nq, (uint32_t)k,
                                     d_keys, d_dists, d_found, (hipStream_t)stream, block_bytes / 8, block_bytes / 4));
    });
}

// reference vs_index/usearch.rs:1179-1205
void vs_f32_to_b1x8(const float* v, size_t n, uint8_t* out) {
    const size_t nb = (n + 7) / 8;
    for (size_t j = 0; j < nb; ++j) {
        uint8_t byte = 0;
        for (size_t i = 0; i < 8 && j * 8 + i < n; ++i)
            if (v[j * 8 + i] > 0.0f) byte |= (uint8_t)(1u << i);
        out[j] = byte;
    }
}

// reference distance.rs:58-105
int vs_distance_valid(float v, int metric, size_t dim) {
    switch (metric) {
        case VS_METRIC_COS: return v >= 0.0f && v <= 2.0f;
        case VS_METRIC_L2SQ: return v >= 0.0f;
        case VS_METRIC_IP: return !std::isnan(v);
        case VS_METRIC_HAMMING: return v >= 0.0f && std::isfinite(v) && v == std::trunc(v) && v <= (float)dim;
        default: return 0;
    }
}

// reference similarity.rs:28-35
float vs_similarity_score(float d, int metric, size_t dim) {
    switch (metric) {
        case VS_METRIC_COS:
        case VS_METRIC_IP: return (2.0f - d) / 2.0f;
        case VS_METRIC_L2SQ: return 1.0f / (1.0f + d);
        default: return 1.0f - d / (float)dim;
    }
}

}  // extern "C"

