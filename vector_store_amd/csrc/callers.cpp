// callers.cpp -- libvs_callers.so (include/vs_callers.h): the reference's search load loop
// (crates/benchmark/src/main.rs:435-525) over the C ABI of the engine, blocking or with queries in flight.
#include "../../include/vs_callers.h"
#include "../../include/vs_hnsw_debug.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <unordered_set>
#include <vector>

#include "bench_util.hpp"

using vsb::SearchMeasure;
using Clock = std::chrono::steady_clock;

namespace {
double recall_of(const uint64_t* truth, size_t k, const uint64_t* found, size_t n) {
    size_t hit = 0;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < k; ++j)
            if (truth[j] == found[i]) {
                ++hit;
                break;
            }
    return k ? (double)hit / (double)k : 0.0;
}
// one answer into the record (the first `cap` completed calls; slots are handed out by an atomic counter)
void keep(vs_callers_record* rec, std::atomic<size_t>& next, size_t qi, size_t k, const uint64_t* keys, const float* dist, size_t found) {
    if (!rec || !rec->cap) return;
    const size_t at = next.fetch_add(1, std::memory_order_relaxed);
    if (at >= rec->cap) return;
    rec->query[at] = (uint32_t)qi;
    rec->found[at] = (uint32_t)found;
    for (size_t i = 0; i < k; ++i) {
        rec->keys[at * k + i] = i < found ? keys[i] : ~0ull;
        rec->distances[at * k + i] = i < found ? dist[i] : __builtin_inff();
    }
}
}  // namespace

extern "C" int vs_callers_run(vs_hnsw* h, const float* queries, size_t nq, size_t dim, size_t k, const uint64_t* truth,
                              unsigned threads, unsigned inflight, double seconds, vs_callers_result* out) {
    return vs_callers_run_recorded(h, queries, nq, dim, k, truth, threads, inflight, seconds, out, nullptr);
}

extern "C" int vs_callers_run_recorded(vs_hnsw* h, const float* queries, size_t nq, size_t dim, size_t k, const uint64_t* truth,
                                       unsigned threads, unsigned inflight, double seconds, vs_callers_result* out, vs_callers_record* rec) {
    if (!h || !queries || !nq || !k || !threads || !out) return VS_ERR_INVALID_ARGUMENT;
    if (!inflight) inflight = 1;
    std::atomic<size_t> rec_next{0};
    std::atomic<bool> stop{false};
    std::vector<SearchMeasure> per(threads);
    std::vector<uint64_t> errors(threads, 0);
    for (auto& m : per) m.with_recall = truth != nullptr;
    uint64_t sv0[4] = {0, 0, 0, 0}, sv1[4] = {0, 0, 0, 0};
    (void)vs_search_service_stats(sv0);
    auto t0 = Clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 7919 + 13);
            if (inflight == 1) {  // one blocking call per query: a thread of the reference's worker pool
                std::vector<uint64_t> keys(k);
                std::vector<float> dist(k);
                while (!stop.load(std::memory_order_relaxed)) {
                    const size_t qi = g() % nq;
                    size_t found = 0;
                    auto s = Clock::now();
                    int rc = vs_hnsw_search(h, queries + qi * dim, dim, k, keys.data(), dist.data(), &found);
                    int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                    if (rc != VS_OK) {
                        ++errors[t];
                        break;
                    }
                    per[t].record(ns, truth ? recall_of(truth + qi * k, k, keys.data(), found) : 0.0);
                    keep(rec, rec_next, qi, k, keys.data(), dist.data(), found);
                }
                return;
            }
            struct Slot {  // the non-blocking entry point, `inflight` queries outstanding per worker
                std::vector<uint64_t> keys;
                std::vector<float> dist;
                size_t found = 0, qi = 0;
                bool busy = false, done = true;
                Clock::time_point start;
                int status = 0;
                std::mutex* mu;
                std::condition_variable* cv;
            };
            std::mutex mu;
            std::condition_variable cv;
            std::vector<Slot> slots(inflight);
            for (auto& sl : slots) {
                sl.keys.resize(k);
                sl.dist.resize(k);
                sl.mu = &mu;
                sl.cv = &cv;
            }
            auto done = [](void* ctx, int status) {
                Slot* sl = (Slot*)ctx;
                std::lock_guard<std::mutex> lk(*sl->mu);
                sl->status = status;
                sl->done = true;
                sl->cv->notify_one();
            };
            size_t outstanding = 0;
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                bool stopping = stop.load(std::memory_order_relaxed);
                for (auto& sl : slots) {
                    if (!sl.done) continue;
                    if (sl.busy) {  // a completed query
                        int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - sl.start).count();
                        if (sl.status != VS_OK) ++errors[t];
                        else {
                            per[t].record(ns, truth ? recall_of(truth + sl.qi * k, k, sl.keys.data(), sl.found) : 0.0);
                            keep(rec, rec_next, sl.qi, k, sl.keys.data(), sl.dist.data(), sl.found);
                        }
                        sl.busy = false;
                        --outstanding;
                    }
                    if (stopping) continue;
                    sl.qi = g() % nq;
                    sl.done = false;
                    sl.busy = true;
                    sl.start = Clock::now();
                    ++outstanding;
                    lk.unlock();
                    int rc = vs_hnsw_search_async(h, queries + sl.qi * dim, dim, k, sl.keys.data(), sl.dist.data(), &sl.found, done, &sl);
                    lk.lock();
                    if (rc != VS_OK) {
                        ++errors[t];
                        sl.done = true;
                        sl.busy = false;
                        --outstanding;
                        stopping = true;
                        stop = true;
                    }
                }
                if (stopping && outstanding == 0) break;
                cv.wait(lk, [&] {
                    for (auto& sl : slots)
                        if (sl.done && sl.busy) return true;
                    return stop.load() && outstanding == 0;
                });
            }
        });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& x : th) x.join();
    const double wall = std::chrono::duration<double>(Clock::now() - t0).count();
    (void)vs_search_service_stats(sv1);
    SearchMeasure all;
    all.with_recall = truth != nullptr;
    for (auto& m : per) all.append(m);
    out->seconds = wall;
    out->queries = all.count;
    out->qps = all.count / wall;
    out->latency_min_ns = all.count ? all.latency_min : 0;
    out->latency_max_ns = all.latency_max;
    out->p01_ns = all.histogram.percentile(1);
    out->p10_ns = all.histogram.percentile(10);
    out->p25_ns = all.histogram.percentile(25);
    out->p50_ns = all.histogram.percentile(50);
    out->p75_ns = all.histogram.percentile(75);
    out->p90_ns = all.histogram.percentile(90);
    out->p99_ns = all.histogram.percentile(99);
    out->recall_avg = truth && all.count ? all.recall_sum / (double)all.count : -1.0;
    out->errors = 0;
    for (uint64_t e : errors) out->errors += e;
    out->launches = sv1[0] - sv0[0];
    out->team_launches = sv1[2] - sv0[2];
    if (rec) rec->n = std::min(rec->cap, rec_next.load());
    return VS_OK;
}

// The reference dispatches every filtered query through spawn_blocking (usearch.rs:937-948): `threads` blocking callers of
// vs_hnsw_filtered_search, each with its own predicate context (the reference's closure takes a table read-lock and evaluates
// the restriction per candidate, usearch.rs:1118-1124; here: key % modulus == 0, counted).  extra[0] = predicate calls,
// extra[1] = results returned, over the whole run.
extern "C" int vs_callers_run_filtered(vs_hnsw* h, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus, unsigned threads,
                                       double seconds, vs_callers_result* out, uint64_t extra[4]) {
    return vs_callers_run_filtered_recorded(h, queries, nq, dim, k, modulus, threads, seconds, out, extra, nullptr);
}

extern "C" int vs_callers_run_filtered_recorded(vs_hnsw* h, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus, unsigned threads,
                                                double seconds, vs_callers_result* out, uint64_t extra[4], vs_callers_record* rec) {
    return vs_callers_run_filtered_keyed(h, queries, nq, dim, k, modulus, 0, threads, seconds, out, extra, rec);
}

extern "C" int vs_callers_run_filtered_keyed(vs_hnsw* h, const float* queries, size_t nq, size_t dim, size_t k, uint64_t modulus, uint64_t filter_key,
                                             unsigned threads, double seconds, vs_callers_result* out, uint64_t extra[4], vs_callers_record* rec) {
    if (!h || !queries || !nq || !k || !threads || !out || !modulus || !extra) return VS_ERR_INVALID_ARGUMENT;
    std::atomic<size_t> rec_next{0};
    struct Ctx {
        uint64_t modulus;
        std::atomic<uint64_t>* calls;
    };
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> calls{0}, results{0};
    std::vector<SearchMeasure> per(threads);
    std::vector<uint64_t> errors(threads, 0);
    auto t0 = Clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 104729 + 7);
            std::vector<uint64_t> keys(k);
            std::vector<float> dist(k);
            uint64_t local_calls = 0;
            struct Local {
                uint64_t modulus;
                uint64_t* calls;
            } ctx{modulus, &local_calls};
            auto pred = [](uint64_t key, void* c) -> int {
                Local* l = (Local*)c;
                ++*l->calls;
                return key % l->modulus == 0 ? 1 : 0;
            };
            while (!stop.load(std::memory_order_relaxed)) {
                const size_t qi = g() % nq;
                size_t found = 0;
                auto s = Clock::now();
                int rc = filter_key ? vs_hnsw_filtered_search_keyed(h, queries + qi * dim, dim, k, pred, &ctx, filter_key, keys.data(), dist.data(), &found)
                                    : vs_hnsw_filtered_search(h, queries + qi * dim, dim, k, pred, &ctx, keys.data(), dist.data(), &found);
                int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                if (rc != VS_OK) {
                    ++errors[t];
                    break;
                }
                for (size_t i = 0; i < found; ++i)
                    if (keys[i] % modulus != 0) ++errors[t];  // a result the predicate rejects
                results += found;
                per[t].record(ns, 0.0);
                keep(rec, rec_next, qi, k, keys.data(), dist.data(), found);
            }
            calls += local_calls;
        });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop = true;
    for (auto& x : th) x.join();
    if (rec) rec->n = std::min(rec->cap, rec_next.load());
    const double wall = std::chrono::duration<double>(Clock::now() - t0).count();
    SearchMeasure all;
    for (auto& m : per) all.append(m);
    out->seconds = wall;
    out->queries = all.count;
    out->qps = all.count / wall;
    out->latency_min_ns = all.count ? all.latency_min : 0;
    out->latency_max_ns = all.latency_max;
    out->p01_ns = all.histogram.percentile(1);
    out->p10_ns = all.histogram.percentile(10);
    out->p25_ns = all.histogram.percentile(25);
    out->p50_ns = all.histogram.percentile(50);
    out->p75_ns = all.histogram.percentile(75);
    out->p90_ns = all.histogram.percentile(90);
    out->p99_ns = all.histogram.percentile(99);
    out->recall_avg = -1.0;
    out->errors = 0;
    for (uint64_t e : errors) out->errors += e;
    out->launches = out->team_launches = 0;
    extra[0] = calls.load();
    extra[1] = results.load();
    extra[2] = extra[3] = 0;
    return VS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The reference's mixed workloads through a dispatch actor (include/vs_callers.h).
namespace {
void fill_result(vs_callers_result* out, SearchMeasure& all, double wall) {
    out->seconds = wall;
    out->queries = all.count;
    out->qps = wall > 0 ? all.count / wall : 0.0;
    out->latency_min_ns = all.count ? all.latency_min : 0;
    out->latency_max_ns = all.latency_max;
    out->p01_ns = all.histogram.percentile(1);
    out->p10_ns = all.histogram.percentile(10);
    out->p25_ns = all.histogram.percentile(25);
    out->p50_ns = all.histogram.percentile(50);
    out->p75_ns = all.histogram.percentile(75);
    out->p90_ns = all.histogram.percentile(90);
    out->p99_ns = all.histogram.percentile(99);
    out->recall_avg = -1.0;
    out->errors = 0;
    out->launches = out->team_launches = 0;
}
}  // namespace

extern "C" int vs_mixed_run(vs_actor* actor, const vs_mixed_options* o, const float* queries, size_t nq, const float* vectors, size_t nv,
                            size_t dim, vs_mixed_result* out) {
    if (!actor || !o || !out || !dim || !o->k) return VS_ERR_INVALID_ARGUMENT;
    const unsigned searchers = o->plain_callers + o->filtered_callers;
    if (searchers && (!queries || !nq)) return VS_ERR_INVALID_ARGUMENT;
    if ((o->modify == VS_MIXED_INSERT || o->modify == VS_MIXED_UPDATE) && (!vectors || !nv)) return VS_ERR_INVALID_ARGUMENT;
    if (o->filtered_callers && !o->modulus) return VS_ERR_INVALID_ARGUMENT;
    const unsigned producers = o->modify == VS_MIXED_NONE ? 0u : (o->producers ? o->producers : 1u);
    static std::atomic<uint64_t> run_counter{0};
    const uint64_t run_salt = producers ? run_counter.fetch_add(1, std::memory_order_relaxed) : 0;
    const size_t k = o->k;
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> next_item{0}, adds_applied{0}, removes_applied{0}, calls{0}, results{0}, errors{0};
    std::vector<SearchMeasure> per_plain(o->plain_callers), per_filtered(o->filtered_callers), per_item(producers);
    const auto t0 = Clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < o->plain_callers; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 7919 + 13);
            std::vector<uint64_t> keys(k);
            std::vector<float> dist(k);
            while (!stop.load(std::memory_order_relaxed)) {
                const size_t qi = g() % nq;
                size_t found = 0;
                const auto s = Clock::now();
                const int rc = vs_actor_ann(actor, o->partition, queries + qi * dim, dim, k, keys.data(), dist.data(), &found);
                const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                if (rc != VS_OK) {
                    ++errors;
                    break;
                }
                per_plain[t].record(ns, 0.0);
            }
        });
    for (unsigned t = 0; t < o->filtered_callers; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 104729 + 7);
            std::vector<uint64_t> keys(k);
            std::vector<float> dist(k);
            struct Local {
                uint64_t modulus, calls;
            } ctx{o->modulus, 0};
            auto pred = [](uint64_t key, void* c) -> int {
                Local* l = (Local*)c;
                ++l->calls;
                return key % l->modulus == 0 ? 1 : 0;
            };
            while (!stop.load(std::memory_order_relaxed)) {
                const size_t qi = g() % nq;
                size_t found = 0;
                const auto s = Clock::now();
                const int rc = o->filter_key ? vs_actor_filtered_ann_keyed(actor, o->partition, queries + qi * dim, dim, k, pred, &ctx, o->filter_key, keys.data(), dist.data(), &found)
                                             : vs_actor_filtered_ann(actor, o->partition, queries + qi * dim, dim, k, pred, &ctx, keys.data(), dist.data(), &found);
                const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                if (rc != VS_OK) {
                    ++errors;
                    break;
                }
                for (size_t i = 0; i < found; ++i)
                    if (keys[i] % o->modulus != 0) ++errors;  // a result the predicate rejects
                results += found;
                per_filtered[t].record(ns, 0.0);
            }
            calls += ctx.calls;
        });
    for (unsigned t = 0; t < producers; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 15485863 + 29);
            std::vector<float> vec(dim);
            // item `it` carries vectors[it % nv]; from the second pass over the pool on, one coordinate is moved so that no two items
            // carry the same vector (exact duplicates make equal distances, which is a different workload: DESIGN.md section 4.7)
            auto vector_of = [&](uint64_t it) -> const float* {
                const float* src = vectors + (it % nv) * dim;
                const uint64_t pass = it / nv + run_salt * 64;  // (the runs of one process do not repeat each other's vectors either)
                if (!pass) return src;
                std::memcpy(vec.data(), src, dim * sizeof(float));
                vec[(size_t)((it * 2654435761ull) % dim)] += 0.02f * (float)(pass % 61 + 1) * ((pass & 1) ? 1.f : -1.f);
                vec[(size_t)((it * 40503ull + pass) % dim)] += 0.01f * (float)((pass / 61) % 97 + 1);
                return vec.data();
            };
            while (!stop.load(std::memory_order_relaxed)) {
                const uint64_t it = next_item.fetch_add(1, std::memory_order_relaxed);
                if (o->max_items && it >= o->max_items) break;
                int applied = 0, rc = VS_OK;
                const auto s = Clock::now();
                switch (o->modify) {
                    case VS_MIXED_INSERT:
                        rc = vs_actor_add_vector_wait(actor, o->partition, o->first_new_key + it, vector_of(it), dim, &applied);
                        adds_applied += (uint64_t)applied;
                        break;
                    case VS_MIXED_UPDATE: {
                        const uint64_t key = o->existing_keys ? g() % o->existing_keys : 0;
                        rc = vs_actor_remove_vector(actor, o->partition, key);  // RemoveBeforeAddValue: no marker
                        if (rc == VS_OK) rc = vs_actor_add_vector_wait(actor, o->partition, key, vector_of(it), dim, &applied);
                        adds_applied += (uint64_t)applied;
                        break;
                    }
                    default:
                        rc = vs_actor_remove_vector_wait(actor, o->partition, o->delete_from + it, &applied);
                        removes_applied += (uint64_t)applied;
                        break;
                }
                const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                if (rc != VS_OK) {
                    ++errors;
                    break;
                }
                per_item[t].record(ns, 0.0);
            }
        });
    // the run ends after `seconds`, or -- a modify-only run with max_items -- when the producers are through
    const bool until_items = o->max_items && !searchers && producers;
    if (until_items) {
        for (auto& x : th) x.join();
    } else {
        std::this_thread::sleep_for(std::chrono::duration<double>(o->seconds));
        stop = true;
        for (auto& x : th) x.join();
    }
    // whatever is staged inside the index is indexed before the clock stops (Count is answered by the actor; a search reaches the index)
    // -- ALWAYS, also in a modify-only run without queries: *_wait returns once the index has STAGED the operation, so without this
    // the last few thousand operations would be counted but not indexed when the clock stops (round-5 advisor); the probe is the
    // first query, or -- no queries given -- a unit vector
    if (producers) {
        std::vector<uint64_t> keys(k ? k : 1);
        std::vector<float> dist(k ? k : 1), probe;
        if (!(queries && nq)) {
            probe.assign(dim, 0.f);
            if (dim) probe[0] = 1.f;
        }
        size_t found = 0;
        (void)vs_actor_ann(actor, o->partition, queries && nq ? queries : probe.data(), dim, k ? k : 1, keys.data(), dist.data(), &found);
    }
    const double wall = std::chrono::duration<double>(Clock::now() - t0).count();
    SearchMeasure plain, filtered, item;
    for (auto& m : per_plain) plain.append(m);
    for (auto& m : per_filtered) filtered.append(m);
    for (auto& m : per_item) item.append(m);
    out->seconds = wall;
    out->items = item.count;
    out->adds_applied = adds_applied.load();
    out->removes_applied = removes_applied.load();
    out->predicate_calls = calls.load();
    out->filtered_results = results.load();
    out->errors = errors.load();
    fill_result(&out->item, item, wall);
    fill_result(&out->plain, plain, wall);
    fill_result(&out->filtered, filtered, wall);
    return VS_OK;
}
