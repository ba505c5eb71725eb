// engine_service.hpp -- part of the engine's single translation unit (included by engine.hip, in this order: engine_base.hpp,
// engine_pods.hpp, the Engine itself in engine.hip, engine_service.hpp, engine_abi.hpp).  The single-query dispatcher (SearchService) and the lone-query entry points of Engine (pods first, then the dispatcher).
#pragma once

namespace vs {

// ---------------------------------------------------------------------------------------------
// SearchService: one dispatcher thread per device turns the stream of single-query calls
// (vs_hnsw_search / vs_hnsw_search_async, one vector per FFI call as the reference issues them,
// usearch.rs:212) into kernel launches.  Requests queue under a mutex; the dispatcher drains the queue
// into one of kSlots pipeline slots (pinned staging, own stream), so up to kSlots batches are in flight
// and a batch is simply "whatever queued since the last launch" -- no timer, no per-caller HIP calls.
// Lightly loaded the latency is one graph walk; under load batches grow and the GPU saturates.
// ---------------------------------------------------------------------------------------------
struct SearchReq {
    Engine* e;
    std::vector<float> q;
    size_t k;
    uint64_t* keys;
    float* dist;
    size_t* found;
    void (*cb)(void*, int);
    void* ctx;
};

class SearchService {
   public:
    static SearchService& get(int device) {
        static std::mutex mu;
        static std::unordered_map<int, SearchService*> all;  // leaked on purpose: outlives static teardown
        std::lock_guard<std::mutex> g(mu);
        SearchService*& s = all[device];
        if (!s) s = new SearchService(device);
        return *s;
    }
    void submit(SearchReq&& r) {
        {
            std::lock_guard<std::mutex> g(mu_);
            pending_.push_back(std::move(r));
            n_pending_.store(pending_.size(), std::memory_order_release);
        }
        cv_.notify_one();
    }

   private:
    // Pipeline slots.  Blocking callers (the reference's num_workers() + 1 threads, one query each) come back in ones and
    // twos as their results are delivered: with two slots most of them found both busy and waited out half a walk on
    // average (1.5 walks per round trip); with eight a free slot is there when the query is, and the round trip is one
    // walk (team kernels of different streams run side by side: GPU_MAX_HW_QUEUES).  VS_HNSW_SERVICE_SLOTS: 1..16.
    static constexpr int kSlots = 16;
    int n_slots_ = 8;
    // While a batch is in flight the dispatcher polls its event and the queue instead of sleeping in timer steps (a
    // 20 us condition-variable wait is 70 us of timer slack and wake-up on Linux): VS_HNSW_SERVICE_SPIN=0 sleeps instead.
    bool spin_ = true;
    std::atomic<size_t> n_pending_{0};
    static constexpr size_t kMaxBatch = 8192;
    static constexpr size_t kZeroCopyBatch = 256;
    static constexpr size_t kHeavyLoad = 512;  // queries in flight beyond which only two slots are used
    struct Slot {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        float* h_q = nullptr;
        float* d_q = nullptr;
        // results of one batch live in ONE block, device and pinned host alike: [keys nb*k u64 | dist nb*k f32 | found nb u32],
        // so a batch costs one copy in and one copy out
        char* d_out = nullptr;
        char* h_out = nullptr;
        uint64_t *d_k = nullptr, *h_k = nullptr;
        float *d_d = nullptr, *h_d = nullptr;
        uint32_t *d_f = nullptr, *h_f = nullptr;
        size_t q_bytes = 0, out_bytes = 0;
        std::vector<SearchReq> reqs;
        bool busy = false;
        int status = VS_OK;
        std::string err;
    };
    int device_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<SearchReq> pending_;
    Slot slots_[kSlots];

    explicit SearchService(int device) : device_(device) {
        if (const char* v = std::getenv("VS_HNSW_SERVICE_SLOTS")) n_slots_ = std::min(kSlots, std::max(1, std::atoi(v)));
        if (const char* v = std::getenv("VS_HNSW_SERVICE_SPIN")) spin_ = v[0] != '0';
        std::thread([this] { run(); }).detach();
    }

    static void grow(Slot& s, size_t nq, size_t dim, size_t k) {
        const size_t qb = nq * dim * 4, ob = nq * k * 12 + nq * 4;
        if (qb > s.q_bytes) {
            graveyard().bury(s.d_q, s.h_q);
            s.q_bytes = qb + qb / 2;
            HIP_OK(hipHostMalloc((void**)&s.h_q, s.q_bytes, hipHostMallocDefault));
            HIP_OK(hipMalloc((void**)&s.d_q, s.q_bytes));
        }
        if (ob > s.out_bytes) {
            graveyard().bury(s.d_out, s.h_out);
            s.out_bytes = ob + ob / 2;
            HIP_OK(hipHostMalloc((void**)&s.h_out, s.out_bytes, hipHostMallocDefault));
            HIP_OK(hipMalloc((void**)&s.d_out, s.out_bytes));
        }
        s.d_k = (uint64_t*)s.d_out;
        s.h_k = (uint64_t*)s.h_out;
        s.d_d = (float*)(s.d_out + nq * k * 8);
        s.h_d = (float*)(s.h_out + nq * k * 8);
        s.d_f = (uint32_t*)(s.d_out + nq * k * 12);
        s.h_f = (uint32_t*)(s.h_out + nq * k * 12);
    }

    void launch(Slot& s) {
        s.status = VS_OK;
        s.err.clear();
        try {
            Engine* e = s.reqs[0].e;
            const size_t nb = s.reqs.size(), k = s.reqs[0].k, dim = e->dim;
            if (!s.st) {
                s.st = device_streams(device_).for_slot((int)(&s - slots_));
                HIP_OK(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
            }
            grow(s, nb, dim, k);
            for (size_t i = 0; i < nb; ++i) std::memcpy(s.h_q + i * dim, s.reqs[i].q.data(), dim * 4);
            size_t load = 0;  // this batch + the batches of the other slots still in flight
            for (int i = 0; i < n_slots_; ++i) load += slots_[i].busy ? slots_[i].reqs.size() : 0;
            // Small batches skip the copy engine: the kernel reads its queries from, and writes its results to, the
            // pinned host block directly (device-mapped) -- 3 KB in and 124 B out per query over PCIe, two API calls
            // and two copy-engine latencies less per launch (the dispatcher thread is what bounds small batches).
            const bool zero_copy = nb <= kZeroCopyBatch;
            if (!zero_copy) HIP_OK(hipMemcpyAsync(s.d_q, s.h_q, nb * dim * 4, hipMemcpyHostToDevice, s.st));
            tl_pipe_no_second = true;  // deliver() serves the rare query the pipelined walk hands over
            struct Reset {
                ~Reset() { tl_pipe_no_second = false; }
            } reset;
            if (zero_copy)
                e->search_device(s.h_q, nb, k, s.h_k, s.h_d, s.h_f, s.st, load);
            else
                e->search_device(s.d_q, nb, k, s.d_k, s.d_d, s.d_f, s.st, load);
            const bool team = e->team_mode == 1 || e->team_mode == 3 || (e->team_mode == 0 && std::max(nb, load) <= 3 * e->team_max_nq);  // 8- or 4-wave teams
            n_batches += 1;
            n_queries += nb;
            if (team) {
                n_team_batches += 1;
                n_team_queries += nb;
            }
            if (!zero_copy) HIP_OK(hipMemcpyAsync(s.h_out, s.d_out, nb * k * 12 + nb * 4, hipMemcpyDeviceToHost, s.st));
            HIP_OK(hipEventRecord(s.ev, s.st));
        } catch (const Fail& f) {
            s.status = f.code;
            s.err = f.msg;
        } catch (const std::exception& x) {
            s.status = VS_ERR_DEVICE;
            s.err = x.what();
        }
    }

    void deliver(Slot& s) {
        const size_t k = s.reqs.empty() ? 0 : s.reqs[0].k;
        for (size_t i = 0; i < s.reqs.size(); ++i) {
            SearchReq& r = s.reqs[i];
            int status = s.status;
            if (status == VS_OK && s.h_f[i] == kWalkFailed) {
                // The walk outgrew its workspace (e.g. fewer live members than the beam after mass removes: `top` never
                // fills and the walk floods the graph): rank exhaustively, as the header promises for every host entry point.
                try {
                    *r.found = r.e->rank_all(r.q.data(), k, r.keys, r.dist);
                    n_ranked_fallbacks += 1;
                } catch (const Fail& f) {
                    status = f.code;
                    s.err = f.msg;
                } catch (const std::exception& x) {
                    status = VS_ERR_DEVICE;
                    s.err = x.what();
                }
            } else if (status == VS_OK && s.h_f[i] == kPipeRedoFound) {
                // two equal distances met where their order matters: the team form of the fused-list kernel answers, as for a batch
                try {
                    tl_no_pipe = true;
                    size_t f = 0;
                    r.e->search_host(r.q.data(), 1, k, r.keys, r.dist, &f, false);
                    tl_no_pipe = false;
                    if (f == (size_t)-1) f = r.e->rank_all(r.q.data(), k, r.keys, r.dist);
                    *r.found = f;
                    n_pipe_redone += 1;
                } catch (const Fail& f) {
                    tl_no_pipe = false;
                    status = f.code;
                    s.err = f.msg;
                } catch (const std::exception& x) {
                    tl_no_pipe = false;
                    status = VS_ERR_DEVICE;
                    s.err = x.what();
                }
            } else if (status == VS_OK) {
                const size_t f = std::min<size_t>(s.h_f[i], k);
                std::memcpy(r.keys, s.h_k + i * k, f * 8);
                std::memcpy(r.dist, s.h_d + i * k, f * 4);
                *r.found = f;
            }
            if (status != VS_OK) {
                *r.found = 0;
                g_async_err = s.err;
                g_err = s.err;  // vs_hnsw_last_error() inside the completion callback
            }
            r.cb(r.ctx, status);
        }
        s.reqs.clear();
    }

    void run() {
        (void)hipSetDevice(device_);
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            bool progressed = false;
            // reap
            for (int si = 0; si < n_slots_; ++si) {
                Slot& s = slots_[si];
                if (!s.busy) continue;
                bool done = s.status != VS_OK || hipEventQuery(s.ev) == hipSuccess;
                if (done) {
                    lk.unlock();
                    deliver(s);
                    lk.lock();
                    s.busy = false;
                    progressed = true;
                }
            }
            // launch
            if (!pending_.empty()) {
                // under load (the non-blocking entry point with thousands of queries in flight) two batches in flight keep
                // the chip full and large batches are the efficient ones: the other slots are for the trickle of lone callers
                size_t in_flight = 0;
                int busy_n = 0;
                for (int si = 0; si < n_slots_; ++si)
                    if (slots_[si].busy) {
                        in_flight += slots_[si].reqs.size();
                        ++busy_n;
                    }
                for (int si = 0; si < n_slots_; ++si) {
                    Slot& s = slots_[si];
                    if (s.busy || pending_.empty()) continue;
                    if (busy_n >= 2 && in_flight >= kHeavyLoad) break;
                    Engine* e = pending_.front().e;
                    const size_t k = pending_.front().k;
                    std::deque<SearchReq> rest;
                    while (!pending_.empty()) {
                        SearchReq& r = pending_.front();
                        if (r.e == e && r.k == k && s.reqs.size() < kMaxBatch) s.reqs.push_back(std::move(r));
                        else rest.push_back(std::move(r));
                        pending_.pop_front();
                    }
                    pending_.swap(rest);
                    n_pending_.store(pending_.size(), std::memory_order_release);
                    s.busy = true;
                    in_flight += s.reqs.size();
                    ++busy_n;
                    lk.unlock();
                    launch(s);
                    lk.lock();
                    progressed = true;
                }
            }
            if (progressed) continue;
            bool any_busy = false, any_free = false;
            {
                size_t in_flight = 0;
                int busy_n = 0;
                for (int si = 0; si < n_slots_; ++si) {
                    any_busy |= slots_[si].busy;
                    any_free |= !slots_[si].busy;
                    if (slots_[si].busy) {
                        in_flight += slots_[si].reqs.size();
                        ++busy_n;
                    }
                }
                if (busy_n >= 2 && in_flight >= kHeavyLoad) any_free = false;  // (no launch before one of them is back)
            }
            if (!any_busy) {
                cv_.wait(lk, [this] { return !pending_.empty(); });
                continue;
            }
            if (!spin_) {
                cv_.wait_for(lk, std::chrono::microseconds(20));
                continue;
            }
            // poll the events of the batches in flight and the queue (`busy` is written by this thread only); a batch
            // that takes longer than 2 ms (large batches of the non-blocking entry point) is slept on in timer steps
            lk.unlock();
            bool woke = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 1; !woke; ++it) {
                if (any_free && n_pending_.load(std::memory_order_acquire) > 0) woke = true;
                for (int si = 0; si < n_slots_ && !woke; ++si)
                    if (slots_[si].busy && (slots_[si].status != VS_OK || hipEventQuery(slots_[si].ev) == hipSuccess)) woke = true;
                if (woke || ((it & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))) break;
                for (int p = 0; p < 16; ++p) __builtin_ia32_pause();
            }
            lk.lock();
            if (!woke) cv_.wait_for(lk, std::chrono::microseconds(50));
        }
    }

   public:
    static thread_local std::string g_async_err;
    // launches / queries, and how many of them went to the team kernel (process-wide; vs_search_service_stats)
    static inline std::atomic<unsigned long long> n_batches{0}, n_team_batches{0}, n_queries{0}, n_team_queries{0};
    static inline std::atomic<unsigned long long> n_pipe_redone{0};  // lone queries the pipelined walk handed over (served by the team kernels)
    static inline std::atomic<unsigned long long> n_ranked_fallbacks{0};  // queries whose walk reported kWalkFailed and were ranked exhaustively
};
thread_local std::string SearchService::g_async_err;

void Engine::search_async(const float* q, size_t k, uint64_t* keys, float* dist, size_t* found,
                          void (*cb)(void*, int), void* ctx) {
    SearchReq r;
    r.e = this;
    r.q.assign(q, q + dim);  // inputs are borrowed for the duration of the call only
    r.k = k;
    r.keys = keys;
    r.dist = dist;
    r.found = found;
    r.cb = cb;
    r.ctx = ctx;
    SearchService::get(device).submit(std::move(r));
}

// A lone plain query through a pod (pipe_pod.hpp): posted to a resident workgroup from the caller's own thread -- no dispatcher hop, no
// launch, no event -- and answered into the caller's pinned block.  false: not served here (no pod free, or not a query the pipelined
// walk takes): the dispatcher serves it.
bool Engine::search_one_pod(const float* q, size_t k, uint64_t* keys, float* dist, size_t* found) {
    uint32_t ef;
    check_search(k, ef);
    // Integer storage (round 5): i8 lone plain queries take the EXACT pipelined walk of filtered queries -- usearch's tie order, a round
    // handed over where two orders could differ (pipe_device.hpp) -- with no filter: every live member is admitted.  The same pods as the
    // filtered rounds serve them.  b1 (a few hundred distinct distances: nearly every walk is handed over) and order_mode 1 stay with the
    // usearch-order walk.  VS_HNSW_INT_PODS=0: as before.
    static const bool int_pods = !(std::getenv("VS_HNSW_INT_PODS") && std::getenv("VS_HNSW_INT_PODS")[0] == '0');
    const bool exact_kind = usearch_order();
    // b1 (round 6): WALK PODS -- the usearch-order team walk itself as a resident kernel (kernels_walk.hip): the same walk as the
    // dispatcher's launch, minus the launch, the stream and the dispatcher's hop.  Its LDS instances: beams up to 256, slots the tags
    // of the instance tell apart, an index whose walks have not been outgrowing the instance.  VS_HNSW_B1_PODS=0: as before.
    static const bool b1_pods = !(std::getenv("VS_HNSW_B1_PODS") && std::getenv("VS_HNSW_B1_PODS")[0] == '0');
    const uint32_t b1_inst = ef <= 128 ? (uint32_t)WALK_LDS_128 : (uint32_t)WALK_LDS_256;
    const bool walk_pod = exact_kind && b1_pods && order_mode == 0 && scalar == VS_SCALAR_B1 && ef <= 256 && walk_domain_override == 0 &&
                          std::max<uint64_t>(slots_atomic.load(std::memory_order_acquire), 1) <= (1ull << walk_instance_domain_bits(b1_inst)) && !lds_walk_bad[b1_inst].load();
    if (exact_kind && !walk_pod && !(int_pods && order_mode == 0 && scalar == VS_SCALAR_I8)) return false;
    if (!pod_pool(device).enabled || needs_global_walk(ef) || (!walk_pod && !pipe_usable(ef)) || team_mode == 2 || team_mode == 3 ||
        stress_small_table || force_wide_tags)
        return false;
    use_device();
    housekeeping();
    const size_t n = slots_atomic.load(std::memory_order_acquire), lay = layout_slots();
    if (!n) return false;
    Lease w(device);
    const size_t space = batch_space_bytes(lay);
    if (w->ws.bytes < space || w->ws_zeroed != (lay + 31) / 32) {
        char* p = (char*)w->ws.ensure(space);
        HIP_OK(hipMemsetAsync(p, 0, w->ws.bytes, w->stream));
        HIP_OK(hipStreamSynchronize(w->stream));
        w->ws_zeroed = (lay + 31) / 32;
    }
    // pinned, device-mapped: [counters, flag 64 B | keys k x 8 | dist k x 4 | the query]
    const size_t q_off = (64 + k * 12 + 63) & ~(size_t)63;
    const size_t pin_need = q_off + (size_t)dim * 4;
    if (w->pin_bytes < pin_need) {
        if (w->pin) graveyard().bury(nullptr, w->pin);
        w->pin = nullptr;
        w->pin_bytes = 0;
        HIP_OK(hipHostMalloc((void**)&w->pin, pin_need, hipHostMallocDefault));
        w->pin_bytes = pin_need;
    }
    uint32_t* h_cnt = (uint32_t*)w->pin;
    uint32_t* h_done = (uint32_t*)w->pin + 8;
    uint64_t* h_k = (uint64_t*)(w->pin + 64);
    float* h_d = (float*)(h_k + k);
    float* h_q = (float*)(w->pin + q_off);
    std::memcpy(h_q, q, (size_t)dim * 4);
    PipeQuery pq{};
    pq.query = h_q;
    pq.slots = (uint32_t)n;
    pq.k = (uint32_t)k;
    pq.round_id = ++w->round_seq ? w->round_seq : ++w->round_seq;
    pq.cnt = h_cnt;
    pq.keys = h_k;
    pq.space = (char*)w->ws.p;
    if (exact_kind) pq.budget = 0xFFFFFFu;  // (no verdict is ever missing: nothing is listed)
    __atomic_store_n(h_done, 0u, __ATOMIC_RELEASE);
    const auto t_in = std::chrono::steady_clock::now();
    PodRelease pod{device, pod_submit(walk_pod ? 4 : exact_kind ? 1 : 0, ef, lay, pq)};
    if (!pod.t) return false;
    const int dbg_pod = pod.t.pod;
    const uint32_t dbg_slot = pod.t.slot;
    const uint64_t dbg_gen = pod.t.gen;
    const double dbg_age_ms = std::chrono::duration<double, std::milli>(t_in - pod_pool(device).pods[dbg_pod].opened).count();
    // (a walk is the better part of a millisecond)
    static std::atomic<int> waiting{0};
    const auto t0 = std::chrono::steady_clock::now();
    bool lost = false;
    uint32_t looks = 0;
    if (!wait_for_device_flag(
            [&] {
                if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == pq.round_id) return true;
                if ((++looks & 1023u) == 0u && pod_pool(device).lost_post(pod.t)) lost = __atomic_load_n(h_done, __ATOMIC_ACQUIRE) != pq.round_id;
                return lost;
            },
            waiting, wait_typical_us[0], 20.0)) {
        w.retire();
        fail(VS_ERR_DEVICE, "a posted query was not answered");
    }
    pod.done();
    if (lost) return false;  // (the dispatcher serves it)
    {
        PodPool& pp = pod_pool(device);
        const auto t_out = std::chrono::steady_clock::now();
        pp.plain_queries.fetch_add(1, std::memory_order_relaxed);
        pp.plain_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_out - t_in).count(), std::memory_order_relaxed);
        pp.plain_wait_ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_out - t0).count(), std::memory_order_relaxed);
        pp.plain_gpu_ticks.fetch_add(h_cnt[4], std::memory_order_relaxed);
        static const bool pod_debug = std::getenv("VS_HNSW_POD_DEBUG") != nullptr;
        const double wait_us = std::chrono::duration<double, std::micro>(t_out - t0).count();
        if (pod_debug && wait_us - h_cnt[4] * 0.01 > 3000.0)
            fprintf(stderr, "[pod] slow answer: waited %.0f us, device %.0f us, pod %d slot %u gen %llu, %.1f ms after the pod was opened\n", wait_us, h_cnt[4] * 0.01,
                    dbg_pod, dbg_slot, (unsigned long long)dbg_gen, dbg_age_ms);
    }
    const uint32_t f = h_cnt[2];
    if (f == kPipeRedoFound || f == kWalkFailed) {
        // two equal distances met where their order matters: the team form of the fused-list kernel answers, as for a batch
        struct NoPipe {
            NoPipe() { tl_no_pipe = true; }
            ~NoPipe() { tl_no_pipe = false; }
        } no_pipe_here;
        size_t ff = 0;
        search_host(q, 1, k, keys, dist, &ff, false);
        if (ff == (size_t)-1) ff = rank_all(q, k, keys, dist);
        *found = ff;
        if (f == kPipeRedoFound) SearchService::n_pipe_redone += 1;
        return true;
    }
    const size_t ff = std::min<size_t>(f, k);
    std::memcpy(keys, h_k, ff * 8);
    std::memcpy(dist, h_d, ff * 4);
    *found = ff;
    return true;
}

int Engine::search_one(const float* q, size_t k, uint64_t* keys, float* dist, size_t* found) {
    if (search_one_pod(q, k, keys, dist, found)) return VS_OK;
    struct Waiter {
        std::mutex m;
        std::condition_variable c;
        bool done = false;
        int status = VS_OK;
        std::string err;
    } w;
    search_async(q, k, keys, dist, found,
                 [](void* p, int status) {
                     Waiter* w = (Waiter*)p;
                     std::lock_guard<std::mutex> g(w->m);
                     w->status = status;
                     if (status != VS_OK) w->err = SearchService::g_async_err;
                     w->done = true;
                     w->c.notify_one();
                 },
                 &w);
    std::unique_lock<std::mutex> lk(w.m);
    w.c.wait(lk, [&] { return w.done; });
    if (w.status != VS_OK) g_err = w.err;
    return w.status;
}


}  // namespace vs
