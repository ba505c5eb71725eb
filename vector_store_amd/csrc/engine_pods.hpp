// engine_pods.hpp -- part of the engine's single translation unit (included by engine.hip, in this order: engine_base.hpp,
// engine_pods.hpp, the Engine itself in engine.hip, engine_service.hpp, engine_abi.hpp).  Pods: resident launches of the pipelined walk that blocking callers post their queries to (pipe_pod.hpp): the pool, holds, freezes.
#pragma once

namespace vs {

// ---- pods (pipe_pod.hpp): resident launches of the pipelined walk that blocking callers post their queries to ---------------------------
// Up to kPods per device, each on a stream of its own (a pod never ends while it has callers: nothing else may queue behind it) and
// kSlots workgroups wide -- 3 x 64 leaves a quarter of the chip to everything else, and every workgroup of an open pod is resident (a
// workgroup that is not cannot poll its slot).  With the engine's 16 streams and the process's default stream that is the 20 hardware
// queues the library asks for (HwQueuesDefault).  A keeper thread advances the pods' heartbeat, closes a pod that has been idle for
// VS_HNSW_POD_IDLE_US (20 ms) or open for VS_HNSW_POD_AGE_MS (1 s, 2 s when no other pod is free to take its callers over: a
// device-wide synchronisation anywhere in the process -- hipFree, hipDeviceSynchronize -- waits for every kernel, pods included), and
// hands a closed pod back once every workgroup has said it left.  It makes no HIP call: it cannot be held up by one.
struct Pod {
    enum State { kFree, kOpen, kClosing };
    static constexpr uint32_t kSlots = 64;
    State state = kFree;
    const void* owner = nullptr;  // the index whose view the launch carries
    int mode = 0;                 // 0 = plain lone queries, 1 = filtered queries (exact walks and exploring rounds: the slot says which)
    uint32_t efcap = 0;           // 256 / 512: the kernel instance
    size_t index_slots = 0;       // what the callers' visited bitmaps are laid out for: the index's CAPACITY when the pod was opened
    bool frozen = false;          // its index is being modified: no posts (the caller launches instead); the view in `ctl` is rewritten before it thaws
    hipStream_t st = nullptr;
    PodCtl* ctl = nullptr;        // pinned
    PodSlot* slots = nullptr;     // pinned
    PipeQuery* stage = nullptr;   // device: workgroup b's copy of the query it is answering (what a batch launch reads from its table)
    uint32_t n = 0, n_busy = 0;
    uint64_t gen = 0;
    bool busy[kSlots] = {};
    uint32_t seq[kSlots] = {};
    std::chrono::steady_clock::time_point opened, last_used;
};
struct PodTicket {
    int pod = -1;
    uint32_t slot = 0;
    uint64_t gen = 0;  // which opening of the pod
    explicit operator bool() const { return pod >= 0; }
};
struct PodPool {
    static constexpr int kPods = 3;
    std::mutex mu;
    std::condition_variable keeper_cv;
    Pod pods[kPods];
    bool keeper_started = false;
    bool enabled = true;
    int holds = 0;  // > 0: somebody is about to synchronise the device (reserve, stats, export, a drop): no pod may open until it is through
    uint32_t n_slots = Pod::kSlots;
    int idle_us = 20000, max_age_ms = 1000;
    std::atomic<uint64_t> n_opened{0}, n_served{0}, n_closed{0};
    std::atomic<uint64_t> plain_queries{0}, plain_ns{0}, plain_wait_ns{0}, plain_gpu_ticks{0};  // where a posted plain query's time goes (probes)
    PodPool() {
        if (const char* v = std::getenv("VS_HNSW_PODS")) enabled = v[0] != '0';
        if (const char* v = std::getenv("VS_HNSW_POD_SLOTS")) n_slots = (uint32_t)std::min<int>(Pod::kSlots, std::max(1, std::atoi(v)));
        if (const char* v = std::getenv("VS_HNSW_POD_IDLE_US")) idle_us = std::max(100, std::atoi(v));
        if (const char* v = std::getenv("VS_HNSW_POD_AGE_MS")) max_age_ms = std::max(10, std::atoi(v));
    }
    void close_locked(Pod& p) {
        p.state = Pod::kClosing;
        __atomic_store_n(&p.ctl->closed, 1u, __ATOMIC_SEQ_CST);
        n_closed.fetch_add(1, std::memory_order_relaxed);
    }
    static bool all_left(const Pod& p) {
        for (uint32_t i = 0; i < p.n; ++i)
            if (!__atomic_load_n(&p.slots[i].left, __ATOMIC_ACQUIRE)) return false;
        return true;
    }
    void keep() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            bool any = false;
            const auto now = std::chrono::steady_clock::now();
            int free_pods = 0;
            for (Pod& p : pods) free_pods += p.state == Pod::kFree ? 1 : 0;
            for (Pod& p : pods) {
                if (p.state != Pod::kFree) __atomic_fetch_add(&p.ctl->heartbeat, 1u, __ATOMIC_RELAXED);
                // (a busy pod that has reached its age goes once a free pod can take its callers over -- or at twice the age)
                const auto age = now - p.opened;
                const bool aged = age > std::chrono::milliseconds(max_age_ms) && (p.n_busy == 0 || free_pods > 0 || age > std::chrono::milliseconds(2 * max_age_ms));
                if (p.state == Pod::kOpen && ((p.n_busy == 0 && now - p.last_used > std::chrono::microseconds(idle_us)) || aged)) {
                    close_locked(p);
                    if (p.n_busy) --free_pods;  // (its callers will open one)
                }
                if (p.state == Pod::kClosing && p.n_busy == 0 && all_left(p)) p.state = Pod::kFree;
                any |= p.state != Pod::kFree;
            }
            if (!any) {
                keeper_cv.wait(lk);
                continue;
            }
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            lk.lock();
        }
    }
    // Close the pods of one index (nullptr: all of them) and wait until their workgroups have left: before anything that moves the
    // index's arenas, and before a device-wide synchronisation.  quiesce(nullptr) is called under a Hold: with callers posting to
    // any index of the device, a pod freed here would be reopened by the next caller before the others have gone, and the three would
    // never be free together (advisor finding, round 4).
    void quiesce(const void* owner) {
        std::unique_lock<std::mutex> lk(mu);
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            bool pending = false;
            for (Pod& p : pods) {
                if (p.state == Pod::kFree || (owner && p.owner != owner)) continue;
                if (p.state == Pod::kOpen) close_locked(p);
                if (p.n_busy == 0 && all_left(p)) p.state = Pod::kFree;
                else pending = true;
            }
            if (!pending) return;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) fail(VS_ERR_DEVICE, "a pod of resident walks did not close");
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(20));
            lk.lock();
        }
    }
    void release(PodTicket t) {
        std::lock_guard<std::mutex> g(mu);
        Pod& p = pods[t.pod];
        if (p.gen != t.gen) return;
        p.busy[t.slot] = false;
        --p.n_busy;
        p.last_used = std::chrono::steady_clock::now();
    }
    // A query was posted to a slot whose workgroup had decided to leave just before (a pod whose host stood still for seconds ends
    // by itself: kernels_pipe.hip): the workgroup stores `left` as its last act and touches nothing afterwards, so "left, and no
    // answer" means the post was never seen -- the caller serves the query by a launch (advisor finding, round 4: it used to wait out
    // 20 s and fail).  The pod is closed.
    bool lost_post(PodTicket t) {
        std::lock_guard<std::mutex> g(mu);
        Pod& p = pods[t.pod];
        if (p.gen != t.gen || !p.slots) return false;
        if (!__atomic_load_n(&p.slots[t.slot].left, __ATOMIC_ACQUIRE)) return false;
        if (p.state == Pod::kOpen) close_locked(p);
        return true;
    }
    bool all_free() {
        std::lock_guard<std::mutex> g(mu);
        for (const Pod& p : pods)
            if (p.state != Pod::kFree) return false;
        return true;
    }
    // An index is being modified (adds, removes: they never overlap its searches, usearch.rs:590-612): its pods take no posts and have
    // no query in flight while it lasts; thaw() hands them the new entry point / top level / removed flag (PodCtl) -- or closes them,
    // when what they were launched with no longer holds (`layout`: the capacity their callers' workspaces are laid out for).
    void freeze(const void* owner) {
        std::unique_lock<std::mutex> lk(mu);
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            bool pending = false;
            for (Pod& p : pods) {
                if (p.state == Pod::kFree || p.owner != owner) continue;
                p.frozen = true;
                if (p.n_busy) pending = true;
            }
            if (!pending) return;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) fail(VS_ERR_DEVICE, "queries posted to a pod did not finish");
            lk.unlock();
            std::this_thread::sleep_for(std::chrono::microseconds(10));
            lk.lock();
        }
    }
    void thaw(const void* owner, size_t layout, uint32_t entry_slot, int32_t max_level, uint32_t has_removed) {
        std::lock_guard<std::mutex> g(mu);
        for (Pod& p : pods) {
            if (p.owner != owner || p.state == Pod::kFree) continue;  // (every open pod of the owner, frozen or not, gets the new view)
            if (p.state == Pod::kOpen && p.index_slots != layout) close_locked(p);
            if (p.state == Pod::kOpen) {
                __atomic_store_n(&p.ctl->entry_slot, entry_slot, __ATOMIC_RELAXED);
                __atomic_store_n(&p.ctl->max_level, max_level, __ATOMIC_RELAXED);
                __atomic_store_n(&p.ctl->has_removed, has_removed, __ATOMIC_RELEASE);  // (a post's release store follows before any workgroup looks)
                p.last_used = std::chrono::steady_clock::now();
            }
            p.frozen = false;
        }
    }
};
// Taken around a device-wide synchronisation (and whatever it protects: an arena move, frees): closes every pod of the device, keeps
// them closed -- pod_submit answers "no pod" meanwhile and its caller launches as before pods existed -- and frees what was parked.
struct PodHold {
    PodPool& pp;
    explicit PodHold(PodPool& pool) : pp(pool) {
        {
            std::lock_guard<std::mutex> g(pp.mu);
            ++pp.holds;
        }
        try {
            pp.quiesce(nullptr);
            // (parked blocks go NOW, not only when the hold ends: reserve reads its HBM budget under the hold, and would count them as used)
            graveyard().drain();
        } catch (...) {
            std::lock_guard<std::mutex> g(pp.mu);
            --pp.holds;
            throw;
        }
    }
    ~PodHold() {
        graveyard().drain();  // (no pod is open: these frees wait for ordinary kernels only)
        std::lock_guard<std::mutex> g(pp.mu);
        --pp.holds;
    }
};
static PodPool& pod_pool(int dev) {
    static std::mutex mu;
    static std::unordered_map<int, PodPool*> all;  // leaked with the process (the keeper thread outlives static teardown)
    std::lock_guard<std::mutex> g(mu);
    PodPool*& p = all[dev];
    if (!p) {
        p = new PodPool();
        static std::once_flag at_exit;
        std::call_once(at_exit, [] {
            std::atexit([] {  // the runtime's teardown waits for every kernel: tell the pods to go first
                std::lock_guard<std::mutex> g2(mu);
                for (auto& kv : all) {
                    try {
                        kv.second->enabled = false;
                        kv.second->quiesce(nullptr);
                    } catch (...) {
                    }
                }
            });
        });
    }
    return *p;
}

// Parked blocks are freed when no pod is open on the device and none can open meanwhile (a free then waits for ordinary kernels only).
static void drain_graveyard_if_idle(int dev) {
    if (g_buried.load(std::memory_order_relaxed) == 0 || graveyard().n.load(std::memory_order_relaxed) == 0) return;
    PodPool& pp = pod_pool(dev);
    {
        std::lock_guard<std::mutex> g(pp.mu);
        for (const Pod& p : pp.pods)
            if (p.state != Pod::kFree) return;
        ++pp.holds;
    }
    graveyard().drain();
    std::lock_guard<std::mutex> g(pp.mu);
    --pp.holds;
}


}  // namespace vs
