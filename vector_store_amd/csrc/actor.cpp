// actor.cpp -- libvs_actor.so: the reference's per-index dispatch actor restated in C++ over the C ABI.
// See include/vs_actor.h for the contract and the reference lines each piece mirrors.
#include "../../include/vs_actor.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

constexpr size_t kReserveIncrementGlobal = 1000000;  // usearch.rs:442
constexpr size_t kReserveIncrementLocal = 1000;      // usearch.rs:443

// worker.rs:44-118: bounded queue (channel_size) feeding `workers` threads.
class WorkerPool {
   public:
    WorkerPool(size_t workers, size_t channel) : cap_(channel) {
        for (size_t i = 0; i < workers; ++i) th_.emplace_back([this] { run(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void spawn(std::function<void()> f) {  // blocks while the channel is full (async_channel::bounded)
        std::unique_lock<std::mutex> lk(mu_);
        space_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(f));
        cv_.notify_one();
    }

   private:
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
                space_.notify_one();
            }
            f();
        }
    }
    size_t cap_;
    std::mutex mu_;
    std::condition_variable cv_, space_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};

// usearch.rs:515-624: only one family of operations is in flight at a time.
enum class Mode { Reserve, Insert, Remove, Search };
struct Operation {
    Mode mode = Mode::Insert;
    std::mutex mu;
    std::condition_variable cv;
    size_t counter = 0, max_counter = 0, switches = 0;
    static bool exclusive(Mode m) { return m == Mode::Reserve || m == Mode::Remove; }
    void permit(Mode m) {  // called on the actor thread only
        std::unique_lock<std::mutex> lk(mu);
        if (mode != m) {
            cv.wait(lk, [&] { return counter == 0; });  // the in-flight family drains first
            mode = m;
            ++switches;
        }
        if (exclusive(m)) cv.wait(lk, [&] { return counter == 0; });
        ++counter;
        max_counter = std::max(max_counter, counter);
    }
    void release() {
        std::lock_guard<std::mutex> g(mu);
        if (--counter == 0) cv.notify_all();
    }
};

struct Partition {  // usearch.rs:626-670
    uint64_t id;
    void* idx = nullptr;
    const vs_actor_index_vtable* vt = nullptr;
    bool owned = true;  // false: adopted (vs_actor_adopt_partition), the caller stops it
    std::atomic<size_t> size{0}, capacity{0};
    size_t increment, free_threshold;
    ~Partition() {
        if (idx && owned) vt->stop(idx);  // PartitionState::stop (usearch.rs:667-669) + drop
    }
    bool needs_more_capacity(size_t& want) const {
        size_t cap = capacity.load(), sz = size.load();
        if (cap - sz < free_threshold) {
            want = cap + increment;
            return true;
        }
        return false;
    }
};

struct Msg {
    enum Kind { Add, Remove, RemovePartition, Ann, FilteredAnn, Count } kind;
    uint64_t partition = 0, primary_id = 0;
    std::vector<float> v;
    size_t k = 0;
    vs_hnsw_predicate pred = nullptr;
    void* pctx = nullptr;
    uint64_t filter_key = 0;
    uint64_t* keys = nullptr;
    float* dist = nullptr;
    size_t* found = nullptr;
    std::shared_ptr<std::promise<int>> tx;  // oneshot (searches; modify messages: the in-progress marker, value = applied)
};

// `impl UsearchIndex for ThreadedUsearchIndex` (usearch.rs:162-251) = the HIP engine's C ABI
int hip_create(const vs_hnsw_options* o, void** out) { return vs_hnsw_create(o, (vs_hnsw**)out); }
void hip_stop(void* h) { vs_hnsw_free((vs_hnsw*)h); }
int hip_reserve(void* h, size_t cap, size_t threads) { return vs_hnsw_reserve((vs_hnsw*)h, cap, threads); }
size_t hip_capacity(void* h) { return vs_hnsw_capacity((vs_hnsw*)h); }
int hip_add(void* h, uint64_t key, const float* v, size_t dim) { return vs_hnsw_add((vs_hnsw*)h, key, v, dim); }
int hip_remove(void* h, uint64_t key, int* removed) { return vs_hnsw_remove((vs_hnsw*)h, key, removed); }
int hip_search(void* h, const float* q, size_t dim, size_t k, uint64_t* keys, float* d, size_t* found) {
    return vs_hnsw_search((vs_hnsw*)h, q, dim, k, keys, d, found);
}
int hip_filtered(void* h, const float* q, size_t dim, size_t k, vs_hnsw_predicate p, void* ctx, uint64_t* keys, float* d, size_t* found) {
    return vs_hnsw_filtered_search((vs_hnsw*)h, q, dim, k, p, ctx, keys, d, found);
}
int hip_filtered_keyed(void* h, const float* q, size_t dim, size_t k, vs_hnsw_predicate p, void* ctx, uint64_t fk, uint64_t* keys, float* d,
                       size_t* found) {
    return vs_hnsw_filtered_search_keyed((vs_hnsw*)h, q, dim, k, p, ctx, fk, keys, d, found);
}
const vs_actor_index_vtable kHipIndex = {hip_create, hip_stop,   hip_reserve,  hip_capacity,       hip_add,
                                         hip_remove, hip_search, hip_filtered, vs_hnsw_last_error, hip_filtered_keyed};

}  // namespace

struct vs_actor {
    vs_actor_options opt;
    vs_actor_index_vtable vt;
    size_t workers, channel;
    std::unique_ptr<WorkerPool> pool;
    Operation op;
    std::mutex mu;
    std::condition_variable cv, space_search, space_modify;
    std::deque<Msg> q_search, q_modify;
    bool closed = false;
    std::thread th;
    std::map<uint64_t, std::shared_ptr<Partition>> partitions;
    std::mutex part_mu;  // partitions map is owned by the actor thread; introspection takes this
    std::atomic<size_t> index_size{0};
    std::atomic<int> allocate_can{1};
    int allocate_prev = 1;
    std::atomic<uint64_t> c_adds{0}, c_dropped{0}, c_reserves{0}, c_searches{0}, c_removes{0}, c_errors{0};
    std::mutex err_mu;
    std::string last_index_error;

    void send(Msg&& m, bool search) {
        std::unique_lock<std::mutex> lk(mu);
        auto& q = search ? q_search : q_modify;
        (search ? space_search : space_modify).wait(lk, [&] { return closed || q.size() < channel; });
        if (closed) {
            if (m.tx) m.tx->set_value(VS_ERR_INVALID_ARGUMENT);
            return;
        }
        q.push_back(std::move(m));
        cv.notify_one();
    }

    bool recv(Msg& out) {  // vs_index/mod.rs:30-45: search first
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return closed || !q_search.empty() || !q_modify.empty(); });
        if (!q_search.empty()) {
            out = std::move(q_search.front());
            q_search.pop_front();
            space_search.notify_one();
            return true;
        }
        if (!q_modify.empty()) {
            out = std::move(q_modify.front());
            q_modify.pop_front();
            space_modify.notify_one();
            return true;
        }
        return false;  // closed and drained
    }

    std::shared_ptr<Partition> find(uint64_t id) {
        std::lock_guard<std::mutex> g(part_mu);  // (vs_actor_adopt_partition inserts from a caller's thread)
        auto it = partitions.find(id);
        return it == partitions.end() ? nullptr : it->second;
    }

    void run() {
        Msg msg;
        while (recv(msg)) {
            // check_memory_allocation (usearch.rs:1156-1177): only AddVector is gated
            if (msg.kind == Msg::Add) {
                int can = allocate_can.load();
                bool drop = !can;
                allocate_prev = can;
                if (drop) {
                    ++c_dropped;
                    if (msg.tx) msg.tx->set_value(0);
                    continue;
                }
            }
            // preprocess (usearch.rs:743-895)
            std::shared_ptr<Partition> part;
            switch (msg.kind) {
                case Msg::Add:
                    part = find(msg.partition);
                    if (!part) {  // lazily created on the first AddVector of the partition
                        void* h = nullptr;
                        if (vt.create(&opt.index, &h) != VS_OK) {
                            ++c_errors;
                            if (msg.tx) msg.tx->set_value(0);
                            continue;
                        }
                        part = new_partition(msg.partition, h, true);
                        std::lock_guard<std::mutex> g(part_mu);
                        partitions[msg.partition] = part;
                    }
                    break;
                case Msg::Remove:
                    part = find(msg.partition);
                    if (!part) {
                        if (msg.tx) msg.tx->set_value(0);
                        continue;
                    }
                    break;
                case Msg::RemovePartition: {
                    part = find(msg.partition);
                    if (part) {
                        op.permit(Mode::Remove);  // nothing of this partition may be in flight
                        index_size -= part->size.load();
                        {
                            std::lock_guard<std::mutex> g(part_mu);
                            partitions.erase(msg.partition);
                        }
                        op.release();
                    }
                    continue;
                }
                case Msg::Ann:
                case Msg::FilteredAnn:
                    part = find(msg.partition);
                    if (!part) {  // unknown / empty partition => Ok(([], []))  (usearch.rs:787-802)
                        *msg.found = 0;
                        msg.tx->set_value(VS_OK);
                        continue;
                    }
                    break;
                case Msg::Count:
                    *msg.found = index_size.load();
                    msg.tx->set_value(VS_OK);
                    continue;
            }
            // dispatch_task (usearch.rs:897-943)
            size_t want = 0;
            if (msg.kind == Msg::Add && part->needs_more_capacity(want)) {
                op.permit(Mode::Reserve);
                if (part->needs_more_capacity(want)) {
                    pool->spawn([this, part, want] {
                        if (vt.reserve(part->idx, want, workers) == VS_OK)
                            part->capacity = vt.capacity(part->idx);
                        else
                            ++c_errors;  // error!("unable to reserve index capacity ...")
                        ++c_reserves;
                        op.release();
                    });
                } else {
                    op.release();
                }
            }
            Mode m = msg.kind == Msg::Add ? Mode::Insert : msg.kind == Msg::Remove ? Mode::Remove : Mode::Search;
            op.permit(m);
            auto shared = std::make_shared<Msg>(std::move(msg));
            pool->spawn([this, part, shared] {
                process(*part, *shared);
                op.release();
            });
        }
        pool.reset();  // joins the workers after the queue drained
        std::lock_guard<std::mutex> g(part_mu);
        partitions.clear();
    }

    std::shared_ptr<Partition> new_partition(uint64_t id, void* h, bool owned) {
        auto part = std::make_shared<Partition>();
        part->id = id;
        part->idx = h;
        part->vt = &vt;
        part->owned = owned;
        part->increment = opt.reserve_increment ? opt.reserve_increment : (opt.local ? kReserveIncrementLocal : kReserveIncrementGlobal);
        // The reference uses perf::channel_size() (3 x workers, usearch.rs:650), although up to
        // channel + workers adds can be in flight; the margin here covers all of them so that
        // no add can meet "Reserve capacity ahead of insertions!".
        part->free_threshold = channel + workers + 1;
        return part;
    }

    void process(Partition& p, Msg& m) {  // usearch.rs:950-1002
        const size_t dim = opt.index.dimensions;
        switch (m.kind) {
            case Msg::Add: {
                const bool ok = vt.add(p.idx, m.primary_id, m.v.data(), m.v.size()) == VS_OK;
                if (ok) {
                    ++p.size;
                    ++index_size;
                    ++c_adds;
                } else {
                    ++c_errors;  // warn!("add: unable to add embedding"), the vector is not indexed
                }
                if (m.tx) m.tx->set_value(ok ? 1 : 0);  // (the in-progress marker drops with the message)
                break;
            }
            case Msg::Remove: {
                int removed = 0;
                if (vt.remove(p.idx, m.primary_id, &removed) != VS_OK) {
                    ++c_errors;
                    removed = 0;
                } else if (removed) {
                    --p.size;
                    --index_size;
                    ++c_removes;
                }
                if (m.tx) m.tx->set_value(removed ? 1 : 0);
                break;
            }
            case Msg::Ann:
            case Msg::FilteredAnn: {
                int rc;
                if (m.v.size() != dim) {  // validate_dimensions (usearch.rs:1051-1065) -> HTTP 400 upstream
                    rc = VS_ERR_DIMENSION;
                    g_err = "wrong embedding dimension";
                    *m.found = 0;
                } else if (m.kind == Msg::Ann) {
                    rc = vt.search(p.idx, m.v.data(), dim, m.k, m.keys, m.dist, m.found);
                } else if (m.filter_key && vt.filtered_search_keyed) {
                    rc = vt.filtered_search_keyed(p.idx, m.v.data(), dim, m.k, m.pred, m.pctx, m.filter_key, m.keys, m.dist, m.found);
                } else {
                    rc = vt.filtered_search(p.idx, m.v.data(), dim, m.k, m.pred, m.pctx, m.keys, m.dist, m.found);
                }
                if (rc != VS_OK && rc != VS_ERR_DIMENSION) {
                    std::lock_guard<std::mutex> g(err_mu);
                    last_index_error = vt.last_error();  // (thread-local in the index: read on the thread that made the call)
                }
                ++c_searches;
                m.tx->set_value(rc);
                break;
            }
            default: break;
        }
    }
};

extern "C" {

const char* vs_actor_last_error(void) { return g_err.c_str(); }

int vs_actor_create(const vs_actor_options* o, vs_actor** out) { return vs_actor_create_with(o, &kHipIndex, out); }

int vs_actor_create_with(const vs_actor_options* o, const vs_actor_index_vtable* vt, vs_actor** out) {
    if (!o || !out || !o->index.dimensions || !vt || !vt->create || !vt->stop || !vt->reserve || !vt->capacity || !vt->add || !vt->remove ||
        !vt->search || !vt->filtered_search || !vt->last_error) {
        g_err = "invalid options";
        return VS_ERR_INVALID_ARGUMENT;
    }
    try {
        std::unique_ptr<vs_actor> a(new vs_actor());
        a->opt = *o;
        a->vt = *vt;
        a->workers = o->workers ? o->workers : std::max(1u, std::thread::hardware_concurrency());
        a->channel = a->workers * 3;  // perf.rs:20-25
        a->pool.reset(new WorkerPool(a->workers, a->channel));
        vs_actor* raw = a.get();
        a->th = std::thread([raw] { raw->run(); });
        *out = a.release();
        return VS_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return VS_ERR_DEVICE;
    }
}

void vs_actor_stop(vs_actor* a) {
    if (!a) return;
    {
        std::lock_guard<std::mutex> g(a->mu);
        a->closed = true;
    }
    a->cv.notify_all();
    a->space_search.notify_all();
    a->space_modify.notify_all();
    if (a->th.joinable()) a->th.join();
    delete a;
}

int vs_actor_add_vector(vs_actor* a, uint64_t partition, uint64_t primary_id, const float* v, size_t dim) {
    if (!a || !v) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::Add;
    m.partition = partition;
    m.primary_id = primary_id;
    m.v.assign(v, v + dim);
    a->send(std::move(m), false);
    return VS_OK;
}

int vs_actor_remove_vector(vs_actor* a, uint64_t partition, uint64_t primary_id) {
    if (!a) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::Remove;
    m.partition = partition;
    m.primary_id = primary_id;
    a->send(std::move(m), false);
    return VS_OK;
}

int vs_actor_adopt_partition(vs_actor* a, uint64_t partition, void* index, size_t size) {
    if (!a || !index) return VS_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(a->part_mu);
    if (a->partitions.count(partition)) {
        g_err = "the partition exists";
        return VS_ERR_INVALID_ARGUMENT;
    }
    auto part = a->new_partition(partition, index, false);
    part->size = size;
    part->capacity = a->vt.capacity(index);
    a->partitions[partition] = part;
    a->index_size += size;
    return VS_OK;
}

static int modify_wait(vs_actor* a, Msg&& m, int* applied) {
    m.tx = std::make_shared<std::promise<int>>();
    auto fut = m.tx->get_future();
    a->send(std::move(m), false);
    const int v = fut.get();
    if (applied) *applied = v > 0 ? 1 : 0;
    return v < 0 ? v : VS_OK;
}

int vs_actor_add_vector_wait(vs_actor* a, uint64_t partition, uint64_t primary_id, const float* v, size_t dim, int* applied) {
    if (!a || !v) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::Add;
    m.partition = partition;
    m.primary_id = primary_id;
    m.v.assign(v, v + dim);
    return modify_wait(a, std::move(m), applied);
}

int vs_actor_remove_vector_wait(vs_actor* a, uint64_t partition, uint64_t primary_id, int* applied) {
    if (!a) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::Remove;
    m.partition = partition;
    m.primary_id = primary_id;
    return modify_wait(a, std::move(m), applied);
}

int vs_actor_remove_partition(vs_actor* a, uint64_t partition) {
    if (!a) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::RemovePartition;
    m.partition = partition;
    a->send(std::move(m), false);
    return VS_OK;
}

static int round_trip(vs_actor* a, Msg&& m) {
    m.tx = std::make_shared<std::promise<int>>();
    auto fut = m.tx->get_future();
    a->send(std::move(m), true);
    int rc = fut.get();
    if (rc != VS_OK && rc != VS_ERR_DIMENSION) {
        std::lock_guard<std::mutex> g(a->err_mu);
        g_err = a->last_index_error;
    }
    return rc;
}

int vs_actor_ann(vs_actor* a, uint64_t partition, const float* q, size_t dim, size_t k, uint64_t* keys, float* dist,
                 size_t* found) {
    if (!a || !q || !keys || !dist || !found || !k) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::Ann;
    m.partition = partition;
    m.v.assign(q, q + dim);
    m.k = k;
    m.keys = keys;
    m.dist = dist;
    m.found = found;
    return round_trip(a, std::move(m));
}

int vs_actor_filtered_ann(vs_actor* a, uint64_t partition, const float* q, size_t dim, size_t k, vs_hnsw_predicate pred,
                          void* ctx, uint64_t* keys, float* dist, size_t* found) {
    if (!a || !q || !keys || !dist || !found || !k || !pred) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::FilteredAnn;
    m.partition = partition;
    m.v.assign(q, q + dim);
    m.k = k;
    m.pred = pred;
    m.pctx = ctx;
    m.keys = keys;
    m.dist = dist;
    m.found = found;
    return round_trip(a, std::move(m));
}

int vs_actor_filtered_ann_keyed(vs_actor* a, uint64_t partition, const float* q, size_t dim, size_t k, vs_hnsw_predicate pred, void* ctx,
                                uint64_t filter_key, uint64_t* keys, float* dist, size_t* found) {
    if (!a || !q || !keys || !dist || !found || !k || !pred) return VS_ERR_INVALID_ARGUMENT;
    Msg m;
    m.kind = Msg::FilteredAnn;
    m.partition = partition;
    m.v.assign(q, q + dim);
    m.k = k;
    m.pred = pred;
    m.pctx = ctx;
    m.filter_key = filter_key;
    m.keys = keys;
    m.dist = dist;
    m.found = found;
    return round_trip(a, std::move(m));
}

size_t vs_actor_count(vs_actor* a) {
    if (!a) return 0;
    size_t n = 0;
    Msg m;
    m.kind = Msg::Count;
    m.found = &n;
    round_trip(a, std::move(m));
    return n;
}

void vs_actor_set_allocate(vs_actor* a, int can) {
    if (a) a->allocate_can = can ? 1 : 0;
}

size_t vs_actor_partition_capacity(vs_actor* a, uint64_t partition) {
    if (!a) return 0;
    std::lock_guard<std::mutex> g(a->part_mu);
    auto it = a->partitions.find(partition);
    return it == a->partitions.end() ? 0 : it->second->capacity.load();
}

size_t vs_actor_partitions(vs_actor* a) {
    if (!a) return 0;
    std::lock_guard<std::mutex> g(a->part_mu);
    return a->partitions.size();
}

void vs_actor_counters(vs_actor* a, uint64_t out[8]) {
    out[0] = a->c_adds;
    out[1] = a->c_dropped;
    out[2] = a->c_reserves;
    out[3] = a->c_searches;
    out[4] = a->c_removes;
    std::lock_guard<std::mutex> g(a->op.mu);
    out[5] = a->op.switches;
    out[6] = a->op.max_counter;
    out[7] = a->c_errors;
}

}  // extern "C"
