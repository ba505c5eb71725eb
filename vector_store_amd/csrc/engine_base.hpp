// engine_base.hpp -- part of the engine's single translation unit (included by engine.hip, in this order: engine_base.hpp,
// engine_pods.hpp, the Engine itself in engine.hip, engine_service.hpp, engine_abi.hpp).  Errors, HBM arenas that grow in place, parked frees, leased streams + scratch, waiting for device flags.
#pragma once

namespace vs {


// Filtered searches run one walk launch per caller on the caller's own stream (the predicate is the caller's), and the
// reference runs every filtered query on a blocking thread (usearch.rs:937-948): dozens of small kernels must be able to run
// side by side.  ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels that share a
// queue run one after the other: 17 callers got 2.8 x one caller's rate.  The library therefore asks for 20 queues unless the
// process has chosen a value -- effective when it is loaded before the HIP runtime initialises (a Rust service linking it; the
// Python binding and bench.py set the variable themselves before touching the GPU).  Measured at 2M x 768, 10 % selective
// filter, 17 blocking callers: 35 -> 210 queries/s (scripts/probe/filtered_probe.py).
// 20, not more: the device has 24 hardware queue slots for user queues; once a process holds more queues than that (24 of
// its own + the runtime's internal ones) the hardware scheduler time-slices them, and EVERY kernel of the process runs
// ~20 % slower from then on, busy queue or idle (scripts/probe/aftermath_probe.py: the 10,000-query batch kernel 13.5 -> 16.3 ms
// after 17 filtered callers had each opened their stream; 12 / 16 / 20 queues: unchanged).
struct HwQueuesDefault {
    HwQueuesDefault() { setenv("GPU_MAX_HW_QUEUES", "20", 0); }
};
static HwQueuesDefault g_hw_queues_default;

thread_local std::string g_err;

struct Fail {
    int code;
    std::string msg;
};
[[noreturn]] static void fail(int code, std::string msg) { throw Fail{code, std::move(msg)}; }
#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) ::vs::fail(VS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// hipFree / hipHostFree synchronise the whole device: with resident pods (below) they block until every open pod's kernel ends -- up to a
// pod's age limit, a second or two -- and the buffers that regrow sit on search paths (a larger batch, a larger k, another index's
// first query).  A buffer that is replaced while the process runs is therefore PARKED here and freed when no pod is open on any device
// (drain_graveyard, called where a stall costs nothing: under a pod hold, and by leases taken while every pod is free).
static std::atomic<size_t> g_buried{0};  // blocks parked on any device (a cheap "anything to do?" for the search paths)
struct Graveyard {
    std::mutex mu;
    std::vector<void*> dev, host;
    std::atomic<size_t> n{0};
    void bury(void* device_ptr, void* host_ptr) {
        std::lock_guard<std::mutex> g(mu);
        if (device_ptr) dev.push_back(device_ptr);
        if (host_ptr) host.push_back(host_ptr);
        g_buried.fetch_add((device_ptr ? 1 : 0) + (host_ptr ? 1 : 0), std::memory_order_relaxed);
        n.store(dev.size() + host.size(), std::memory_order_relaxed);
    }
    void drain() {  // the caller knows that no pod is open (or accepts the wait)
        std::vector<void*> d, h;
        {
            std::lock_guard<std::mutex> g(mu);
            d.swap(dev);
            h.swap(host);
            n.store(0, std::memory_order_relaxed);
            g_buried.fetch_sub(d.size() + h.size(), std::memory_order_relaxed);
        }
        for (void* p : d) (void)hipFree(p);
        for (void* p : h) (void)hipHostFree(p);
    }
};
static Graveyard& graveyard() {  // of the calling thread's current device (a free synchronises the device the block lives on)
    static std::mutex mu;
    static std::unordered_map<int, Graveyard*> all;  // leaked with the process
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    Graveyard*& p = all[dev];
    if (!p) p = new Graveyard();
    return *p;
}

struct DeviceBuf {  // grow-only device scratch
    void* p = nullptr;
    size_t bytes = 0;
    void* ensure(size_t n) {
        if (n > bytes) {
            if (p) graveyard().bury(p, nullptr);
            p = nullptr;
            bytes = 0;
            size_t want = n + n / 4 + 256;
            HIP_OK(hipMalloc(&p, want));
            bytes = want;
        }
        return p;
    }
    ~DeviceBuf() {
        if (p) (void)hipFree(p);
    }
};

// HBM arena that grows in place.  Above kVmmThreshold the arena is a reserved virtual range with physical chunks
// mapped behind it on demand (hipMemAddressReserve / hipMemCreate / hipMemMap): growing maps one more chunk --
// no second copy of the index in HBM while it grows, no D2D copy, stable addresses -- so an index can keep
// following the reference's "+1,000,000" growth policy (usearch.rs:442, :908-921) up to the whole 288 GB.
// When the virtual range itself runs out, the SAME physical chunks are remapped into a range twice as large
// (still no copy).  Small arenas (thousands of per-partition indexes) stay on hipMalloc + copy.
constexpr size_t kVmmThreshold = 64ull << 20;
struct Arena {
    void* base = nullptr;
    size_t bytes = 0;  // usable bytes behind base
    bool vmm = false;
    size_t va_bytes = 0;
    struct Chunk {
        hipMemGenericAllocationHandle_t h;
        size_t off, size;
    };
    std::vector<Chunk> chunks;
    static inline std::atomic<unsigned long long> copied_bytes{0};  // D2D bytes moved by growth (process-wide)

    static bool vmm_supported(int device) {
        static std::mutex mu;
        static std::unordered_map<int, int> cache;
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(device);
        if (it != cache.end()) return it->second != 0;
        int v = 0;
        const char* off = std::getenv("VS_HNSW_NO_VMM");
        if (!(off && off[0] == '1') &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeVirtualMemoryManagementSupported, device) != hipSuccess)
            v = 0;
        if (off && off[0] == '1') v = 0;
        cache[device] = v;
        return v != 0;
    }
    static hipMemAllocationProp prop(int device) {
        hipMemAllocationProp p = {};
        p.type = hipMemAllocationTypePinned;
        p.location.type = hipMemLocationTypeDevice;
        p.location.id = device;
        return p;
    }
    static size_t granularity(int device) {  // the runtime reports 4 KiB; chunks are kept 2 MiB-aligned (large-page friendly)
        hipMemAllocationProp p = prop(device);
        size_t g = 0;
        HIP_OK(hipMemGetAllocationGranularity(&g, &p, hipMemAllocationGranularityRecommended));
        return std::max<size_t>(g, 2ull << 20);
    }
    static size_t round_up(size_t v, size_t g) { return (v + g - 1) / g * g; }

    // Additional HBM that resize(want) would take.
    size_t extra_needed(size_t want, int device) const {
        want = std::max<size_t>(want, 1);
        if (vmm) return want > bytes ? want - bytes : 0;
        if (want >= kVmmThreshold && vmm_supported(device)) return want;  // transition: old block freed after the copy
        return want == bytes ? 0 : want;
    }

    void set_access(char* p, size_t n, int device) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        HIP_OK(hipMemSetAccess(p, n, &acc, 1));
    }
    void map_chunk(size_t size, int device) {  // at the end of the mapped range
        hipMemAllocationProp p = prop(device);
        Chunk c{};
        c.off = bytes;
        c.size = size;
        hipError_t e = hipMemCreate(&c.h, size, &p, 0);
        if (e == hipErrorOutOfMemory) ::vs::fail(VS_ERR_OUT_OF_MEMORY, "not enough HBM to grow the index");
        HIP_OK(e);
        e = hipMemMap((char*)base + c.off, size, 0, c.h, 0);
        if (e != hipSuccess) {
            (void)hipMemRelease(c.h);
            HIP_OK(e);
        }
        chunks.push_back(c);
        bytes += size;
        // Access is always (re)set over the WHOLE mapped range: on ROCm 7.2 hipMemSetAccess on the sub-range of a
        // later chunk intermittently returns "invalid argument" (scripts/probe/vmm_probe3.cpp: 20 x 12 growth steps
        // fail with sub-ranges, pass with the whole range, with and without remapping).
        try {
            set_access((char*)base, bytes, device);
        } catch (...) {  // leave the arena as it was
            (void)hipMemUnmap((char*)base + c.off, size);
            (void)hipMemRelease(c.h);
            chunks.pop_back();
            bytes -= size;
            throw;
        }
    }

    // Make [0, want) usable; the first `keep` bytes survive.  Returns the (possibly new) base.
    void* resize(size_t want, size_t keep, int device) {
        want = std::max<size_t>(want, 1);
        if (!vmm && !(want >= kVmmThreshold && vmm_supported(device))) {  // plain block + copy
            if (want == bytes && base) return base;
            void* np = nullptr;
            hipError_t e = hipMalloc(&np, want);
            if (e == hipErrorOutOfMemory) ::vs::fail(VS_ERR_OUT_OF_MEMORY, "not enough HBM to grow the index");
            HIP_OK(e);
            keep = std::min(keep, std::min(want, bytes));
            if (base && keep) {
                HIP_OK(hipMemcpy(np, base, keep, hipMemcpyDeviceToDevice));
                copied_bytes += keep;
            }
            if (base) graveyard().bury(base, nullptr);  // (freed when no pod is open: a free synchronises the device)
            base = np;
            bytes = want;
            return base;
        }
        const size_t g = granularity(device);
        const size_t need = round_up(want, g);
        if (!vmm) {  // first time above the threshold: move the plain block behind a virtual range
            void* old = base;
            const size_t old_bytes = bytes;
            void* va = nullptr;
            const size_t vb = round_up(std::max<size_t>(2 * need, 256ull << 20), g);
            HIP_OK(hipMemAddressReserve(&va, vb, 0, nullptr, 0));
            base = va;
            va_bytes = vb;
            bytes = 0;
            vmm = true;
            try {
                map_chunk(need, device);
            } catch (...) {
                (void)hipMemAddressFree(va, vb);
                base = old;
                bytes = old_bytes;
                va_bytes = 0;
                vmm = false;
                throw;
            }
            keep = std::min(keep, std::min(want, old_bytes));
            if (old && keep) {
                HIP_OK(hipMemcpy(base, old, keep, hipMemcpyDeviceToDevice));
                copied_bytes += keep;
            }
            if (old) graveyard().bury(old, nullptr);
            return base;
        }
        if (need < bytes) {  // give whole chunks beyond the new end back
            while (!chunks.empty() && chunks.back().off >= need) {
                Chunk c = chunks.back();
                HIP_OK(hipMemUnmap((char*)base + c.off, c.size));
                HIP_OK(hipMemRelease(c.h));
                bytes = c.off;
                chunks.pop_back();
            }
            return base;
        }
        if (need == bytes) return base;
        if (need > va_bytes) {  // remap the same physical chunks into a larger range: no copy
            void* va = nullptr;
            const size_t vb = round_up(2 * need, g);
            HIP_OK(hipMemAddressReserve(&va, vb, 0, nullptr, 0));
            for (const Chunk& c : chunks) {
                HIP_OK(hipMemUnmap((char*)base + c.off, c.size));
                HIP_OK(hipMemMap((char*)va + c.off, c.size, 0, c.h, 0));
            }
            if (bytes) set_access((char*)va, bytes, device);
            HIP_OK(hipMemAddressFree(base, va_bytes));
            base = va;
            va_bytes = vb;
        }
        map_chunk(need - bytes, device);
        return base;
    }

    void release() {
        if (vmm) {
            for (const Chunk& c : chunks) {
                (void)hipMemUnmap((char*)base + c.off, c.size);
                (void)hipMemRelease(c.h);
            }
            if (base) (void)hipMemAddressFree(base, va_bytes);
        } else if (base) {
            (void)hipFree(base);
        }
        base = nullptr;
        bytes = va_bytes = 0;
        chunks.clear();
        vmm = false;
    }
};

// Stream + scratch leased per host call.  Shared by every index on the device, so thousands
// of per-partition handles (reference usearch.rs:704-705,766-778) do not each own a stream.
struct WorkCtx {
    hipStream_t stream = nullptr;
    DeviceBuf a, b, c, d, e, f;
    // filtered search with a lazily evaluated predicate (Engine::filtered_lazy): pinned staging for the lists and verdicts
    // of a round, and the event a crowd of callers sleeps on
    char* pin = nullptr;
    size_t pin_bytes = 0;
    hipEvent_t ev = nullptr;
    // batched rounds (Engine::filtered_batched): this context's visited bitmap / log / spill slots -- all zero between rounds --, and
    // what it was laid out for
    DeviceBuf ws;
    size_t ws_zeroed = 0;  // bitmap words the workspace is laid out (and all zero) for: the visited log behind the bitmap holds slot numbers, and a
                           // bitmap that grows by a word over them would read those as visited members
    uint32_t round_seq = 0;
    uint32_t ask_dirty = 1u << 17;  // verdict bytes of `pin` that may be non-zero (filtered_ask reads them as "answered" flags and clears them first)
};

// The engine's streams on one device: a fixed set, shared by the leased contexts (from index 0 up) and the single-query
// dispatcher's pipeline slots (from the top down).  The process has GPU_MAX_HW_QUEUES hardware queues (20, see HwQueuesDefault);
// streams beyond that share queues, and two walks on one queue take turns even when other queues are idle -- with a stream per
// context AND per slot, 17 filtered callers after some unfiltered traffic made 26 streams and ran at 329 QPS where a fresh
// process runs at 430.  Contexts beyond the set share a stream with an earlier one (their launches then run in stream order,
// each on its own buffers).  VS_HNSW_STREAMS: 4..20 (default 16: room for the caller's own streams).
// The single-query dispatcher looks at every answer on the host anyway: it asks for no second-chance launch behind the pipelined walk
// (tl_pipe_no_second) and serves a handed-over query itself, with the pipelined walk off (tl_no_pipe).
static thread_local bool tl_pipe_no_second = false, tl_no_pipe = false;
static std::atomic<uint64_t> g_streams_created{0};  // HIP streams the engine has made in this process (every device): a fixed set per device, never one per index
struct DeviceStreams {
    static constexpr int kMax = 20;
    std::mutex mu;
    hipStream_t st[kMax] = {};
    int count = 16;
    unsigned next_ctx = 0;
    DeviceStreams() {
        if (const char* v = std::getenv("VS_HNSW_STREAMS")) count = std::min(kMax, std::max(4, std::atoi(v)));
    }
    hipStream_t at(int i) {  // mu held
        i = ((i % count) + count) % count;
        if (!st[i]) {
            HIP_OK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
            g_streams_created.fetch_add(1, std::memory_order_relaxed);
        }
        return st[i];
    }
    hipStream_t for_new_context() {
        std::lock_guard<std::mutex> g(mu);
        return at((int)(next_ctx++));
    }
    hipStream_t round_robin() {  // batched filtered rounds: launches take the device's streams in turn
        std::lock_guard<std::mutex> g(mu);
        return at((int)(next_rr++));
    }
    unsigned next_rr = 0;
    hipStream_t for_slot(int slot) {
        std::lock_guard<std::mutex> g(mu);
        return at(count - 1 - slot);
    }
};
static DeviceStreams& device_streams(int dev) {
    static std::mutex mu;
    static std::unordered_map<int, std::unique_ptr<DeviceStreams>> all;  // leaked with the process
    std::lock_guard<std::mutex> g(mu);
    auto& p = all[dev];
    if (!p) p.reset(new DeviceStreams());
    return *p;
}

struct DevicePool {
    std::mutex mu;
    std::vector<std::unique_ptr<WorkCtx>> idle;
};
static DevicePool& pool(int dev) {
    static std::mutex mu;
    static std::unordered_map<int, std::unique_ptr<DevicePool>> pools;
    std::lock_guard<std::mutex> g(mu);
    auto& p = pools[dev];
    if (!p) p.reset(new DevicePool());
    return *p;
}
struct Lease {
    DevicePool& pl;
    std::unique_ptr<WorkCtx> ctx;
    explicit Lease(int dev) : pl(pool(dev)) {
        {
            std::lock_guard<std::mutex> g(pl.mu);
            if (!pl.idle.empty()) {
                ctx = std::move(pl.idle.back());
                pl.idle.pop_back();
            }
        }
        if (!ctx) {
            ctx.reset(new WorkCtx());
            ctx->stream = device_streams(dev).for_new_context();  // the context keeps it: its buffers are only ever used in this stream's order
        }
    }
    ~Lease() {
        if (!ctx) return;
        std::lock_guard<std::mutex> g(pl.mu);
        pl.idle.push_back(std::move(ctx));
    }
    // A wait on the device timed out: whatever was launched or posted may still write into this context's buffers, so it never goes
    // back to the pool (leaked, with its buffers: the caller is about to report a device failure anyway).
    void retire() { (void)ctx.release(); }
    WorkCtx* operator->() { return ctx.get(); }
};

// Cores this process may keep busy: the hardware's, or the cgroup's CPU quota when that is less (a container's quota is enforced per
// 100 ms period: a crowd of callers that polls its way through the quota is stopped -- every thread at once -- until the period ends).
static int usable_cores() {
    static const int n = [] {
        int hw = (int)std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && quota[0] != 'm') hw = std::min<int>(hw, std::max<long>(1, std::atol(quota) / period));
            std::fclose(f);
        } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            long quota = -1, period = 100000;
            if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
            std::fclose(g);
            if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(h, "%ld", &period) != 1) period = 100000;
                std::fclose(h);
            }
            if (quota > 0 && period > 0) hw = std::min<int>(hw, std::max<long>(1, quota / period));
        }
        return hw;
    }();
    return n;
}
// A caller waiting for a flag the device sets (pinned memory): spinning while the waiting callers are few against the cores, otherwise
// asleep for most of what such a wait took lately (`typical_us`, a moving average the caller keeps), then awake for the last stretch
// (up to three waiters per core), then in short steps.  false: `limit_s` seconds have passed.
template <class Ready>
static bool wait_for_device_flag(Ready ready, std::atomic<int>& waiting, std::atomic<uint32_t>& typical_us, double limit_s) {
    struct Count {
        std::atomic<int>& w;
        explicit Count(std::atomic<int>& x) : w(x) { w.fetch_add(1, std::memory_order_relaxed); }
        ~Count() { w.fetch_sub(1, std::memory_order_relaxed); }
    } count(waiting);
    const auto t0 = std::chrono::steady_clock::now();
    const auto gone_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    const int cores = usable_cores();
    const int spin_below = std::max(1, cores / 2);
    bool slept = false;
    for (uint32_t it = 0;; ++it) {
        if (ready()) break;
        const int w = waiting.load(std::memory_order_relaxed);
        if (w <= spin_below) {
            for (int p = 0; p < 8; ++p) __builtin_ia32_pause();
        } else {
            const uint32_t typ = typical_us.load(std::memory_order_relaxed);
            if (!slept && typ > 200u) {
                // the bulk of the expected wait asleep (a sleep ends ~60 us late: the timer's slack and the wake-up) ...
                slept = true;
                const double left = 0.85 * typ - 60.0 - gone_us();
                if (left >= 20.0) std::this_thread::sleep_for(std::chrono::microseconds((uint32_t)left));
                continue;
            }
            // ... and the last stretch awake (round 5): a waiter is here for about a sixth of its wait, so three waiters per core keep half
            // of the cores busy -- and none of them sleeps 20-80 us past its answer (17 callers on 16 cores: 71 us per query).  Past
            // 1.25 x the usual wait, or with more waiters than that, short sleeps as before.
            if (slept && w <= 3 * cores && ((it & 15u) != 15u || gone_us() < 1.25 * typ)) {
                for (int p = 0; p < 8; ++p) __builtin_ia32_pause();
            } else {
                slept = true;
                std::this_thread::sleep_for(std::chrono::microseconds(std::max<uint32_t>(20u, typ / 12u)));
            }
        }
        if ((it & 63u) == 63u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) return false;
    }
    const uint32_t took = (uint32_t)std::min<double>(1e7, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    const uint32_t typ = typical_us.load(std::memory_order_relaxed);
    // (down fast, up slowly: on an index whose filters differ in selectivity a short round must not sleep through most of a long one's time)
    typical_us.store(!typ ? took : took < typ ? (typ + took) / 2u : (typ * 7u + took) / 8u, std::memory_order_relaxed);
    return true;
}


}  // namespace vs
