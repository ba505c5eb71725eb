// ranks.cpp -- libvs_ranks.so: key-range shards, one process per GPU, per-shard top-k exchanged by ONE RCCL
// ncclAllGather per batch and merged on every rank (include/vs_ranks.h; BASELINE.json configs[3]).
#include "../../include/vs_ranks.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
thread_local std::string g_err;
constexpr uint64_t kIdxMask = (1ull << 48) - 1;
constexpr uint32_t kWalkFailedFound = 0xFFFFFFFFu;  // vs_hnsw_search_batch_device: "the walk outgrew its workspace, no answer"

// One thread per (query, position): rows of this rank's block that carry no answer -- the whole block when the local
// search could not be launched (all = 1), else the queries flagged kWalkFailed -- are filled with (free key, +inf) before
// they are gathered, so no rank ever merges what a previous batch left in the buffer; *failed counts such queries.
__global__ void blank_unanswered_rows(uint64_t* keys, float* dist, const uint32_t* found, uint32_t nq, uint32_t k, int all,
                                      unsigned long long* failed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * k) return;
    const uint32_t q = i / k;
    if (all || found[q] == kWalkFailedFound) {
        keys[i] = ~0ull;
        dist[i] = __builtin_inff();
        if (i % k == 0) atomicAdd(failed, 1ull);
    }
}

struct Fail {
    int code;
    std::string msg;
};
#define HIP_OK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess) throw Fail{VS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)}; \
    } while (0)
#define NCCL_OK(expr)                                                                            \
    do {                                                                                         \
        ncclResult_t e__ = (expr);                                                               \
        if (e__ != ncclSuccess) throw Fail{VS_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(e__)}; \
    } while (0)
#define VS_OK_OR_THROW(expr)                                              \
    do {                                                                  \
        int e__ = (expr);                                                 \
        if (e__ != VS_OK) throw Fail{e__, std::string(vs_hnsw_last_error())}; \
    } while (0)

template <class F>
int guarded(F&& f) {
    try {
        f();
        return VS_OK;
    } catch (const Fail& x) {
        g_err = x.msg;
        return x.code;
    } catch (const std::exception& x) {
        g_err = x.what();
        return VS_ERR_DEVICE;
    }
}
}  // namespace

// VS_RANKS_HOSTSHM: the all-gather through a POSIX shared-memory segment instead of RCCL -- for ranks that share ONE
// device (RCCL refuses to pair those: "duplicate GPU"), i.e. the one-GPU boxes the tests run on.  Everything around the
// exchange is the product path unchanged: the walk writes this rank's packed block in place, the exchange fills the other
// ranks' blocks of the same receive buffer on the communicator stream, the packed merge follows, two slots pipeline it.
// Per slot and generation g: wait until every rank has consumed generation g - 1 of the slot, copy the block out (D2H),
// publish "arrived", wait for every rank's arrival, copy the other blocks in (H2D), publish "consumed".  The waits run
// as host functions on the communicator stream and give up after kWaitSeconds (the handle then reports the failure).
struct HostExchange {
    static constexpr size_t kCtlBytes = 8192, kBlockCap = 8u << 20, kMaxWorld = 64;
    static constexpr int kWaitSeconds = 60;
    struct Ctl {
        std::atomic<uint64_t> arrived[2][kMaxWorld], consumed[2][kMaxWorld];
        std::atomic<uint32_t> attached;
    };
    static_assert(sizeof(Ctl) <= kCtlBytes, "control block");
    std::string name;
    char* base = nullptr;
    size_t bytes = 0;
    int rank = 0, world = 1;
    bool registered = false;
    uint64_t gen[2] = {0, 0};
    std::atomic<bool> failed{false};

    Ctl* ctl() const { return reinterpret_cast<Ctl*>(base); }
    char* block(int slot, int r) const { return base + kCtlBytes + ((size_t)slot * (size_t)world + (size_t)r) * kBlockCap; }

    void open(const uint8_t* id, int rank_, int world_) {
        rank = rank_;
        world = world_;
        if (world > (int)kMaxWorld) throw Fail{VS_ERR_UNSUPPORTED, "host exchange: at most 64 ranks"};
        char hex[40];
        for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", id[(i * 7 + 3) % VS_RANKS_ID_BYTES] ^ id[i]);
        name = std::string("/vs_ranks_") + hex;
        bytes = kCtlBytes + 2 * (size_t)world * kBlockCap;
        int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0) throw Fail{VS_ERR_DEVICE, "shm_open " + name + " failed"};
        if (ftruncate(fd, (off_t)bytes) != 0) {
            close(fd);
            throw Fail{VS_ERR_OUT_OF_MEMORY, "ftruncate of the exchange segment failed"};
        }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) throw Fail{VS_ERR_OUT_OF_MEMORY, "mmap of the exchange segment failed"};
        base = (char*)p;
        registered = hipHostRegister(base, bytes, hipHostRegisterDefault) == hipSuccess;  // pageable copies work too, only slower
        ctl()->attached.fetch_add(1);
        // collective, like ncclCommInitRank: every rank of the world has attached when this returns
        const auto t0 = std::chrono::steady_clock::now();
        while (ctl()->attached.load(std::memory_order_acquire) < (uint32_t)world) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kWaitSeconds)) {
                close_segment();
                throw Fail{VS_ERR_DEVICE, "host exchange: the other ranks did not attach within " + std::to_string(kWaitSeconds) + " s"};
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    void close_segment() {
        if (!base) return;
        if (registered) (void)hipHostUnregister(base);
        const bool last = ctl()->attached.fetch_sub(1) == 1;
        munmap(base, bytes);
        base = nullptr;
        if (last || rank == 0) shm_unlink(name.c_str());
    }

    struct Wait {
        HostExchange* x;
        int slot, what;  // 0: every rank consumed generation g - 1; 1: publish arrival, wait for everyone's; 2: publish consumption
        uint64_t g;
    };
    static void host_step(void* p) {
        Wait* w = (Wait*)p;
        HostExchange& x = *w->x;
        Ctl* c = x.ctl();
        auto all_at_least = [&](std::atomic<uint64_t>* v, uint64_t want) {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                bool ok = true;
                for (int r = 0; r < x.world; ++r) ok = ok && v[r].load(std::memory_order_acquire) >= want;
                if (ok) return;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kWaitSeconds)) {
                    x.failed = true;
                    return;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
        };
        if (w->what == 0) all_at_least(c->consumed[w->slot], w->g);
        else if (w->what == 1) {
            c->arrived[w->slot][x.rank].store(w->g + 1, std::memory_order_release);
            all_at_least(c->arrived[w->slot], w->g + 1);
        } else c->consumed[w->slot][x.rank].store(w->g + 1, std::memory_order_release);
        delete w;
    }
    void all_gather(int slot, char* gathered, size_t block_bytes, hipStream_t st) {
        if (failed.load()) throw Fail{VS_ERR_DEVICE, "host exchange: a rank did not arrive within " + std::to_string(kWaitSeconds) + " s"};
        if (block_bytes > kBlockCap) throw Fail{VS_ERR_UNSUPPORTED, "host exchange: batch above 8 MiB per rank"};
        const uint64_t g = gen[slot]++;
        HIP_OK(hipLaunchHostFunc(st, host_step, new Wait{this, slot, 0, g}));
        HIP_OK(hipMemcpyAsync(block(slot, rank), gathered + (size_t)rank * block_bytes, block_bytes, hipMemcpyDeviceToHost, st));
        HIP_OK(hipLaunchHostFunc(st, host_step, new Wait{this, slot, 1, g}));
        for (int r = 0; r < world; ++r)
            if (r != rank) HIP_OK(hipMemcpyAsync(gathered + (size_t)r * block_bytes, block(slot, r), block_bytes, hipMemcpyHostToDevice, st));
        HIP_OK(hipLaunchHostFunc(st, host_step, new Wait{this, slot, 2, g}));
    }
};

struct vs_ranks {
    vs_hnsw* shard = nullptr;
    int exchange = VS_RANKS_RCCL;
    HostExchange host;
    int rank = 0, world = 1;
    uint64_t total_rows = 0, per = 1;
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;
    unsigned long long* d_failed = nullptr;  // queries this rank contributed no answer to (see blank_unanswered_rows)
    struct Slot {
        char* gathered = nullptr;  // world x block bytes; this rank's block is written by its walk, in place
        uint32_t* local_found = nullptr;
        size_t bytes = 0, found_n = 0;
        hipEvent_t walked = nullptr, merged = nullptr;
    } slot[2];

    void ensure(Slot& s, size_t block, size_t nq) {
        const size_t want = block * (size_t)world;
        if (want > s.bytes) {
            if (s.gathered) HIP_OK(hipFree(s.gathered));
            s.gathered = nullptr;
            HIP_OK(hipMalloc((void**)&s.gathered, want + want / 4));
            s.bytes = want + want / 4;
        }
        if (nq > s.found_n) {
            if (s.local_found) HIP_OK(hipFree(s.local_found));
            s.local_found = nullptr;
            HIP_OK(hipMalloc((void**)&s.local_found, (nq + nq / 4) * 4));
            s.found_n = nq + nq / 4;
        }
        if (!s.walked) {
            HIP_OK(hipEventCreateWithFlags(&s.walked, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&s.merged, hipEventDisableTiming));
        }
    }

    // walk (or exact search) on `st` into this rank's block -> all-gather + merge on comm_stream
    void submit(int si, bool exact, const float* d_q, size_t nq, size_t dim, size_t k, uint64_t* d_keys, float* d_dist,
                uint32_t* d_found, hipStream_t st) {
        if (si < 0 || si > 1) throw Fail{VS_ERR_INVALID_ARGUMENT, "slot must be 0 or 1"};
        if (!nq || !k) throw Fail{VS_ERR_INVALID_ARGUMENT, "empty batch"};
        Slot& s = slot[si];
        const size_t block = (nq * k * 12 + 15) / 16 * 16;  // [keys u64 | distances f32], 16-byte aligned blocks
        if (s.merged) HIP_OK(hipStreamWaitEvent(st, s.merged, 0));  // this slot's previous merge has read the blocks the walk is about to rewrite
        ensure(s, block, nq);
        char* mine = s.gathered + (size_t)rank * block;
        // A rank whose local search fails must still enter the collective -- the others are already on their way into it
        // and would wait for ever -- so the failure is kept, an empty block is gathered, and the error is reported after
        // the all-gather and the merge are enqueued.
        int local_rc = exact ? vs_hnsw_exact_search_batch_device(shard, d_q, nq, dim, k, (uint64_t*)mine, (float*)(mine + nq * k * 8),
                                                                 s.local_found, st)
                             : vs_hnsw_search_batch_device(shard, d_q, nq, dim, k, (uint64_t*)mine, (float*)(mine + nq * k * 8),
                                                           s.local_found, st);
        std::string local_err = local_rc == VS_OK ? std::string() : std::string(vs_hnsw_last_error());
        const uint32_t cells = (uint32_t)(nq * k);
        hipLaunchKernelGGL(blank_unanswered_rows, dim3((cells + 255) / 256), dim3(256), 0, st, (uint64_t*)mine, (float*)(mine + nq * k * 8),
                           s.local_found, (uint32_t)nq, (uint32_t)k, local_rc == VS_OK ? 0 : 1, d_failed);
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventRecord(s.walked, st));
        HIP_OK(hipStreamWaitEvent(comm_stream, s.walked, 0));
        // (a world of one enters the collective too: RCCL's initialisation, the in-place all-gather and its ordering against the walk and
        // the merge run on every box, not only where several GPUs are)
        if (world > 1 && exchange == VS_RANKS_HOSTSHM) host.all_gather(si, s.gathered, block, comm_stream);
        else if (comm) NCCL_OK(ncclAllGather(mine, s.gathered, block, ncclChar, comm, comm_stream));  // in place: sendbuff = recvbuff + rank * count
        VS_OK_OR_THROW(vs_topk_merge_packed_device(s.gathered, (size_t)world, block, nq, k, d_keys, d_dist, d_found, comm_stream));
        HIP_OK(hipEventRecord(s.merged, comm_stream));
        if (local_rc != VS_OK) throw Fail{local_rc, "local shard search failed (an empty block was gathered so that the other ranks go on): " + local_err};
    }
};

extern "C" {

const char* vs_ranks_last_error(void) { return g_err.c_str(); }

int vs_ranks_unique_id(uint8_t id[VS_RANKS_ID_BYTES]) {
    return guarded([&] {
        static_assert(sizeof(ncclUniqueId) == VS_RANKS_ID_BYTES, "ncclUniqueId size");
        ncclUniqueId u;
        NCCL_OK(ncclGetUniqueId(&u));
        std::memcpy(id, &u, sizeof u);
    });
}

int vs_ranks_create(vs_hnsw* shard, int rank, int world, const uint8_t id[VS_RANKS_ID_BYTES], uint64_t total_rows, vs_ranks** out) {
    const char* x = std::getenv("VS_RANKS_EXCHANGE");
    return vs_ranks_create_ex(shard, rank, world, id, total_rows, x && !std::strcmp(x, "hostshm") ? VS_RANKS_HOSTSHM : VS_RANKS_RCCL, out);
}

int vs_ranks_create_ex(vs_hnsw* shard, int rank, int world, const uint8_t id[VS_RANKS_ID_BYTES], uint64_t total_rows, int exchange,
                       vs_ranks** out) {
    return guarded([&] {
        if (exchange != VS_RANKS_RCCL && exchange != VS_RANKS_HOSTSHM) throw Fail{VS_ERR_INVALID_ARGUMENT, "unknown exchange"};
        if (!shard || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id))
            throw Fail{VS_ERR_INVALID_ARGUMENT, "invalid argument"};
        vs_ranks* r = new vs_ranks();
        r->shard = shard;
        r->rank = rank;
        r->world = world;
        r->exchange = exchange;
        r->total_rows = total_rows ? total_rows : 1;
        r->per = (r->total_rows + (uint64_t)world - 1) / (uint64_t)world;
        try {
            HIP_OK(hipStreamCreateWithFlags(&r->comm_stream, hipStreamNonBlocking));
            HIP_OK(hipMalloc((void**)&r->d_failed, 8));
            HIP_OK(hipMemset(r->d_failed, 0, 8));
            if (world > 1 && exchange == VS_RANKS_HOSTSHM) {
                r->host.open(id, rank, world);
            } else if (exchange == VS_RANKS_RCCL) {
                if (world > 1) {
                    ncclUniqueId u;
                    std::memcpy(&u, id, sizeof u);
                    NCCL_OK(ncclCommInitRank(&r->comm, world, u, rank));
                } else {
                    // A world of one runs the collective too -- but it does not NEED it (the block is gathered in place): where RCCL
                    // cannot initialise (a container without shared memory or a network interface) the handle serves without a
                    // communicator, says so once, and reports rccl_ranks = 0 (advisor finding, round 4: it used to fail there).
                    ncclUniqueId u;
                    ncclResult_t e = id ? ncclSuccess : ncclGetUniqueId(&u);
                    if (id) std::memcpy(&u, id, sizeof u);
                    if (e == ncclSuccess) e = ncclCommInitRank(&r->comm, 1, u, 0);
                    if (e != ncclSuccess) {
                        r->comm = nullptr;
                        std::fprintf(stderr, "[vs_ranks] a world of one could not initialise RCCL (%s): serving without a communicator\n", ncclGetErrorString(e));
                    }
                }
            }
        } catch (...) {
            vs_ranks_free(r);
            throw;
        }
        *out = r;
    });
}

void vs_ranks_free(vs_ranks* r) {
    if (!r) return;
    if (r->comm_stream) (void)hipStreamSynchronize(r->comm_stream);
    for (auto& s : r->slot) {
        if (s.gathered) (void)hipFree(s.gathered);
        if (s.local_found) (void)hipFree(s.local_found);
        if (s.walked) (void)hipEventDestroy(s.walked);
        if (s.merged) (void)hipEventDestroy(s.merged);
    }
    r->host.close_segment();
    if (r->d_failed) (void)hipFree(r->d_failed);
    if (r->comm) (void)ncclCommDestroy(r->comm);
    if (r->comm_stream) (void)hipStreamDestroy(r->comm_stream);
    delete r;
}

int vs_ranks_world(const vs_ranks* r, int* rank, int* world, int* comm_ranks) {
    return guarded([&] {
        if (!r) throw Fail{VS_ERR_INVALID_ARGUMENT, "null handle"};
        if (rank) *rank = r->rank;
        if (world) *world = r->world;
        if (comm_ranks) {
            int n = 1;
            if (r->comm) NCCL_OK(ncclCommCount(r->comm, &n));
            else if (r->host.base) n = (int)r->host.ctl()->attached.load();
            *comm_ranks = n;
        }
    });
}

int vs_ranks_exchange_kind(const vs_ranks* r) { return r ? r->exchange : -1; }

int vs_ranks_rccl_ranks(const vs_ranks* r, int* n) {
    return guarded([&] {
        if (!r || !n) throw Fail{VS_ERR_INVALID_ARGUMENT, "null argument"};
        *n = 0;
        if (r->comm) NCCL_OK(ncclCommCount(r->comm, n));
    });
}

int vs_ranks_unanswered(vs_ranks* r, uint64_t* queries) {
    return guarded([&] {
        if (!r || !queries) throw Fail{VS_ERR_INVALID_ARGUMENT, "null argument"};
        HIP_OK(hipDeviceSynchronize());
        unsigned long long v = 0;
        HIP_OK(hipMemcpy(&v, r->d_failed, 8, hipMemcpyDeviceToHost));
        *queries = v;
    });
}

int vs_ranks_owner(const vs_ranks* r, uint64_t key) {
    if (!r) return -1;
    const uint64_t o = (key & kIdxMask) / r->per;
    return (int)(o < (uint64_t)r->world ? o : (uint64_t)r->world - 1);
}

void vs_ranks_range(const vs_ranks* r, uint64_t* first_row, uint64_t* end_row) {
    if (!r) return;
    const uint64_t lo = std::min(r->total_rows, (uint64_t)r->rank * r->per);
    if (first_row) *first_row = lo;
    if (end_row) *end_row = std::min(r->total_rows, lo + r->per);
}

int vs_ranks_add_batch(vs_ranks* r, const uint64_t* keys, const float* vectors, size_t n, size_t dim, size_t* added) {
    return guarded([&] {
        if (!r || (n && (!keys || !vectors))) throw Fail{VS_ERR_INVALID_ARGUMENT, "null argument"};
        std::vector<uint64_t> mk;
        std::vector<float> mv;
        bool all = true;
        for (size_t i = 0; i < n; ++i) all = all && vs_ranks_owner(r, keys[i]) == r->rank;
        if (all) {  // the usual case: the caller already routes by range
            if (n) VS_OK_OR_THROW(vs_hnsw_add_batch(r->shard, keys, vectors, n, dim));
            if (added) *added = n;
            return;
        }
        for (size_t i = 0; i < n; ++i)
            if (vs_ranks_owner(r, keys[i]) == r->rank) {
                mk.push_back(keys[i]);
                mv.insert(mv.end(), vectors + i * dim, vectors + (i + 1) * dim);
            }
        if (!mk.empty()) VS_OK_OR_THROW(vs_hnsw_add_batch(r->shard, mk.data(), mv.data(), mk.size(), dim));
        if (added) *added = mk.size();
    });
}

int vs_ranks_search_submit(vs_ranks* r, int slot, const float* d_q, size_t nq, size_t dim, size_t k, uint64_t* d_keys, float* d_dist,
                           uint32_t* d_found, void* stream) {
    return guarded([&] {
        if (!r || !d_q || !d_keys || !d_dist) throw Fail{VS_ERR_INVALID_ARGUMENT, "null argument"};
        r->submit(slot, false, d_q, nq, dim, k, d_keys, d_dist, d_found, (hipStream_t)stream);
    });
}

int vs_ranks_wait(vs_ranks* r, int slot, void* stream) {
    return guarded([&] {
        if (!r || slot < 0 || slot > 1) throw Fail{VS_ERR_INVALID_ARGUMENT, "invalid argument"};
        if (r->slot[slot].merged) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, r->slot[slot].merged, 0));
        if (r->host.failed.load()) throw Fail{VS_ERR_DEVICE, "host exchange: a rank did not arrive in time"};
    });
}

int vs_ranks_search_batch_device(vs_ranks* r, const float* d_q, size_t nq, size_t dim, size_t k, uint64_t* d_keys, float* d_dist,
                                 uint32_t* d_found, void* stream) {
    return guarded([&] {
        if (!r || !d_q || !d_keys || !d_dist) throw Fail{VS_ERR_INVALID_ARGUMENT, "null argument"};
        r->submit(0, false, d_q, nq, dim, k, d_keys, d_dist, d_found, (hipStream_t)stream);
        HIP_OK(hipStreamWaitEvent((hipStream_t)stream, r->slot[0].merged, 0));
    });
}

int vs_ranks_exact_search_batch_device(vs_ranks* r, const float* d_q, size_t nq, size_t dim, size_t k, uint64_t* d_keys,
                                       float* d_dist, uint32_t* d_found, void* stream) {
    return guarded([&] {
        if (!r || !d_q || !d_keys || !d_dist) throw Fail{VS_ERR_INVALID_ARGUMENT, "null argument"};
        r->submit(0, true, d_q, nq, dim, k, d_keys, d_dist, d_found, (hipStream_t)stream);
        HIP_OK(hipStreamWaitEvent((hipStream_t)stream, r->slot[0].merged, 0));
    });
}

}  // extern "C"
