// vs_bench -- native benchmark driver over the C ABI (include/vs_hnsw.h): the counterpart of the
// reference's `vector-search-benchmark` (crates/benchmark/src/main.rs) for the engine alone.
//
//   vs_bench selftest
//   vs_bench gen          --data-dir D --n N --dim D [--queries Q] [--neighbors K] [--dist lowrank|gaussian]
//                         [--rank R] [--metric cos|l2sq|ip] [--seed S]
//   vs_bench build-index  --data-dir D [--metric M] [--connectivity C] [--expansion-add E]
//                         [--concurrency C [--max-vectors N]]   (one vector per call from C threads)
//   vs_bench search       --data-dir D --limit K --duration SEC --concurrency C [--inflight D] [--expansion-search E]
//                         [--metric M] [--connectivity C] [--expansion-add E]
//   vs_bench search-http  --data-dir D --limit K --duration SEC --concurrency C [--host H] [--port P]
//                         [--keyspace K] [--index I]      (against a running vs_httpd / vector-store: POST .../ann)
//
// `search` is the reference's search-http / search-cql loop (main.rs:435-525): `concurrency` workers
// each pick a random query and issue ONE query per call (vs_hnsw_search, as client.ann does), recording
// latency into a 10,000-bucket histogram and recall against query.ibin.  The reference measures a
// running service; here the index is built in-process first (its wall time is what `build-index`
// reports: main.rs:285-306 times CREATE INDEX until SERVING).
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cinttypes>
#include <condition_variable>
#include <mutex>
#include <cstdlib>
#include <iostream>
#include <random>
#include <thread>

#include "../../include/vs_hnsw_debug.h"
#include "bench_util.hpp"

using Clock = std::chrono::steady_clock;
using namespace vsb;

struct Args {
    std::map<std::string, std::string> kv;
    std::string get(const std::string& k, const std::string& d = "") const {
        auto it = kv.find(k);
        return it == kv.end() ? d : it->second;
    }
    long num(const std::string& k, long d) const { return kv.count(k) ? std::atol(kv.at(k).c_str()) : d; }
    double real(const std::string& k, double d) const { return kv.count(k) ? std::atof(kv.at(k).c_str()) : d; }
};

static int metric_code(const std::string& m) {
    if (m == "cos") return VS_METRIC_COS;
    if (m == "l2sq" || m == "euclidean") return VS_METRIC_L2SQ;
    if (m == "ip" || m == "dot") return VS_METRIC_IP;
    throw std::runtime_error("unknown metric " + m);
}

static void ok(int rc, const char* what) {
    if (rc != VS_OK) throw std::runtime_error(std::string(what) + ": " + vs_hnsw_last_error());
}

static vs_hnsw* make_index(const Args& a, size_t dim, size_t capacity) {
    vs_hnsw_options o{};
    o.dimensions = dim;
    o.connectivity = (size_t)a.num("connectivity", 16);
    o.expansion_add = (size_t)a.num("expansion-add", 128);
    o.expansion_search = (size_t)a.num("expansion-search", 64);
    o.metric = metric_code(a.get("metric", "cos"));
    o.quantization = VS_SCALAR_F32;
    o.device = -1;
    vs_hnsw* h = nullptr;
    ok(vs_hnsw_create(&o, &h), "create");
    ok(vs_hnsw_reserve(h, capacity, 0), "reserve");
    return h;
}

// Synthetic rows: i.i.d. Gaussian, or a `rank`-d Gaussian latent through a fixed random map + noise.
static void synth(float* out, size_t rows, size_t dim, const std::string& dist, size_t rank, uint64_t seed) {
    std::vector<float> w;
    if (dist == "lowrank") {
        std::mt19937_64 g(99);
        std::normal_distribution<float> nd;
        w.resize(rank * dim);
        for (auto& x : w) x = nd(g) / std::sqrt((float)rank);
    }
    unsigned T = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            std::vector<float> z(rank);
            for (size_t r = t; r < rows; r += T) {
                std::mt19937_64 g(seed * 0x9E3779B97F4A7C15ull + r);
                std::normal_distribution<float> nd;
                float* row = out + r * dim;
                if (dist == "lowrank") {
                    for (auto& x : z) x = nd(g);
                    for (size_t c = 0; c < dim; ++c) {
                        float s = 0.05f * nd(g);
                        for (size_t j = 0; j < rank; ++j) s += z[j] * w[j * dim + c];
                        row[c] = s;
                    }
                } else {
                    for (size_t c = 0; c < dim; ++c) row[c] = nd(g);
                }
            }
        });
    for (auto& x : th) x.join();
}

static int cmd_selftest() {
    // fbin / ibin round trip
    const char* dir = std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp";
    std::string base = std::string(dir) + "/vs_bench_selftest";
    std::string mk = "mkdir -p " + base;
    if (std::system(mk.c_str())) return 1;
    std::vector<float> f = {1.5f, -2.f, 3.f, 4.f, 5.f, 6.f};
    std::vector<int32_t> t = {5, 1, 9, 2, 0, 7};
    write_fbin(base + "/query.fbin", f.data(), 2, 3);
    write_ibin(base + "/query.ibin", t.data(), 2, 3);
    write_dataset_toml(base, DatasetConfig{});
    auto cfg = read_dataset_toml(base);
    auto q = load_queries(base, cfg, 2);
    bool good = q.size() == 2 && q[1].query == std::vector<float>({4.f, 5.f, 6.f}) && q[0].neighbors.count(5) &&
                q[0].neighbors.count(1) && !q[0].neighbors.count(9) && q[1].neighbors.size() == 2;
    std::ifstream raw(base + "/query.fbin", std::ios::binary);
    unsigned char hdr[8];
    raw.read((char*)hdr, 8);
    good = good && hdr[0] == 2 && hdr[1] == 0 && hdr[4] == 3 && hdr[7] == 0;  // little-endian u32 count, dim
    // recall
    uint64_t found[3] = {5, 42, 1};
    good = good && recall(q[0].neighbors, found, 3) == 1.0 && recall(q[1].neighbors, found, 3) == 0.0;
    // histogram: 1..100 ms, 10,000 buckets, percentiles as the reference computes them
    Histogram h;
    for (int i = 1; i <= 100; ++i) h.record((int64_t)i * 1000000);  // 1ms .. 100ms
    good = good && h.percentile(50) == 1000000 + Histogram::kStepNs * (int64_t)std::llround(49e6 / Histogram::kStepNs + 1);
    good = good && h.percentile(1) == Histogram::kMinNs + Histogram::kStepNs;  // first sample sits in bucket 1
    h.record(500);          // below the window -> bucket 0
    h.record(200000000);    // above the window -> overflow bucket
    // exactly 100 ms rounds into the last index as well (same arithmetic as the reference, main.rs:569-577)
    good = good && h.buckets.front() == 1 && h.buckets.back() == 2 && h.count == 102;
    Histogram over;
    for (int i = 0; i < 100; ++i) over.record(200000000);
    good = good && over.percentile(99) == INT64_MAX;  // Duration::MAX in the reference
    SearchMeasure a, b;
    a.record(2000000, 0.5);
    b.record(4000000, 1.0);
    a.append(b);
    good = good && a.count == 2 && a.latency_min == 2000000 && a.latency_max == 4000000 && a.recall_min == 0.5 &&
           a.recall_sum == 1.5;
    std::cout << (good ? "selftest ok" : "selftest FAILED") << std::endl;
    return good ? 0 : 1;
}

static int cmd_gen(const Args& a) {
    const std::string dir = a.get("data-dir");
    const size_t n = (size_t)a.num("n", 100000), dim = (size_t)a.num("dim", 768), nq = (size_t)a.num("queries", 1000);
    const size_t k = (size_t)a.num("neighbors", 100), rank = (size_t)a.num("rank", 24);
    const std::string dist = a.get("dist", "lowrank");
    const uint64_t seed = (uint64_t)a.num("seed", 1234);
    if (dir.empty()) throw std::runtime_error("--data-dir is required");
    if (std::system(("mkdir -p " + dir).c_str())) return 1;
    std::vector<float> base(n * dim), q(nq * dim);
    synth(base.data(), n, dim, dist, rank, seed);
    synth(q.data(), nq, dim, dist, rank, seed + 3087);
    DatasetConfig cfg;
    write_fbin(dir + "/" + cfg.data_fbin, base.data(), (uint32_t)n, (uint32_t)dim);
    write_fbin(dir + "/" + cfg.query_fbin, q.data(), (uint32_t)nq, (uint32_t)dim);
    // ground truth: exact search of the engine (ids are row indices, fbin.rs:86)
    vs_hnsw* h = make_index(a, dim, n);
    std::vector<uint64_t> keys(n);
    for (size_t i = 0; i < n; ++i) keys[i] = i;
    ok(vs_hnsw_add_batch(h, keys.data(), base.data(), n, dim), "add_batch");
    std::vector<int32_t> truth(nq * k, -1);
    const size_t kk = std::min<size_t>(k, 256);
    std::vector<uint64_t> tk(nq * kk);
    std::vector<float> td(nq * kk);
    std::vector<size_t> tf(nq);
    ok(vs_hnsw_exact_search_batch(h, q.data(), nq, dim, kk, tk.data(), td.data(), tf.data()), "exact");
    for (size_t i = 0; i < nq; ++i)
        for (size_t j = 0; j < std::min(kk, tf[i]); ++j) truth[i * k + j] = (int32_t)tk[i * kk + j];
    write_ibin(dir + "/" + cfg.query_ibin, truth.data(), (uint32_t)nq, (uint32_t)k);
    write_dataset_toml(dir, cfg);
    vs_hnsw_free(h);
    std::cout << "wrote " << n << " x " << dim << " vectors, " << nq << " queries, " << k << " neighbours to " << dir << std::endl;
    return 0;
}

static vs_hnsw* build(const Args& a, const std::string& dir, const DatasetConfig& cfg, size_t& dim_out, double& secs) {
    Matrix d = read_bin(dir + "/" + cfg.data_fbin, false);
    vs_hnsw* h = make_index(a, d.dim, d.count);
    std::vector<uint64_t> keys(d.count);
    for (size_t i = 0; i < d.count; ++i) keys[i] = i;
    auto t0 = Clock::now();
    ok(vs_hnsw_add_batch(h, keys.data(), d.f.data(), d.count, d.dim), "add_batch");
    secs = std::chrono::duration<double>(Clock::now() - t0).count();
    dim_out = d.dim;
    std::cout << "index build: " << d.count << " vectors in " << secs << " s = " << (double)d.count / secs << " vectors/s" << std::endl;
    return h;
}

// One vector per call from `concurrency` threads: how the reference's service feeds the index
// (monitor_items -> actor -> worker pool -> usearch::Index::add, one vector per FFI call).
static int cmd_build_single(const Args& a, unsigned conc) {
    const std::string dir = a.get("data-dir");
    DatasetConfig cfg = read_dataset_toml(dir);
    Matrix d = read_bin(dir + "/" + cfg.data_fbin, false);
    const size_t n = std::min<size_t>(d.count, (size_t)a.num("max-vectors", d.count));
    vs_hnsw* h = make_index(a, d.dim, n);
    std::atomic<size_t> next{0};
    std::atomic<int> errors{0};
    auto t0 = Clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < conc; ++t)
        th.emplace_back([&] {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= n) break;
                if (vs_hnsw_add(h, i, d.f.data() + i * d.dim, d.dim) != VS_OK) errors.fetch_add(1);
            }
        });
    for (auto& x : th) x.join();
    const size_t indexed = vs_hnsw_size(h);  // barrier: staged vectors are inserted before the clock stops
    double secs = std::chrono::duration<double>(Clock::now() - t0).count();
    if (indexed != n - (size_t)errors.load()) std::cerr << "size mismatch: " << indexed << std::endl;
    std::cout << "index build (one vector per call, concurrency " << conc << "): " << n << " vectors in " << secs << " s = "
              << (double)n / secs << " vectors/s, errors " << errors.load() << std::endl;
    vs_hnsw_free(h);
    return errors.load() ? 1 : 0;
}

static int cmd_build(const Args& a) {
    const std::string dir = a.get("data-dir");
    if (a.kv.count("concurrency")) return cmd_build_single(a, (unsigned)a.num("concurrency", 16));
    size_t dim;
    double secs;
    vs_hnsw* h = build(a, dir, read_dataset_toml(dir), dim, secs);
    uint64_t st[8];
    ok(vs_hnsw_stats(h, st, 0), "stats");
    std::cout << "distance evaluations per add: " << (double)st[3] / (double)std::max<uint64_t>(st[5], 1) << std::endl;
    vs_hnsw_free(h);
    return 0;
}

static int cmd_search(const Args& a) {
    const std::string dir = a.get("data-dir");
    const size_t limit = (size_t)a.num("limit", 10);
    const double duration = a.real("duration", 10.0);
    const unsigned conc = (unsigned)a.num("concurrency", 64);
    if (limit < 1 || limit > 10000) throw std::runtime_error("--limit must be in 1..=10000");  // main.rs:192,221
    DatasetConfig cfg = read_dataset_toml(dir);
    size_t dim;
    double secs;
    vs_hnsw* h = build(a, dir, cfg, dim, secs);
    ok(vs_hnsw_set_expansion_search(h, (size_t)a.num("expansion-search", 64)), "set ef");
    std::vector<Query> queries = load_queries(dir, cfg, limit);
    std::atomic<bool> stop{false};
    std::vector<SearchMeasure> per(conc);
    std::vector<std::string> errors(conc);
    const unsigned inflight = (unsigned)std::max<long>(1, a.num("inflight", 1));
    auto t0 = Clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < conc; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 7919 + 13);
            if (inflight == 1) {  // blocking call per query: what a thread of the reference's worker pool does
                std::vector<uint64_t> keys(limit);
                std::vector<float> dist(limit);
                while (!stop.load(std::memory_order_relaxed)) {
                    const Query& q = queries[g() % queries.size()];
                    size_t found = 0;
                    auto s = Clock::now();
                    int rc = vs_hnsw_search(h, q.query.data(), dim, limit, keys.data(), dist.data(), &found);
                    int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                    if (rc != VS_OK) {
                        errors[t] = vs_hnsw_last_error();
                        break;
                    }
                    per[t].record(ns, recall(q.neighbors, keys.data(), found));
                }
                return;
            }
            // --inflight D: the non-blocking entry point, D queries outstanding per worker (an async runtime)
            struct Slot {
                std::vector<uint64_t> keys;
                std::vector<float> dist;
                size_t found = 0;
                const Query* q = nullptr;
                Clock::time_point start;
                int status = 0;
                bool done = true;
                std::mutex* mu;
                std::condition_variable* cv;
            };
            std::mutex mu;
            std::condition_variable cv;
            std::vector<Slot> slots(inflight);
            for (auto& sl : slots) {
                sl.keys.resize(limit);
                sl.dist.resize(limit);
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
                    if (sl.q) {  // a completed query
                        int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - sl.start).count();
                        if (sl.status != VS_OK) errors[t] = "async search failed";
                        else per[t].record(ns, recall(sl.q->neighbors, sl.keys.data(), sl.found));
                        sl.q = nullptr;
                        --outstanding;
                    }
                    if (stopping) continue;
                    sl.q = &queries[g() % queries.size()];
                    sl.done = false;
                    sl.start = Clock::now();
                    ++outstanding;
                    lk.unlock();
                    int rc = vs_hnsw_search_async(h, sl.q->query.data(), dim, limit, sl.keys.data(), sl.dist.data(), &sl.found,
                                                  done, &sl);
                    lk.lock();
                    if (rc != VS_OK) {
                        errors[t] = vs_hnsw_last_error();
                        sl.done = true;
                        sl.q = nullptr;
                        --outstanding;
                        stopping = true;
                        stop = true;
                    }
                }
                if (stopping && outstanding == 0) break;
                cv.wait(lk, [&] {
                    for (auto& sl : slots)
                        if (sl.done && sl.q) return true;
                    return stop.load() && outstanding == 0;
                });
            }
        });
    std::this_thread::sleep_for(std::chrono::duration<double>(duration));
    stop = true;
    for (auto& x : th) x.join();
    double wall = std::chrono::duration<double>(Clock::now() - t0).count();
    SearchMeasure all;
    for (auto& m : per) all.append(m);
    for (auto& e : errors)
        if (!e.empty()) std::cerr << "search error: " << e << std::endl;
    std::cout << "concurrency: " << conc << (inflight > 1 ? " x inflight " + std::to_string(inflight) : std::string()) << "\n" << all.report(wall);
    uint64_t st[8];
    ok(vs_hnsw_stats(h, st, 0), "stats");
    if (st[2]) std::cout << "distance evaluations per query: " << (double)st[0] / (double)st[2] << std::endl;
    uint64_t sv[4];
    if (vs_search_service_stats(sv) == 0 && sv[0])
        std::cout << "kernel launches: " << sv[0] << " (mean batch " << (double)sv[1] / (double)sv[0] << "), team-kernel launches: " << sv[2]
                  << " (" << sv[3] << " queries)" << std::endl;
    vs_hnsw_free(h);
    return 0;
}

// ---- search-http: the reference's `search-http` scenario (main.rs:435-525) against a running /ann server
// (vs_httpd, or the reference service): `concurrency` keep-alive connections, one query per request.
struct HttpConn {
    int fd = -1;
    std::string buf;
    bool connect_to(const std::string& host, int port) {
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) return false;
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        sockaddr_in sa{};
        sa.sin_family = AF_INET;
        sa.sin_port = htons((uint16_t)port);
        if (inet_pton(AF_INET, host.c_str(), &sa.sin_addr) != 1) return false;
        return ::connect(fd, (sockaddr*)&sa, sizeof sa) == 0;
    }
    // Sends one request, returns the status code and the body.
    int roundtrip(const std::string& req, std::string& body) {
        size_t off = 0;
        while (off < req.size()) {
            ssize_t n = ::send(fd, req.data() + off, req.size() - off, MSG_NOSIGNAL);
            if (n <= 0) return -1;
            off += (size_t)n;
        }
        buf.clear();
        size_t he = std::string::npos, clen = 0;
        char tmp[16384];
        for (;;) {
            if (he == std::string::npos) {
                he = buf.find("\r\n\r\n");
                if (he != std::string::npos) {
                    std::string head = buf.substr(0, he);
                    for (auto& ch : head) ch = (char)std::tolower((unsigned char)ch);
                    size_t p = head.find("content-length:");
                    clen = p == std::string::npos ? 0 : (size_t)std::strtoull(head.c_str() + p + 15, nullptr, 10);
                }
            }
            if (he != std::string::npos && buf.size() >= he + 4 + clen) break;
            ssize_t n = ::recv(fd, tmp, sizeof tmp, 0);
            if (n <= 0) return -1;
            buf.append(tmp, (size_t)n);
        }
        body.assign(buf, he + 4, clen);
        return std::atoi(buf.c_str() + 9);  // "HTTP/1.1 200 ..."
    }
    ~HttpConn() {
        if (fd >= 0) ::close(fd);
    }
};

static int cmd_search_http(const Args& a) {
    const std::string dir = a.get("data-dir");
    const size_t limit = (size_t)a.num("limit", 10);
    const double duration = a.real("duration", 10.0);
    const unsigned conc = (unsigned)a.num("concurrency", 64);
    const std::string host = a.get("host", "127.0.0.1");
    const int port = (int)a.num("port", 6080);
    const std::string base = "/api/v1/indexes/" + a.get("keyspace", "vsb_keyspace") + "/" + a.get("index", "vsb_index");
    DatasetConfig cfg = read_dataset_toml(dir);
    std::vector<Query> queries = load_queries(dir, cfg, limit);
    // wait for SERVING, as the reference's wait_for_index_ready does (vs.rs:17-39)
    {
        const std::string req = "GET " + base + "/status HTTP/1.1\r\nhost: " + host + "\r\n\r\n";
        for (int tries = 0;; ++tries) {
            HttpConn c;
            std::string body;
            if (c.connect_to(host, port) && c.roundtrip(req, body) == 200 && body.find("\"SERVING\"") != std::string::npos) break;
            if (tries > 3600) throw std::runtime_error("index never reached SERVING");
            std::this_thread::sleep_for(std::chrono::milliseconds(500));
        }
    }
    // one pre-rendered request per query (the client's JSON encoding is not what is measured)
    std::vector<std::string> reqs(queries.size());
    for (size_t i = 0; i < queries.size(); ++i) {
        std::string body = "{\"vector\":[";
        char num[32];
        for (size_t j = 0; j < queries[i].query.size(); ++j) {
            std::snprintf(num, sizeof num, j ? ",%.9g" : "%.9g", (double)queries[i].query[j]);
            body += num;
        }
        body += "],\"limit\":" + std::to_string(limit) + "}";
        reqs[i] = "POST " + base + "/ann HTTP/1.1\r\nhost: " + host + "\r\ncontent-type: application/json\r\ncontent-length: " +
                  std::to_string(body.size()) + "\r\n\r\n" + body;
    }
    std::atomic<bool> stop{false};
    std::vector<SearchMeasure> per(conc);
    std::vector<std::string> errors(conc);
    auto t0 = Clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < conc; ++t)
        th.emplace_back([&, t] {
            std::mt19937_64 g(t * 7919 + 13);
            HttpConn c;
            if (!c.connect_to(host, port)) {
                errors[t] = "connect failed";
                return;
            }
            std::string body;
            std::vector<uint64_t> keys;
            while (!stop.load(std::memory_order_relaxed)) {
                const size_t qi = g() % queries.size();
                auto s = Clock::now();
                int code = c.roundtrip(reqs[qi], body);
                int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - s).count();
                if (code != 200) {
                    errors[t] = "HTTP " + std::to_string(code) + ": " + body.substr(0, 200);
                    break;
                }
                keys.clear();
                size_t p = body.find("\"primary_keys\"");
                p = p == std::string::npos ? p : body.find('[', p);
                if (p != std::string::npos) {
                    const char* q = body.c_str() + p + 1;
                    while (*q && *q != ']') {
                        char* e = nullptr;
                        unsigned long long v = std::strtoull(q, &e, 10);
                        if (e == q) break;
                        keys.push_back(v);
                        q = *e == ',' ? e + 1 : e;
                    }
                }
                per[t].record(ns, recall(queries[qi].neighbors, keys.data(), keys.size()));
            }
        });
    std::this_thread::sleep_for(std::chrono::duration<double>(duration));
    stop = true;
    for (auto& x : th) x.join();
    double wall = std::chrono::duration<double>(Clock::now() - t0).count();
    SearchMeasure all;
    for (auto& m : per) all.append(m);
    for (auto& e : errors)
        if (!e.empty()) std::cerr << "search-http error: " << e << std::endl;
    std::cout << "search-http concurrency: " << conc << "\n" << all.report(wall);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "usage: vs_bench selftest | gen | build-index | search | search-http  (see the header of vs_bench.cpp)" << std::endl;
        return 2;
    }
    Args a;
    for (int i = 2; i < argc; ++i) {
        std::string k = argv[i];
        if (k.rfind("--", 0) != 0) continue;
        k = k.substr(2);
        std::string v = (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) ? argv[++i] : "1";
        a.kv[k] = v;
    }
    try {
        std::string c = argv[1];
        if (c == "selftest") return cmd_selftest();
        if (c == "gen") return cmd_gen(a);
        if (c == "build-index") return cmd_build(a);
        if (c == "search") return cmd_search(a);
        if (c == "search-http") return cmd_search_http(a);
        std::cerr << "unknown command " << c << std::endl;
        return 2;
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
}
