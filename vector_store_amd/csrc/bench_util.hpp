// bench_util.hpp -- host-side pieces of the benchmark driver (vs_bench), the counterpart of the
// reference's crates/benchmark: dataset files, latency histogram, search report, recall.
//
// Reproduces:
//   fbin  = u32 count, u32 dim (little endian) + count*dim f32 LE, row major     crates/benchmark/src/data/fbin.rs:30-45,69-99
//   ibin  = same header + count*dim i32 LE; the first `limit` ids of a row are used   fbin.rs:109-111,137-142
//   ids   = row indices 0..count                                                 fbin.rs:86
//   dataset.toml: [fbin] data_fbin / query_fbin / query_ibin                     fbin.rs:23-28, data/mod.rs:102-125
//   histogram: 10,000 linear buckets over 1..100 ms (+ underflow, overflow)       main.rs:539-604
//   report: count, QPS = count / wall, min, P1 P10 P25 P50 P75 P90 P99, max, recall min/avg/max   main.rs:606-697
//   recall = |neighbors ∩ found| / |neighbors|                                     db.rs:308
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace vsb {

struct Matrix {
    uint32_t count = 0, dim = 0;
    std::vector<float> f;    // fbin payload
    std::vector<int32_t> i;  // ibin payload
};

inline void write_header(std::ofstream& o, uint32_t count, uint32_t dim) {
    o.write(reinterpret_cast<const char*>(&count), 4);  // x86-64: little endian, as the format requires
    o.write(reinterpret_cast<const char*>(&dim), 4);
}

inline void write_fbin(const std::string& path, const float* data, uint32_t count, uint32_t dim) {
    std::ofstream o(path, std::ios::binary);
    if (!o) throw std::runtime_error("cannot write " + path);
    write_header(o, count, dim);
    o.write(reinterpret_cast<const char*>(data), (std::streamsize)count * dim * 4);
}

inline void write_ibin(const std::string& path, const int32_t* data, uint32_t count, uint32_t dim) {
    std::ofstream o(path, std::ios::binary);
    if (!o) throw std::runtime_error("cannot write " + path);
    write_header(o, count, dim);
    o.write(reinterpret_cast<const char*>(data), (std::streamsize)count * dim * 4);
}

inline Matrix read_bin(const std::string& path, bool ints) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    Matrix m;
    in.read(reinterpret_cast<char*>(&m.count), 4);
    in.read(reinterpret_cast<char*>(&m.dim), 4);
    if (!in) throw std::runtime_error("short header in " + path);
    const size_t n = (size_t)m.count * m.dim;
    if (ints) {
        m.i.resize(n);
        in.read(reinterpret_cast<char*>(m.i.data()), (std::streamsize)n * 4);
    } else {
        m.f.resize(n);
        in.read(reinterpret_cast<char*>(m.f.data()), (std::streamsize)n * 4);
    }
    if ((size_t)in.gcount() != n * 4) throw std::runtime_error("short payload in " + path);
    return m;
}

// The three file names of the [fbin] table of dataset.toml.
struct DatasetConfig {
    std::string data_fbin = "data.fbin", query_fbin = "query.fbin", query_ibin = "query.ibin";
};

inline DatasetConfig read_dataset_toml(const std::string& dir) {
    DatasetConfig c;
    std::ifstream in(dir + "/dataset.toml");
    if (!in) return c;
    std::string line, table;
    while (std::getline(in, line)) {
        auto hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        auto trim = [](std::string s) {
            size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
            return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
        };
        line = trim(line);
        if (line.empty()) continue;
        if (line.front() == '[') {
            table = trim(line.substr(1, line.find(']') - 1));
            continue;
        }
        auto eq = line.find('=');
        if (eq == std::string::npos || table != "fbin") continue;
        std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
        if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
        if (key == "data_fbin") c.data_fbin = val;
        if (key == "query_fbin") c.query_fbin = val;
        if (key == "query_ibin") c.query_ibin = val;
    }
    return c;
}

inline void write_dataset_toml(const std::string& dir, const DatasetConfig& c) {
    std::ofstream o(dir + "/dataset.toml");
    o << "[fbin]\ndata_fbin = \"" << c.data_fbin << "\"\nquery_fbin = \"" << c.query_fbin << "\"\nquery_ibin = \""
      << c.query_ibin << "\"\n";
}

struct Query {
    std::vector<float> query;
    std::unordered_set<int64_t> neighbors;  // the first `limit` ground-truth ids of the row
};

inline std::vector<Query> load_queries(const std::string& dir, const DatasetConfig& c, size_t limit) {
    Matrix q = read_bin(dir + "/" + c.query_fbin, false), t = read_bin(dir + "/" + c.query_ibin, true);
    if (q.count != t.count) throw std::runtime_error("query.fbin and query.ibin disagree on the row count");
    limit = std::min<size_t>(limit, t.dim);
    std::vector<Query> out(q.count);
    for (uint32_t r = 0; r < q.count; ++r) {
        out[r].query.assign(q.f.begin() + (size_t)r * q.dim, q.f.begin() + (size_t)(r + 1) * q.dim);
        for (size_t j = 0; j < limit; ++j) out[r].neighbors.insert(t.i[(size_t)r * t.dim + j]);
    }
    return out;
}

inline double recall(const std::unordered_set<int64_t>& neighbors, const uint64_t* found, size_t n) {
    if (neighbors.empty()) return 0.0;
    size_t hit = 0;
    for (size_t i = 0; i < n; ++i) hit += neighbors.count((int64_t)found[i]);
    return (double)hit / (double)neighbors.size();
}

// Latencies in nanoseconds.
struct Histogram {
    static constexpr size_t kBuckets = 10000;
    static constexpr int64_t kMinNs = 1000000, kMaxNs = 100000000;
    static constexpr int64_t kStepNs = (kMaxNs - kMinNs) / (int64_t)kBuckets;  // 9,900 ns
    std::vector<uint64_t> buckets = std::vector<uint64_t>(kBuckets + 2, 0);
    uint64_t count = 0;

    void record(int64_t ns) {
        size_t idx;
        if (ns < kMinNs) idx = 0;
        else if (ns > kMaxNs) idx = kBuckets + 1;
        else idx = (size_t)std::llround((double)(ns - kMinNs) / (double)kStepNs) + 1;
        ++buckets[idx];
        ++count;
    }
    // INT64_MAX stands for the reference's Duration::MAX (beyond the histogram window).
    int64_t percentile(double p) const {
        const uint64_t want = (uint64_t)((double)count * p / 100.0);
        uint64_t sum = 0;
        for (size_t idx = 0; idx < buckets.size(); ++idx) {
            sum += buckets[idx];
            if (sum >= want) {
                if (idx == buckets.size() - 1) return INT64_MAX;
                return kMinNs + kStepNs * (int64_t)idx;
            }
        }
        return INT64_MAX;
    }
    void append(const Histogram& o) {
        for (size_t i = 0; i < buckets.size(); ++i) buckets[i] += o.buckets[i];
        count += o.count;
    }
};

struct SearchMeasure {
    size_t count = 0;
    Histogram histogram;
    int64_t latency_min = INT64_MAX, latency_max = 0;
    bool with_recall = true;
    double recall_min = 100.0, recall_max = 0.0, recall_sum = 0.0;

    void record(int64_t ns, double rec) {
        ++count;
        histogram.record(ns);
        latency_min = std::min(latency_min, ns);
        latency_max = std::max(latency_max, ns);
        if (with_recall) {
            recall_min = std::min(recall_min, rec);
            recall_max = std::max(recall_max, rec);
            recall_sum += rec;
        }
    }
    void append(const SearchMeasure& o) {
        count += o.count;
        histogram.append(o.histogram);
        latency_min = std::min(latency_min, o.latency_min);
        latency_max = std::max(latency_max, o.latency_max);
        if (with_recall && o.with_recall) {
            recall_min = std::min(recall_min, o.recall_min);
            recall_max = std::max(recall_max, o.recall_max);
            recall_sum += o.recall_sum;
        }
    }
    static std::string fmt_ns(int64_t ns) {
        if (ns == INT64_MAX) return ">100ms";
        char b[64];
        if (ns >= 1000000) snprintf(b, sizeof b, "%.1fms", ns / 1e6);
        else snprintf(b, sizeof b, "%.1fus", ns / 1e3);
        return b;
    }
    // One JSON object (machine readable) after the human-readable lines of the reference's log().
    std::string report(double seconds) const {
        std::ostringstream o;
        char b[256];
        snprintf(b, sizeof b, "duration: %.1fs\nqueries: %zu\nQPS: %.1f\n", seconds, count, count / seconds);
        o << b << "latency min: " << fmt_ns(latency_min) << "\n";
        for (int p : {1, 10, 25, 50, 75, 90, 99}) o << "latency P" << (p < 10 ? "0" : "") << p << ": " << fmt_ns(histogram.percentile(p)) << "\n";
        o << "latency max: " << fmt_ns(latency_max) << "\n";
        if (with_recall && count) {
            snprintf(b, sizeof b, "recall min: %.1f\nrecall avg: %.1f\nrecall max: %.1f\n", recall_min * 100.0,
                     recall_sum * 100.0 / (double)count, recall_max * 100.0);
            o << b;
        }
        return o.str();
    }
};

}  // namespace vsb
