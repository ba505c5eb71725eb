// vs_httpd -- the `/ann` HTTP surface of the reference, served natively over libvs_hnsw (row f-1 of SURVEY.md
// section 8).  Same wire format, status codes and defaults as crates/vector-store's REST API for the hot path, so
// that the reference's benchmark client (`crates/benchmark search-http`, and its status polling, vs.rs:17-39) can
// talk to the engine unchanged:
//
//   POST /api/v1/indexes/{keyspace}/{index}/ann      httproutes.rs:661-904, httpapi/src/lib.rs:369-409
//   GET  /api/v1/indexes/{keyspace}/{index}/status   httpapi/src/lib.rs:192-207
//   GET  /api/v1/indexes                             httpapi/src/lib.rs:84-91
//   GET  /api/v1/info, GET /api/v1/status            httpapi/src/lib.rs:232-240, 296-309
//
//   400 malformed body / wrong vector size / bad filter, 404 unknown index, 500 engine error,
//   503 {"reason": "INDEX_BUILDING", ...} while the index loads; +-inf saturate to +-f32::MAX (lib.rs:397-409);
//   `limit` defaults to 1 (lib.rs:289-293).
//
// What is NOT here (it stays in the Rust service): the table cache, CQL types, TLS, metrics.  The only primary-key
// column is an integer (`--pk-column`, default "id") holding the row index = the low 48 bits of the PrimaryId, the
// way the reference's fbin loader numbers rows (benchmark data/fbin.rs:86); filters are evaluated on that column
// (all twelve restriction forms of lib.rs:323-366 that one integer column admits).
//
// Shape: T event-loop threads (epoll, SO_REUSEPORT listeners, HTTP/1.1 keep-alive).  An unfiltered query is handed
// to vs_hnsw_search_async -- no thread blocks on the GPU; the engine's dispatcher batches whatever the connections
// have in flight -- and the completion comes back to the connection's loop through an eventfd.  Filtered queries
// (host predicate) run on a small pool.  Mirrors vector_store_amd/httpd.py, which stays as the readable twin.
//
//   vs_httpd --data-dir D [--keyspace vsb_keyspace] [--index vsb_index] [--metric cos|l2sq|ip] [--host 127.0.0.1]
//            [--port 6080] [--threads 4] [--expansion-search 64] [--connectivity 16] [--expansion-add 128]
//            [--pk-column id] [--max-vectors N]
#include <arpa/inet.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <signal.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cfloat>
#include <chrono>
#include <cinttypes>
#include <cerrno>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/vs_hnsw.h"
#include "bench_util.hpp"

namespace {

// ------------------------------------------------------------------------------------------ JSON (requests only)
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    bool is_int = false;  // a number written without fraction / exponent that fits int64
    int64_t i = 0;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<float> fvec;  // the value of a key named "vector" when it is an array of numbers only (fast path)
    bool is_fvec = false;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const {
        for (auto& kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class JsonParser {
   public:
    JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
    JVal parse() {
        JVal v = value(0);
        ws();
        if (p_ != e_) throw JsonError("trailing characters");
        return v;
    }

   private:
    const char *p_, *e_;
    void ws() {
        while (p_ < e_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    JVal value(int depth) {
        if (depth > 32) throw JsonError("nesting too deep");
        ws();
        if (p_ >= e_) throw JsonError("unexpected end");
        JVal v;
        char c = *p_;
        if (c == '{') {
            v.kind = JVal::Obj;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == '}') {
                ++p_;
                return v;
            }
            for (;;) {
                ws();
                if (p_ >= e_ || *p_ != '"') throw JsonError("expected a key");
                std::string k = string();
                ws();
                if (p_ >= e_ || *p_ != ':') throw JsonError("expected ':'");
                ++p_;
                if (k == "vector") {
                    JVal fv;
                    if (float_array(fv)) {
                        v.obj.emplace_back(std::move(k), std::move(fv));
                    } else {
                        v.obj.emplace_back(std::move(k), value(depth + 1));
                    }
                } else {
                    v.obj.emplace_back(std::move(k), value(depth + 1));
                }
                ws();
                if (p_ < e_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < e_ && *p_ == '}') {
                    ++p_;
                    return v;
                }
                throw JsonError("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JVal::Arr;
            ++p_;
            ws();
            if (p_ < e_ && *p_ == ']') {
                ++p_;
                return v;
            }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p_ < e_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < e_ && *p_ == ']') {
                    ++p_;
                    return v;
                }
                throw JsonError("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = JVal::Str;
            v.str = string();
            return v;
        }
        if (lit("true")) {
            v.kind = JVal::Bool;
            v.b = true;
            return v;
        }
        if (lit("false")) {
            v.kind = JVal::Bool;
            return v;
        }
        if (lit("null")) return v;
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p_;
            bool integral = true;
            if (*p_ == '-') ++p_;
            while (p_ < e_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) {
                if (*p_ == '.' || *p_ == 'e' || *p_ == 'E') integral = false;
                ++p_;
            }
            std::string t(s, p_);
            char* end = nullptr;
            v.kind = JVal::Num;
            v.num = std::strtod(t.c_str(), &end);
            if (end != t.c_str() + t.size() || t == "-") throw JsonError("bad number");
            if (integral && t.size() <= 20) {  // every int64 (serde reads `limit` as a usize: 2^62 is a valid request)
                errno = 0;
                const long long i = std::strtoll(t.c_str(), nullptr, 10);
                if (errno != ERANGE) {
                    v.is_int = true;
                    v.i = i;
                }
            }
            return v;
        }
        throw JsonError("unexpected character");
    }
    // "[n, n, ...]" of plain numbers straight into floats (a 768-d query is 768 numbers: no per-element JVal).
    // Decimal -> f32 by Clinger's fast path (mantissa < 2^53, |exp10| <= 22: one exactly rounded double operation,
    // then one rounding to f32 -- exact for the <= 9 significant digits a round-tripping f32 needs); anything else
    // goes through strtod.  On any non-number element the cursor is restored and the generic parser takes over.
    bool float_array(JVal& out) {
        const char* save = p_;
        ws();
        if (p_ >= e_ || *p_ != '[') {
            p_ = save;
            return false;
        }
        ++p_;
        out.kind = JVal::Arr;
        out.is_fvec = true;
        out.fvec.reserve(1024);
        ws();
        if (p_ < e_ && *p_ == ']') {
            ++p_;
            return true;
        }
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                     1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        for (;;) {
            ws();
            const char* s = p_;
            bool neg = false;
            if (p_ < e_ && *p_ == '-') {
                neg = true;
                ++p_;
            }
            uint64_t mant = 0;
            int digits = 0, exp10 = 0;
            bool any = false, slow = false;
            while (p_ < e_ && *p_ >= '0' && *p_ <= '9') {
                any = true;
                if (digits < 18) {
                    mant = mant * 10 + (uint64_t)(*p_ - '0');
                    digits += (mant != 0);
                } else {
                    ++exp10;
                    slow = true;
                }
                ++p_;
            }
            if (p_ < e_ && *p_ == '.') {
                ++p_;
                while (p_ < e_ && *p_ >= '0' && *p_ <= '9') {
                    any = true;
                    if (digits < 18) {
                        mant = mant * 10 + (uint64_t)(*p_ - '0');
                        digits += (mant != 0);
                        --exp10;
                    } else {
                        slow = true;
                    }
                    ++p_;
                }
            }
            if (!any) {
                p_ = save;
                out = JVal();
                return false;
            }
            if (p_ < e_ && (*p_ == 'e' || *p_ == 'E')) {
                ++p_;
                bool eneg = false;
                if (p_ < e_ && (*p_ == '+' || *p_ == '-')) eneg = *p_++ == '-';
                int ev = 0;
                bool edig = false;
                while (p_ < e_ && *p_ >= '0' && *p_ <= '9') {
                    edig = true;
                    if (ev < 10000) ev = ev * 10 + (*p_ - '0');
                    ++p_;
                }
                if (!edig) throw JsonError("bad number");
                exp10 += eneg ? -ev : ev;
            }
            double d;
            if (!slow && mant < (1ull << 53) && exp10 >= -22 && exp10 <= 22) {
                d = (double)mant;
                d = exp10 < 0 ? d / p10[-exp10] : d * p10[exp10];
                if (neg) d = -d;
            } else {
                d = std::strtod(std::string(s, p_).c_str(), nullptr);
            }
            out.fvec.push_back((float)d);
            ws();
            if (p_ < e_ && *p_ == ',') {
                ++p_;
                continue;
            }
            if (p_ < e_ && *p_ == ']') {
                ++p_;
                return true;
            }
            p_ = save;  // not a plain array of numbers (or malformed): let the generic parser decide / complain
            out = JVal();
            return false;
        }
    }
    bool lit(const char* w) {
        size_t n = std::strlen(w);
        if ((size_t)(e_ - p_) >= n && !std::memcmp(p_, w, n)) {
            p_ += n;
            return true;
        }
        return false;
    }
    std::string string() {
        ++p_;  // opening quote
        std::string out;
        while (p_ < e_ && *p_ != '"') {
            if (*p_ == '\\') {
                if (++p_ >= e_) break;
                switch (*p_) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (e_ - p_ < 5) throw JsonError("bad escape");
                        unsigned cp = (unsigned)std::strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) {
                            out += (char)(0xC0 | (cp >> 6));
                            out += (char)(0x80 | (cp & 0x3F));
                        } else {
                            out += (char)(0xE0 | (cp >> 12));
                            out += (char)(0x80 | ((cp >> 6) & 0x3F));
                            out += (char)(0x80 | (cp & 0x3F));
                        }
                        break;
                    }
                    default: out += *p_;
                }
                ++p_;
            } else {
                out += *p_++;
            }
        }
        if (p_ >= e_) throw JsonError("unterminated string");
        ++p_;
        return out;
    }
};

std::string json_escape(const std::string& s) {
    std::string o;
    for (char c : s) {
        if (c == '"' || c == '\\') {
            o += '\\';
            o += c;
        } else if ((unsigned char)c < 0x20) {
            char b[8];
            std::snprintf(b, sizeof b, "\\u%04x", c);
            o += b;
        } else {
            o += c;
        }
    }
    return o;
}

// serialize_saturated_f32 (httpapi/src/lib.rs:397-409); shortest text that reads back to the same f32.
void put_f32(std::string& o, float v) {
    if (v == INFINITY) v = FLT_MAX;
    else if (v == -INFINITY) v = -FLT_MAX;
    char b[32];
    for (int prec = 6; prec <= 9; ++prec) {
        std::snprintf(b, sizeof b, "%.*g", prec, (double)v);
        if (std::strtof(b, nullptr) == v) break;
    }
    o += b;
    if (!std::strpbrk(b, ".eEn")) o += ".0";  // serde writes floats with a fraction
}

// ------------------------------------------------------------------------------------------ the served index
struct BadRequest : std::runtime_error {
    using std::runtime_error::runtime_error;
};

constexpr uint64_t kRowMask = (1ull << 48) - 1;
constexpr int kIdleSeconds = 120;  // keep-alive connections idle for longer are closed (reqwest's pool default is 90 s)
constexpr size_t kMaxBuffered = (64u << 20) + (1u << 20);  // unread request bytes one connection may hold (one maximal body + headers)

struct Test {
    enum Op { Eq, In, Lt, Le, Gt, Ge } op;
    int64_t v = 0;
    std::unordered_set<int64_t> set;
    bool operator()(int64_t x) const {
        switch (op) {
            case Eq: return x == v;
            case In: return set.count(x) != 0;
            case Lt: return x < v;
            case Le: return x <= v;
            case Gt: return x > v;
            default: return x >= v;
        }
    }
};

int64_t int_of(const JVal& v) {
    if (v.kind != JVal::Num || !v.is_int) throw BadRequest("filter values on the primary key column must be integers");
    return v.i;
}

// PostIndexAnnFilter -> tests on the row index (key & (2^48 - 1)).
std::vector<Test> compile_filter(const JVal& flt, const std::string& pk) {
    std::vector<Test> tests;
    const JVal* rs = flt.get("restrictions");
    if (!rs || rs->kind != JVal::Arr) throw BadRequest("filter.restrictions must be an array");
    for (const JVal& r : rs->arr) {
        const JVal *typ = r.get("type"), *lhs = r.get("lhs"), *rhs = r.get("rhs");
        if (!typ || typ->kind != JVal::Str || !lhs || !rhs) throw BadRequest("restriction needs type, lhs and rhs");
        std::string op = typ->str;
        const bool tuple = op.rfind("()", 0) == 0;
        const JVal* scalar = rhs;
        std::vector<const JVal*> many;
        if (tuple) {
            if (lhs->kind != JVal::Arr || lhs->arr.size() != 1 || lhs->arr[0].kind != JVal::Str || lhs->arr[0].str != pk)
                throw BadRequest("unknown column(s) in filter");
            if (op.size() < 5 || op.compare(op.size() - 2, 2, "()") != 0) throw BadRequest("unknown restriction type: " + typ->str);
            op = op.substr(2, op.size() - 4);  // "()<=()" -> "<="
            if (rhs->kind != JVal::Arr || rhs->arr.empty()) throw BadRequest("tuple restriction needs values");
            if (op == "IN") {
                for (const JVal& t : rhs->arr) {
                    if (t.kind != JVal::Arr || t.arr.size() != 1) throw BadRequest("tuple arity mismatch");
                    many.push_back(&t.arr[0]);
                }
            } else {
                if (rhs->arr.size() != 1) throw BadRequest("tuple arity mismatch");
                scalar = &rhs->arr[0];
            }
        } else {
            if (lhs->kind != JVal::Str || lhs->str != pk) throw BadRequest("unknown column in filter: " + (lhs->kind == JVal::Str ? lhs->str : "?"));
            if (op == "IN") {
                if (rhs->kind != JVal::Arr) throw BadRequest("IN needs an array");
                for (const JVal& t : rhs->arr) many.push_back(&t);
            }
        }
        Test t;
        if (op == "==") t.op = Test::Eq;
        else if (op == "IN") t.op = Test::In;
        else if (op == "<") t.op = Test::Lt;
        else if (op == "<=") t.op = Test::Le;
        else if (op == ">") t.op = Test::Gt;
        else if (op == ">=") t.op = Test::Ge;
        else throw BadRequest("unknown restriction type: " + typ->str);
        if (t.op == Test::In)
            for (const JVal* m : many) t.set.insert(int_of(*m));
        else
            t.v = int_of(*scalar);
        tests.push_back(std::move(t));
    }
    return tests;
}

// A name for the filter (vs_hnsw_filtered_search_keyed).  The engine's memory is keyed by a 64-bit id, and the restrictions come
// straight from request bodies: a bare hash of them would let a client craft restrictions that collide with another filter's id and
// have the engine answer one filter with the other's verdicts (round-5 advisor).  So the id is NOT a hash: the server keeps the
// canonical serialisation of every filter it has named and hands out ids from a counter -- two filters share an id exactly when their
// canonical forms are equal, byte for byte.  (The predicate below is a function of (these tests, the key's row number); row numbers
// never change in this server, so nothing here has to call vs_hnsw_filter_forget_keys -- a host whose restrictions read mutable
// columns does, see include/vs_hnsw.h.)
std::string filter_canonical(const std::vector<Test>& tests) {
    std::string c;
    for (const Test& t : tests) {
        c += (char)('a' + (int)t.op);
        if (t.op == Test::In) {
            std::vector<int64_t> members(t.set.begin(), t.set.end());  // (a set: the order of its members must not matter)
            std::sort(members.begin(), members.end());
            c += std::to_string(members.size());
            for (int64_t m : members) {
                c += ',';
                c += std::to_string(m);
            }
        } else {
            c += std::to_string(t.v);
        }
        c += ';';
    }
    return c;
}

class FilterNames {
    static constexpr size_t kMaxNames = 256;      // canonical forms remembered (the engine itself keeps 4 memories per index)
    static constexpr size_t kMaxCanonical = 4096;  // longer filters (huge IN lists) stay unnamed: no memory, always exact
    std::mutex mu_;
    std::unordered_map<std::string, std::pair<uint64_t, uint64_t>> ids_;  // canonical -> (id, last use)
    uint64_t next_ = 1, clock_ = 0;

public:
    // 0: unnamed.  `retired` (may be null) receives the id of a name that fell out, so that its memory can be dropped.
    uint64_t name_of(const std::vector<Test>& tests, uint64_t* retired = nullptr) {
        std::string c = filter_canonical(tests);
        if (c.size() > kMaxCanonical) return 0;
        std::lock_guard<std::mutex> g(mu_);
        auto it = ids_.find(c);
        if (it != ids_.end()) {
            it->second.second = ++clock_;
            return it->second.first;
        }
        if (ids_.size() >= kMaxNames) {
            auto lru = ids_.begin();
            for (auto j = ids_.begin(); j != ids_.end(); ++j)
                if (j->second.second < lru->second.second) lru = j;
            if (retired) *retired = lru->second.first;
            ids_.erase(lru);
        }
        const uint64_t id = next_++;  // never re-used: a retired name's memory cannot be mistaken for a new filter's
        ids_.emplace(std::move(c), std::make_pair(id, ++clock_));
        return id;
    }
};

struct Served {
    std::string keyspace, name, pk = "id";
    vs_hnsw* h = nullptr;
    size_t dim = 0;
    int metric = VS_METRIC_COS;
    size_t connectivity = 16, expansion_add = 128, expansion_search = 64;
    std::atomic<int> serving{0};  // 0 = BOOTSTRAPPING, 1 = SERVING
    std::atomic<size_t> count{0};
    std::atomic<double> progress{0.0};
    FilterNames names;  // per index: ids of the filters its requests have carried
};

const char* similarity_name(int m) {
    return m == VS_METRIC_COS ? "COSINE" : m == VS_METRIC_L2SQ ? "EUCLIDEAN" : m == VS_METRIC_IP ? "DOT_PRODUCT" : "HAMMING";
}

// ------------------------------------------------------------------------------------------ HTTP plumbing
struct Response {
    int code = 200;
    bool json = true;
    std::string body;
};

const char* reason(int code) {
    switch (code) {
        case 200: return "OK";
        case 400: return "Bad Request";
        case 404: return "Not Found";
        case 405: return "Method Not Allowed";
        case 413: return "Payload Too Large";
        case 500: return "Internal Server Error";
        case 503: return "Service Unavailable";
        default: return "Error";
    }
}

struct Worker;

struct Conn {
    int fd = -1;
    uint64_t gen = 0;
    std::string in, out;
    size_t out_off = 0;
    bool busy = false;   // a request is with the engine
    bool close_after = false;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();  // last byte received
};

struct Pending {  // one query in flight
    Worker* w;
    int fd;
    uint64_t gen;
    Served* s;
    size_t k;
    std::vector<float> q;
    std::vector<uint64_t> keys;
    std::vector<float> dist;
    size_t found = 0;
    int status = VS_OK;
    std::string err;
    bool keep_alive = true;
    std::vector<Test> tests;  // filtered queries
    bool blocking = false;    // served by the pool through a blocking call (filters; limits beyond the LDS beam)
};

std::string ann_body(const Pending& p) {
    std::string o;
    o.reserve(64 + p.found * 40);
    o += "{\"primary_keys\":{\"";
    o += json_escape(p.s->pk);
    o += "\":[";
    for (size_t i = 0; i < p.found; ++i) {
        if (i) o += ',';
        o += std::to_string(p.keys[i] & kRowMask);
    }
    o += "]},\"distances\":[";
    for (size_t i = 0; i < p.found; ++i) {
        if (i) o += ',';
        put_f32(o, p.dist[i]);
    }
    o += "],\"similarity_scores\":[";
    for (size_t i = 0; i < p.found; ++i) {
        if (i) o += ',';
        put_f32(o, vs_similarity_score(p.dist[i], p.s->metric, p.s->dim));
    }
    o += "]}";
    return o;
}

struct FilterPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Pending>> q;
    std::vector<std::thread> th;
    void start(int n, std::function<void(std::unique_ptr<Pending>)> done) {
        for (int i = 0; i < n; ++i)
            th.emplace_back([this, done] {
                for (;;) {
                    std::unique_ptr<Pending> p;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [this] { return !q.empty(); });
                        p = std::move(q.front());
                        q.pop_front();
                    }
                    auto pred = [](uint64_t key, void* ctx) -> int {
                        const auto* tests = static_cast<const std::vector<Test>*>(ctx);
                        const int64_t row = (int64_t)(key & kRowMask);
                        for (const Test& t : *tests)
                            if (!t(row)) return 0;
                        return 1;
                    };
                    if (p->tests.empty())  // no filter: a limit beyond the LDS beam (exhaustive ranking inside the engine)
                        p->status = vs_hnsw_search(p->s->h, p->q.data(), p->q.size(), p->k, p->keys.data(), p->dist.data(), &p->found);
                    else
                    {
                        uint64_t retired = 0;
                        const uint64_t name = p->s->names.name_of(p->tests, &retired);
                        if (retired) (void)vs_hnsw_filter_forget(p->s->h, retired, nullptr);
                        p->status = vs_hnsw_filtered_search_keyed(p->s->h, p->q.data(), p->q.size(), p->k, pred, &p->tests, name, p->keys.data(), p->dist.data(),
                                                                  &p->found);
                    }
                    if (p->status != VS_OK) p->err = vs_hnsw_last_error();
                    done(std::move(p));
                }
            });
        for (auto& t : th) t.detach();
    }
    void submit(std::unique_ptr<Pending> p) {
        {
            std::lock_guard<std::mutex> g(mu);
            q.push_back(std::move(p));
        }
        cv.notify_one();
    }
};

struct Server {
    std::map<std::pair<std::string, std::string>, Served*> indexes;
    std::string engine = std::string("hip-hnsw-") + vs_hnsw_version();
    FilterPool filters;
};

struct Worker {
    Server* srv = nullptr;
    int ep = -1, lfd = -1, evfd = -1;
    uint64_t next_gen = 1;
    std::unordered_map<int, Conn> conns;
    std::mutex done_mu;
    std::vector<std::unique_ptr<Pending>> done;

    void post(std::unique_ptr<Pending> p) {  // any thread
        {
            std::lock_guard<std::mutex> g(done_mu);
            done.push_back(std::move(p));
        }
        uint64_t one = 1;
        (void)!write(evfd, &one, 8);
    }

    static void on_search_done(void* ctx, int status) {  // engine dispatcher thread: must not block
        std::unique_ptr<Pending> p(static_cast<Pending*>(ctx));
        p->status = status;
        if (status != VS_OK) p->err = vs_hnsw_last_error();
        Worker* w = p->w;
        w->post(std::move(p));
    }

    void queue_response(Conn& c, const Response& r, bool keep_alive) {
        char head[256];
        int n = std::snprintf(head, sizeof head,
                              "HTTP/1.1 %d %s\r\ncontent-type: %s\r\ncontent-length: %zu\r\n%s\r\n", r.code, reason(r.code),
                              r.json ? "application/json" : "text/plain; charset=utf-8", r.body.size(),
                              keep_alive ? "" : "connection: close\r\n");
        c.out.append(head, (size_t)n);
        c.out += r.body;
        if (!keep_alive) c.close_after = true;
    }

    void close_conn(int fd) {
        epoll_ctl(ep, EPOLL_CTL_DEL, fd, nullptr);
        ::close(fd);
        conns.erase(fd);
    }

    // Returns false when the connection was closed.
    bool flush(Conn& c) {
        while (c.out_off < c.out.size()) {
            ssize_t n = ::send(c.fd, c.out.data() + c.out_off, c.out.size() - c.out_off, MSG_NOSIGNAL);
            if (n > 0) {
                c.out_off += (size_t)n;
            } else if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
                epoll_event ev{};
                ev.events = EPOLLIN | EPOLLOUT;
                ev.data.fd = c.fd;
                epoll_ctl(ep, EPOLL_CTL_MOD, c.fd, &ev);
                return true;
            } else {
                close_conn(c.fd);
                return false;
            }
        }
        c.out.clear();
        c.out_off = 0;
        if (c.close_after) {
            close_conn(c.fd);
            return false;
        }
        epoll_event ev{};
        ev.events = EPOLLIN;
        ev.data.fd = c.fd;
        epoll_ctl(ep, EPOLL_CTL_MOD, c.fd, &ev);
        return true;
    }

    Response status_of(const Served& s) {
        char b[160];
        std::snprintf(b, sizeof b, "{\"status\":\"%s\",\"count\":%zu,\"build_progress\":%.2f}", s.serving ? "SERVING" : "BOOTSTRAPPING",
                      s.count.load(), s.progress.load());
        return {200, true, b};
    }

    // One complete request.  Returns true when the answer is deferred (query in flight).
    bool handle(Conn& c, const std::string& method, const std::string& path, const char* body, size_t blen, bool keep_alive) {
        auto reply = [&](int code, bool json, std::string text) {
            queue_response(c, Response{code, json, std::move(text)}, keep_alive);
            return false;
        };
        std::vector<std::string> seg;
        {
            size_t i = 0;
            std::string p = path.substr(0, path.find('?'));
            while (i < p.size()) {
                size_t j = p.find('/', i);
                if (j == std::string::npos) j = p.size();
                if (j > i) seg.push_back(p.substr(i, j - i));
                i = j + 1;
            }
        }
        if (seg.size() < 3 || seg[0] != "api" || seg[1] != "v1") return reply(404, false, "not found");
        if (seg.size() == 3 && seg[2] == "info") {
            if (method != "GET") return reply(405, false, "method not allowed");
            return reply(200, true, "{\"engine\":\"" + json_escape(srv->engine) + "\",\"service\":\"vector-store\",\"version\":\"0.1.0\"}");
        }
        if (seg.size() == 3 && seg[2] == "status") {
            if (method != "GET") return reply(405, false, "method not allowed");
            return reply(200, true, "\"SERVING\"");
        }
        if (seg.size() == 3 && seg[2] == "indexes") {
            if (method != "GET") return reply(405, false, "method not allowed");
            std::string o = "[";
            bool first = true;
            for (auto& kv : srv->indexes) {
                const Served& s = *kv.second;
                if (!first) o += ',';
                first = false;
                o += "{\"keyspace\":\"" + json_escape(s.keyspace) + "\",\"index\":\"" + json_escape(s.name) +
                     "\",\"options\":{\"type\":\"vector\",\"dimensions\":" + std::to_string(s.dim) +
                     ",\"maximum_node_connections\":" + std::to_string(s.connectivity) +
                     ",\"construction_beam_width\":" + std::to_string(s.expansion_add) +
                     ",\"search_beam_width\":" + std::to_string(s.expansion_search) + ",\"similarity_function\":\"" +
                     similarity_name(s.metric) + "\",\"quantization\":\"F32\"}}";
            }
            return reply(200, true, o + "]");
        }
        if (seg.size() == 6 && seg[2] == "indexes") {
            auto it = srv->indexes.find({seg[3], seg[4]});
            if (it == srv->indexes.end()) return reply(404, false, "missing index: " + seg[3] + "." + seg[4]);
            Served& s = *it->second;
            if (seg[5] == "status") {
                if (method != "GET") return reply(405, false, "method not allowed");
                Response r = status_of(s);
                queue_response(c, r, keep_alive);
                return false;
            }
            if (seg[5] == "ann") {
                if (method != "POST") return reply(405, false, "method not allowed");
                if (!s.serving)
                    return reply(503, true, "{\"reason\":\"INDEX_BUILDING\",\"message\":\"index " + json_escape(seg[3] + "." + seg[4]) +
                                                " is BOOTSTRAPPING\"}");
                std::unique_ptr<Pending> p(new Pending());
                try {
                    JVal req;
                    try {
                        req = JsonParser(body, blen).parse();
                    } catch (const JsonError& e) {
                        throw BadRequest(std::string("malformed request: ") + e.what());
                    }
                    if (req.kind != JVal::Obj) throw BadRequest("malformed request: expected an object");
                    const JVal* vec = req.get("vector");
                    if (!vec) throw BadRequest("malformed request: missing field `vector`");
                    if (vec->kind != JVal::Arr) throw BadRequest("vector must be an array of numbers");
                    if (vec->is_fvec) {
                        p->q = vec->fvec;
                    } else {
                        p->q.reserve(vec->arr.size());
                        for (const JVal& x : vec->arr) {
                            if (x.kind != JVal::Num) throw BadRequest("vector must be an array of numbers");
                            p->q.push_back((float)x.num);
                        }
                    }
                    size_t limit = 1;  // Limit::default() (lib.rs:289-293)
                    if (const JVal* l = req.get("limit")) {
                        if (l->kind != JVal::Num || !l->is_int || l->i < 1) throw BadRequest("limit must be a positive integer");
                        limit = (size_t)l->i;
                    }
                    if (p->q.size() != s.dim)  // validator.rs:12-26 -> 400
                        throw BadRequest("wrong embedding dimension: got " + std::to_string(p->q.size()) + ", index has " + std::to_string(s.dim));
                    const JVal* flt = req.get("filter");
                    if (flt && flt->kind != JVal::Null) {
                        if (flt->kind != JVal::Obj) throw BadRequest("filter must be an object");
                        p->tests = compile_filter(*flt, s.pk);
                    }
                    // The engine never returns more than the members it holds, so `limit` is clamped to that before anything is
                    // sized by it: an absurd limit (2^62) must cost a 200 with `count` hits, not a bad_alloc on the event loop
                    // (advisor finding, round 1: one unauthenticated request took the server down).
                    p->k = std::min<size_t>(limit, std::max<size_t>(1, vs_hnsw_size(s.h)));
                } catch (const BadRequest& e) {
                    return reply(400, false, e.what());
                }
                p->w = this;
                p->fd = c.fd;
                p->gen = c.gen;
                p->s = &s;
                p->keep_alive = keep_alive;
                p->keys.resize(p->k);
                p->dist.resize(p->k);
                c.busy = true;
                p->blocking = !p->tests.empty() || p->k > 512;  // the reference accepts any limit (httproutes.rs:842-847)
                if (p->blocking) {
                    srv->filters.submit(std::move(p));
                    return true;
                }
                Pending* raw = p.release();
                int rc = vs_hnsw_search_async(s.h, raw->q.data(), raw->q.size(), raw->k, raw->keys.data(), raw->dist.data(), &raw->found,
                                              &Worker::on_search_done, raw);
                if (rc != VS_OK) {  // rejected synchronously: the callback will not run
                    std::unique_ptr<Pending> back(raw);
                    c.busy = false;
                    std::string msg = vs_hnsw_last_error();
                    return reply(rc == VS_ERR_DIMENSION ? 400 : 500, false, "index.ann request error: " + msg);
                }
                return true;
            }
        }
        return reply(404, false, "not found");
    }

    // Parse as many complete requests as the connection may start (one at a time: answers keep their order).
    void pump(int fd) {
        auto it = conns.find(fd);
        if (it == conns.end()) return;
        Conn& c = it->second;
        while (!c.busy && !c.close_after) {
            size_t he = c.in.find("\r\n\r\n");
            if (he == std::string::npos) {
                if (c.in.size() > 65536) {
                    queue_response(c, Response{400, false, "header too large"}, false);
                }
                break;  // the rest of the header is still arriving
            }
            size_t le = c.in.find("\r\n");
            std::string line = c.in.substr(0, le);
            size_t s1 = line.find(' '), s2 = line.rfind(' ');
            if (s1 == std::string::npos || s2 == s1) {
                queue_response(c, Response{400, false, "bad request line"}, false);
                break;
            }
            std::string method = line.substr(0, s1), path = line.substr(s1 + 1, s2 - s1 - 1), ver = line.substr(s2 + 1);
            size_t clen = 0;
            bool keep = ver != "HTTP/1.0";
            size_t pos = le + 2;
            while (pos < he) {
                size_t e = c.in.find("\r\n", pos);
                std::string h = c.in.substr(pos, e - pos);
                pos = e + 2;
                size_t colon = h.find(':');
                if (colon == std::string::npos) continue;
                std::string name = h.substr(0, colon), val = h.substr(colon + 1);
                for (auto& ch : name) ch = (char)std::tolower((unsigned char)ch);
                size_t a = val.find_first_not_of(" \t");
                val = a == std::string::npos ? "" : val.substr(a);
                if (name == "content-length") clen = (size_t)std::strtoull(val.c_str(), nullptr, 10);
                if (name == "connection") {
                    for (auto& ch : val) ch = (char)std::tolower((unsigned char)ch);
                    if (val.find("close") != std::string::npos) keep = false;
                    if (val.find("keep-alive") != std::string::npos) keep = true;
                }
            }
            if (clen > (64u << 20)) {
                queue_response(c, Response{413, false, "body too large"}, false);
                break;
            }
            if (c.in.size() < he + 4 + clen) break;  // body still arriving
            bool deferred = false;
            try {
                deferred = handle(c, method, path, c.in.data() + he + 4, clen, keep);
            } catch (const std::exception& e) {  // nothing may unwind into the event loop: answer 500 and go on
                c.busy = false;
                queue_response(c, Response{500, false, std::string("internal error: ") + e.what()}, false);
            }
            c.in.erase(0, he + 4 + clen);
            if (deferred) break;
        }
        if (!c.out.empty()) flush(c);  // answers queued by this pass (after it `c` may be gone)
    }

    void complete(std::unique_ptr<Pending> p) {
        try {
            complete_unguarded(std::move(p));
        } catch (const std::exception&) {  // a response that cannot be built (out of memory) drops that answer, not the server
        }
    }
    void complete_unguarded(std::unique_ptr<Pending> p) {
        auto it = conns.find(p->fd);
        if (it == conns.end() || it->second.gen != p->gen) return;  // the client went away
        Conn& c = it->second;
        c.busy = false;
        bool in_range = true;  // Distance::try_from (distance.rs:58-105): an out-of-range distance fails the request (usearch.rs:219)
        for (size_t i = 0; i < p->found && p->status == VS_OK; ++i) in_range = in_range && vs_distance_valid(p->dist[i], p->s->metric, p->s->dim);
        if (p->status == VS_OK && !in_range) queue_response(c, Response{500, false, "index.ann request error: ann: search failed (distance out of range)"}, p->keep_alive);
        else if (p->status == VS_OK) queue_response(c, Response{200, true, ann_body(*p)}, p->keep_alive);
        else queue_response(c, Response{p->status == VS_ERR_DIMENSION ? 400 : 500, false, "index.ann request error: " + p->err}, p->keep_alive);
        const int fd = c.fd;
        if (flush(c)) pump(fd);  // a pipelined request may already be waiting
    }

    void run() {
        epoll_event evs[256];
        auto swept = std::chrono::steady_clock::now();
        for (;;) {
            int n = epoll_wait(ep, evs, 256, 1000);
            const auto now = std::chrono::steady_clock::now();
            if (now - swept > std::chrono::seconds(5)) {  // connections that sent nothing for kIdleSeconds (half requests, dead peers)
                swept = now;
                std::vector<int> idle;
                for (auto& kv : conns)
                    if (!kv.second.busy && kv.second.out.empty() && now - kv.second.last > std::chrono::seconds(kIdleSeconds)) idle.push_back(kv.first);
                for (int fd : idle) close_conn(fd);
            }
            for (int i = 0; i < n; ++i) {
                int fd = evs[i].data.fd;
                if (fd == lfd) {
                    for (;;) {
                        int cfd = accept4(lfd, nullptr, nullptr, SOCK_NONBLOCK | SOCK_CLOEXEC);
                        if (cfd < 0) break;
                        int one = 1;
                        setsockopt(cfd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                        Conn& c = conns[cfd];
                        c = Conn();
                        c.fd = cfd;
                        c.gen = next_gen++;
                        epoll_event ev{};
                        ev.events = EPOLLIN;
                        ev.data.fd = cfd;
                        epoll_ctl(ep, EPOLL_CTL_ADD, cfd, &ev);
                    }
                } else if (fd == evfd) {
                    uint64_t v;
                    (void)!read(evfd, &v, 8);
                    std::vector<std::unique_ptr<Pending>> batch;
                    {
                        std::lock_guard<std::mutex> g(done_mu);
                        batch.swap(done);
                    }
                    for (auto& p : batch) complete(std::move(p));
                } else {
                    auto it = conns.find(fd);
                    if (it == conns.end()) continue;
                    Conn& c = it->second;
                    if (evs[i].events & (EPOLLHUP | EPOLLERR)) {
                        close_conn(fd);
                        continue;
                    }
                    if (evs[i].events & EPOLLOUT) {
                        if (!flush(c)) continue;
                    }
                    if (evs[i].events & EPOLLIN) {
                        char buf[65536];
                        bool closed = false;
                        for (;;) {
                            ssize_t r = ::recv(fd, buf, sizeof buf, 0);
                            if (r > 0) {
                                c.last = std::chrono::steady_clock::now();
                                if (c.in.size() + (size_t)r > kMaxBuffered) {  // a pipelined flood while a request is in flight
                                    closed = true;
                                    break;
                                }
                                c.in.append(buf, (size_t)r);
                                if ((size_t)r < sizeof buf) break;
                            } else if (r == 0) {
                                closed = true;
                                break;
                            } else {
                                if (errno != EAGAIN && errno != EWOULDBLOCK) closed = true;
                                break;
                            }
                        }
                        if (closed && !c.busy && c.out.empty()) {
                            close_conn(fd);
                            continue;
                        }
                        if (closed) c.close_after = true;
                        pump(fd);
                    }
                }
            }
        }
    }
};

struct Args {
    std::map<std::string, std::string> kv;
    std::string get(const std::string& k, const std::string& d = "") const {
        auto it = kv.find(k);
        return it == kv.end() ? d : it->second;
    }
    long num(const std::string& k, long d) const { return kv.count(k) ? std::atol(kv.at(k).c_str()) : d; }
};

int metric_code(const std::string& m) {
    if (m == "cos") return VS_METRIC_COS;
    if (m == "l2sq" || m == "euclidean") return VS_METRIC_L2SQ;
    if (m == "ip" || m == "dot") return VS_METRIC_IP;
    throw std::runtime_error("unknown metric " + m);
}

}  // namespace

// Parser / filter checks that need no GPU (tests/test_bench_driver.py runs them on CPU).
static int selftest() {
    int bad = 0;
    auto expect = [&](bool ok, const char* what) {
        if (!ok) {
            ++bad;
            std::cerr << "selftest FAILED: " << what << std::endl;
        }
    };
    // the float fast path agrees with strtof bit for bit
    std::vector<std::string> texts = {"0", "-0", "0.0", "1", "-1", "0.1", "0.5", "3.4028235e38", "1e-45", "1.17549435e-38", "1.5e-40",
                                      "123456789", "0.000001", "1E5", "1e+5", "2.5e-3", "16777217", "0.30000001192092896", "9007199254740993",
                                      "0.1234567890123456789012", "1e23", "-7.0e-23", "4.9406564584124654e-324", "1e39"};
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < 20000; ++i) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        uint32_t bits = (uint32_t)x;
        float f;
        std::memcpy(&f, &bits, 4);
        if (!std::isfinite(f)) continue;
        char b[40];
        std::snprintf(b, sizeof b, i % 3 == 0 ? "%.9g" : i % 3 == 1 ? "%.8e" : "%.12f", (double)f);
        if (std::strlen(b) < 38) texts.push_back(b);
    }
    std::string doc = "{\"vector\":[";
    for (size_t i = 0; i < texts.size(); ++i) doc += (i ? ", " : " ") + texts[i];
    doc += " ],\"limit\":7}";
    JVal v = JsonParser(doc.data(), doc.size()).parse();
    const JVal* vec = v.get("vector");
    expect(vec && vec->is_fvec && vec->fvec.size() == texts.size(), "fast vector path taken");
    if (vec && vec->is_fvec)
        for (size_t i = 0; i < texts.size(); ++i) {
            float want = std::strtof(texts[i].c_str(), nullptr), got = vec->fvec[i];
            if (std::memcmp(&want, &got, 4) != 0) {
                ++bad;
                std::cerr << "selftest FAILED: float " << texts[i] << " -> " << got << ", strtof " << want << std::endl;
            }
        }
    expect(v.get("limit") && v.get("limit")->is_int && v.get("limit")->i == 7, "limit");
    // a vector with a non-number falls back to the generic parser (and is then rejected by the route)
    std::string mixed = "{\"vector\":[1, true, 2]}";
    JVal m = JsonParser(mixed.data(), mixed.size()).parse();
    expect(m.get("vector") && !m.get("vector")->is_fvec && m.get("vector")->arr.size() == 3 && m.get("vector")->arr[1].kind == JVal::Bool, "mixed vector");
    for (const char* broken : {"{not json", "{\"vector\":[1,}", "{\"vector\":[1 2]}", "[1,2", "{\"a\":1}x", "{\"vector\":[1e]}"}) {
        bool threw = false;
        try {
            JsonParser(broken, std::strlen(broken)).parse();
        } catch (const JsonError&) {
            threw = true;
        }
        expect(threw, broken);
    }
    // mutated bodies: the parser either returns a value or throws JsonError -- never reads out of bounds (run under
    // -fsanitize=address,undefined by `make -C vector_store_amd/csrc sanitize-httpd`)
    {
        const std::string seed_doc = "{\"vector\":[0.25,-1e-3,3,4.5e+2,7],\"limit\":3,\"filter\":{\"restrictions\":[{\"type\":\"()IN()\",\"lhs\":[\"id\"],"
                                     "\"rhs\":[[1],[2]]},{\"type\":\"<\",\"lhs\":\"id\",\"rhs\":9}],\"allow_filtering\":true},\"x\":\"a\\u00e9\\n\"}";
        uint64_t r = 0x9E3779B97F4A7C15ull;
        auto rnd = [&] {
            r ^= r << 13;
            r ^= r >> 7;
            r ^= r << 17;
            return r;
        };
        const char alphabet[] = "{}[]\",:.-+eE0123456789 tfn\\u\"";
        size_t parsed = 0, rejected = 0;
        for (int it = 0; it < 20000; ++it) {
            std::string d = seed_doc;
            const int edits = 1 + (int)(rnd() % 4);
            for (int e = 0; e < edits; ++e) {
                const size_t pos = rnd() % d.size();
                switch (rnd() % 3) {
                    case 0: d[pos] = alphabet[rnd() % (sizeof alphabet - 1)]; break;
                    case 1: d.erase(pos, 1 + rnd() % 3); break;
                    default: d.insert(pos, 1, alphabet[rnd() % (sizeof alphabet - 1)]);
                }
                if (d.empty()) d = "x";
            }
            // exact-size heap buffer: an over-read is an ASAN report
            std::unique_ptr<char[]> buf(new char[d.size()]);
            std::memcpy(buf.get(), d.data(), d.size());
            try {
                JVal v = JsonParser(buf.get(), d.size()).parse();
                ++parsed;
                if (v.kind == JVal::Obj)
                    if (const JVal* f = v.get("filter"))
                        if (f->kind == JVal::Obj) {
                            try {
                                (void)compile_filter(*f, "id");
                            } catch (const BadRequest&) {
                            }
                        }
            } catch (const JsonError&) {
                ++rejected;
            }
        }
        expect(parsed > 100 && rejected > 1000, "mutation run saw both outcomes");
    }
    // filters: every restriction form one integer column admits (httpapi/src/lib.rs:323-366)
    auto rows = [&](const std::string& restrictions) {
        std::string f = "{\"restrictions\":" + restrictions + ",\"allow_filtering\":true}";
        JVal fv = JsonParser(f.data(), f.size()).parse();
        std::vector<Test> tests = compile_filter(fv, "id");
        std::vector<int> out;
        for (int r = 0; r < 30; ++r) {
            bool okr = true;
            for (auto& t : tests) okr = okr && t(r);
            if (okr) out.push_back(r);
        }
        return out;
    };
    using V = std::vector<int>;
    expect(rows("[{\"type\":\"<\",\"lhs\":\"id\",\"rhs\":3}]") == V({0, 1, 2}), "<");
    expect(rows("[{\"type\":\"<=\",\"lhs\":\"id\",\"rhs\":3}]") == V({0, 1, 2, 3}), "<=");
    expect(rows("[{\"type\":\">\",\"lhs\":\"id\",\"rhs\":26}]") == V({27, 28, 29}), ">");
    expect(rows("[{\"type\":\">=\",\"lhs\":\"id\",\"rhs\":27},{\"type\":\"<\",\"lhs\":\"id\",\"rhs\":29}]") == V({27, 28}), ">= and <");
    expect(rows("[{\"type\":\"==\",\"lhs\":\"id\",\"rhs\":15}]") == V({15}), "==");
    expect(rows("[{\"type\":\"IN\",\"lhs\":\"id\",\"rhs\":[1,12,23]}]") == V({1, 12, 23}), "IN");
    expect(rows("[{\"type\":\"()==()\",\"lhs\":[\"id\"],\"rhs\":[7]}]") == V({7}), "()==()");
    expect(rows("[{\"type\":\"()IN()\",\"lhs\":[\"id\"],\"rhs\":[[7],[9]]}]") == V({7, 9}), "()IN()");
    expect(rows("[{\"type\":\"()<()\",\"lhs\":[\"id\"],\"rhs\":[2]}]") == V({0, 1}), "()<()");
    expect(rows("[{\"type\":\"()<=()\",\"lhs\":[\"id\"],\"rhs\":[1]}]") == V({0, 1}), "()<=()");
    expect(rows("[{\"type\":\"()>()\",\"lhs\":[\"id\"],\"rhs\":[27]}]") == V({28, 29}), "()>()");
    expect(rows("[{\"type\":\"()>=()\",\"lhs\":[\"id\"],\"rhs\":[28]}]") == V({28, 29}), "()>=()");
    for (const char* rej : {"[{\"type\":\"<\",\"lhs\":\"ck\",\"rhs\":3}]", "[{\"type\":\"~\",\"lhs\":\"id\",\"rhs\":3}]",
                            "[{\"type\":\"<\",\"lhs\":\"id\",\"rhs\":1.5}]", "[{\"type\":\"()<()\",\"lhs\":[\"id\",\"x\"],\"rhs\":[1,2]}]"}) {
        bool threw = false;
        try {
            rows(rej);
        } catch (const BadRequest&) {
            threw = true;
        }
        expect(threw, rej);
    }
    // number rendering: shortest text that reads back, +-inf saturate (httpapi/src/lib.rs:397-409)
    auto text = [](float f) {
        std::string o;
        put_f32(o, f);
        return o;
    };
    expect(text(0.5f) == "0.5" && text(1.0f) == "1.0" && text(-1.0f) == "-1.0" && text(9.0f) == "9.0", "plain floats");
    expect(std::strtof(text(0.1f).c_str(), nullptr) == 0.1f && text(0.1f) == "0.1", "0.1");
    expect(std::strtof(text(INFINITY).c_str(), nullptr) == FLT_MAX && std::strtof(text(-INFINITY).c_str(), nullptr) == -FLT_MAX, "saturation");
    std::cout << (bad ? "selftest FAILED" : "selftest ok") << std::endl;
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    signal(SIGPIPE, SIG_IGN);
    if (argc > 1 && std::string(argv[1]) == "selftest") return selftest();
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (k.rfind("--", 0) != 0) continue;
        k = k.substr(2);
        std::string v = (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) ? argv[++i] : "1";
        a.kv[k] = v;
    }
    try {
        const std::string dir = a.get("data-dir");
        if (dir.empty()) {
            std::cerr << "usage: vs_httpd --data-dir D [--keyspace K] [--index I] [--metric cos|l2sq|ip] [--host H] [--port P] [--threads T]\n";
            return 2;
        }
        static Served s;
        s.keyspace = a.get("keyspace", "vsb_keyspace");
        s.name = a.get("index", "vsb_index");
        s.pk = a.get("pk-column", "id");
        s.metric = metric_code(a.get("metric", "cos"));
        s.connectivity = (size_t)a.num("connectivity", 16);
        s.expansion_add = (size_t)a.num("expansion-add", 128);
        s.expansion_search = (size_t)a.num("expansion-search", 64);
        auto cfg = vsb::read_dataset_toml(dir);
        static vsb::Matrix base = vsb::read_bin(dir + "/" + cfg.data_fbin, false);
        size_t n = base.count;
        if (a.num("max-vectors", 0) > 0) n = std::min<size_t>(n, (size_t)a.num("max-vectors", 0));
        s.dim = base.dim;
        vs_hnsw_options o{};
        o.dimensions = s.dim;
        o.connectivity = s.connectivity;
        o.expansion_add = s.expansion_add;
        o.expansion_search = s.expansion_search;
        o.metric = s.metric;
        o.quantization = VS_SCALAR_F32;
        o.device = -1;
        if (vs_hnsw_create(&o, &s.h) != VS_OK) throw std::runtime_error(std::string("create: ") + vs_hnsw_last_error());

        static Server srv;
        srv.indexes[{s.keyspace, s.name}] = &s;

        const int T = (int)std::max(1l, a.num("threads", 4));
        const std::string host = a.get("host", "127.0.0.1");
        const int port = (int)a.num("port", 6080);
        static std::vector<std::unique_ptr<Worker>> workers;
        for (int t = 0; t < T; ++t) {
            std::unique_ptr<Worker> w(new Worker());
            w->srv = &srv;
            w->ep = epoll_create1(EPOLL_CLOEXEC);
            w->evfd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
            w->lfd = socket(AF_INET, SOCK_STREAM | SOCK_NONBLOCK | SOCK_CLOEXEC, 0);
            int one = 1;
            setsockopt(w->lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
            setsockopt(w->lfd, SOL_SOCKET, SO_REUSEPORT, &one, sizeof one);
            sockaddr_in sa{};
            sa.sin_family = AF_INET;
            sa.sin_port = htons((uint16_t)port);
            if (inet_pton(AF_INET, host.c_str(), &sa.sin_addr) != 1) throw std::runtime_error("bad --host " + host);
            if (bind(w->lfd, (sockaddr*)&sa, sizeof sa) != 0 || listen(w->lfd, 1024) != 0)
                throw std::runtime_error("cannot listen on " + host + ":" + std::to_string(port) + ": " + std::strerror(errno));
            epoll_event ev{};
            ev.events = EPOLLIN;
            ev.data.fd = w->lfd;
            epoll_ctl(w->ep, EPOLL_CTL_ADD, w->lfd, &ev);
            ev.data.fd = w->evfd;
            epoll_ctl(w->ep, EPOLL_CTL_ADD, w->evfd, &ev);
            workers.push_back(std::move(w));
        }
        srv.filters.start(4, [](std::unique_ptr<Pending> p) {
            Worker* w = p->w;
            w->post(std::move(p));
        });
        // Build in the background: BOOTSTRAPPING -> SERVING is what `build-index` polls (benchmark vs.rs:17-39).
        std::thread([n] {
            try {
                if (vs_hnsw_reserve(s.h, n, 0) != VS_OK) throw std::runtime_error(vs_hnsw_last_error());
                const size_t step = 1u << 18;
                std::vector<uint64_t> keys(step);
                for (size_t i = 0; i < n; i += step) {
                    const size_t m = std::min(step, n - i);
                    for (size_t j = 0; j < m; ++j) keys[j] = i + j;
                    if (vs_hnsw_add_batch(s.h, keys.data(), base.f.data() + i * s.dim, m, s.dim) != VS_OK)
                        throw std::runtime_error(vs_hnsw_last_error());
                    s.count = i + m;
                    s.progress = 100.0 * (double)(i + m) / (double)n;
                }
                s.progress = 100.0;
                s.serving = 1;
                std::cerr << "vs_httpd: index " << s.keyspace << "." << s.name << " SERVING, " << n << " vectors" << std::endl;
            } catch (const std::exception& e) {
                std::cerr << "vs_httpd: build failed: " << e.what() << std::endl;
                std::_Exit(1);
            }
        }).detach();
        std::cerr << "vs_httpd: listening on " << host << ":" << port << " (" << T << " threads), engine " << srv.engine << std::endl;
        std::vector<std::thread> th;
        for (auto& w : workers) th.emplace_back([&w] { w->run(); });
        for (auto& t : th) t.join();
    } catch (const std::exception& e) {
        std::cerr << "vs_httpd: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
