"""vs_httpd -- the native /ann HTTP surface -- against the same wire-format checks as the
Python twin (tests/test_httpd.py: httproutes.rs:661-904, httpapi/src/lib.rs), over real sockets, plus the native
`vs_bench search-http` client (the reference's search-http scenario, crates/benchmark main.rs:435-525)."""
import http.client
import json
import os
import socket
import struct
import subprocess
import time

import numpy as np
import pytest

from tests import kat_runner as K

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HTTPD = os.path.join(ROOT, "vector_store_amd", "vs_httpd")
VS_BENCH = os.path.join(ROOT, "vector_store_amd", "vs_bench")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _write_fbin(path, rows, dim):
    rows = np.asarray(rows, dtype="<f4").reshape(-1, dim)
    with open(path, "wb") as f:
        f.write(struct.pack("<II", rows.shape[0], dim))
        f.write(rows.tobytes())


class Server:
    def __init__(self, data_dir, metric, extra=()):
        self.port = _free_port()
        self.proc = subprocess.Popen([HTTPD, "--data-dir", data_dir, "--keyspace", "ks", "--index", "idx", "--metric", metric,
                                      "--port", str(self.port), "--threads", "2", *extra], stderr=subprocess.PIPE)
        deadline = time.time() + 120
        while True:
            try:
                st, body = self.get("/api/v1/indexes/ks/idx/status")
                if st == 200 and json.loads(body)["status"] == "SERVING":
                    break
            except OSError:
                pass
            if self.proc.poll() is not None:
                raise RuntimeError("vs_httpd exited: " + self.proc.stderr.read().decode())
            if time.time() > deadline:
                self.close()
                raise RuntimeError("vs_httpd never reached SERVING")
            time.sleep(0.1)

    def request(self, method, path, body=None, raw=None):
        c = http.client.HTTPConnection("127.0.0.1", self.port, timeout=30)
        data = raw if raw is not None else (json.dumps(body).encode() if body is not None else None)
        c.request(method, path, body=data, headers={"content-type": "application/json"} if data is not None else {})
        r = c.getresponse()
        out = r.status, r.read().decode()
        c.close()
        return out

    def get(self, path):
        return self.request("GET", path)

    def ann(self, body=None, raw=None, index="idx"):
        return self.request("POST", f"/api/v1/indexes/ks/{index}/ann", body, raw)

    def close(self):
        self.proc.kill()
        self.proc.wait()


def _serve_kat(tmp_path, name, metric):
    t = K.KAT[name]
    rows = sorted(t["base"], key=lambda r: r["key"])
    first = rows[0]["key"] if rows else 0
    assert [r["key"] for r in rows] == list(range(first, first + len(rows)))  # fbin numbers rows 0..n-1 (fbin.rs:86)
    d = tmp_path / name
    d.mkdir()
    _write_fbin(d / "data.fbin", [r["vector"] for r in rows], t["dim"])
    return Server(str(d), metric), t, first


def test_wire_format_status_codes_and_scores(tmp_path):
    s, t, off = _serve_kat(tmp_path, "B3_l2sq_1d_scores", "l2sq")
    try:
        st, body = s.ann({"vector": t["query"], "limit": 3})
        assert st == 200
        body = json.loads(body)
        assert body["primary_keys"] == {"id": [0, 1, 2]}
        assert body["distances"] == [0.0, 1.0, 9.0]
        assert body["similarity_scores"] == pytest.approx([1.0, 0.5, 0.1], abs=1e-5)
        assert json.loads(s.ann({"vector": t["query"]})[1])["primary_keys"] == {"id": [0]}  # default limit 1
        assert json.loads(s.get("/api/v1/indexes/ks/idx/status")[1]) == {"status": "SERVING", "count": 3, "build_progress": 100.0}
        info = json.loads(s.get("/api/v1/info")[1])
        assert info["service"] == "vector-store" and info["engine"].startswith("hip-hnsw-")
        assert json.loads(s.get("/api/v1/status")[1]) == "SERVING"
        lst = json.loads(s.get("/api/v1/indexes")[1])
        assert lst[0]["keyspace"] == "ks" and lst[0]["index"] == "idx" and lst[0]["options"]["type"] == "vector"
        assert lst[0]["options"]["dimensions"] == 1 and lst[0]["options"]["similarity_function"] == "EUCLIDEAN"
        assert s.ann({"vector": [0.0]}, index="nope")[0] == 404
        assert s.get("/api/v1/indexes/ks/nope/status")[0] == 404
        assert s.ann({"vector": [0.0, 1.0]})[0] == 400       # wrong dimension (validator.rs:12-26)
        assert s.ann({"limit": 3})[0] == 400                  # no vector
        assert s.ann({"vector": [0.0], "limit": 0})[0] == 400
        assert s.ann({"vector": [0.0], "limit": 1.5})[0] == 400
        assert s.ann({"vector": [True]})[0] == 400
        assert s.ann(raw=b"{not json")[0] == 400
        st, body = s.ann(raw=b'{"vector":[1e999],"limit":3}')   # inf query: l2sq = +inf is in range (distance.rs:72-75) ...
        assert st == 200 and [np.float32(x) for x in json.loads(body)["distances"]] == [np.finfo(np.float32).max] * 3   # ... saturated (lib.rs:397-409)
        assert json.loads(body)["similarity_scores"] == [0.0, 0.0, 0.0]
        assert s.get("/api/v1/nothing")[0] == 404
        # an absurd limit is clamped to the members the index holds: 200 with every member, the server stays up
        for huge in (4611686018427387904, 10 ** 9, 2 ** 63 - 1):
            st, body = s.ann({"vector": [0.0], "limit": huge})
            assert st == 200 and len(json.loads(body)["distances"]) == 3, (huge, st, body)
        assert json.loads(s.get("/api/v1/status")[1]) == "SERVING"
        # keep-alive: several requests on one connection, answers in order
        c = http.client.HTTPConnection("127.0.0.1", s.port, timeout=30)
        for q, want in (([0.0], 0), ([1.2], 1), ([2.9], 2)):
            c.request("POST", "/api/v1/indexes/ks/idx/ann", body=json.dumps({"vector": q}), headers={"content-type": "application/json"})
            assert json.loads(c.getresponse().read())["primary_keys"] == {"id": [want]}
        c.close()
    finally:
        s.close()


def test_empty_index_and_dot_product_scores(tmp_path):
    s, t, _ = _serve_kat(tmp_path, "B4_empty", "l2sq")
    try:
        st, body = s.ann({"vector": t["query"], "limit": 10})
        assert st == 200 and json.loads(body) == {"primary_keys": {"id": []}, "distances": [], "similarity_scores": []}
    finally:
        s.close()
    s, t, off = _serve_kat(tmp_path, "B6_ip_winner", "ip")
    try:
        body = json.loads(s.ann({"vector": t["query"], "limit": 1})[1])   # similarity.rs:94-100
        assert body["primary_keys"] == {"id": [4 - off]} and body["distances"] == [-1.0] and body["similarity_scores"] == [1.5]
        # inf * 0 = NaN for the rows orthogonal to the query: a NaN distance among the hits fails the request as
        # Distance::try_from does (distance.rs:76-83 -> 500); whether a NaN hit survives the walk is not defined (it is not in
        # usearch either), but the body is never invalid JSON
        st, body = s.ann(raw=b'{"vector":[1e999,0,0],"limit":4}')
        assert st in (200, 500)
        if st == 200:
            assert all(np.isfinite(x) for x in json.loads(body)["distances"])
        assert json.loads(s.ann({"vector": t["query"], "limit": 1})[1])["primary_keys"] == {"id": [4 - off]}   # still serving
    finally:
        s.close()


def test_filters_on_the_key_column(tmp_path):
    s, t, _ = _serve_kat(tmp_path, "B11_filter_30", "l2sq")
    try:
        def ann(restrictions):
            st, body = s.ann({"vector": t["query"], "limit": 100, "filter": {"restrictions": restrictions, "allow_filtering": True}})
            assert st == 200, body
            return sorted(json.loads(body)["primary_keys"]["id"])

        assert ann([{"type": "<", "lhs": "id", "rhs": 3}]) == [0, 1, 2]
        assert ann([{"type": "<=", "lhs": "id", "rhs": 3}]) == [0, 1, 2, 3]
        assert ann([{"type": ">", "lhs": "id", "rhs": 26}]) == [27, 28, 29]
        assert ann([{"type": ">=", "lhs": "id", "rhs": 27}, {"type": "<", "lhs": "id", "rhs": 29}]) == [27, 28]
        assert ann([{"type": "==", "lhs": "id", "rhs": 15}]) == [15]
        assert ann([{"type": "IN", "lhs": "id", "rhs": [1, 12, 23]}]) == [1, 12, 23]
        assert ann([{"type": "()==()", "lhs": ["id"], "rhs": [7]}]) == [7]
        assert ann([{"type": "()IN()", "lhs": ["id"], "rhs": [[7], [9]]}]) == [7, 9]
        assert ann([{"type": "()<()", "lhs": ["id"], "rhs": [2]}]) == [0, 1]
        assert ann([{"type": "()>=()", "lhs": ["id"], "rhs": [28]}]) == [28, 29]
        st, body = s.ann({"vector": t["query"], "limit": 1000})       # any limit is accepted (httproutes.rs:842-847)
        assert st == 200 and sorted(json.loads(body)["primary_keys"]["id"]) == list(range(30))
        d = json.loads(body)["distances"]
        assert d == sorted(d)
        st, _ = s.ann({"vector": t["query"], "limit": 5, "filter": {"restrictions": [{"type": "<", "lhs": "ck", "rhs": 3}]}})
        assert st == 400
        st, _ = s.ann({"vector": t["query"], "limit": 5, "filter": {"restrictions": [{"type": "~", "lhs": "id", "rhs": 3}]}})
        assert st == 400
        st, _ = s.ann({"vector": t["query"], "limit": 5, "filter": {"restrictions": [{"type": "<", "lhs": "id", "rhs": 1.5}]}})
        assert st == 400
    finally:
        s.close()


def test_search_http_client_against_the_native_server(tmp_path):
    """vs_bench gen -> vs_httpd (BOOTSTRAPPING -> SERVING) -> vs_bench search-http: recall as through the C ABI."""
    d = str(tmp_path / "ds")
    subprocess.check_call([VS_BENCH, "gen", "--data-dir", d, "--n", "30000", "--dim", "96", "--queries", "300", "--neighbors", "10"])
    s = Server(d, "cos", extra=("--expansion-search", "128"))
    try:
        out = subprocess.check_output([VS_BENCH, "search-http", "--data-dir", d, "--limit", "10", "--duration", "2", "--concurrency", "32",
                                       "--port", str(s.port), "--keyspace", "ks", "--index", "idx"], stderr=subprocess.STDOUT).decode()
        assert "error" not in out, out
        stats = {l.split(":")[0].strip(): l.split(":")[1].strip() for l in out.splitlines() if ":" in l}
        assert float(stats["QPS"]) > 1000, out
        assert float(stats["recall avg"]) > 90.0, out
        st = json.loads(s.get("/api/v1/indexes/ks/idx/status")[1])
        assert st == {"status": "SERVING", "count": 30000, "build_progress": 100.0}
    finally:
        s.close()


def test_server_survives_rude_clients(tmp_path):
    """Half-sent requests, abrupt closes while a query is in flight, garbage, oversized headers: the server keeps serving."""
    s, t, _ = _serve_kat(tmp_path, "B11_filter_30", "l2sq")
    try:
        body = json.dumps({"vector": t["query"], "limit": 5}).encode()
        head = f"POST /api/v1/indexes/ks/idx/ann HTTP/1.1\r\nhost: x\r\ncontent-type: application/json\r\ncontent-length: {len(body)}\r\n\r\n".encode()
        for i in range(60):
            c = socket.create_connection(("127.0.0.1", s.port), timeout=5)
            kind = i % 6
            if kind == 0:
                c.sendall(head[: len(head) // 2])                  # half a header, then gone
            elif kind == 1:
                c.sendall(head + body[: len(body) // 2])           # half a body
            elif kind == 2:
                c.sendall(head + body)                             # complete request, closed before the answer
            elif kind == 3:
                c.sendall(b"\x00\xff garbage \r\n\r\n")
            elif kind == 4:
                c.sendall(b"GET /" + b"a" * 70000 + b" HTTP/1.1\r\n")   # oversized request line
            else:
                c.sendall((head + body) * 3)                       # three pipelined requests, read only the first
                c.recv(4096)
            c.close()
        for _ in range(5):
            st, out = s.ann({"vector": t["query"], "limit": 5})
            assert st == 200 and len(json.loads(out)["primary_keys"]["id"]) == 5
        assert s.proc.poll() is None
    finally:
        s.close()
