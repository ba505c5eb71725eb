"""/ann HTTP surface: wire format, status codes and score arithmetic of the
reference's REST API (httproutes.rs:661-904, httpapi/src/lib.rs), exercised on CPU with the oracle as
the index behind the route and on GPU with the HIP engine."""
import math

import numpy as np
import pytest
from fastapi.testclient import TestClient

import oracle
from oracle import OracleIndex
from tests import kat_runner as K
from vector_store_amd import httpd

F32_MAX = float(np.finfo(np.float32).max)


def _client(ix, dim, metric, **kw):
    served = httpd.ServedIndex(ix, dim, metric, **kw)
    return TestClient(httpd.create_app({("ks", "idx"): served}, engine_version="test-engine")), served


def _oracle_index(name):
    t = K.KAT[name]
    ix = OracleIndex(t["dim"], oracle.METRICS[t["metric"]])
    ix.reserve(64)
    for row in t["base"]:
        ix.add(row["key"], np.asarray(row["vector"], dtype=np.float32))
    return ix, t


def check_surface(make_index):
    # B3: distances 0/1/9 -> similarity 1.0/0.5/0.1 (tests/integration/vs_index.rs:1795-1886)
    ix, t = make_index("B3_l2sq_1d_scores")
    c, served = _client(ix, t["dim"], 1)
    r = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"], "limit": 3})
    assert r.status_code == 200
    body = r.json()
    assert body["primary_keys"] == {"id": [0, 1, 2]}
    assert body["distances"] == [0.0, 1.0, 9.0]
    assert body["similarity_scores"] == pytest.approx([1.0, 0.5, 0.1], abs=1e-5)
    assert body["similarity_scores"][0] > body["similarity_scores"][1] > body["similarity_scores"][2]
    # default limit is 1 (httpapi/src/lib.rs:289-293)
    assert c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"]}).json()["primary_keys"] == {"id": [0]}
    # status / info / list
    assert c.get("/api/v1/indexes/ks/idx/status").json() == {"status": "SERVING", "count": 3, "build_progress": 100.0}
    assert c.get("/api/v1/info").json() == {"engine": "test-engine", "service": "vector-store", "version": "0.1.0"}
    assert c.get("/api/v1/status").json() == "SERVING"
    lst = c.get("/api/v1/indexes").json()
    assert lst[0]["keyspace"] == "ks" and lst[0]["options"]["type"] == "vector" and lst[0]["options"]["dimensions"] == 1
    assert lst[0]["options"]["similarity_function"] == "EUCLIDEAN"
    # errors
    assert c.post("/api/v1/indexes/ks/nope/ann", json={"vector": [0.0]}).status_code == 404
    assert c.get("/api/v1/indexes/ks/nope/status").status_code == 404
    assert c.post("/api/v1/indexes/ks/idx/ann", json={"vector": [0.0, 1.0]}).status_code == 400  # wrong dimension
    assert c.post("/api/v1/indexes/ks/idx/ann", json={"limit": 3}).status_code == 400
    assert c.post("/api/v1/indexes/ks/idx/ann", json={"vector": [0.0], "limit": 0}).status_code == 400
    assert c.post("/api/v1/indexes/ks/idx/ann", content=b"{not json").status_code == 400
    served.status = "BOOTSTRAPPING"
    r = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": [0.0]})
    assert r.status_code == 503 and r.json()["reason"] == "INDEX_BUILDING"
    # B4: empty index -> empty result (vs_index.rs:1919-1951)
    ix, t = make_index("B4_empty")
    c, _ = _client(ix, t["dim"], 1)
    r = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"], "limit": 10})
    assert r.status_code == 200 and r.json() == {"primary_keys": {"id": []}, "distances": [], "similarity_scores": []}
    # B6: dot product winner, similarity 1.5 (similarity.rs:94-100)
    ix, t = make_index("B6_ip_winner")
    c, _ = _client(ix, t["dim"], 2)
    body = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"], "limit": 1}).json()
    assert body["primary_keys"] == {"id": [4]} and body["distances"] == [-1.0] and body["similarity_scores"] == [1.5]
    # B11: filters on the key column over the 30-row fixture (ids 0..29, query [1,2,3], limit 100)
    ix, t = make_index("B11_filter_30")
    c, _ = _client(ix, t["dim"], 1)

    def ann(restrictions):
        r = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"], "limit": 100,
                                                       "filter": {"restrictions": restrictions, "allow_filtering": True}})
        assert r.status_code == 200, r.text
        return sorted(r.json()["primary_keys"]["id"])

    assert ann([{"type": "<", "lhs": "id", "rhs": 3}]) == [0, 1, 2]
    assert ann([{"type": "<=", "lhs": "id", "rhs": 3}]) == [0, 1, 2, 3]
    assert ann([{"type": ">", "lhs": "id", "rhs": 26}]) == [27, 28, 29]
    assert ann([{"type": ">=", "lhs": "id", "rhs": 27}, {"type": "<", "lhs": "id", "rhs": 29}]) == [27, 28]
    assert ann([{"type": "==", "lhs": "id", "rhs": 15}]) == [15]
    assert ann([{"type": "IN", "lhs": "id", "rhs": [1, 12, 23]}]) == [1, 12, 23]
    assert ann([{"type": "()==()", "lhs": ["id"], "rhs": [7]}]) == [7]
    assert ann([{"type": "()IN()", "lhs": ["id"], "rhs": [[7], [9]]}]) == [7, 9]
    assert ann([{"type": "()<()", "lhs": ["id"], "rhs": [2]}]) == [0, 1]
    r = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"], "limit": 1000})   # any limit (httproutes.rs:842-847)
    assert r.status_code == 200 and sorted(r.json()["primary_keys"]["id"]) == list(range(30))
    r = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": t["query"], "limit": 5,
                                                   "filter": {"restrictions": [{"type": "<", "lhs": "ck", "rhs": 3}]}})
    assert r.status_code == 400


def test_surface_with_oracle_behind_the_route():
    check_surface(_oracle_index)


def test_non_finite_values_saturate():
    class Inf:
        def size(self):
            return 2

        def search(self, q, k):
            return np.array([1, 2], dtype=np.uint64), np.array([math.inf, -math.inf], dtype=np.float32)

    c, _ = _client(Inf(), 2, 2)
    body = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": [0.0, 0.0], "limit": 2}).json()
    assert body["distances"] == [F32_MAX, -F32_MAX]  # httpapi/src/lib.rs:397-409
    assert all(math.isfinite(x) for x in body["similarity_scores"])


def test_epoch_bits_are_stripped_from_primary_keys():
    class One:
        def size(self):
            return 1

        def search(self, q, k):
            return np.array([(9 << 48) | 41], dtype=np.uint64), np.array([0.25], dtype=np.float32)

    c, _ = _client(One(), 1, 0)
    body = c.post("/api/v1/indexes/ks/idx/ann", json={"vector": [1.0]}).json()
    assert body["primary_keys"] == {"id": [41]} and body["similarity_scores"] == [0.875]


@pytest.mark.gpu
def test_surface_with_hip_engine_behind_the_route():
    import vector_store_amd as vs

    def make(name):
        t = K.KAT[name]
        ix = vs.HipUsearchIndex(t["dim"], vs.METRICS[t["metric"]])
        ix.reserve(64)
        for row in t["base"]:
            ix.add(row["key"], np.asarray(row["vector"], dtype=np.float32))
        return ix, t

    check_surface(make)
