"""Thin `/ann` HTTP surface over the engine -- row f-1 of SURVEY.md section 8.

Keeps the wire format of the reference's REST API for the hot path so that its benchmark client
(`crates/benchmark search-http`, `vs.rs:17-39` status polling) can talk to the engine unchanged:

  POST /api/v1/indexes/{keyspace}/{index}/ann      httproutes.rs:661-904, httpapi/src/lib.rs:369-409
  GET  /api/v1/indexes/{keyspace}/{index}/status   httpapi/src/lib.rs:192-207
  GET  /api/v1/indexes                             httpapi/src/lib.rs:84-91
  GET  /api/v1/info, GET /api/v1/status            httpapi/src/lib.rs:232-240, 296-309

Status codes as the reference: 400 wrong vector size / malformed body, 404 unknown index, 500 engine
error, 503 {"reason": "INDEX_BUILDING", "message": ...} while an index is loading.  Distances and
similarity scores saturate +-inf to +-f32::MAX (lib.rs:386-409).

What is NOT here (stays in the Rust service): the table cache, CQL types, TLS, metrics.  The only
primary-key column is an integer `pk_column` (default "id") holding the row index = the low 48 bits
of the PrimaryId, exactly how the reference's fbin loader numbers rows (benchmark data/fbin.rs:86).
Filters are evaluated on that column only (all twelve restriction forms of lib.rs:323-366 that a
single integer column admits).

The index object is anything with the `UsearchIndex` surface (search / filtered_search / size); the
CPU tests inject a stand-in, the server entry point uses the HIP engine.
"""
from __future__ import annotations

import argparse
import asyncio
import math
import threading
from dataclasses import dataclass, field

import numpy as np
from fastapi import FastAPI, Request
from fastapi.responses import JSONResponse, PlainTextResponse

F32_MAX = float(np.finfo(np.float32).max)
SIMILARITY = {0: "COSINE", 1: "EUCLIDEAN", 2: "DOT_PRODUCT", 3: "HAMMING"}


def saturate(x: float) -> float:
    """serialize_saturated_f32 (httpapi/src/lib.rs:397-409)."""
    if x == math.inf:
        return F32_MAX
    if x == -math.inf:
        return -F32_MAX
    return float(x)


def similarity_score(d: float, metric: int, dim: int) -> float:
    """SimilarityScore::from (similarity.rs:28-35); f32 arithmetic."""
    d = np.float32(d)
    if metric in (0, 2):
        return float((np.float32(2.0) - d) / np.float32(2.0))
    if metric == 1:
        return float(np.float32(1.0) / (np.float32(1.0) + d))
    return float(np.float32(1.0) - d / np.float32(dim))


@dataclass
class ServedIndex:
    index: object
    dim: int
    metric: int
    pk_column: str = "id"
    status: str = "SERVING"  # INITIALIZING | BOOTSTRAPPING | SERVING
    build_progress: float = 100.0
    options: dict = field(default_factory=dict)


class BadRequest(Exception):
    pass


def _cmp_int(v):
    if isinstance(v, bool) or not isinstance(v, int):
        raise BadRequest("filter values on the primary key column must be integers")
    return v


def compile_filter(flt: dict, pk_column: str):
    """PostIndexAnnFilter -> predicate(key).  Row index = key & (2^48 - 1)."""
    tests = []
    for r in flt.get("restrictions", []):
        typ, lhs, rhs = r.get("type"), r.get("lhs"), r.get("rhs")
        tuple_form = typ.startswith("()")
        if tuple_form:
            if lhs != [pk_column]:
                raise BadRequest(f"unknown column(s) in filter: {lhs}")
            op = typ.replace("()", "")
            rhs = [x[0] for x in rhs] if op == "IN" else rhs[0]
        else:
            if lhs != pk_column:
                raise BadRequest(f"unknown column in filter: {lhs}")
            op = typ
        if op == "==":
            v = _cmp_int(rhs); tests.append(lambda x, v=v: x == v)
        elif op == "IN":
            s = {_cmp_int(v) for v in rhs}; tests.append(lambda x, s=s: x in s)
        elif op == "<":
            v = _cmp_int(rhs); tests.append(lambda x, v=v: x < v)
        elif op == "<=":
            v = _cmp_int(rhs); tests.append(lambda x, v=v: x <= v)
        elif op == ">":
            v = _cmp_int(rhs); tests.append(lambda x, v=v: x > v)
        elif op == ">=":
            v = _cmp_int(rhs); tests.append(lambda x, v=v: x >= v)
        else:
            raise BadRequest(f"unknown restriction type: {typ}")
    mask = (1 << 48) - 1
    return lambda key: all(t(int(key) & mask) for t in tests)


def create_app(indexes: dict, engine_version: str = "hip-hnsw", node_status: str = "SERVING") -> FastAPI:
    """indexes: {(keyspace, index_name): ServedIndex}"""
    app = FastAPI(title="vector-store ANN surface over the MI355X HNSW engine")

    def find(ks, name):
        return indexes.get((ks, name))

    @app.get("/api/v1/indexes")
    async def get_indexes():
        out = []
        for (ks, name), s in indexes.items():
            opts = {"type": "vector", "dimensions": s.dim, "maximum_node_connections": s.options.get("connectivity", 16),
                    "construction_beam_width": s.options.get("expansion_add", 128),
                    "search_beam_width": s.options.get("expansion_search", 64),
                    "similarity_function": SIMILARITY[s.metric], "quantization": "F32"}
            out.append({"keyspace": ks, "index": name, "options": opts})
        return out

    @app.get("/api/v1/indexes/{keyspace}/{index}/status")
    async def get_status(keyspace: str, index: str):
        s = find(keyspace, index)
        if s is None:
            return PlainTextResponse(f"missing index: {keyspace}.{index}", status_code=404)
        return {"status": s.status, "count": int(s.index.size()), "build_progress": s.build_progress}

    @app.get("/api/v1/info")
    async def get_info():
        return {"engine": engine_version, "service": "vector-store", "version": "0.1.0"}

    @app.get("/api/v1/status")
    async def get_node_status():
        return JSONResponse(node_status)

    @app.post("/api/v1/indexes/{keyspace}/{index}/ann")
    async def post_ann(keyspace: str, index: str, request: Request):
        s = find(keyspace, index)
        if s is None:
            return PlainTextResponse(f"missing index: {keyspace}.{index}", status_code=404)
        if node_status != "SERVING":
            return JSONResponse({"reason": "NODE_BOOTSTRAPPING"}, status_code=503)
        if s.status != "SERVING":
            return JSONResponse({"reason": "INDEX_BUILDING", "message": f"index {keyspace}.{index} is {s.status}"},
                                status_code=503)
        try:
            body = await request.json()
            vector = body["vector"]
            if not isinstance(vector, list) or not all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in vector):
                raise BadRequest("vector must be an array of numbers")
            limit = body.get("limit", 1)  # Limit::default() == 1 (lib.rs:289-293)
            if isinstance(limit, bool) or not isinstance(limit, int) or limit < 1:
                raise BadRequest("limit must be a positive integer")
            if len(vector) != s.dim:  # validator.rs:12-26 -> 400
                raise BadRequest(f"wrong embedding dimension: got {len(vector)}, index has {s.dim}")
            flt = body.get("filter")
            pred = compile_filter(flt, s.pk_column) if flt else None
        except BadRequest as e:
            return PlainTextResponse(str(e), status_code=400)
        except Exception as e:  # malformed JSON / missing fields
            return PlainTextResponse(f"malformed request: {e}", status_code=400)
        q = np.asarray(vector, dtype=np.float32)
        try:
            loop = asyncio.get_running_loop()
            if pred is None and limit <= 512 and hasattr(s.index, "search_async"):  # beyond the LDS beam: blocking exhaustive path
                fut = loop.create_future()

                def on_done(keys, dist, status):
                    loop.call_soon_threadsafe(fut.set_result, (keys.copy(), dist.copy(), status))

                hold = s.index.search_async(q, limit, on_done)
                keys, dist, status = await fut
                del hold
                if status != 0:
                    raise RuntimeError(f"engine status {status}")
            elif pred is None:
                keys, dist = await loop.run_in_executor(None, s.index.search, q, limit)
            else:
                keys, dist = await loop.run_in_executor(None, s.index.filtered_search, q, limit, pred)
        except Exception as e:
            code = getattr(e, "code", None)
            if code == -2:  # VS_ERR_DIMENSION
                return PlainTextResponse(str(e), status_code=400)
            return PlainTextResponse(f"index.ann request error: {e}", status_code=500)
        mask = (1 << 48) - 1
        return {
            "primary_keys": {s.pk_column: [int(k) & mask for k in keys]},
            "distances": [saturate(float(d)) for d in dist],
            "similarity_scores": [saturate(similarity_score(float(d), s.metric, s.dim)) for d in dist],
        }

    return app


def serve_dataset(data_dir: str, keyspace: str, index_name: str, metric: str, host: str, port: int,
                  expansion_search: int):
    """Loads data.fbin (benchmark dataset layout), builds the index on the GPU in the background (status
    BOOTSTRAPPING -> SERVING, which is what `build-index` polls, benchmark vs.rs:17-39) and serves."""
    import uvicorn

    import vector_store_amd as vs
    from vector_store_amd import datasets

    files = datasets.dataset_files(data_dir)
    base = datasets.read_fbin(files["data_fbin"])
    n, dim = base.shape
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], 16, 128, expansion_search)
    served = ServedIndex(ix, dim, vs.METRICS[metric], status="BOOTSTRAPPING", build_progress=0.0,
                         options={"expansion_search": expansion_search})

    def build():
        ix.reserve(n)
        step = 1 << 18
        for i in range(0, n, step):
            ix.add_batch(np.arange(i, min(i + step, n), dtype=np.uint64), base[i:i + step])
            served.build_progress = 100.0 * min(i + step, n) / n
        served.status = "SERVING"

    threading.Thread(target=build, daemon=True).start()
    app = create_app({(keyspace, index_name): served}, engine_version=f"hip-hnsw-{vs.version()}")
    uvicorn.run(app, host=host, port=port, log_level="warning")


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--data-dir", required=True)
    ap.add_argument("--keyspace", default="vsb_keyspace")
    ap.add_argument("--index", default="vsb_index")
    ap.add_argument("--metric", default="cos", choices=["cos", "l2sq", "ip"])
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=6080)
    ap.add_argument("--expansion-search", type=int, default=64)
    a = ap.parse_args()
    serve_dataset(a.data_dir, a.keyspace, a.index, a.metric, a.host, a.port, a.expansion_search)


if __name__ == "__main__":
    main()
