"""The benchmark-harness row of the scope table: dataset file formats,
histogram / report arithmetic and recall of the native driver `vs_bench`, and the C-ABI export check."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VS_BENCH = os.path.join(ROOT, "vector_store_amd", "vs_bench")
LIB = os.path.join(ROOT, "vector_store_amd", "libvs_hnsw.so")


def test_c_abi_exports_every_declared_symbol():
    """The library loads and exports exactly what include/vs_hnsw.h (the boundary) and include/vs_hnsw_debug.h (counters, test hooks)
    declare (no compute calls here).  The boundary header carries no counters and no test hooks (round-5 review, item 9)."""
    header = open(os.path.join(ROOT, "include", "vs_hnsw.h")).read()
    debug = open(os.path.join(ROOT, "include", "vs_hnsw_debug.h")).read()
    pattern = r"^VS_API [^;(]*?\b(vs_[a-z0-9_]+)\("
    boundary = set(re.findall(pattern, header, flags=re.M))
    declared = boundary | set(re.findall(pattern, debug, flags=re.M))
    assert len(boundary) >= 25
    assert not [s for s in boundary if s.endswith("_stats") or s.endswith("_stats2") or s.endswith("_info")], "counters belong in vs_hnsw_debug.h"
    assert "bit 0" not in header and "VS_DEBUG_TINY_VISITED" in debug  # the hooks are named in the debug header only
    for trait_call in ("vs_hnsw_create", "vs_hnsw_reserve", "vs_hnsw_capacity", "vs_hnsw_add", "vs_hnsw_remove", "vs_hnsw_search", "vs_hnsw_filtered_search",
                       "vs_hnsw_filter_forget", "vs_hnsw_filter_forget_keys"):
        assert trait_call in boundary, trait_call
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert declared <= exported, declared - exported
    assert {s for s in exported if s.startswith("vs_")} == declared  # nothing undeclared leaks out
    import vector_store_amd as vs
    assert vs.version() and isinstance(vs.version(), str)
    for sym in declared:
        getattr(vs.lib(), sym)
    # every other header of include/ against its library (actor, in-process shards, one-process-per-GPU ranks)
    for hdr, so in (("vs_actor.h", "libvs_actor.so"), ("vs_shards.h", "libvs_shards.so"), ("vs_ranks.h", "libvs_ranks.so")):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        want = set(re.findall(r"^VS_API [^;(]*?\b(vs_[a-z0-9_]+)\(", text, flags=re.M))
        assert want, hdr
        out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "vector_store_amd", so)], text=True)
        have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
        assert want <= have, (hdr, want - have)
        assert {x for x in have if x.startswith("vs_")} == want, (hdr, have - want)
    from vector_store_amd import ranks
    assert ranks.lib().vs_ranks_create  # loads (librccl resolves) without a GPU


def test_host_helpers_match_reference_kats():
    """f32_to_b1x8 / Distance::try_from / SimilarityScore through the product's C ABI (host-only calls)."""
    import vector_store_amd as vs
    from tests import kat_runner as K
    for c in K.KAT["B14_b1_packing"]["cases"]:
        assert vs.f32_to_b1x8(np.asarray(c["input"], dtype=np.float32)).tolist() == c["expect"]
    t = K.KAT["B15_ranges"]
    for m in ("l2sq", "cos", "ip", "hamming"):
        for v in t[m]["ok"]:
            assert vs.distance_valid(K.special(v), vs.METRICS[m], t[m].get("dim", 0)), (m, v)
        for v in t[m]["err"]:
            assert not vs.distance_valid(K.special(v), vs.METRICS[m], t[m].get("dim", 0)), (m, v)
    for c in K.KAT["B16_scores"]["cases"]:
        got = vs.similarity_score(c["d"], vs.METRICS[c["metric"]], c.get("dim", 0))
        assert got == pytest.approx(np.float32(c["score"]), abs=1e-6), c


def test_vs_bench_selftest():
    assert subprocess.run([VS_BENCH, "selftest"], capture_output=True, text=True).stdout.strip() == "selftest ok"


def test_fbin_ibin_python_roundtrip(tmp_path):
    from vector_store_amd import datasets
    a = np.random.default_rng(0).standard_normal((7, 5)).astype(np.float32)
    t = np.arange(21, dtype=np.int32).reshape(7, 3)
    datasets.write_fbin(str(tmp_path / "data.fbin"), a)
    datasets.write_ibin(str(tmp_path / "query.ibin"), t)
    raw = open(tmp_path / "data.fbin", "rb").read()
    assert raw[:8] == bytes([7, 0, 0, 0, 5, 0, 0, 0]) and len(raw) == 8 + 7 * 5 * 4
    assert np.array_equal(datasets.read_fbin(str(tmp_path / "data.fbin")), a)
    assert np.array_equal(datasets.read_ibin(str(tmp_path / "query.ibin")), t)
    (tmp_path / "dataset.toml").write_text('[fbin]\ndata_fbin = "d.fbin"  # comment\nquery_fbin = "q.fbin"\n')
    f = datasets.dataset_files(str(tmp_path))
    assert f["data_fbin"].endswith("d.fbin") and f["query_ibin"].endswith("query.ibin")


@pytest.mark.gpu
def test_vs_bench_gen_build_search(tmp_path):
    from vector_store_amd import datasets
    d = str(tmp_path)
    subprocess.check_call([VS_BENCH, "gen", "--data-dir", d, "--n", "20000", "--dim", "96", "--queries", "200",
                           "--neighbors", "20"])
    f = datasets.dataset_files(d)
    base, q, truth = datasets.read_fbin(f["data_fbin"]), datasets.read_fbin(f["query_fbin"]), datasets.read_ibin(f["query_ibin"])
    assert base.shape == (20000, 96) and q.shape == (200, 96) and truth.shape == (200, 20)
    bn = base / np.linalg.norm(base, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    want = np.argsort(1.0 - qn[:20] @ bn.T, axis=1, kind="stable")[:, :20]
    assert all(set(want[i].tolist()) == set(truth[i].tolist()) for i in range(20))
    out = subprocess.check_output([VS_BENCH, "build-index", "--data-dir", d], text=True)
    assert "vectors/s" in out
    out = subprocess.check_output([VS_BENCH, "search", "--data-dir", d, "--limit", "10", "--duration", "2",
                                   "--concurrency", "32", "--expansion-search", "128"], text=True)
    kv = dict(ln.split(": ", 1) for ln in out.splitlines() if ": " in ln)
    assert int(kv["queries"]) > 100 and float(kv["QPS"]) > 50
    assert float(kv["recall avg"]) >= 90.0
    for p in ("P01", "P10", "P25", "P50", "P75", "P90", "P99"):
        assert f"latency {p}" in kv


def test_vs_httpd_selftest():
    """Request parser (bit-exact float fast path, malformed bodies), filter compiler and number rendering of the native
    /ann server; no GPU needed."""
    httpd = os.path.join(os.path.dirname(VS_BENCH), "vs_httpd")
    out = subprocess.check_output([httpd, "selftest"]).decode()
    assert "selftest ok" in out

