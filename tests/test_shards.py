"""In-process multi-GPU handle (include/vs_shards.h).  On a 1-GPU box the same device is listed several times:
the routing, the concurrent per-shard work and the merge are what is under test."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests import kat_runner as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_library_exports_its_header():
    header = open(os.path.join(ROOT, "include", "vs_shards.h")).read()
    declared = set(re.findall(r"^VS_API [^;(]*?\b(vs_shards_[a-z0-9_]+)\(", header, flags=re.M))
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "vector_store_amd", "libvs_shards.so")], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert declared == {s for s in exported if s.startswith("vs_shards_")} and len(declared) == 16


def _factory(devices):
    def make(metric, dim, **kw):
        import vector_store_amd as vs
        from vector_store_amd.shards import ShardedIndex
        return ShardedIndex(dim, vs.METRICS[metric], devices=devices, **kw)
    return make


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)])
def test_reference_kats_through_the_sharded_handle(devices):
    f = _factory(devices)
    for name in ("B2_l2sq_3d_http", "B3_l2sq_1d_scores", "B4_empty", "B5_cos_winners", "B6_ip_winner", "B7_l2_winner"):
        K.run_simple(f, name)
    K.run_b1(f)
    K.run_b11(f)
    K.run_b13(f)


@pytest.mark.gpu
def test_sharded_search_equals_exact_on_every_shard_and_balances():
    import vector_store_amd as vs
    from vector_store_amd.shards import ShardedIndex
    n, dim, k = 40000, 64, 10
    rng = np.random.default_rng(0)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    data = (rng.standard_normal((n + 200, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n + 200, dim))).astype(np.float32)
    base, q = data[:n], data[n:]
    keys = np.arange(n, dtype=np.uint64) | np.uint64(7 << 48)
    sh = ShardedIndex(dim, vs.COS, devices=(0, 0, 0, 0))
    assert sh.shards() == 4
    sh.reserve(n)
    assert sh.capacity() >= n
    sh.add_batch(keys[: n // 2], base[: n // 2])
    for i in range(n // 2, n // 2 + 50):
        sh.add(int(keys[i]), base[i])
    sh.add_batch(keys[n // 2 + 50:], base[n // 2 + 50:])
    assert sh.size() == n
    owners = np.bincount([sh.owner(int(x)) for x in keys[::97]], minlength=4)
    assert owners.min() > 0.8 * owners.mean()  # 4096-row stripes: balanced
    assert sh.owner(5) == sh.owner(5 | (9 << 48))  # the epoch does not move a row
    sh.set_expansion_search(128)
    gk, gd, gf = sh.search_batch(q, k)
    bn = base / np.linalg.norm(base, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    truth = np.argsort(1.0 - qn @ bn.T, axis=1, kind="stable")[:, :k]
    rec = np.mean([len(set(truth[i].tolist()) & set((gk[i] & np.uint64((1 << 48) - 1)).tolist())) / k for i in range(len(q))])
    one = vs.HipUsearchIndex(dim, vs.COS)
    one.reserve(n)
    one.add_batch(keys, base)
    one.set_expansion_search(128)
    ok_, _, _ = one.search_batch(q, k)
    rec_one = np.mean([len(set(truth[i].tolist()) & set((ok_[i] & np.uint64((1 << 48) - 1)).tolist())) / k for i in range(len(q))])
    assert (gf == k).all() and rec >= rec_one - 0.005 and rec >= 0.95, (rec, rec_one)
    assert all(np.all(gd[i, :-1] <= gd[i, 1:]) for i in range(len(q)))
    # single-query entry point == batch entry point
    for i in range(0, 200, 23):
        k1, d1 = sh.search(q[i], k)
        assert k1.tolist() == gk[i].tolist() and np.allclose(d1, gd[i])
    # remove / duplicate / stats
    assert sh.remove(int(keys[123])) and not sh.remove(int(keys[123])) and sh.size() == n - 1
    with pytest.raises(vs.VsError, match="Duplicate"):
        sh.add(int(keys[5]), base[5])
    st = sh.stats()
    assert st["added"] == n - 4 and st["visited_overflow"] == 0  # the first member of each shard is its entry point
    fk, fd = sh.filtered_search(q[0], 5, lambda key: (int(key) & 0xFFFF) % 1000 == 1)
    assert len(fk) == 5 and all((int(x) & 0xFFFF) % 1000 == 1 for x in fk) and np.all(fd[:-1] <= fd[1:])
