"""The dispatch actor exercised the way the reference's own unit tests
exercise it (crates/vector-store/src/vs_index/usearch.rs:1298-1607)."""
import os
import re
import subprocess
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_actor_library_exports_its_header():
    header = open(os.path.join(ROOT, "include", "vs_actor.h")).read()
    declared = set(re.findall(r"^VS_API [^;(]*?\b(vs_actor_[a-z0-9_]+)\(", header, flags=re.M))
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "vector_store_amd", "libvs_actor.so")], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert declared == {s for s in exported if s.startswith("vs_actor_")} and len(declared) == 13


def wait_for_count(actor, expected, timeout=120.0):  # generous: the first GPU call of a fresh box loads 30 MB of code objects
    t0 = time.time()
    while actor.count() != expected:
        assert time.time() - t0 < timeout, (actor.count(), expected)
        time.sleep(0.001)


@pytest.mark.gpu
def test_add_or_replace_size_ann():
    """usearch.rs:1298-1458"""
    from vector_store_amd import L2SQ
    from vector_store_amd.actor import IndexActor
    a = IndexActor(3, L2SQ)
    p = 7
    a.add_vector(p, 1, [1., 1., 1.])
    a.add_vector(p, 2, [2., -2., 2.])
    a.add_vector(p, 3, [3., 3., 3.])
    wait_for_count(a, 3)
    keys, d = a.ann(p, [2.2, -2.2, 2.2], 1)
    assert keys.tolist() == [2] and len(d) == 1
    a.remove_vector(p, 3)
    wait_for_count(a, 2)
    a.add_vector(p, 3, [2.1, -2.1, 2.1])
    wait_for_count(a, 3)
    t0 = time.time()
    while a.ann(p, [2.2, -2.2, 2.2], 1)[0].tolist() != [3]:
        assert time.time() - t0 < 120
    a.remove_vector(p, 3)
    wait_for_count(a, 2)
    keys, d = a.ann(p, [2.2, -2.2, 2.2], 1)
    assert keys.tolist() == [2]
    # unknown partition => empty result (usearch.rs:787-802); wrong dimension => error (validator.rs:12-26)
    assert len(a.ann(99, [0., 0., 0.], 5)[0]) == 0
    with pytest.raises(Exception) as e:
        a.ann(p, [1., 2.], 1)
    assert e.value.code == -2
    # first add reserved +1,000,000 (usearch.rs:442, 655-665)
    assert a.partition_capacity(p) == 1_000_000
    a.stop()


@pytest.mark.gpu
def test_allocate_parameter_works():
    """usearch.rs:1460-1524: adds are dropped while the memory guard says Cannot."""
    from vector_store_amd import L2SQ
    from vector_store_amd.actor import IndexActor
    a = IndexActor(3, L2SQ)
    a.set_allocate(False)
    a.add_vector(1, 1, [1., 1., 1.])
    assert a.count() == 0  # Count is a round trip through the same actor: the add has been seen and dropped
    a.set_allocate(True)
    a.add_vector(1, 1, [1., 1., 1.])
    wait_for_count(a, 1)
    assert a.counters()["adds_dropped"] == 1
    a.stop()


@pytest.mark.gpu
def test_concurrent_add_and_search():
    """usearch.rs:1526-1607: 2 x cores tasks x 50 adds and as many searches, concurrently, no error."""
    from vector_store_amd import L2SQ
    from vector_store_amd.actor import IndexActor
    dim, tasks, per = 1024, 16, 50
    a = IndexActor(dim, L2SQ, workers=8)
    z = np.zeros(dim, dtype=np.float32)
    errs = []

    def adder(t):
        try:
            for i in range(per):
                a.add_vector(0, t * per + i, z)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    def searcher():
        try:
            for _ in range(per):
                keys, d = a.ann(0, z, 5)
                assert len(keys) <= 5 and all(float(x) == 0.0 for x in d)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=adder, args=(t,)) for t in range(tasks)] + [threading.Thread(target=searcher) for _ in range(tasks)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    wait_for_count(a, tasks * per)
    c = a.counters()
    assert c["adds"] == tasks * per and c["searches"] == tasks * per and c["errors"] == 0
    assert c["mode_switches"] >= 1  # adds and searches alternated in families, never mixed
    assert c["max_in_flight"] <= 8 * 3 + 8 + 1  # channel (3 x workers) + workers
    a.stop()


@pytest.mark.gpu
def test_local_index_growth_and_partitions():
    """Local (per-partition-key) indexes: one handle per partition, +1,000 slots per reserve
    (usearch.rs:443, 640-644, 766-778); RemovePartition drops the handle (usearch.rs:888-893)."""
    from vector_store_amd import COS
    from vector_store_amd.actor import IndexActor
    dim = 16
    a = IndexActor(dim, COS, workers=4, local=True)
    rng = np.random.default_rng(0)
    data = rng.standard_normal((1500, dim)).astype(np.float32)
    for i in range(1500):
        a.add_vector(i % 3, i, data[i])
    wait_for_count(a, 1500)
    assert a.partitions() == 3
    assert all(a.partition_capacity(p) == 1000 for p in range(3))  # 500 each: well above the free threshold
    for i in range(1500, 2600):
        a.add_vector(0, i, data[i % 1500] + 1e-3 * (i // 1500))
    wait_for_count(a, 2600)
    assert a.partition_capacity(0) == 2000 and a.counters()["reserves"] == 4
    keys, d = a.ann(1, data[1], 1)
    assert keys.tolist() == [1]
    keys, _ = a.filtered_ann(2, data[2], 10, lambda key: key % 2 == 0)
    assert len(keys) == 10 and all(int(k) % 2 == 0 and int(k) % 3 == 2 for k in keys)
    a.remove_partition(1)
    t0 = time.time()
    while a.partitions() != 2:
        assert time.time() - t0 < 120
    assert len(a.ann(1, data[1], 1)[0]) == 0 and a.count() == 2600 - 500
    a.stop()
