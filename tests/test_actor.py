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
    assert declared == {s for s in exported if s.startswith("vs_actor_")} and len(declared) == 18


def _oracle_actor(dim, metric, n, seed=0, workers=4, ef=64):
    """The SAME actor library over the CPU oracle (vs_actor_create_with + oracle.trait_vtable()): what bench.py's cpu_baseline.mixed
    times, and what makes the actor's own logic testable without a GPU."""
    import oracle
    from vector_store_amd.actor import IndexActor
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    o = oracle.OracleIndex(dim, metric, 16, 128, ef)
    o.reserve(n + 4096)
    o.add_batch(np.arange(n, dtype=np.uint64), base, threads=4)
    a = IndexActor(dim, metric, 16, 128, ef, workers=workers, index_vtable=oracle.trait_vtable())
    a.adopt_partition(0, o.h, n)
    return a, o, base, rng


def test_actor_over_the_cpu_oracle_adopts_an_index_and_keeps_the_markers():
    """vs_actor_create_with / vs_actor_adopt_partition / the in-progress markers (benches/pipeline.rs:685-712) on CPU."""
    import oracle
    a, o, base, rng = _oracle_actor(32, oracle.L2SQ, 2000)
    try:
        assert a.count() == 2000 and a.partition_capacity(0) == 2000 + 4096
        q = base[17] + 0.001
        keys, d = a.ann(0, q, 5)
        ok, od = o.search(q, 5)
        assert keys.tolist() == ok.tolist() and np.array_equal(d, od)
        # update = RemoveBeforeAddValue + AddVector (monitor_items.rs:301-313); the marker drops once the index call has run
        a.remove_vector(0, 17)
        assert a.add_vector_wait(0, 17, base[18] * 2) is True
        assert a.count() == 2000 and o.size() == 2000
        assert a.add_vector_wait(0, 17, base[18]) is False        # duplicate key: the add is swallowed (usearch.rs:1028-1030)
        assert a.remove_vector_wait(0, 17) is True and a.remove_vector_wait(0, 17) is False
        assert a.count() == 1999
        assert a.remove_vector_wait(5, 1) is False                 # unknown partition
        keys, _ = a.filtered_ann(0, q, 5, lambda k: k % 2 == 0)
        assert len(keys) == 5 and all(k % 2 == 0 for k in keys.tolist())
        c = a.counters()
        assert c["adds"] == 1 and c["removes"] == 2 and c["errors"] == 1
    finally:
        a.stop()


def test_mixed_driver_runs_the_references_pipeline_scenarios_over_the_oracle():
    """libvs_callers' vs_mixed_run (cdc_insert / cdc_update / cdc_delete / search_while_updating, benches/pipeline.rs:508-1292)
    through the actor over the CPU oracle: item accounting, predicate-respecting answers, no errors."""
    import oracle
    from vector_store_amd import callers
    n = 3000
    a, o, base, rng = _oracle_actor(24, oracle.COS, n)
    try:
        q = rng.standard_normal((64, 24)).astype(np.float32)
        fresh = rng.standard_normal((256, 24)).astype(np.float32)
        r = callers.mixed_run(a, q, fresh, modify=callers.INSERT, first_new_key=1 << 40, max_items=300, seconds=30)
        assert r["items"] == 300 and r["adds_applied"] == 300 and r["errors"] == 0 and o.size() == n + 300
        r = callers.mixed_run(a, q, fresh, modify=callers.UPDATE, existing_keys=n, max_items=200, producers=4, seconds=30)
        assert r["items"] == 200 and r["adds_applied"] == 200 and r["errors"] == 0 and o.size() == n + 300
        r = callers.mixed_run(a, q, fresh, modify=callers.DELETE, delete_from=0, max_items=100, seconds=30)
        assert r["removes_applied"] == 100 and o.size() == n + 200
        r = callers.mixed_run(a, q, fresh, modify=callers.UPDATE, existing_keys=n, plain_callers=3, filtered_callers=2, modulus=3, seconds=0.5)
        assert r["errors"] == 0 and r["items"] > 0 and r["plain"]["count"] > 0 and r["filtered"]["count"] > 0
        assert r["filtered"]["predicate_calls_per_query"] > 0
        c = a.counters()
        assert c["mode_switches"] > 2 * r["items"] - 4  # every modification is a family of its own between families of searches (usearch.rs:590-612)
    finally:
        a.stop()


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


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_thousand_partition_handles():
    """A local (per-partition-key) index is one usearch handle per partition, made on the partition's first add and grown by
    1,000 slots at a time (reference usearch.rs:443, 704-705, 766-778) -- a table with thousands of partitions means
    thousands of live handles.  2,000 partitions x 200 rows through the dispatch actor: making and reserving the handles is
    quick, an idle handle costs little HBM, no handle owns a HIP stream, and every partition answers its own rows."""
    import torch
    import vector_store_amd as vs
    from vector_store_amd.actor import IndexActor
    parts, per, dim = 2000, 200, 64
    rng = np.random.default_rng(5)
    data = rng.standard_normal((per, dim)).astype(np.float32)
    vs.HipUsearchIndex(dim, vs.COS).reserve(1000)  # (the device's one-time costs are not the handles')
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    streams0 = vs.streams_created()
    a = IndexActor(dim, vs.COS, workers=8, local=True)
    t0 = time.time()
    for p in range(parts):                      # the first add of a partition makes its handle and reserves 1,000 slots
        a.add_vector(p, (p << 20), data[0] + 1e-3 * p)
    wait_for_count(a, parts)
    t_handles = time.time() - t0
    assert a.partitions() == parts and a.counters()["reserves"] == parts
    assert all(a.partition_capacity(p) == 1000 for p in (0, 1, parts // 2, parts - 1))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    per_handle = (free0 - free1) / parts
    assert t_handles <= 5.0, t_handles                       # create + reserve of 2,000 handles
    assert per_handle <= 512 * 1024, per_handle              # HBM of a handle with 1,000 reserved slots of 64-d f32
    for p in range(parts):
        for i in range(1, per):
            a.add_vector(p, (p << 20) + i, data[i] + 1e-3 * p)
    wait_for_count(a, parts * per, timeout=400.0)
    assert a.counters()["errors"] == 0
    hits = 0
    for p in range(parts):                      # one search per partition: its own row, not a neighbour partition's
        keys, d = a.ann(p, data[p % per] + 1e-3 * p, 3)
        hits += int(len(keys) == 3 and int(keys[0]) == (p << 20) + p % per)
    assert hits == parts, hits
    # the engine's streams are a fixed set per device (leased contexts and dispatcher slots share them): none per handle
    assert vs.streams_created() - streams0 <= 20, (streams0, vs.streams_created())
    a.stop()
