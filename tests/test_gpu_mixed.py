"""The reference's MIXED call pattern: single adds and removes -- one vector per call,
usearch.rs:191-201, :1019-1049 -- alternating with families of searches under the actor's permits (usearch.rs:515-624, :897-948),
as crates/vector-store/benches/pipeline.rs drives them (cdc_update, search_while_updating ...).  Every answer is compared with the
CPU oracle fed the same sequence (KAT B1's shape -- add / remove / re-add between searches, usearch.rs:1298-1458 -- at 100,000
members)."""
import threading

import numpy as np
import pytest

import oracle
from oracle import OracleIndex
from tests.parity_util import assert_same_results, lattice

pytestmark = pytest.mark.gpu


def vs():
    import vector_store_amd as v
    return v


USEARCH_ORDER = 16  # vs_hnsw_options.reserved bit 4: the usearch-order walk for every search (exact also where distances TIE)


def _same_start(n, dim, extra, seed, ef=64):
    """An oracle-built graph of exactly representable vectors and the engine holding the same graph.  The oracle builds on thread
    slot 1, so the level generator of slot 0 -- the one a single-threaded replay uses, and the engine's own stream -- is still fresh.
    Integer coordinates make every distance exact -- and equal distances common (a query ties two of its ~1,000 candidates one time
    in four), so the engine answers with the usearch-order walk, whose order among equal distances is usearch's; the
    fused-list kernel and the pipelined walk are compared with the team kernels under the same modifications in
    tests/test_gpu_round5_fixes.py."""
    v = vs()
    data = lattice(n + extra, dim, seed, span=500)
    o = OracleIndex(dim, oracle.L2SQ, 16, 128, ef)
    o.reserve(n + extra)
    for i in range(n):
        o.add(i, data[i], thread=1)
    ix = v.HipUsearchIndex(dim, v.L2SQ, 16, 128, ef, _stress=USEARCH_ORDER)
    ix.reserve(n + extra)
    ix.import_graph(o.export_graph())
    return v, o, ix, data


def _script(n, extra, steps, seed):
    """A deterministic list of single modifications, in the order the reference issues them: an update is RemoveBeforeAddValue then
    AddVector (monitor_items.rs:301-313) -- two messages, and a search may be served between them --, an insert is one add, a delete
    one remove.  ("remove", key) / ("add", key, row)."""
    rng = np.random.default_rng(seed)
    alive = set(range(n))
    next_new, next_row = n + (1 << 20), n
    ops = []
    for _ in range(steps):
        r = rng.random()
        if r < 0.6 and next_row < n + extra:
            key = int(rng.integers(0, n))
            if key not in alive:
                continue
            ops.append(("remove", key))
            ops.append(("add", key, next_row))
            next_row += 1
        elif r < 0.8 and next_row < n + extra:
            ops.append(("add", next_new, next_row))
            alive.add(next_new)
            next_new += 1
            next_row += 1
        else:
            key = int(rng.integers(0, n))
            if key not in alive:
                continue
            alive.discard(key)
            ops.append(("remove", key))
    return ops


def _apply_to_oracle(o, op, data):
    if op[0] == "add":
        o.add(op[1], data[op[2]])
    else:
        assert o.remove(op[1])


def test_interleaved_single_adds_and_removes_with_searches_equal_the_oracle():
    """A thousand single modifications through the dispatch actor while 8 threads search (plain and filtered) through it: every answer
    equals the oracle's in one of the states its call overlapped (a search is a family of its own between two modifications, so that
    state exists): ids and distance bits, no exception (tests/parity_util.py, exact)."""
    from vector_store_amd.actor import IndexActor
    n, dim, k, extra = 100_000, 8, 10, 1200
    v, o, ix, data = _same_start(n, dim, extra, seed=31)
    ops = _script(n, extra, 700, seed=5)   # ~1,100 add / remove messages
    queries = lattice(512, dim, 77, span=500)
    a = IndexActor(dim, v.L2SQ, 16, 128, 64, workers=8)
    a.adopt_partition(0, ix.h, ix.size())
    done_ops = [0]      # modifications applied so far (written by the modifier only)
    stop = threading.Event()
    records = [[] for _ in range(8)]
    errors = []

    def searcher(t):
        rng = np.random.default_rng(1000 + t)
        try:
            while not stop.is_set():
                qi = int(rng.integers(0, len(queries)))
                filtered = t >= 6
                before = done_ops[0]
                if filtered:
                    keys, d = a.filtered_ann(0, queries[qi], k, lambda key: key % 3 == 0)
                else:
                    keys, d = a.ann(0, queries[qi], k)
                records[t].append((before, done_ops[0], qi, filtered, keys.copy(), d.copy()))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=searcher, args=(t,)) for t in range(8)]
    [x.start() for x in th]
    try:
        for op in ops:
            if op[0] == "add":
                assert a.add_vector_wait(0, op[1], data[op[2]])
            else:
                assert a.remove_vector_wait(0, op[1])
            done_ops[0] += 1
    finally:
        stop.set()
        [x.join() for x in th]
    assert not errors, errors
    total = sum(len(r) for r in records)
    assert total >= 2000, total
    # replay: after e modifications the oracle answers every search whose call overlapped that state
    # (the counter is advanced by the modifier's thread AFTER its message was processed: a search may already have seen the modification
    # that was in flight when it returned -- one more state than the counter said)
    by_state = {}
    for r in records:
        for i, rec in enumerate(r):
            r[i] = rec = (rec[0], min(rec[1] + 1, len(ops))) + rec[2:]
            for e in range(rec[0], rec[1] + 1):
                by_state.setdefault(e, []).append(rec)
    matched = set()

    def check(e):
        for rec in by_state.get(e, ()):
            if id(rec) in matched:
                continue
            _, _, qi, filtered, keys, d = rec
            wk, wd = o.filtered_search(queries[qi], k, lambda key: key % 3 == 0) if filtered else o.search(queries[qi], k)
            try:
                assert_same_results(keys, d, wk, wd, exact=True, what=f"state {e}")
                matched.add(id(rec))
            except AssertionError:
                if e == rec[1]:  # the last state the call overlapped
                    raise

    check(0)
    for e, op in enumerate(ops, start=1):
        _apply_to_oracle(o, op, data)
        check(e)
    assert len(matched) == total
    # same members, same graph shape at the end
    assert ix.size() == o.size()
    go, gg = o.export_graph(), ix.export_graph()
    assert (go["keys"] == gg["keys"]).all() and (go["levels"] == gg["levels"]).all()
    assert go["entry_slot"] == gg["entry_slot"] and go["max_level"] == gg["max_level"]
    differ = int((np.sort(go["adj0"], axis=1) != np.sort(gg["adj0"], axis=1)).any(axis=1).sum())
    assert differ <= 20, differ  # (rows that involve an exact tie may differ: test_sequential_adds_build_the_oracle_graph)
    a.stop()


def test_staged_removes_and_adds_are_applied_in_call_order():
    """vs_hnsw_remove is logged like vs_hnsw_add and applied with the next observation (round 5).  What every call returns, and what
    the index then holds, is what the calls would have done one by one (usearch.rs:1298-1458's sequence, many times over): a
    removed key can be added again at once, its slot is re-used in FIFO order, a key staged and removed before any flush ends up
    removed, duplicates and unknown keys are told apart against the STAGED state."""
    v = vs()
    dim, n = 8, 3000
    data = lattice(n + 800, dim, 3, span=500)
    ix = v.HipUsearchIndex(dim, v.L2SQ)
    o = OracleIndex(dim, oracle.L2SQ)
    ix.reserve(n + 800)
    o.reserve(n + 800)
    ix.add_batch(np.arange(n, dtype=np.uint64), data[:n])
    o.import_graph(ix.export_graph())
    # the oracle runs the calls one by one; the engine stages all of them and applies them at the first observation
    row = n
    rng = np.random.default_rng(9)
    staged = 0
    for step in range(600):
        key = int(rng.integers(0, n))
        r = rng.random()
        if r < 0.45:     # update: RemoveBeforeAddValue + AddVector
            want = o.remove(key)
            assert ix.remove(key) == want
            if want:
                o.add(key, data[row])
                ix.add(key, data[row])
                row += 1
        elif r < 0.65:   # delete (an unknown / already removed key answers False)
            assert ix.remove(key) == o.remove(key)
        elif r < 0.8:    # a new key that is removed again before anything observes it; removing it twice answers False
            nk = (7 << 48) | step
            ix.add(nk, data[row])
            o.add(nk, data[row])
            row += 1
            assert ix.remove(nk) and o.remove(nk)
            assert not ix.remove(nk)
        else:            # a duplicate is refused against the STAGED state, whatever it is made of
            nk = (9 << 48) | step
            ix.add(nk, data[row])
            o.add(nk, data[row])
            row += 1
            with pytest.raises(v.VsError, match="Duplicate"):
                ix.add(nk, data[row])
        staged += 1
        if step == 299:
            assert ix.size() == o.size()   # a barrier in the middle: the first half is applied, the second half staged on top of it
    assert ix.modify_stats()["flushes"] <= 4   # (nothing above observed the index but the two size() calls)
    assert ix.size() == o.size()
    go, gg = o.export_graph(), ix.export_graph()
    assert (go["keys"] == gg["keys"]).all()    # the same slots re-used in the same (FIFO) order, new slots in call order
    live = {int(x) for x in go["keys"] if int(x) != oracle.FREE_KEY}
    for qi in range(50):
        q = data[qi] + 1.0
        gk, gd = ix.search(q, 10)
        assert len(gk) == 10 and set(gk.tolist()) <= live
        ek, ed = ix.exact_search_batch(q[None, :], 10)[:2]
        wk, wd = o.exact_search(q, 10)
        assert np.array_equal(np.asarray(ed)[0][:10], wd)   # the stored vectors are the oracle's, slot for slot
