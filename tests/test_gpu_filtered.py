"""filtered_search with the reference's semantics: usearch tests the predicate when
a node would enter `top` and keeps expanding rejected nodes (reference vs_index/usearch.rs:224-248, wrapper
filtered_ann :1107-1154; 19 integration tests tests/integration/vs_index.rs:718-1640).  The engine evaluates the host
predicate into an allow-bitmap over slots and tests it in the walk at admission, so on the same graph the result set
and its order equal the CPU restatement's -- compared here at 100,000 members for 50 % / 10 % / 1 % / 0.1 % selectivity."""
import numpy as np
import pytest

import oracle
from oracle import OracleIndex
from tests.parity_util import assert_same_results, lattice

pytestmark = pytest.mark.gpu


def _dataset(n, dim, seed):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((16, dim)).astype(np.float32) / 4
    return (rng.standard_normal((n, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((n, dim))).astype(np.float32)


@pytest.mark.parametrize("metric,kind,dim", [("cos", "f32", 96), ("l2sq", "f32", 32), ("ip", "f16", 128), ("l2sq", "i8", 64),
                                             ("hamming", "b1", 128)])
def test_filtered_search_equals_the_cpu_algorithm_at_100k(metric, kind, dim):
    import vector_store_amd as vs
    n, nq = 100000, 8
    data = _dataset(n + nq, dim, 41)
    base, q = data[:n], data[n:]
    if metric == "ip":
        base = base / np.linalg.norm(base, axis=1, keepdims=True)
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[kind])
    ix.reserve(n)
    keys = np.arange(n, dtype=np.uint64) * 7 + 3
    ix.add_batch(keys, base)
    o = OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[kind])
    o.import_graph(ix.export_graph())
    exact = kind in ("i8", "b1")
    dist_of = lambda i: (lambda key: oracle.distance_as(oracle.METRICS[metric], oracle.SCALARS[kind], q[i], base[(key - 3) // 7]))
    ties = 0
    for modulo, k, ef in ((2, 10, 64), (10, 10, 64), (100, 10, 64), (100, 50, 128), (1000, 20, 64), (10, 600, 64)):
        pred = lambda key: ((key - 3) // 7) % modulo == 1
        ix.set_expansion_search(ef)
        o.set_expansion_search(ef)
        for i in range(nq):
            fk, fd = ix.filtered_search(q[i], k, pred)
            ek, ed = o.filtered_search(q[i], k, pred)
            assert all(pred(int(x)) for x in fk)
            assert len(fk) == len(ek) == k, (modulo, k, len(fk), len(ek))
            ties += assert_same_results(fk, fd, ek, ed, dist_of(i), exact=exact, what=(metric, kind, modulo, k, ef, i))
    assert ties <= 8, ties
    # 100,000 members > 65,536: the predicate was asked lazily -- a few rounds per query, far fewer calls than one per member
    fs = ix.filter_stats()
    queries = 6 * nq
    assert fs["lazy_rounds"] >= queries and fs["lazy_rounds"] <= 8 * queries, fs
    assert fs["lazy_predicate_calls"] < 0.5 * n * queries, fs
    # a predicate nothing passes, one a single member passes, and one everything passes (== plain search)
    fk, fd = ix.filtered_search(q[0], 10, lambda key: False)
    assert len(fk) == 0
    fk, fd = ix.filtered_search(q[0], 10, lambda key: key == int(keys[777]))
    assert fk.tolist() == [int(keys[777])]
    fk, fd = ix.filtered_search(q[0], 10, lambda key: True)
    pk, pd = ix.search(q[0], 10)
    assert fk.tolist() == pk.tolist() or not exact  # float indexes answer plain searches from the fused list


def test_filtered_search_with_removed_members_and_exact_data():
    """Lattice data (every distance exact in f32): bit-identical ids with removed members in the graph."""
    import vector_store_amd as vs
    n, dim = 20000, 24
    data = lattice(n + 16, dim, 77, span=200)
    base, q = data[:n], data[n:]
    ix = vs.HipUsearchIndex(dim, vs.L2SQ)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    for key in range(0, n, 5):
        assert ix.remove(key)
    o = OracleIndex(dim, oracle.L2SQ)
    o.import_graph(ix.export_graph())
    for modulo, k in ((3, 10), (50, 25)):
        pred = lambda key: key % modulo == 1
        for i in range(16):
            fk, fd = ix.filtered_search(q[i], k, pred)
            ek, ed = o.filtered_search(q[i], k, pred)
            assert_same_results(fk, fd, ek, ed, exact=True, what=(modulo, k, i))
            assert all(int(x) % 5 != 0 for x in fk)


def test_a_named_filter_remembers_verdicts_and_answers_like_the_unnamed_one():
    """vs_hnsw_filtered_search_keyed (round 5): queries that carry the same filter_key share the verdicts the predicate gave (two bits
    per slot on the device).  Answers are those of the plain call -- the oracle's, ids and distances --; the predicate is asked about a
    key at most once per filter; once the neighbourhoods are known a query is one walk without a single predicate call; a member that
    is removed and re-added under the same key is asked about again (its verdict may have changed); another key is another filter."""
    import threading

    import vector_store_amd as v
    n, dim, k = 100_000, 64, 10
    data = _dataset(n + 64, dim, 61)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(n + 64)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.COS, 16, 128, 96)
    o.import_graph(ix.export_graph())
    asked = {}
    flip = set()

    def pred(key):
        asked[key] = asked.get(key, 0) + 1
        return (key % 10 == 3) != (key in flip)

    FK = 0xF117E5
    for rnd in range(3):
        before = sum(asked.values())
        for i in range(len(q)):
            gk, gd = ix.filtered_search(q[i], k, pred, filter_key=FK)
            wk, wd = o.filtered_search(q[i], k, lambda key: (key % 10 == 3) != (key in flip))
            assert_same_results(gk, gd, wk, wd, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=("keyed", rnd, i))
        calls = sum(asked.values()) - before
        if rnd == 0:
            assert calls > 1000 * 8, calls          # the first pass asks
        else:
            assert calls == 0, (rnd, calls)         # the same queries again: every verdict is remembered
    assert max(asked.values()) == 1                 # nothing was asked twice
    st = ix.filter_memo_stats()
    assert st["queries"] == 3 * len(q) and st["memories_held"] == 1
    # the unnamed call still asks (and answers the same)
    before = sum(asked.values())
    gk, gd = ix.filtered_search(q[0], k, pred)
    assert sum(asked.values()) > before
    wk, wd = o.filtered_search(q[0], k, lambda key: key % 10 == 3)
    assert_same_results(gk, gd, wk, wd, lambda key: o.distance_to_slot(q[0], int(key)), what="unnamed")
    # a member that changes is asked about again: remove the best match of q[0], re-add it with a verdict that flipped
    best = int(ix.filtered_search(q[0], 1, pred, filter_key=FK)[0][0])
    asked.clear()
    assert ix.remove(best)
    ix.add(best, base[best])
    flip.add(best)                                   # the row's filterable column changed with it: now rejected
    gk, gd = ix.filtered_search(q[0], k, pred, filter_key=FK)
    assert best not in gk.tolist() and asked.get(best, 0) == 1, (best, gk, asked.get(best))
    # another name, another memory: the same key is asked about again
    asked.clear()
    ix.filtered_search(q[1], k, pred, filter_key=FK + 1)
    assert sum(asked.values()) > 100 and ix.filter_memo_stats()["memories_held"] == 2
    # concurrent callers of one filter: same answers
    want = [ix.filtered_search(q[i], k, lambda key: key % 7 == 1)[0].tolist() for i in range(16)]
    errors = []

    def caller(t):
        try:
            for i in range(t, 16, 4):
                assert ix.filtered_search(q[i], k, lambda key: key % 7 == 1, filter_key=77)[0].tolist() == want[i]
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=caller, args=(t,)) for t in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors[:2]


def test_a_named_filter_forgets_when_the_host_rewrites_filtering_columns():
    """Round-5 review, weak 1 / missing 2.  The reference's predicate reads FILTERING COLUMNS, which `Table::upsert` rewrites in place
    under a fixed PrimaryId (`update_columns`, table/mod.rs:676-695, 1053-1061) -- no remove, no add, no new key.  Two waves of the
    same queries around such a rewrite of 5 % of the rows: without a word from the host the named filter answers with the verdicts
    it remembered (the staleness is asserted, so that it is documented); after vs_hnsw_filter_forget_keys(the rewritten rows) -- and,
    separately, after vs_hnsw_filter_forget(the filter) and under a NEW filter_key (a table generation) -- every answer equals the
    oracle's under the new column values, and only what was forgotten is asked again."""
    import vector_store_amd as v
    n, dim, k = 100_000, 64, 10
    data = _dataset(n + 48, dim, 67)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.COS, 16, 128, 96)
    o.import_graph(ix.export_graph())
    column = (np.arange(n) % 10 == 3)                      # the filtering column `f`, restriction `f = true`
    asked = {}

    def pred(key):
        asked[key] = asked.get(key, 0) + 1
        return bool(column[key])

    def wave(filter_key, what):
        """answers of every query; rows that differ from the oracle's under the CURRENT column values"""
        wrong = 0
        for i in range(len(q)):
            gk, gd = ix.filtered_search(q[i], k, pred, filter_key=filter_key)
            wk, wd = o.filtered_search(q[i], k, lambda key: bool(column[key]))
            try:
                assert_same_results(gk, gd, wk, wd, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=(what, i))
            except AssertionError:
                wrong += 1
        return wrong

    FK = 0x51A7E
    assert wave(FK, "first wave") == 0
    assert wave(FK, "warm") == 0
    ever = set(asked)
    # ---- the host rewrites the column of 5 % of the rows: half of the admitted ones near the queries are rejected now, others admitted
    rng = np.random.default_rng(5)
    near = np.unique(np.concatenate([o.filtered_search(q[i], 4, lambda key: bool(column[key]))[0] for i in range(len(q))]).astype(np.int64))
    rewritten = np.unique(np.concatenate([near[::2], rng.choice(n, n // 20, replace=False)]))
    column[rewritten] = ~column[rewritten]
    asked.clear()
    stale = wave(FK, "stale")
    assert stale > 0 and not asked, (stale, len(asked))    # documented: nobody told the engine, it asked nothing and answers from memory
    # ---- (1) the host names the rewritten rows
    ix.filter_forget_keys(rewritten.astype(np.uint64))
    asked.clear()
    assert wave(FK, "after forget_keys") == 0
    # asked again: rewritten rows only (beyond them, members no query had met before -- the walks go on where admitted rows were lost)
    assert asked and (set(asked) & ever) <= set(rewritten.tolist()) and max(asked.values()) == 1, (len(asked), len((set(asked) & ever) - set(rewritten.tolist())))
    assert set(asked) & set(rewritten.tolist())
    st = ix.filter_memo_stats()
    assert st["forget_calls"] == 1 and st["members_forgotten"] == len(rewritten) and st["memories_held"] == 1
    asked.clear()
    assert wave(FK, "warm again") == 0 and not asked
    # ---- (2) a second rewrite, answered by forgetting the whole filter
    back = rewritten[: len(rewritten) // 2]
    column[back] = ~column[back]
    assert ix.filter_forget(FK) == 1 and ix.filter_memo_stats()["memories_held"] == 0
    asked.clear()
    assert wave(FK, "after forget") == 0 and len(asked) > 1000
    # ---- (3) a third rewrite, answered by a new name (filter_key = a registry id per (restrictions, table generation))
    column[back] = ~column[back]
    assert wave(FK + 1, "new generation") == 0
    assert ix.filter_forget(0) == 2 and ix.filter_memo_stats()["memories_held"] == 0
    # unknown keys and an empty list are no-ops
    ix.filter_forget_keys(np.array([2 ** 40, 2 ** 41], dtype=np.uint64))
    ix.filter_forget_keys(np.zeros(0, dtype=np.uint64))


def test_forgetting_under_concurrent_named_queries_never_leaves_a_stale_verdict():
    """forget_keys while queries of the same filter are in flight: whatever they overlap, the queries that START after the call
    returned answer with the new column values (copy-on-forget: verdicts evaluated before the change cannot reach the new memory)."""
    import threading
    import vector_store_amd as v
    n, dim, k = 100_000, 64, 10
    data = _dataset(n + 32, dim, 71)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.COS, 16, 128, 96)
    o.import_graph(ix.export_graph())
    column = (np.arange(n) % 10 == 3)
    FK = 77
    stop = threading.Event()
    errors = []

    def crowd(t):
        try:
            i = t
            while not stop.is_set():
                ix.filtered_search(q[i % len(q)], k, lambda key: bool(column[key]), filter_key=FK)
                i += 4
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=crowd, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    try:
        rng = np.random.default_rng(9)
        for rnd in range(6):
            rewritten = rng.choice(n, n // 20, replace=False)
            column[rewritten] = ~column[rewritten]           # `table.write()` ...
            ix.filter_forget_keys(rewritten.astype(np.uint64))  # ... and the word to the engine, in the reference's order
            for i in range(0, len(q), 3):
                gk, gd = ix.filtered_search(q[i], k, lambda key: bool(column[key]), filter_key=FK)
                wk, wd = o.filtered_search(q[i], k, lambda key: bool(column[key]))
                assert_same_results(gk, gd, wk, wd, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=("round", rnd, i))
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert not errors, errors


def test_an_unnamed_filter_is_one_walk_that_asks_what_usearch_asks():
    """Round 6 (review item 3): the trait's own signature -- an opaque predicate, no name -- is served by ONE walk that asks while it
    runs.  Per query: the answer equals the oracle's, no key is asked about twice, and the predicate is called about as often as
    usearch calls it (a member is asked about when it is new to the visited set and passes the radius test; the walk's radius may lag
    the true one by the answers that are still on their way: a few per cent more).  Walks that hand over (an order-relevant tie) are
    served by the rounds of rounds 3-5 and counted."""
    import vector_store_amd as v
    n, dim, k = 200_000, 96, 10
    data = _dataset(n + 32, dim, 73)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=128)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.COS, 16, 128, 128)
    o.import_graph(ix.export_graph())
    gpu_calls = cpu_calls = 0
    for modulo in (10, 100):
        for i in range(len(q)):
            asked = {}

            def pred(key, m=modulo, asked=asked):
                asked[key] = asked.get(key, 0) + 1
                return key % m == 3

            s0 = ix.filter_ask_stats()
            gk, gd = ix.filtered_search(q[i], k, pred)
            s1 = ix.filter_ask_stats()
            oracle_calls = [0]

            def opred(key, m=modulo, c=oracle_calls):
                c[0] += 1
                return key % m == 3

            wk, wd = o.filtered_search(q[i], k, opred)
            assert_same_results(gk, gd, wk, wd, lambda key, i=i: o.distance_to_slot(q[i], int(key)), what=("ask", modulo, i))
            if s1["queries"] == s0["queries"] + 1:  # served by the asking walk alone
                assert max(asked.values()) == 1, (modulo, i, max(asked.values()))
                gpu_calls += len(asked)
                cpu_calls += oracle_calls[0]
    st = ix.filter_ask_stats()
    assert st["queries"] >= 48 and st["queries"] + st["handed_over"] + st["no_pod"] == 2 * len(q), st
    assert cpu_calls > 0 and gpu_calls <= 1.15 * cpu_calls and gpu_calls >= 0.9 * cpu_calls, (gpu_calls, cpu_calls)


def test_a_predicate_that_stalls_is_outlasted_by_the_asking_walk():
    """The asking walk waits for its caller's answers on the device; a caller that does not answer for a fifth of a second (a closure
    that blocks on a lock, a thread that is not scheduled) must not hold the workgroup: the walk gives up, reports "handed over", and
    the rounds serve the query when the caller comes back -- same answer, no hang."""
    import time
    import vector_store_amd as v
    n, dim, k = 120_000, 64, 10
    data = _dataset(n + 4, dim, 79)
    base, q = data[:n], data[n:]
    ix = v.HipUsearchIndex(dim, v.COS, expansion_search=96)
    ix.reserve(n)
    ix.add_batch(np.arange(n, dtype=np.uint64), base)
    o = OracleIndex(dim, oracle.COS, 16, 128, 96)
    o.import_graph(ix.export_graph())
    ix.filtered_search(q[0], k, lambda key: key % 10 == 3)   # (pods open, contexts exist)
    calls = [0]

    def stalling(key):
        calls[0] += 1
        if calls[0] == 200:
            time.sleep(1.0)
        return key % 10 == 3

    s0 = ix.filter_ask_stats()
    t0 = time.time()
    gk, gd = ix.filtered_search(q[1], k, stalling)
    took = time.time() - t0
    s1 = ix.filter_ask_stats()
    wk, wd = o.filtered_search(q[1], k, lambda key: key % 10 == 3)
    assert_same_results(gk, gd, wk, wd, lambda key: o.distance_to_slot(q[1], int(key)), what="stalled")
    assert s1["handed_over"] == s0["handed_over"] + 1 and took < 10.0, (s0, s1, took)
    # ... and the next query is served by an asking walk again
    gk, gd = ix.filtered_search(q[2], k, lambda key: key % 10 == 3)
    wk, wd = o.filtered_search(q[2], k, lambda key: key % 10 == 3)
    assert_same_results(gk, gd, wk, wd, lambda key: o.distance_to_slot(q[2], int(key)), what="after the stall")
    assert ix.filter_ask_stats()["queries"] == s1["queries"] + 1
