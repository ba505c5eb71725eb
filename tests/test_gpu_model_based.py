"""Model-based test: a random sequence of the trait's operations (add / remove / re-add under a new epoch /
reserve growth / search / filtered search) applied to the HIP engine, checked after every phase against
(a) a plain dict model of what must be stored and (b) the CPU algorithm walking the engine's own graph."""
import numpy as np
import pytest

import oracle
from oracle import OracleIndex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric,quant,seed", [("l2sq", "f32", 1), ("cos", "f32", 2), ("ip", "f16", 3), ("cos", "i8", 4)])
def test_random_operation_sequences(metric, quant, seed):
    import vector_store_amd as vs
    rng = np.random.default_rng(seed)
    dim = 20
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric], quantization=vs.SCALARS[quant])
    model = {}  # key -> vector
    cap = 0
    next_row = 0

    def check():
        assert ix.size() == len(model)
        g = ix.export_graph()
        o = OracleIndex(dim, oracle.METRICS[metric], quantization=oracle.SCALARS[quant])
        o.import_graph(g)
        assert o.size() == len(model)
        live = {int(k) for k in g["keys"].tolist() if int(k) != 0xFFFFFFFFFFFFFFFF}
        assert live == set(model)
        k = min(len(model) + 3, 250)
        o.set_expansion_search(max(k, 64))
        ix.set_expansion_search(64)
        for _ in range(6):
            q = rng.standard_normal(dim).astype(np.float32)
            gk, gd = ix.search(q, k)
            ok_, od_ = o.search(q, k)
            assert len(gk) == len(ok_) and set(gk.tolist()) <= set(model)
            assert np.allclose(gd, od_, rtol=1e-5, atol=1e-5)
            assert sorted(gk.tolist()) == sorted(ok_.tolist()) or np.allclose(np.sort(gd), np.sort(od_), rtol=1e-5, atol=1e-5)
            assert len(set(gk.tolist())) == len(gk)
            # filtered: keys whose row index is even
            fk, fd = ix.filtered_search(q, 5, lambda key: (key & 0xFFFF) % 2 == 0)
            o.set_expansion_search(64)
            ek, ed = o.filtered_search(q, 5, lambda key: (key & 0xFFFF) % 2 == 0)
            o.set_expansion_search(max(k, 64))
            assert all((int(x) & 0xFFFF) % 2 == 0 for x in fk)
            assert len(fk) == min(5, sum(1 for x in model if (x & 0xFFFF) % 2 == 0))
            assert np.all(fd[:-1] <= fd[1:])
            # the predicate gates admission inside the traversal: same result set and order as the CPU algorithm
            assert np.allclose(fd, ed, rtol=1e-5, atol=1e-5)
            assert fk.tolist() == ek.tolist(), (fk, ek, fd, ed)

    for phase in range(12):
        op = rng.choice(["add", "add", "remove", "update", "grow"])
        if op == "grow" or cap - ix.size() < 40:
            cap = cap + int(rng.integers(50, 120))
            ix.reserve(cap)
            assert ix.capacity() == cap
        if op == "add":
            n = int(rng.integers(1, 30))
            keys = np.arange(next_row, next_row + n, dtype=np.uint64)
            vecs = rng.standard_normal((n, dim)).astype(np.float32)
            next_row += n
            if rng.random() < 0.5:
                ix.add_batch(keys, vecs)
            else:
                for i in range(n):
                    ix.add(int(keys[i]), vecs[i])
            model.update({int(k): v for k, v in zip(keys, vecs)})
        elif op == "remove" and model:
            for key in rng.choice(sorted(model), size=min(len(model), int(rng.integers(1, 8))), replace=False).tolist():
                assert ix.remove(key)
                assert not ix.remove(key)
                del model[key]
        elif op == "update" and model:
            # the reference re-keys an updated row: RemoveBeforeAddValue(old epoch) then AddVector(new epoch)
            for key in rng.choice(sorted(model), size=min(len(model), 4), replace=False).tolist():
                assert ix.remove(key)
                del model[key]
                new_key = ((key >> 48) + 1) << 48 | (key & 0xFFFFFFFFFFFF)
                v = rng.standard_normal(dim).astype(np.float32)
                ix.add(new_key, v)
                model[new_key] = v
        with pytest.raises(vs.VsError):
            if model:
                ix.add(next(iter(model)), np.zeros(dim, dtype=np.float32))  # duplicate key
            else:
                raise vs.VsError(-4, "nothing to duplicate")
        check()
