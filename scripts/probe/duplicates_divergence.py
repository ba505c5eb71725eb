import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import vector_store_amd as vs, oracle
rng = np.random.default_rng(5)
dim = 64
w = rng.standard_normal((16, dim)).astype(np.float32) / 4
copies, uniq = 50, 12
u = (rng.standard_normal((uniq, 16)).astype(np.float32) @ w + 0.05 * rng.standard_normal((uniq, dim)).astype(np.float32))
base = np.repeat(u, copies, axis=0)
perm = rng.permutation(len(base)); base = base[perm]; ident = np.repeat(np.arange(uniq), copies)[perm]
n = len(base)
for metric in ("l2sq", "cos"):
    o = oracle.OracleIndex(dim, oracle.METRICS[metric]); o.reserve(n)
    ix = vs.HipUsearchIndex(dim, vs.METRICS[metric]); ix.reserve(n)
    first = None
    for i in range(n):
        o.add(i, base[i]); ix.add(i, base[i]); ix.size()
        if first is None and i % 10 == 9:
            go, gg = o.export_graph(), ix.export_graph()
            diff = [s for s in range(i + 1) if set(go["adj0"][s].tolist()) != set(gg["adj0"][s].tolist())]
            if diff:
                first = (i, diff[:5])
    go, gg = o.export_graph(), ix.export_graph()
    deg_o = [(go["adj0"][s] != 0xFFFFFFFF).sum() for s in range(n)]
    deg_g = [(gg["adj0"][s] != 0xFFFFFFFF).sum() for s in range(n)]
    same_o = np.mean([np.mean([ident[x] == ident[s] for x in go["adj0"][s] if x != 0xFFFFFFFF]) for s in range(n)])
    same_g = np.mean([np.mean([ident[x] == ident[s] for x in gg["adj0"][s] if x != 0xFFFFFFFF]) for s in range(n)])
    print(metric, "first divergence", first, "| mean degree oracle", np.mean(deg_o), "gpu", np.mean(deg_g), "| share of links to own copies: oracle", round(same_o, 3), "gpu", round(same_g, 3))
    if first:
        s = first[1][0]
        print("   slot", s, "oracle row", sorted(x for x in go["adj0"][s].tolist() if x != 0xFFFFFFFF)[:40])
        print("   slot", s, "gpu    row", sorted(x for x in gg["adj0"][s].tolist() if x != 0xFFFFFFFF)[:40])
