#!/bin/bash
# run_prof.sh OUT: the phases of a hop of the asking walk and of the exact (named) walk, profile build in scripts/probe/var_P
OUT=$1
VS_LIB_DIR=$PWD/scripts/probe/var_P VS_HNSW_ASK_DEBUG=1 timeout 300 python scripts/probe/ask_probe.py --threads 1 --seconds 0.5 > $OUT.ask 2>&1
VS_LIB_DIR=$PWD/scripts/probe/var_P VS_HNSW_ASK_DEBUG=1 timeout 300 python scripts/probe/ask_probe.py --threads 1 --seconds 0.5 --named 7 > $OUT.exact 2>&1
python - $OUT <<'PY'
import re, sys
for kind in ("ask", "exact"):
    rows = []
    for ln in open(sys.argv[1] + "." + kind):
        if ln.startswith("[" + kind + "]"):
            v = [int(x) for x in re.findall(r"\d+", ln.split("]", 1)[1])]
            if 1500 < v[0] < 3000:  # the 10 % walks
                rows.append(v)
    if not rows: print(kind, "no rows"); continue
    n = len(rows); hops = sum(r[0] for r in rows) / n
    names = ["hops", "ticks", "(16)", "head", "decide", "pop", "entry", "atomics", "verdicts", "pushes", "schedule", "wait_entry"]
    print(kind, "walks", n, "hops %.0f" % hops, "ms %.2f" % (sum(r[1] for r in rows) / n / 1e5),
          " clocks per hop:", {names[i]: round(sum(r[i] for r in rows) / n * 16 / hops) for i in range(3, 12)})
PY
