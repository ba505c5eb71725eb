#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf /tmp/bp; mkdir -p /tmp/bp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -- python3 scripts/probe/block_search_probe.py ${1:-1000000} ip child > /tmp/bp/out.txt 2>/tmp/bp/err.txt
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(x in n for x in ("block_","exact_","split_","prepare_","row_norm")):
        print(n.split("(")[0][-40:], r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3), "total_ms %.1f" % (float(r["TotalDurationNs"])/1e6))
PY
