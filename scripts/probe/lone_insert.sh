#!/bin/bash
# scripts/probe/lone_insert.sh <tag> [vectors]: timing of lone inserts, then a rocprofv3 kernel trace of the same (kernel stats of the flush's launches)
set -u
TAG=$1; N=${2:-1000000}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=/tmp/li_${TAG}_$$_$RANDOM
mkdir -p "$T" "$O"; cd "$R"
timeout 600 python3 scripts/probe/lone_insert.py --vectors $N > "$O/${TAG}_plain.log" 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$T" -- python3 scripts/probe/lone_insert.py --vectors $N --n 220 > "$O/${TAG}_prof.log" 2>&1
f=$(find "$T" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$O/${TAG}_kernel_stats.csv"
t=$(find "$T" -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && tail -n 400 "$t" | cut -d, -f8-12,14- > "$O/${TAG}_trace_tail.csv"
cat "$O/${TAG}_plain.log"; tail -3 "$O/${TAG}_prof.log"; head -30 "$O/${TAG}_kernel_stats.csv" | cut -c1-200
