#!/bin/bash
# rocprofv3 --kernel-trace --stats of the pipelined walk as LAUNCHES (VS_HNSW_PODS=0: a pod is one launch that lasts as long as its callers
# keep it busy, so its duration says nothing about a query): one lone filtered caller at 10 % selectivity, then one lone plain caller.
#   scripts/profile_pipe.sh <tag> [vectors]
set -u
TAG=$1; N=${2:-10000000}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=/tmp/prof_$TAG
mkdir -p "$T" "$O/summary"; cd "$R"
export VS_HNSW_PODS=0 PIPE_PROBE_FAST=1 PIPE_PROBE_MODS=10
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/filtered" -- python3 scripts/probe/pipe_probe.py $N 200 1 3 > "$O/${TAG}_filtered.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/plain" -- python3 scripts/probe/callers_probe.py $N 200 3 f32 1x1 > "$O/${TAG}_plain.log" 2>&1
for leg in filtered plain; do
  f=$(find "$T/$leg" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E "Name|pipe_walk|hnsw_walk_kernel|export_round|apply_verdicts|hnsw_search_kernel" "$f" > "$O/summary/${TAG}_${leg}_kernel_stats.csv"
done
grep -a "pipe:\|threads" "$O/${TAG}_filtered.log" "$O/${TAG}_plain.log"
cat "$O/summary/${TAG}_filtered_kernel_stats.csv" "$O/summary/${TAG}_plain_kernel_stats.csv" | cut -c1-260
