#!/bin/bash
# Profile one bench.py workload on the GPU box and leave only the small summaries in gpurun_out/summary/.
#   scripts/profile_round.sh <tag> <ef> [bench.py args...]      e.g.  scripts/profile_round.sh r01_h 200
# Separate rocprofv3 runs (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; --pmc the four TCC hit / miss / request counters), then
# scripts/summarise_profiles.py.  Raw traces stay in /tmp (a 10M build's traces exceed what gpurun copies back).
set -u
TAG=$1; EF=$2; shift 2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=/tmp/prof_${TAG}_$$_$RANDOM
mkdir -p "$T" "$O/summary"; cd "$R"
python3 bench.py "$@" > "$O/summary/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats" -- python3 bench.py --ef "$EF" --steps 10 --warmup 2 --cpu-seconds 0 --boundary-seconds 0 --mixed-seconds 0 --no-side-records "$@" > "$O/summary/${TAG}_bench_under_rocprof.json" 2> "$O/${TAG}_stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$T/pmc_$c" -- python3 bench.py --ef "$EF" --steps 3 --warmup 1 --cpu-seconds 0 --boundary-seconds 0 --mixed-seconds 0 --no-side-records "$@" > "$O/${TAG}_pmc_$c.json" 2> "$O/${TAG}_pmc_$c.err"
done
# one more pass: L2 hits / misses and the memory-side read requests (four TCC slots: they fit one pass)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d "$T/pmc_TCC" -- python3 bench.py --ef "$EF" --steps 3 --warmup 1 --cpu-seconds 0 --boundary-seconds 0 --mixed-seconds 0 --no-side-records "$@" > "$O/${TAG}_pmc_TCC.json" 2> "$O/${TAG}_pmc_TCC.err"
python3 scripts/summarise_profiles.py --stats-dir "$T/stats" --fetch-dir "$T/pmc_FETCH_SIZE" --write-dir "$T/pmc_WRITE_SIZE" --tcc-dir "$T/pmc_TCC" \
  --bench-json "$O/summary/${TAG}_bench_under_rocprof.json" --out "$O/summary" --tag "$TAG"
cat "$O/summary/${TAG}_traffic.json" "$O/summary/${TAG}_traffic_insert.json" "$O/summary/${TAG}_traffic_mfma.json"; grep -E "hnsw_search|hnsw_insert|exact_dist_mfma" "$O/summary/${TAG}_kernel_stats.csv"
