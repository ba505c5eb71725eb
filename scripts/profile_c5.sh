#!/bin/bash
# Profile BASELINE.json configs[4] (q = 256 batches, 10M x 768 inner product) on the GPU box: rocprofv3 kernel stats, then
# --pmc FETCH_SIZE, WRITE_SIZE and SQ_VALU_MFMA_BUSY_CYCLES in separate passes, reduced to gpurun_out/summary/<tag>_c5_*.
#   scripts/profile_c5.sh r02 [vectors]
set -u
TAG=$1; N=${2:-10000000}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=/tmp/prof_c5_${TAG}_$$_$RANDOM
mkdir -p "$T" "$O/summary"; cd "$R"
rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats" -- python3 scripts/c5_batched_ip.py "$N" > "$O/summary/${TAG}_c5_batched_ip.json" 2> "$O/${TAG}_c5_stats.err"
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$T/pmc_$c" -- python3 scripts/c5_batched_ip.py "$N" > "$O/${TAG}_c5_pmc_$c.json" 2> "$O/${TAG}_c5_pmc_$c.err"
done
python3 scripts/summarise_c5.py --dir "$T" --vectors "$N" --out "$O/summary" --tag "$TAG"
cat "$O/summary/${TAG}_c5_kernels.json"
