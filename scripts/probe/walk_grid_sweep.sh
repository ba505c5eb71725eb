#!/bin/bash
# How the usearch-order walk's rate depends on the number of resident workgroups.  VS_HNSW_WALK_GRID caps the grid of
# the LDS instances; VS_HNSW_WALK_PER_CU overrides the occupancy query (work is drawn from a shared counter, so
# oversubscription only costs idle launches).
#   KINDS=i8 N=10000000 EF=208 GRIDS="768 1024 1280 1536" bash scripts/probe/walk_grid_sweep.sh
cd ${GRAFT_REPO_ROOT:-.}
for q in ${KINDS:-i8 b1}; do
  for g in ${GRIDS:-1024 1536 2048 2560}; do
    VS_HNSW_WALK_GRID=$g timeout 600 python3 bench.py --vectors ${N:-1000000} --quantization $q --ef ${EF:-128} --cpu-seconds 0 --boundary-seconds 0 --no-side-records --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print('$q grid $g', 'qps %.0f' % r['value'], 'recall', r['recall_at_10'], 'ms %.2f' % rf['kernel_ms'])"
  done
done
