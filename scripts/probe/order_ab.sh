#!/bin/bash
# A/B: fused-list kernel vs usearch-order walk, same index, same beam (1M x 768, ef 128)
cd ${GRAFT_REPO_ROOT:-.}
for q in ${KINDS:-i8 b1 f32 f16}; do
  for o in fused usearch; do
    VS_HNSW_ORDER=$o python3 bench.py --vectors ${N:-1000000} --quantization $q --ef ${EF:-128} --cpu-seconds 0 --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print('$q $o', 'qps %.0f' % r['value'], 'recall', r['recall_at_10'], 'ms %.2f' % rf['kernel_ms'], 'frac %.3f' % rf['frac'], 'E_q %.0f H_q %.0f' % (rf['evals_per_query'], rf['hops_per_query']))"
  done
done
