#!/bin/bash
# crowds of unnamed-filter callers: the rounds (default beyond cores + cores / 4) against asking walks with sleeping callers
OUT=$1
for cfg in "0 30" "16 30" "16 60" "16 15"; do set -- $cfg
  echo "== VS_HNSW_ASK_CROWD=$1 VS_HNSW_ASK_SLEEP_US=$2" >> $OUT
  VS_HNSW_ASK_CROWD=$1 VS_HNSW_ASK_SLEEP_US=$2 timeout 400 python scripts/probe/ask_probe.py --threads 17,64,128 --seconds 2 2>&1 | grep '^{"mod' >> $OUT
done
cat $OUT
