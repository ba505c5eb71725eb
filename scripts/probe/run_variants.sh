#!/bin/bash
# run_variants.sh OUT NAMES...: scripts/probe/ask_probe.py once per development build in scripts/probe/var_NAME (see /tmp/build_variant.sh)
OUT=$1; shift
for v in "$@"; do
  echo "== variant $v" >> $OUT
  VS_LIB_DIR=$([ "$v" = tree ] && echo "" || echo $PWD/scripts/probe/var_$v) timeout 300 python scripts/probe/ask_probe.py --named ${NAMED:-0} --threads ${THREADS:-1} --seconds ${SECONDS_PER_LEG:-1.0} 2>&1 | grep '^{"mod' >> $OUT
done
cat $OUT
