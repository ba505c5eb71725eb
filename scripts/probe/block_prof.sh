#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rm -rf /tmp/bp; mkdir -p /tmp/bp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp -- python3 scripts/probe/block_search_probe.py 1000000 ip child > /tmp/bp/out.txt 2>/tmp/bp/err.txt
f=$(find /tmp/bp -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-220
