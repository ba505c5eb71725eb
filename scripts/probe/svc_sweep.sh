#!/bin/bash
# Development aid: single-query service throughput, direct (C ABI) and over HTTP, at a few concurrencies.
B=vector_store_amd/vs_bench; D=/tmp/d1m
[ -f $D/data.fbin ] || $B gen --data-dir $D --n 1000000 --dim 768 --queries 10000 --neighbors 10 2>&1 | tail -1
vector_store_amd/vs_httpd --data-dir $D --threads 8 --expansion-search 128 --port 6200 2>/tmp/httpd.err &
for c in 16 64 256 1024; do echo "== http concurrency $c"; timeout 120 $B search-http --data-dir $D --limit 10 --duration 4 --concurrency $c --port 6200 2>&1 | grep "QPS\|P50\|P99\|error" | tr '\n' ' '; echo; done
kill %1
for c in 1 16 64 256 1024; do echo "== direct conc $c";  timeout 300 $B search --data-dir $D --limit 10 --duration 4 --concurrency $c --expansion-search 128 2>&1 | grep "QPS\|P50\|P99\|launches" | tr '\n' ' '; echo; done
echo "== 16 x inflight 256"; timeout 300 $B search --data-dir $D --limit 10 --duration 4 --concurrency 16 --inflight 256 --expansion-search 128 2>&1 | grep "QPS\|P50\|P99\|launches" | tr '\n' ' '; echo
