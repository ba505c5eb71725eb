#!/bin/bash
# A/B of the pipelined walk's compile-time knobs, built ON THE GPU BOX (the in-tree library stays the product build):
#   bash scripts/probe/pipe_variants.sh "base: solo:-DVS_PIPE_SOLO p2:-DVS_PIPE_PARTS=2" [vectors] [threads]
cd ${GRAFT_REPO_ROOT:-.}/vector_store_amd/csrc
for v in $1; do
  name=${v%%:*}; fl=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DVS_AR=0 $fl -c kernels_pipe.hip -o /tmp/pk_0_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libvs_hnsw_v_$name.so engine.o kernels_dispatch.o kernels_misc.o arith_*.o wk_*.o /tmp/pk_0_$name.o pk_1.o pk_2.o pk_3.o pk_4.o pk_5.o ) &
done
wait
cd ../..
for v in $1; do
  name=${v%%:*}
  echo "== $name"
  if [ -n "$LONE" ]; then VS_HNSW_LIB=/tmp/libvs_hnsw_v_$name.so timeout 300 python scripts/probe/callers_probe.py ${2:-2000000} 200 1.5 f32 1x1,17x1 2>&1 | grep -a "threads\|rror"; fi
  if [ -z "$NOFILT" ]; then PIPE_PROBE_FAST=1 PIPE_PROBE_MODS=${MODS:-10} VS_HNSW_LIB=/tmp/libvs_hnsw_v_$name.so timeout 300 python scripts/probe/pipe_probe.py ${2:-2000000} 200 ${3:-1,17} 1.5 2>&1 | grep -a "pipe:\|rror"; fi
done
