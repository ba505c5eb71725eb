#!/bin/bash
# the pipelined walk's kernel sources of scratch_old/mix (an earlier commit's pipe_device.hpp / kernels_pipe.hip beside the current headers)
# linked into the current library: isolates the kernel from the host side in an A/B of lone-caller latency
cd ${GRAFT_REPO_ROOT:-.}
C=vector_store_amd/csrc
( cd scratch_old/mix && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DVS_AR=0 -c kernels_pipe.hip -o /tmp/pk_0_mix.o 2>&1 | grep -i error )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libvs_hnsw_mix.so $C/engine.o $C/kernels_dispatch.o $C/kernels_misc.o $C/arith_*.o $C/wk_*.o /tmp/pk_0_mix.o $C/pk_1.o $C/pk_2.o $C/pk_3.o $C/pk_4.o $C/pk_5.o || exit 1
for lib in /tmp/libvs_hnsw_mix.so vector_store_amd/libvs_hnsw.so; do
  echo "== $lib"
  VS_HNSW_LIB=$lib timeout 300 python scripts/probe/callers_probe.py ${1:-10000000} 200 1.5 f32 1x1,17x1 2>&1 | grep -a "threads\|rror"
done
