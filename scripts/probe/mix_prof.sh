#!/bin/bash
# phase clocks of lone plain queries: the earlier commit's pipelined-walk sources (scratch_old/mix) against the tree's, both built
# with -DVS_WALK_PROFILE on the GPU box
cd ${GRAFT_REPO_ROOT:-.}
C=vector_store_amd/csrc
( cd scratch_old/mix && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DVS_AR=0 -DVS_WALK_PROFILE -c kernels_pipe.hip -o /tmp/pk_0_mixp.o 2>&1 | grep -i " error" ) &
( cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DVS_AR=0 -DVS_WALK_PROFILE -c kernels_pipe.hip -o /tmp/pk_0_newp.o 2>&1 | grep -i " error" ) &
wait
for v in mixp newp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /tmp/libvs_hnsw_$v.so $C/engine.o $C/kernels_dispatch.o $C/kernels_misc.o $C/arith_*.o $C/wk_*.o /tmp/pk_0_$v.o $C/pk_1.o $C/pk_2.o $C/pk_3.o $C/pk_4.o $C/pk_5.o || exit 1
  echo "== $v"
  VS_HNSW_LIB=/tmp/libvs_hnsw_$v.so timeout 300 python scripts/probe/lone_pipe_debug.py ${1:-2000000} 200 2>&1 | grep -a "walk pipe" | tail -6 | cut -c1-420
done
