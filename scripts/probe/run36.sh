for hq in 12 16 20 24; do echo "GPU_MAX_HW_QUEUES=$hq"; GPU_MAX_HW_QUEUES=$hq python scripts/probe/aftermath_probe.py 1000000 2>&1 | grep -v amdgpu.ids | grep "fresh\|filtered\|17 x 1"; done
echo "TEAM_GLOBAL=0"; VS_HNSW_TEAM_GLOBAL=0 python scripts/probe/aftermath_probe.py 1000000 2>&1 | grep -v amdgpu.ids | grep "fresh\|filtered"
