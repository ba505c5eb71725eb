timeout 900 python -m pytest tests/test_gpu_quantized.py -x -q 2>&1 | tail -5
for t in 2 1; do echo "== VS_HNSW_B1_POD_TEAM=$t"; VS_HNSW_B1_POD_TEAM=$t timeout 600 python3 scripts/probe/callers_probe.py 10000000 200 2 b1 1x1,17x1,64x1 2>&1 | grep -v amdgpu.ids; done
