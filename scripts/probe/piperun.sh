timeout 1200 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_pods.py -x -q 2>&1 | tail -3
timeout 600 python3 scripts/probe/callers_probe.py 10000000 200 2 f32 1x1,17x1,64x1 2>&1 | grep -v amdgpu.ids
