(python scripts/probe/callers_probe.py 10000000 200 2 f32 33x1,65x1,17x1,33x1,65x1; VS_HNSW_DIRECT_CALLERS=0 python scripts/probe/callers_probe.py 10000000 200 2 f32 33x1,65x1,17x1,33x1,65x1) 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_callers_probe_4.log; cat gpurun_out/r03_callers_probe_4.log
python -m pytest tests/test_gpu_filtered.py tests/test_gpu_round3_fixes.py -x -q > gpurun_out/r03_gputest_15.log 2>&1; tail -3 gpurun_out/r03_gputest_15.log
python scripts/probe/filtered_probe.py 10000000 200 1,17 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_filtered_probe_3.log; cut -c1-330 gpurun_out/r03_filtered_probe_3.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_quantized.py -x -q > gpurun_out/r03_gputest_16.log 2>&1; tail -3 gpurun_out/r03_gputest_16.log
