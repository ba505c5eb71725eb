python -m pytest tests/test_gpu_filtered.py -x -q > gpurun_out/r03_gputest_17.log 2>&1; tail -3 gpurun_out/r03_gputest_17.log
python scripts/probe/filtered_probe.py 10000000 200 1,17,64,128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_filtered_probe_5.log; cut -c1-330 gpurun_out/r03_filtered_probe_5.log
python scripts/probe/callers_probe.py 10000000 200 2 f32 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_callers_probe_6.log; cat gpurun_out/r03_callers_probe_6.log
