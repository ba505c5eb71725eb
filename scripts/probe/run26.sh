python scripts/probe/callers_probe.py 10000000 200 2 f32 1x1,17x1,33x1,65x1,17x1,33x1,16x256 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_callers_probe_5.log; cat gpurun_out/r03_callers_probe_5.log
python scripts/probe/filtered_probe.py 10000000 200 17,64,128 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_filtered_probe_4.log; cut -c1-330 gpurun_out/r03_filtered_probe_4.log
