python -m pytest tests/test_gpu_filtered.py tests/test_gpu_round3_fixes.py -x -q > gpurun_out/r03_gputest_18.log 2>&1; tail -3 gpurun_out/r03_gputest_18.log
VS_HNSW_LIB=vector_store_amd/libvs_hnsw_dbg.so VS_HNSW_WALK_DEBUG=1 python scripts/probe/filtered_phase_probe.py 10000000 10 > gpurun_out/r03_filtered_phase_2.log 2>&1; grep -v amdgpu.ids gpurun_out/r03_filtered_phase_2.log | tail -5 | cut -c1-400
python scripts/probe/filtered_probe.py 10000000 200 1,17 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_filtered_probe_6.log; cut -c1-330 gpurun_out/r03_filtered_probe_6.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_quantized.py tests/test_gpu_limits.py -x -q > gpurun_out/r03_gputest_19.log 2>&1; tail -3 gpurun_out/r03_gputest_19.log
