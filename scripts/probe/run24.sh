for hq in 48 96; do
  echo "GPU_MAX_HW_QUEUES=$hq"
  GPU_MAX_HW_QUEUES=$hq python scripts/probe/callers_probe.py 10000000 200 2 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$hq python scripts/probe/filtered_probe.py 10000000 200 17,64 2>&1 | grep -v amdgpu.ids | cut -c1-330
done > gpurun_out/r03_hwq_probe.log 2>&1
cat gpurun_out/r03_hwq_probe.log
