python bench.py --boundary-seconds 0 --cpu-seconds 0 --cpu-build-vectors 0 > gpurun_out/r03_bench_noboundary.json 2> gpurun_out/r03_bench_noboundary.err
python - <<'PY'
import json
d=None
for l in open("gpurun_out/r03_bench_noboundary.json"):
    if l.startswith("{"): d=json.loads(l)
print("no boundary:", d["value"], [(c["config"], c.get("ms_per_batch"), round(c["roofline"]["frac"],3)) for c in d["configs"]])
PY
python -m pytest tests/test_gpu_round3_fixes.py tests/test_gpu_quantized.py -q > gpurun_out/r03_gputest_21.log 2>&1; tail -3 gpurun_out/r03_gputest_21.log
(python scripts/probe/dense_probe.py 10000000 280 b1; VS_HNSW_WALK_DENSE=0 python scripts/probe/dense_probe.py 10000000 280 b1) 2>&1 | grep -v amdgpu.ids | grep "^n \|^launch 0" | cut -c1-200
