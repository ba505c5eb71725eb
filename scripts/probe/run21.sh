python -m pytest tests/test_gpu_round3_fixes.py tests/test_gpu_quantized.py tests/test_gpu_team.py -x -q > gpurun_out/r03_gputest_12.log 2>&1; tail -3 gpurun_out/r03_gputest_12.log
(VS_HNSW_SERVICE_SLOTS=2 VS_HNSW_SERVICE_SPIN=0 python scripts/probe/callers_probe.py 10000000 200 2; VS_HNSW_SERVICE_SPIN=0 python scripts/probe/callers_probe.py 10000000 200 2; python scripts/probe/callers_probe.py 10000000 200 2; VS_HNSW_SERVICE_SLOTS=4 python scripts/probe/callers_probe.py 10000000 200 2) > gpurun_out/r03_callers_probe_1.log 2>&1; cat gpurun_out/r03_callers_probe_1.log
VS_HNSW_WALK_DENSE=0 python bench.py --quantization i8 --cpu-seconds 0 --cpu-build-vectors 0 --boundary-seconds 0 --no-side-records --configs none > gpurun_out/r03_i8_dense0.json 2> gpurun_out/r03_i8_dense0.err; python bench.py --quantization i8 --cpu-seconds 0 --cpu-build-vectors 0 --boundary-seconds 0 --no-side-records --configs none > gpurun_out/r03_i8_dense1.json 2> gpurun_out/r03_i8_dense1.err
python - <<'PY'
import json
for f in ("gpurun_out/r03_i8_dense0.json","gpurun_out/r03_i8_dense1.json"):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["value"], d["ef_search"], d["recall_at_10"], d["roofline"].get("kernel"), d["roofline"].get("kernel_ms"))
PY
