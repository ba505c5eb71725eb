for q in i8 b1; do timeout 600 python bench.py --quantization $q --no-side-records --configs none --cpu-seconds 0 --boundary-seconds 0 --mixed-seconds 0 > gpurun_out/r06_h_bench_$q.json 2> gpurun_out/r06_h_bench_$q.err; echo $q rc $?; done
python3 -c "
import json
for q in ('i8','b1'):
    l=json.loads(open('gpurun_out/r06_h_bench_%s.json'%q).read().strip().splitlines()[-1]); print(q, l['value'], l['ms_per_step'], l.get('recall_at_10'))
"
timeout 900 python -m pytest tests/test_gpu_quantized.py -x -q 2>&1 | tail -3
for t in 2; do echo "== VS_HNSW_B1_POD_TEAM=$t"; VS_HNSW_B1_POD_TEAM=$t timeout 600 python3 scripts/probe/callers_probe.py 10000000 200 2 b1 1x1,17x1,64x1 2>&1 | grep -v amdgpu.ids; done
