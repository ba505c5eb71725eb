timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_limits.py -x -q 2>&1 | tail -3
timeout 900 python3 scripts/c5_batched_ip.py 10000000 > gpurun_out/r06_c5_batched_ip_2.json 2> gpurun_out/r06_c5_batched_ip_2.err; echo rc $?
python3 -c "
import json
l=json.loads(open('gpurun_out/r06_c5_batched_ip_2.json').read().strip().splitlines()[-1])
print({k:v for k,v in l.items() if not isinstance(v,(dict,list))})
" | cut -c1-1500
