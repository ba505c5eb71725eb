python -m pytest tests -q -m gpu > gpurun_out/r03_gputest_full_2.log 2>&1; tail -5 gpurun_out/r03_gputest_full_2.log
python bench.py > gpurun_out/r03_bench_4.json 2> gpurun_out/r03_bench_4.err; tail -c 600 gpurun_out/r03_bench_4.json; tail -3 gpurun_out/r03_bench_4.err
