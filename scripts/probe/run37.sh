python bench.py > gpurun_out/r03_bench_6.json 2> gpurun_out/r03_bench_6.err; tail -c 200 gpurun_out/r03_bench_6.json
scripts/profile_round.sh r03_h 200 > gpurun_out/r03_profile_h.log 2>&1; tail -2 gpurun_out/r03_profile_h.log | cut -c1-200
scripts/profile_c5.sh r03 > gpurun_out/r03_profile_c5.log 2>&1; tail -2 gpurun_out/r03_profile_c5.log | cut -c1-300
scripts/profile_round.sh r03_c2 128 --vectors 1000000 > gpurun_out/r03_profile_c2.log 2>&1; tail -2 gpurun_out/r03_profile_c2.log | cut -c1-200
for qz in i8 b1 f16; do python bench.py --quantization $qz --no-side-records --configs none > gpurun_out/r03_h_bench_$qz.json 2> gpurun_out/r03_h_bench_$qz.err; tail -c 200 gpurun_out/r03_h_bench_$qz.json; done
python bench.py --dim 1536 --metric l2sq --no-side-records --configs none > gpurun_out/r03_c3_bench.json 2> gpurun_out/r03_c3_bench.err; tail -c 200 gpurun_out/r03_c3_bench.json
python -m pytest tests -q -m gpu > gpurun_out/r03_gputest_full_3.log 2>&1; tail -3 gpurun_out/r03_gputest_full_3.log
