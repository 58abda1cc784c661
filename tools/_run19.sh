set -u
python -m pytest tests/test_gpu_filter.py tests/test_gpu_cells.py tests/test_gpu_merge_bound.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-hard-workload --router-budget 20000 > gpurun_out/r4_bench_pk.json 2> gpurun_out/r4_bench_pk.err; tail -1 gpurun_out/r4_bench_pk.err
python bench.py --dist uniform --steps 5 --warmup 2 --no-cpu-baseline --router-budget 20000 > gpurun_out/r4_bench_uniform_pk.json 2> gpurun_out/r4_bench_uniform_pk.err; tail -1 gpurun_out/r4_bench_uniform_pk.err
python bench.py --sigma 0.8 --steps 10 --warmup 3 --no-cpu-baseline --no-hard-workload --router-budget 20000 > gpurun_out/r4_bench_s08_pk.json 2> gpurun_out/r4_bench_s08_pk.err; tail -1 gpurun_out/r4_bench_s08_pk.err
