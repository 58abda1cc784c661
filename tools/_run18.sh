set -u
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
( time python bench.py > gpurun_out/r4_bench_ns_default.json 2> gpurun_out/r4_bench_ns_default.err ) 2>&1 | grep real; tail -1 gpurun_out/r4_bench_ns_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hard-workload --router-budget 20000 --emulate-world 2,4,8 > gpurun_out/r4_emulate.json 2> gpurun_out/r4_emulate.err; grep "emulated world [248]:" gpurun_out/r4_emulate.err
