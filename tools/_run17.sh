set -u
python bench.py > gpurun_out/r4_bench_ns_default.json 2> gpurun_out/r4_bench_ns_default.err; tail -2 gpurun_out/r4_bench_ns_default.err
bash tools/profile_bench.sh r4ns --router-budget 20000 --no-hard-workload > gpurun_out/r4_profile.log 2>&1; tail -3 gpurun_out/r4_profile.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hard-workload --router-budget 20000 --emulate-world 2,4,8 > gpurun_out/r4_emulate.json 2> gpurun_out/r4_emulate.err; grep "emulated world [248]:" gpurun_out/r4_emulate.err
python bench.py --config c4 --steps 30 --warmup 5 > gpurun_out/r4_bench_c4.json 2> gpurun_out/r4_bench_c4.err; tail -1 gpurun_out/r4_bench_c4.err
python bench.py --dist uniform --steps 5 --warmup 2 --no-cpu-baseline --router-budget 20000 > gpurun_out/r4_bench_uniform.json 2> gpurun_out/r4_bench_uniform.err; tail -1 gpurun_out/r4_bench_uniform.err
python bench.py --config c5 --steps 30 --warmup 5 > gpurun_out/r4_bench_c5.json 2> gpurun_out/r4_bench_c5.err; tail -1 gpurun_out/r4_bench_c5.err
python bench.py --config c5 --sigma 0.8 --steps 20 --warmup 5 > gpurun_out/r4_bench_c5_sigma08.json 2> gpurun_out/r4_bench_c5_sigma08.err; tail -1 gpurun_out/r4_bench_c5_sigma08.err
