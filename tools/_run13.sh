set -u
python bench.py --config c2 --steps 40 --warmup 5 > gpurun_out/r4_bench_c2.json 2> gpurun_out/r4_bench_c2.err; tail -1 gpurun_out/r4_bench_c2.err
python bench.py --config c3 --steps 40 --warmup 5 > gpurun_out/r4_bench_c3.json 2> gpurun_out/r4_bench_c3.err; tail -1 gpurun_out/r4_bench_c3.err
bash tools/trace_steps.sh c2 --config c2 > /dev/null 2>&1
bash tools/trace_steps.sh c3 --config c3 > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hard-workload --router-budget 20000 --emulate-world 2,4,8 > gpurun_out/r4_emulate.json 2> gpurun_out/r4_emulate.err; grep "emulated world" gpurun_out/r4_emulate.err | tail -20
