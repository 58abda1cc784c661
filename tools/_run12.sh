set -u
python -m pytest tests -m gpu -x -q -p no:cacheprovider -s 2>&1 | grep -E "passed|failed|calls/s|Error" | tail -8
export GPANN_BENCH_FORCE_DEVICE=0
for RA in 1 0; do
  GPANN_BENCH_ROUTE_AHEAD=$RA timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend tcp --points 6000000 --shards 12 --steps 20 --warmup 3 --no-cpu-baseline --router-budget 20000 > gpurun_out/r4_tcp2_ra$RA.json 2> gpurun_out/r4_tcp2_ra$RA.err
  tail -2 gpurun_out/r4_tcp2_ra$RA.err
done
unset GPANN_BENCH_FORCE_DEVICE
GPANN_BENCH_SWEEP="GPANN_FILTER_SAMPLE_UNIT=1,2;GPANN_FILTER_Q8_L1=1,0" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --router-budget 20000 --no-hard-workload > gpurun_out/r4_sweep3.json 2> gpurun_out/r4_sweep3.err
grep sweep gpurun_out/r4_sweep3.err | cut -c16-300
