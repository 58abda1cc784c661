set -u
export GPANN_BENCH_FORCE_DEVICE=0
run2() { tag=$1; shift; env "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --backend tcp --points 6000000 --shards 12 --steps 20 --warmup 3 --no-cpu-baseline --router-budget 20000 > gpurun_out/r4_tcp2_$tag.json 2> gpurun_out/r4_tcp2_$tag.err; tail -1 gpurun_out/r4_tcp2_$tag.err; }
run2 sa1 GPANN_X=1
run2 sa0 GPANN_BENCH_SCAN_AHEAD=0
run2 ra0 GPANN_BENCH_ROUTE_AHEAD=0
unset GPANN_BENCH_FORCE_DEVICE
GPANN_BENCH_FORCE_COMM=1 timeout 600 python bench.py --points 6000000 --shards 12 --steps 20 --warmup 3 --no-cpu-baseline --router-budget 20000 > gpurun_out/r4_rccl1_forced.json 2> gpurun_out/r4_rccl1_forced.err; tail -2 gpurun_out/r4_rccl1_forced.err
