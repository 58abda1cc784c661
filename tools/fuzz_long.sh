#!/bin/bash
# Runs on the GPU box (via gpurun): the five randomised parity tests over seed ranges far beyond the suite's (GPANN_FUZZ_SEEDS), eight processes on the one GPU.
#   gpurun --timeout 700 -- 'bash tools/fuzz_long.sh [scale=1]'      scale 1: ~16 000 configurations in about 3.5 minutes
# Every configuration is compared bit for bit with the CPU oracle; a failure names its seed (profiles/r06/NOTES.md §5: round 6's run found one defect).
set -u
K=${1:-1}
run() { # file, first seed, count
  echo "== $1 seeds $2:$(( $2 + $3 * K ))"
  GPANN_FUZZ_SEEDS=$2:$(( $2 + $3 * K )) timeout 600 python3 -m pytest tests/$1 -q -m gpu -n 8 --tb=short 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -20
}
run test_gpu_fuzz.py 96 6900
run test_gpu_fuzz_deferred.py 40 3500
run test_gpu_fuzz_entry_points.py 40 2500
run test_gpu_fuzz_router.py 30 2000
run test_gpu_fuzz_midsize.py 10 1200
