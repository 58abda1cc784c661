#!/bin/bash
# repeats the QueryAttribution drop-in test; prints pass/fail counts per scan mode (debugging aid)
N=${1:-20}
for mode in default exact; do
  pass=0; fail=0
  for i in $(seq 1 $N); do
    if [ "$mode" = exact ]; then export GPANN_SCAN_MODE=exact; else unset GPANN_SCAN_MODE; fi
    if python -m pytest tests/test_gpu_dropin.py -x -q -k query_attribution < /dev/null > gpurun_out/stress_last.txt 2>&1; then pass=$((pass+1)); else fail=$((fail+1)); grep -n "AssertionError: budget" gpurun_out/stress_last.txt | head -1 | cut -c1-700; fi
  done
  echo "mode $mode: pass $pass fail $fail"
done
