set -u
python -m pytest tests/test_gpu_filter.py tests/test_gpu_cells.py tests/test_gpu_merge_bound.py tests/test_gpu_threads.py tests/test_gpu_bench_line.py -m gpu -x -q -p no:cacheprovider -s 2>&1 | grep -E "passed|failed|calls/s|Error|assert" | tail -8
bash tools/trace_steps.sh ns7 > /dev/null 2>&1; sed -n 14,24p gpurun_out/trace_ns7/timeline.md; grep PassSetup gpurun_out/trace_ns7/timeline.md | head -3
bash tools/_run14.sh 2>&1 | grep -E "CellDist|PassSetup|step 0|emulated world" | head -8
