set -u
python -m pytest tests/test_gpu_router.py tests/test_gpu_golden.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
bash tools/trace_steps.sh ns8 > /dev/null 2>&1; sed -n 4,12p gpurun_out/trace_ns8/timeline.md
