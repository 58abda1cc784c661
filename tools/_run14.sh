set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_emul8
rm -rf "$OUT"; mkdir -p "$OUT"
export GPANN_BENCH_EMULATE_RANKS=0
rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-hard-workload --router-budget 20000 --emulate-world 8 > "$OUT/bench.json" 2> "$OUT/bench.err"
TRACE=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py "$TRACE" 2 > "$OUT/timeline.md" 2>> "$OUT/bench.err"
rm -rf "$OUT/raw"
head -60 "$OUT/timeline.md"; grep "emulated world" "$OUT/bench.err"
