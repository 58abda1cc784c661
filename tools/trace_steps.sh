#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 kernel trace of a short bench.py run, one batch at a time (GPANN_BENCH_ROUTE_AHEAD=0), for
# tools/step_timeline.py.   gpurun -- 'bash tools/trace_steps.sh <tag> [bench args]'  ->  gpurun_out/trace_<tag>/{timeline.md, stats.csv, bench.json}
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export GPANN_BENCH_ROUTE_AHEAD=${GPANN_BENCH_ROUTE_AHEAD:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --router-budget 20000 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
TRACE=$(find "$OUT/raw" -name '*kernel_trace.csv' | head -1)
STATS=$(find "$OUT/raw" -name '*kernel_stats.csv' | head -1)
python3 tools/step_timeline.py "$TRACE" 2 > "$OUT/timeline.md" 2>> "$OUT/bench.err"
[ -n "$STATS" ] && cp "$STATS" "$OUT/stats.csv"
rm -rf "$OUT/raw"
head -70 "$OUT/timeline.md"
tail -3 "$OUT/bench.err"
