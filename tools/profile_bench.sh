#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace/stats pass + two PMC passes (FETCH_SIZE, WRITE_SIZE each in its
# own run, as MI355X_MICROARCH.md §rocprofv3 PMC slots prescribes) over the default bench.py command.
#   gpurun --timeout 1800 -- 'bash tools/profile_bench.sh <tag> [bench args]'
# Summaries are written by tools/summarize_profile.py (run it in the build container afterwards).
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-exact-compare "$@" > "$OUT/stats.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-compare "$@" > "$OUT/fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-compare "$@" > "$OUT/write.json" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-compare "$@" > "$OUT/sq.json" 2> "$OUT/sq.err"
ls -R "$OUT" | head -40
