#!/bin/bash
# Long-run stress of GPANN_SCAN_OVERLAP's exactness (profiles/r06/NOTES.md §1): tools/overlap_race_probe.py over the configurations in which
# round 5 saw wrong lists before the matrix-core kernels took whole SIMDs — 64 / 32 / 48 dimensions, one and two calls ahead, persistent and leaving
# rest-pass workgroups, L2 and inner product — each for SECONDS (default 120) of back-to-back pipelines, every batch compared bit for bit with the plain call.
#   bash tools/overlap_stress.sh [seconds per configuration] [output directory]
SECS=${1:-120}
OUT=${2:-gpurun_out/r6/overlap_stress}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
run() {   # name, probe arguments...
    local name=$1; shift
    timeout $((SECS + 240)) python3 tools/overlap_race_probe.py --seconds "$SECS" "$@" > "$OUT/$name.log" 2>&1
    echo "$name rc=$? $(grep -h '^\[probe\] [0-9]' "$OUT/$name.log" | tail -1)" | tee -a "$OUT/summary.txt"
}
run d64_depth1_persistent   --d 64 --depth 1 --rest-items 0
run d64_depth2_persistent   --d 64 --depth 2 --rest-items 0
run d64_depth2_leaving      --d 64 --depth 2 --rest-items 2
run d64_depth2_no_sample    --d 64 --depth 2 --rest-items 0 --set GPANN_FILTER_SHARD_SAMPLE=0
run d32_depth2_persistent   --d 32 --depth 2 --rest-items 0
run d48_ip_depth2           --d 48 --depth 2 --rest-items 0 --metric 1
run d96_depth2_persistent   --d 96 --depth 2 --rest-items 0
run d128_depth2_persistent  --d 128 --depth 2 --rest-items 0
