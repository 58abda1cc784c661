#!/bin/bash
# What do the product's kernels return when ANOTHER process keeps int8 matrix wavefronts resident on the GPU?  (profiles/r06/NOTES.md §2.)
# tools/microbench/matrix_neighbour — one 128-register wavefront per SIMD on WORKGROUPS compute units (default 128: half the chip, so that the
# product's matrix-core kernels, which need whole compute units, still find room) — runs in the background while parts of the GPU suite run.
#   bash tools/suite_beside_neighbour.sh KIND SECONDS OUTPUT_DIR [pytest arguments ...]      KIND: 0 int8, 1 bf16, 2 no matrix instruction
KIND=${1:-0}; SECS=${2:-240}; OUT=${3:-gpurun_out/r6/suite_beside}; shift 3
cd "$(dirname "$0")/.."
mkdir -p "$OUT"
[ -x tools/microbench/matrix_neighbour ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/matrix_neighbour tools/microbench/matrix_neighbour.hip
./tools/microbench/matrix_neighbour $((SECS + 30)) "$KIND" 256 "${NEIGHBOUR_WGS:-128}" > "$OUT/neighbour_kind$KIND.log" 2>&1 &
nb=$!
sleep 2
timeout "$SECS" python -m pytest "$@" -q -p no:cacheprovider --maxfail=200 > "$OUT/pytest_kind$KIND.log" 2>&1
rc=$?
kill "$nb" 2>/dev/null; wait "$nb" 2>/dev/null
echo "kind $KIND rc=$rc $(tail -1 "$OUT/pytest_kind$KIND.log")" | tee -a "$OUT/summary.txt"
grep -E "^FAILED|^ERROR" "$OUT/pytest_kind$KIND.log" | cut -c1-200 | head -60 >> "$OUT/summary.txt"
