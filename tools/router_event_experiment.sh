#!/bin/bash
# Is RouterBoundKernel's "minima unset" self-check event (3 in ~50 suites of rounds 1-3, never reproduced) what a wavefront looks like that shared a SIMD
# with ANOTHER process's v_mfma_i32_32x32x32_i8 wavefronts?  (profiles/r05/overlap_race.md named that as the first candidate explanation; since round 5
# the kernel takes its SIMDs whole.)  The experiment: the router probe (tools/router_event_probe.py: verify + probe armed, events counted) runs beside
# tools/microbench/matrix_neighbour in another process — one 128-register wavefront per SIMD on every CU, so that two RouterBoundKernel wavefronts of up
# to 192 registers fit beside it — with
# (the probe compares every batch with the CPU oracle; GPANN_ROUTER_VERIFY is NOT armed: its canonical reference kernel is a plain fp32 tile kernel and
# was itself disturbed beside the int8 neighbour in the first run of this experiment — "196 of 10000 minima differ" with the minima the host recomputed
# being the bound path's)
#   * the MEASUREMENT build (make -C gp-ann_amd experiment-shared-simds: the kernel keeps its natural 148 - 164 registers at d <= 64), and
#   * the SHIPPED library (256 registers: its workgroups can only run on compute units without a neighbour wavefront),
# each beside an int8 neighbour, a bf16 neighbour, and alone.  Output: one line per cell (profiles/r06/router_event.md).
#   bash tools/router_event_experiment.sh [seconds per cell] [output directory]
SECS=${1:-25}
OUT=${2:-gpurun_out/r6/router_event}
cd "$(dirname "$0")/.."
mkdir -p "$OUT"
EXP_LIB=gp-ann_amd/lib_experiment/libgpann_hip.so
[ -f "$EXP_LIB" ] || make -C gp-ann_amd -j6 experiment-shared-simds > "$OUT/build.log" 2>&1
[ -x tools/microbench/matrix_neighbour ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/matrix_neighbour tools/microbench/matrix_neighbour.hip
cell() {   # name, library ("" = shipped), neighbour kind ("" = none), d
    local name=$1 lib=$2 kind=$3 d=$4 nb=
    if [ -n "$kind" ]; then
        ./tools/microbench/matrix_neighbour $((SECS + 45)) "$kind" 256 > "$OUT/$name.neighbour.log" 2>&1 &
        nb=$!
        sleep 2
    fi
    GPANN_HIP_LIBRARY=$lib timeout $((SECS + 120)) python3 tools/router_event_probe.py "$SECS" "$d" > "$OUT/$name.log" 2>&1
    local rc=$?
    if [ -n "$nb" ]; then kill "$nb" 2>/dev/null; wait "$nb" 2>/dev/null; fi
    echo "$name rc=$rc $(grep -h 'batches in' "$OUT/$name.log" | tail -1)" | tee -a "$OUT/summary.txt"
}
for d in 32 64; do
    cell "experiment_d${d}_alone"  "$EXP_LIB" ""  $d
    cell "experiment_d${d}_int8"   "$EXP_LIB" "0" $d
    cell "experiment_d${d}_bf16"   "$EXP_LIB" "1" $d
    cell "shipped_d${d}_alone"     ""         ""  $d
    cell "shipped_d${d}_int8"      ""         "0" $d
done
