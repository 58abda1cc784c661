#!/bin/bash
# Debugging aid (GPU box): tries to reproduce the router bound path's rare "minima unset" event IN THE CONTEXT IT OCCURRED IN — the test
# suite (a parent pytest process holding a GPU context, DistributedBench children with 2-3 ranks sharing the GPU and RCCL set-up / tear-down
# right before the reference's QueryAttribution driver) — instead of the stand-alone harness, which never showed it.  Every router call runs
# with GPANN_ROUTER_VERIFY + STRICT (conftest) and GPANN_ROUTER_PROBE=1: each bound block is run twice and every bit of the two runs compared.
#   tools/stress_suite_context.sh OUT_DIR PAIR_LOOPS CONCURRENT_PROCS CONCURRENT_RUNS [FULL_SUITES] [LEGACY_MEMSET] [PROBE_PIPELINES, 0 = none]
# LEGACY_MEMSET=1 (round 5): every router handle sets its bound blocks up the way rounds 1-3 did — three runtime memsets + a fill launch
# (GPANN_ROUTER_LEGACY_MEMSET, csrc/router.hip EnqueueBound) instead of one BoundInitKernel.  All three historical events predate the kernel; if
# the event lives in the runtime's memsets it shows in this variant under the probe and nowhere else.
# Evidence of an event: OUT_DIR/router_verify_report.txt (+ .bin dumps), and the failing pytest log.
set -u
OUT=${1:-gpurun_out/r4_stress}
PAIRS=${2:-40}
PROCS=${3:-4}
RUNS=${4:-20}
FULL=${5:-1}
LEGACY=${6:-0}
mkdir -p "$OUT"
if [ "$LEGACY" = 1 ]; then export GPANN_ROUTER_LEGACY_MEMSET=1; echo "(legacy memset set-up of the bound blocks)" > "$OUT/variant.txt"; fi
export GPANN_ROUTER_PROBE=1
export GPANN_VERIFY_REPORT="$PWD/$OUT/router_verify_report.txt"
events=0
t0=$(date +%s)
echo "== phase A: $FULL full -m gpu suite(s) with the probe armed" | tee "$OUT/summary.txt"
for i in $(seq 1 "$FULL"); do
    if ! timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$OUT/full_$i.log" 2>&1; then
        events=$((events + 1))
        echo "full suite $i FAILED (see full_$i.log)" | tee -a "$OUT/summary.txt"
        tail -30 "$OUT/full_$i.log" >> "$OUT/summary.txt"
    fi
    tail -1 "$OUT/full_$i.log" | tee -a "$OUT/summary.txt"
done
echo "== phase B: $PAIRS x {test_gpu_distributed.py, test_gpu_dropin.py} in one pytest process each" | tee -a "$OUT/summary.txt"
for i in $(seq 1 "$PAIRS"); do
    if ! timeout 600 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_dropin.py -m gpu -x -q -p no:cacheprovider > "$OUT/pair.log" 2>&1; then
        events=$((events + 1))
        cp "$OUT/pair.log" "$OUT/pair_failed_$i.log"
        echo "pair loop $i FAILED (pair_failed_$i.log)" | tee -a "$OUT/summary.txt"
        tail -30 "$OUT/pair.log" >> "$OUT/summary.txt"
    fi
done
tail -1 "$OUT/pair.log" | tee -a "$OUT/summary.txt"
echo "phase B done after $(( $(date +%s) - t0 )) s, events so far: $events" | tee -a "$OUT/summary.txt"
echo "== phase C: $PROCS concurrent loops of $RUNS QueryAttribution driver runs each (processes sharing the GPU)" | tee -a "$OUT/summary.txt"
pids=()
for p in $(seq 1 "$PROCS"); do
    python tools/stress_qa_driver.py "$RUNS" > "$OUT/qa_conc_$p.log" 2>&1 &
    pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid"; done
for p in $(seq 1 "$PROCS"); do
    tail -1 "$OUT/qa_conc_$p.log" | tee -a "$OUT/summary.txt"
    grep -c "router report\|failed:" "$OUT/qa_conc_$p.log" >> "$OUT/summary.txt"
done
# phase D (round 5): the overlapped pipeline at 64 dimensions, thousands of times (tools/overlap_race_probe.py; profiles/r05/overlap_race.md) — and
# the reproducer of the int8 matrix instruction's effect on co-resident wavefronts, if it has been built
PROBE=${7:-4000}
if [ "$PROBE" != 0 ]; then
    echo "== phase D: overlap probe, $PROBE pipelines at d = 64 (one call ahead) and d = 128" | tee -a "$OUT/summary.txt"
    timeout 900 python3 tools/overlap_race_probe.py --d 64 --depth 1 --rest-items 0 --repeats "$PROBE" > "$OUT/probe_d64.log" 2>&1; tail -1 "$OUT/probe_d64.log" | tee -a "$OUT/summary.txt"
    timeout 900 python3 tools/overlap_race_probe.py --d 128 --depth 1 --rest-items 0 --repeats "$PROBE" > "$OUT/probe_d128.log" 2>&1; tail -1 "$OUT/probe_d128.log" | tee -a "$OUT/summary.txt"
    if [ -x tools/microbench/coresidency_repro ]; then (cd tools/microbench && timeout 300 ./coresidency_repro 5) > "$OUT/coresidency_repro.txt" 2>&1; grep -v "^  config" "$OUT/coresidency_repro.txt" | tee -a "$OUT/summary.txt"; fi
fi
echo "total wall $(( $(date +%s) - t0 )) s; suite-level events: $events; report file: $(test -s "$GPANN_VERIFY_REPORT" && wc -l < "$GPANN_VERIFY_REPORT" || echo 0) lines" | tee -a "$OUT/summary.txt"
ls "$OUT" | grep -c "\.bin$" | sed 's/^/binary dumps: /' | tee -a "$OUT/summary.txt"
exit 0
