#!/bin/bash
# Starts gp-ann_amd/bin/RouterHarness N times (a fresh process each time) and counts the runs that did not end clean.
#   tools/run_router_harness.sh <N> <out_prefix> <harness args...>
# Writes <out_prefix>.summary and, per failing run, <out_prefix>.fail<i>.txt (stderr incl. the library's VERIFY report).
N=$1; OUT=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BIN="$ROOT/gp-ann_amd/bin/RouterHarness"
ok=0; bad=0; t0=$(date +%s)
for i in $(seq 1 "$N"); do
    args=("${@/SEED/$i}")   # an argument spelled SEED becomes the run number
    if timeout 120 "$BIN" "${args[@]}" >"$OUT.last.out" 2>"$OUT.last.err"; then
        ok=$((ok + 1))
    else
        rc=$?
        bad=$((bad + 1))
        { echo "run $i rc $rc"; cat "$OUT.last.out" "$OUT.last.err"; } >"$OUT.fail$bad.txt"
    fi
done
t1=$(date +%s)
echo "harness args: $*  runs: $N  clean: $ok  not clean: $bad  seconds: $((t1 - t0))  last: $(tail -1 "$OUT.last.out")" | tee "$OUT.summary"
