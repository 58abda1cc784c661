#!/bin/bash
# Ad-hoc PMC passes over bench.py for the filter kernel's largest launch (the rest pass), run via gpurun:
#   gpurun -- 'bash tools/pmc_filter.sh <tag> [bench args]'
# Prints, per counter, the value of the FilterKernelCells dispatch with the most SQ_WAVE_CYCLES-like weight (the longest one) and writes
# gpurun_out/pmc_<tag>.txt.  Counter sets are collected in separate runs (rocprofv3 --pmc only, no tracing).
set -u
TAG=${1:-run}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --router-budget 20000 "$@" > "$OUT/p$i.json" 2> "$OUT/p$i.err"
done
python3 - "$OUT" <<'PY' | tee "gpurun_out/pmc_$TAG.txt"
import csv, glob, collections, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(dict)   # dispatch id -> counter -> value
        names = {}
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "FilterKernelCells" in k:
                per[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = k.split("(")[0]
        if not per:
            continue
        # the rest pass = the dispatch with the largest first counter among the last quarter of the dispatches (timed steps come last)
        ids = sorted(per, key=lambda x: int(x))
        tail = ids[-8:]
        first = sorted(per[tail[0]])[0]
        best = max(tail, key=lambda x: per[x].get(first, 0.0))
        print(f"# {names[best]}  (dispatch {best})")
        for c, v in sorted(per[best].items()):
            print(f"{c:28s} {v:.4e}")
PY
rm -rf "$OUT"/p*/
