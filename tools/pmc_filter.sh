#!/bin/bash
# Ad-hoc PMC passes over bench.py for the filter kernel (run via gpurun): prints per-counter sums over the FilterKernelDma dispatches.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_filter; rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-compare --router-budget 20000 "$@" > "$OUT/p$i.json" 2> "$OUT/p$i.err"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_filter/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "FilterKernelCells" in k and ("Li0ELi0" in k or "<0, 0" in k):
                agg[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        for c in agg: print(f"{c:28s} {agg[c]:.4e}  over {n[c]} dispatches")
PY
