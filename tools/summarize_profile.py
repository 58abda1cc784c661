#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh) into profiles/<round>/<tag>_*.{md,csv,json}.

usage: python tools/summarize_profile.py <tag> <round-dir>      e.g.  python tools/summarize_profile.py v1 profiles/r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, outdir = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
os.makedirs(outdir, exist_ok=True)


def one(pattern):
    """the NEWEST match: gpurun merges a call's files into gpurun_out/, so a re-run of the same tag leaves the older pass's files beside the new ones"""
    g = glob.glob(pattern, recursive=True)
    return max(g, key=os.path.getmtime) if g else None


lines = [f"# rocprofv3 summary `{tag}`", "",
         "Command (per pass): `rocprofv3 <pass flags> --output-format csv -- python3 bench.py --steps K --warmup 1 --no-cpu-baseline`",
         "(passes: `--kernel-trace --stats`; `--pmc FETCH_SIZE`; `--pmc WRITE_SIZE`; SQ counters — each in its own run).", ""]

bench = json.loads(open(f"{src}/stats.json").read().strip().splitlines()[-1])
lines += ["## bench.py line of the kernel-trace pass", "", "```json", json.dumps(bench), "```", ""]

ks = one(f"{src}/stats/**/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, f"{outdir}/{tag}_kernel_stats.csv")
    rows = list(csv.DictReader(open(ks)))
    lines += ["## --stats per-kernel summary (top 8 by total time)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in rows[:8]:
        lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | {float(r['MaxNs'])/1e6:.3f} | {r['Percentage']} |")
    lines.append("")

kt = one(f"{src}/stats/**/*_kernel_trace.csv")
hot = {}
if kt:
    rows = list(csv.DictReader(open(kt)))
    per = collections.defaultdict(list)
    meta = {}
    for r in rows:
        if "gpann::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0]
            per[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            meta[name] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size_X"], r["Workgroup_Size_X"])
    lines += ["## per-dispatch durations of the library's kernels (kernel-trace)", "",
              "| kernel | dispatches | avg ms | durations ms | VGPR/AGPR/SGPR | LDS B | scratch B | last grid x wg |", "|---|---|---|---|---|---|---|---|"]
    for name, v in per.items():
        m = meta[name]
        shown = ", ".join(f"{x:.3f}" for x in v[:8]) + (" ..." if len(v) > 8 else "")
        lines.append(f"| `{name}` | {len(v)} | {sum(v)/len(v):.3f} | {shown} | {m[0]}/{m[1]}/{m[2]} | {m[3]} | {m[4]} | {m[5]} x {m[6]} |")
        hot[name] = sum(v) / len(v)
    lines.append("")

traffic = {}
for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = one(f"{src}/{sub}/**/*_counter_collection.csv")
    if not f:
        continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname and "gpann::" in r["Kernel_Name"]:
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    traffic[cname] = {k: v for k, v in vals.items()}
if traffic:
    lines += ["## HBM traffic of the hot scan kernel (PMC, per launch)", "",
              "FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md §HBM, on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide",
              "coalesced 16-B-per-lane stream, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.", "",
              "(A kernel that runs several passes of a step — `FilterKernelCells<.., 2, ..>`: sample and rest pass — has launches of very different",
              "sizes: the mean mixes them, the largest launch is the rest pass, which is what `bench.py` reports as `roofline.traffic`.)", "",
              "| kernel | launches | FETCH_SIZE KiB (each) | WRITE_SIZE KiB (each) | corrected HBM bytes / launch (mean) | largest launch |", "|---|---|---|---|---|---|"]
    names = set(traffic.get("FETCH_SIZE", {})) | set(traffic.get("WRITE_SIZE", {}))
    tj = {}
    for name in sorted(names):
        fv = traffic.get("FETCH_SIZE", {}).get(name, [])
        wv = traffic.get("WRITE_SIZE", {}).get(name, [])
        fb = 2 * 1024 * (sum(fv) / len(fv)) if fv else 0
        wb = 1024 * (sum(wv) / len(wv)) if wv else 0
        big = max((2048.0 * f + 1024.0 * w for f, w in zip(fv, wv)), default=0.0)
        lines.append(f"| `{name}` | {len(fv)} | {', '.join(f'{x:.0f}' for x in fv[:6])} | {', '.join(f'{x:.0f}' for x in wv[:6])} | {fb + wb:.4g} | {big:.4g} |")
        tj[name] = {"fetch_kib": fv, "write_kib": wv, "hbm_bytes_per_launch": fb + wb, "hbm_bytes_largest_launch": big, "launches": len(fv),
                    "hbm_bytes_all_launches": 2 * 1024 * sum(fv) + 1024 * sum(wv)}
    lines.append("")
    # the bytes the dominant launch of the FETCH pass's own run had to read: bench.py only reports this file's traffic for runs whose
    # dominant launch reads the same bytes (within 2 %)
    dom = None
    try:
        dom = json.loads(open(f"{src}/fetch.json").read().strip().splitlines()[-1])["roofline"].get("kernel_bytes")
    except Exception:
        dom = bench.get("roofline", {}).get("kernel_bytes")
    json.dump({"workload": bench["config"], "dominant_kernel_bytes": dom, "kernels": tj}, open(f"{outdir}/{tag}_pmc_traffic.json", "w"), indent=1)

f = one(f"{src}/sq/**/*_counter_collection.csv")
if f:
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if any(x in r["Kernel_Name"] for x in ("gpann::ScanTileKernel", "gpann::RouterWalk", "gpann::FilterKernel", "gpann::DenseDist")):
            vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines += ["## SQ counters (last launch of each kernel)", ""]
    for name, cs in vals.items():
        lines.append(f"* `{name}`: " + ", ".join(f"{c}={v[-1]:.4g}" for c, v in sorted(cs.items())))
    lines.append("")

open(f"{outdir}/{tag}_rocprof_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
