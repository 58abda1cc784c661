#!/usr/bin/env python3
"""GPU-side timeline of the last timed steps of a bench.py run from a rocprofv3 --kernel-trace CSV: per kernel of a step its start offset,
duration and the idle gap before it (the gaps are what the host side — Python, ctypes, plan building, synchronous copies — costs).

usage: python tools/step_timeline.py <kernel_trace.csv> [steps=3] [anchor-kernel-substring=BoundInitKernel]
A step starts at the first kernel of the router (the anchor) and ends before the next anchor."""
import csv
import re
import sys

path = sys.argv[1]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
anchor = sys.argv[3] if len(sys.argv) > 3 else "BoundInitKernel"   # (rounds 1-3: FillMinBitsKernel)
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if len(starts) < n_steps + 1:
    sys.exit(f"only {len(starts)} anchors ({anchor}) in the trace")
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("gpann::", ""))[:60]
# steps = between consecutive anchors; the exact-mode comparison / other work after the timed region is cut by taking steps before the last anchors
chosen = starts[-(n_steps + 2):-1]
for si in range(len(chosen) - 1):
    seg = rows[chosen[si]:chosen[si + 1]]
    t0 = int(seg[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    wall = int(rows[chosen[si + 1]]["Start_Timestamp"]) - t0
    print(f"## step {si}: {len(seg)} kernels, GPU busy {busy / 1e6:.3f} ms, anchor-to-anchor {wall / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms")
    print("| # | kernel | start ms | dur ms | gap before ms |")
    print("|---|---|---|---|---|")
    prev_end = t0
    for i, r in enumerate(seg):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e6
        print(f"| {i} | `{short(r['Kernel_Name'])}` | {(s - t0) / 1e6:.3f} | {(e - s) / 1e6:.3f} | {gap:.3f} |")
        prev_end = max(prev_end, e)
    tail = (int(rows[chosen[si + 1]]["Start_Timestamp"]) - prev_end) / 1e6
    print(f"| | (until the next step's first kernel) | | | {tail:.3f} |")
    print()
