#!/usr/bin/env python3
"""Timeline of OVERLAPPED steps from a rocprofv3 --kernel-trace CSV: which launches of which queue ran beside the rest pass.

usage: python tools/overlap_timeline.py <kernel_trace.csv> [periods=2] [rest-kernel-substring]
A period runs from the start of one rest-pass launch (the longest FilterKernelCells instantiation in the trace, or the given substring) to the start
of the next; the periods before the last `skip_tail` rest passes are shown (the timed loop, not the checks after it).  Per launch: its queue, start
offset in the period, duration.  Below each table: how long 1, 2, 3+ launches were in flight, per-queue busy time."""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
n_periods = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [r for r in csv.DictReader(open(path))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qcol = next((c for c in ("Queue_Id", "Stream_Id", "Queue_ID") if c in rows[0]), None)
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("gpann::", ""))[:52]
if len(sys.argv) > 3:
    key = sys.argv[3]
else:   # the FilterKernelCells instantiation with the largest total time = the rest pass
    tot = defaultdict(int)
    for r in rows:
        if "FilterKernelCells" in r["Kernel_Name"]:
            tot[r["Kernel_Name"]] += r["e"] - r["s"]
    key = max(tot, key=tot.get)
rest = [r for r in rows if key in r["Kernel_Name"]]
# the timed loop: the longest run of rest passes whose starts are less than 3 median periods apart
gaps = sorted(b["s"] - a["s"] for a, b in zip(rest, rest[1:]))
med = gaps[len(gaps) // 2]
runs, cur = [], [rest[0]]
for a, b in zip(rest, rest[1:]):
    if b["s"] - a["s"] < 2.5 * med:
        cur.append(b)
    else:
        runs.append(cur)
        cur = [b]
runs.append(cur)
loop = max(runs, key=len)
print(f"rest pass = `{short(key)}`; {len(rest)} launches, the longest back-to-back run has {len(loop)}; median start-to-start {med / 1e6:.3f} ms; queue column: {qcol}")
durs = sorted(r["e"] - r["s"] for r in loop)
print(f"rest-pass duration in that run: median {durs[len(durs) // 2] / 1e6:.3f} ms, min {durs[0] / 1e6:.3f}, max {durs[-1] / 1e6:.3f}; "
      f"gap between one rest pass's end and the next one's start: median {sorted(b['s'] - a['e'] for a, b in zip(loop, loop[1:]))[len(loop) // 2 - 1] / 1e6:.3f} ms\n")
mid = len(loop) // 2
for p in range(n_periods):
    a, b = loop[mid + p], loop[mid + p + 1]
    t0, t1 = a["s"], b["s"]
    seg = [r for r in rows if r["e"] > t0 and r["s"] < t1]
    print(f"## period {p}: {(t1 - t0) / 1e6:.3f} ms from rest-pass start to rest-pass start, {len(seg)} launches in flight during it")
    print("| queue | kernel | start ms | dur ms |")
    print("|---|---|---|---|")
    for r in seg:
        print(f"| {r[qcol] if qcol else '-'} | `{short(r['Kernel_Name'])}` | {(r['s'] - t0) / 1e6:.3f} | {(r['e'] - r['s']) / 1e6:.3f} |")
    # concurrency profile
    ev = []
    for r in seg:
        ev.append((max(r["s"], t0), 1))
        ev.append((min(r["e"], t1), -1))
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        last = t
        depth += d
    hist[min(depth, 3)] += t1 - last
    busy = defaultdict(int)
    for r in seg:
        busy[r[qcol] if qcol else "-"] += min(r["e"], t1) - max(r["s"], t0)
    print(f"\nlaunches in flight: none {hist[0] / 1e6:.3f} ms, one {hist[1] / 1e6:.3f}, two {hist[2] / 1e6:.3f}, three or more {hist[3] / 1e6:.3f}; "
          "busy per queue: " + ", ".join(f"{q}: {v / 1e6:.3f}" for q, v in sorted(busy.items())) + "\n")
