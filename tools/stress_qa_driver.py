"""Debugging aid (GPU box): runs oracle/_ref/QueryAttribution (the reference's driver over the host mirror) N times on the drop-in
test's data and reports every run whose probe lists differ from the first run's (headers carry timings and are ignored)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle

QA = os.path.join(ROOT, "oracle", "_ref", "QueryAttribution")
orc = pyoracle.Oracle()
n, d, nq, S, k = 20000, 32, 200, 5, 10
pts = orc.synth_rows(555, 7, 32, 0.35, 0, n, d, nthreads=8)
queries = orc.synth_rows(556, 7, 32, 0.35, 0, nq, d, nthreads=8)
part, _ = orc.kmeans(pts, S, 555, 0, nthreads=8)
clusters = orc.partition_to_clusters(part)
tmp = tempfile.mkdtemp()


def fbin(path, a):
    with open(path, "wb") as f:
        np.array(a.shape, np.uint32).tofile(f)
        np.ascontiguousarray(a, np.float32).tofile(f)


fbin(tmp + "/base.fbin", pts)
fbin(tmp + "/query.fbin", queries)
with open(tmp + "/base.partition.k=5.KMeans", "w") as f:
    for c in clusters:
        f.write(" ".join(str(int(x)) for x in c) + " \n")


def probe_lines(path):
    return [ln for ln in open(path).read().split("\n") if ln and ln[0].isdigit() and " " in ln and not any(ch.isalpha() for ch in ln)]


first, bad = None, 0
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    out = f"{tmp}/qa{i}"
    r = subprocess.run([QA, tmp + "/base.fbin", tmp + "/query.fbin", f"{tmp}/gt{i}.bin", str(k), tmp + "/base.partition.k=5.KMeans", out, "KMeans", str(S)],
                       capture_output=True, text=True, env=dict(os.environ, GPANN_MAX_ROUTER_BUDGET="100000", GPANN_ROUTER_VERIFY="1", GPANN_ROUTER_STRICT="1"))
    if r.returncode != 0:
        print("run", i, "failed:", r.stderr[-500:])
        bad += 1
        continue
    if any(w in r.stderr for w in ("VERIFY", "minima unset", "EVIDENCE", "PROBE", "self-check")):
        bad += 1
        print("run", i, "router report:", r.stderr[-3000:], flush=True)
    lines = probe_lines(out + ".routes")
    if first is None:
        first = lines
    elif lines != first:
        bad += 1
        diff = sum(1 for a, b in zip(lines, first) if a != b)
        print(f"run {i}: {diff} of {len(first)} probe lines differ from run 0; stderr tail: {r.stderr[-400:]}", flush=True)
print("runs differing from the first:", bad)
