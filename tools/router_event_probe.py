#!/usr/bin/env python3
"""Router batches (budget >= tree: RouterBoundKernel passes 1 + 2, the path whose self-check fired three times in rounds 1-3) in a loop for SECONDS,
every batch compared with the CPU oracle's min_dist / probes, events counted instead of ending the run:
  batches, batches that differ from the oracle, calls that returned an error (by message), and the handles' counters (bound_selfcheck = the "minima
  unset" self-check that sends a block to the canonical re-evaluation — the event of rounds 1-3).
ROUTER_EVENT_MODE=verify arms GPANN_ROUTER_VERIFY + GPANN_ROUTER_PROBE as well (default: neither — the verifier's own canonical kernel is a plain fp32
tile kernel and, beside another process's int8 matrix wavefronts, turned out to be a victim itself: profiles/r06/router_event.md).
Run by tools/router_event_experiment.sh beside tools/microbench/matrix_neighbour in another process, with the shipped library and with the
measurement build whose matrix-core kernels leave room on their SIMDs (GPANN_HIP_LIBRARY).
    python3 tools/router_event_probe.py SECONDS [d]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
for name in ("GPANN_ROUTER_VERIFY", "GPANN_ROUTER_PROBE", "GPANN_ROUTER_STRICT"):
    os.environ.pop(name, None)
if os.environ.get("ROUTER_EVENT_MODE") == "verify":      # (the handles snapshot the environment when they are created)
    os.environ["GPANN_ROUTER_VERIFY"] = "1"
    os.environ["GPANN_ROUTER_PROBE"] = "1"
import pygpann as gp  # noqa: E402
import pyoracle  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
orc = pyoracle.Oracle()
n, nq, S = 40000, 2000, 5
pts = orc.synth_rows(555, 7, 32, 0.35, 0, n, d, nthreads=8)
queries = orc.synth_rows(556, 7, 32, 0.35, 0, nq, d, nthreads=8)
part, _ = orc.kmeans(pts, S, 555, 0, nthreads=8)
clusters = orc.partition_to_clusters(part)
tree = orc.kmtr_train(pts, clusters, 64, 100, 20000, 0, nthreads=8)
want_p, want_md = tree.query_batch(queries, 1 << 30, nthreads=8)
flat = tree.flat()
total = {"bound_calls": 0, "bound_selfcheck": 0, "verify_failures": 0, "probe_failures": 0, "verified": 0, "probed": 0}
batches = bad = 0
errors = {}
router = None
t0 = time.time()
print(f"[router probe] library {gp.LIB_PATH}; tree of {flat['centroids'].shape[0] if isinstance(flat, dict) and 'centroids' in flat else '?'} centroids, d = {d}, {nq} queries per batch", flush=True)
while time.time() - t0 < seconds:
    if batches % 50 == 0:
        if router is not None:
            st = router.stats()
            for key in total:
                total[key] += st[key]
            router.close()
        router = gp.Router(flat, 0)
    batches += 1
    try:
        p, md, _ = router.query_batch(queries, 1 << 30)
    except gp.GpannError as e:
        key = str(e).split(":")[1].strip()[:60] if ":" in str(e) else str(e)[:60]
        errors[key] = errors.get(key, 0) + 1
        if sum(errors.values()) <= 2:
            print(f"[router probe] batch {batches}: {str(e)[:300]}", flush=True)
        continue
    if not (np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))):
        bad += 1
        if bad <= 3:
            q = np.flatnonzero((md.view(np.uint32) != want_md.view(np.uint32)).any(axis=1))
            print(f"[router probe] batch {batches}: {len(q)} queries' min_dist differ from the oracle's, e.g. query {q[:3].tolist()}", flush=True)
st = router.stats()
for key in total:
    total[key] += st[key]
print(f"[router probe] {batches} batches in {time.time() - t0:.1f} s: {bad} batches differ from the oracle, {sum(errors.values())} calls failed {errors}; counters {total}", flush=True)
