"""Debugging aid (GPU box): repeats KMeansTreeRouter::Query batches on a small tree (the drop-in test's shape: many row tiles of the
matrix-core bound straddle shard boundaries) and reports every batch whose min_dist / probes differ from the CPU oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pygpann as gp
import pyoracle

orc = pyoracle.Oracle()
n, d, nq, S = 20000, 32, 200, 5
pts = orc.synth_rows(555, 7, 32, 0.35, 0, n, d, nthreads=8)
queries = orc.synth_rows(556, 7, 32, 0.35, 0, nq, d, nthreads=8)
part, _ = orc.kmeans(pts, S, 555, 0, nthreads=8)
clusters = orc.partition_to_clusters(part)
bad = 0
for budget in (20000, 100000):
    tree = orc.kmtr_train(pts, clusters, 64, 350, budget, 0, nthreads=8)
    want_p, want_md = tree.query_batch(queries, 50000, nthreads=8)
    flat = tree.flat()
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
        router = gp.Router(flat, 0) if rep % 10 == 0 else router
        p, md, _ = router.query_batch(queries, 50000)
        if not (np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))):
            bad += 1
            q = np.flatnonzero((md.view(np.uint32) != want_md.view(np.uint32)).any(axis=1))
            print(f"budget {budget} rep {rep}: {len(q)} queries differ, e.g. q={q[:3]} got {md[q[0]]} want {want_md[q[0]]}", flush=True)
    # and with device training in a fresh store each time
    for rep in range(20):
        shards = gp.Shards(pts, clusters, plain=True)
        dev = gp.Router.train(shards, n, 64, 350, budget)
        p, md, _ = dev.query_batch(queries, 50000)
        if not (np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))):
            bad += 1
            print(f"budget {budget} trained rep {rep}: differs", flush=True)
        dev.close(); shards.close()
print("mismatching batches:", bad)
