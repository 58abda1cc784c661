#!/usr/bin/env python3
"""Regenerates the committed golden fixtures in tests/golden/ (run in the build container, where /root/reference exists):

  ref_topn.npz, ref_partition.npz, ref_clusters.txt(+.out)  — outputs of the REFERENCE's own code (src/topn.h, src/defs.cpp,
      src/metis_io.cpp) through oracle/_ref/ref_core: they pin the oracle.
  hotpath_*.npz — small seeded inputs of the route + scan path with the strict oracle's outputs: GPU parity tests replay them
      (the oracle itself is "parity unpinned" for these).

usage: python tests/golden/make_golden.py
"""
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import pyoracle  # noqa: E402
from ref_core import RefCore  # noqa: E402


def main():
    pyoracle.build()
    ref = RefCore()
    rng = np.random.default_rng(20241020)
    # ---- TopN: ties, fewer items than n, n = 1, duplicates
    cases = []
    for n_items, n, levels in ((50, 10, 4), (7, 10, 3), (200, 1, 5), (300, 64, 9), (40, 40, 2), (128, 13, 100000)):
        d = (rng.integers(0, levels, n_items) / np.float32(levels)).astype(np.float32)
        ids = rng.integers(0, 60, n_items).astype(np.uint32)
        cases.append((d, ids, n))
    asc = ref.topn(cases)
    desc = ref.topn(cases, descending=True)
    np.savez(os.path.join(HERE, "ref_topn.npz"), n_cases=len(cases),
             **{f"d{i}": c[0] for i, c in enumerate(cases)}, **{f"i{i}": c[1] for i, c in enumerate(cases)}, **{f"n{i}": c[2] for i, c in enumerate(cases)},
             **{f"ad{i}": a[0] for i, a in enumerate(asc)}, **{f"ai{i}": a[1] for i, a in enumerate(asc)},
             **{f"dd{i}": a[0] for i, a in enumerate(desc)}, **{f"di{i}": a[1] for i, a in enumerate(desc)})
    # ---- partitions (with an empty shard id -> RemapPartitionIDs compacts)
    part = rng.integers(0, 6, 400).astype(np.int32)
    part[part == 2] = 3
    res = ref.partition(part)
    np.savez(os.path.join(HERE, "ref_partition.npz"), partition=part, num_parts=res["num_parts"], cluster_offsets=res["offsets"],
             cluster_ids=res["ids"], cover_sizes=res["cover_sizes"], cover_flat=res["cover_flat"], remapped=res["remapped"])
    # ---- clusters text format round trip through ReadClusters / WriteClusters
    txt = os.path.join(HERE, "ref_clusters.txt")
    with open(txt, "w") as f:
        f.write("3 1 4 1 5 \n\n9 2 6\n5 3 5 8 9 7 9 \n")
    ref.run("clusters_roundtrip", txt, txt + ".out")

    # ---- hot-path vectors from the strict oracle
    orc = pyoracle.Oracle("strict")
    for name, n, d, metric in (("l2_d128", 3000, 128, 0), ("l2_d100", 2500, 100, 0), ("ip_d200", 2000, 200, 1), ("l2_d7", 1500, 7, 0)):
        pts = orc.synth_rows(555, 7, 16, 0.35, 0, n, d)
        pts[10:40] = pts[10]                                   # distance ties
        queries = orc.synth_rows(556, 7, 16, 0.35, 0, 24, d)
        queries[0] = pts[10]
        S, k = 4, 10
        part = rng.integers(0, S, n)
        clusters = [rng.permutation(np.flatnonzero(part == b)).astype(np.uint32) for b in range(S)]
        clusters[1] = np.concatenate([clusters[1], clusters[0][:50]])   # overlap
        ivf = orc.ivf_build(pts, clusters)
        probes = np.stack([rng.permutation(S) for _ in range(len(queries))]).astype(np.int32)
        sd, si, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, 3, metric)
        gd, gi = orc.ground_truth(pts, queries, k, metric)
        tree = orc.kmtr_train(pts, clusters, 8, 30, 400, metric)
        flat = tree.flat()
        rp, rm = tree.query_batch(queries, 150)
        rp0, rm0 = tree.query_batch(queries, 20)               # too small to open every root
        offs = np.concatenate([[0], np.cumsum([len(c) for c in clusters])]).astype(np.uint64)
        np.savez_compressed(os.path.join(HERE, f"hotpath_{name}.npz"), metric=metric, seed_note="synth seed 555/556, centres 16, sigma 0.35",
                            n=n, d=d, k=k, cluster_ids=np.concatenate(clusters), cluster_offsets=offs, probes=probes, tie_rows=np.array([10, 40]),
                            scan_dist=sd, scan_ids=si, gt_dist=gd, gt_ids=gi, router_budget=np.array([150, 20]), router_probes=rp,
                            router_min_dist=rm, router_probes_small=rp0, router_min_dist_small=rm0,
                            **{"tree_" + key: np.asarray(val) for key, val in flat.items()})
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
