"""Shard synthesis at scale (GPU box): gpann_kmeans — RandomSample + 20 Lloyd rounds, the step every k-means partitioner of the
reference starts with (partition.cpp:23-26) — over a device-resident synthetic point set, beside the CPU oracle's k-means on a
bounded sample of the same rows.   python tools/partition_bench.py [--n 10000000 --d 96 --k 16]   -> one JSON line"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=96)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--cpu-sample", type=int, default=400_000)
    a = ap.parse_args()
    import torch
    import pygpann as gp
    import pyoracle

    pts = torch.empty((a.n, a.d), dtype=torch.float32, device="cuda:0")
    gp.synth_rows_device(pts.data_ptr(), 555, 7, 1024, 0.35, 0, a.n, a.d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    store = gp.Shards(pts.data_ptr(), [np.arange(a.n, dtype=np.uint32)], n=a.n, d=a.d, plain=True)
    t_store = time.perf_counter() - t0
    store.kmeans(a.k, 555)   # warm-up (allocations)
    t0 = time.perf_counter()
    part, cent = store.kmeans(a.k, 555)
    t_gpu = time.perf_counter() - t0
    sizes = np.bincount(part, minlength=len(cent))

    orc = pyoracle.Oracle()
    m = min(a.cpu_sample, a.n)
    sample = pts[:m].cpu().numpy()
    threads = os.cpu_count() or 1
    t0 = time.perf_counter()
    op, oc = orc.kmeans(sample, a.k, 555, 0, nthreads=threads)
    t_cpu = time.perf_counter() - t0
    # same engine check on the sample: the GPU run over exactly these rows must give the oracle's clusters
    s2 = gp.Shards(sample, [np.arange(m, dtype=np.uint32)], plain=True)
    gp2, gc2 = s2.kmeans(a.k, 555)
    same = bool(np.array_equal(gp2, op) and np.array_equal(gc2.view(np.uint32), oc.view(np.uint32)))
    print(json.dumps({
        "workload": f"RandomSample + KMeans (20 rounds), {a.n} x {a.d} f32, k = {a.k}",
        "gpu_seconds": t_gpu, "gpu_store_seconds": t_store, "gpu_point_rounds_per_s": a.n * 20 / t_gpu,
        "clusters": int(len(cent)), "max_cluster": int(sizes.max()), "min_cluster": int(sizes.min()),
        "cpu_sample_rows": m, "cpu_threads": threads, "cpu_seconds": t_cpu, "cpu_point_rounds_per_s": m * 20 / t_cpu,
        "speedup_per_point_round": (a.n * 20 / t_gpu) / (m * 20 / t_cpu), "sample_identical_to_oracle": same}))


if __name__ == "__main__":
    main()
