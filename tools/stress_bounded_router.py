#!/usr/bin/env python3
"""Debugging aid (GPU box): the bounded replay (matrix-core distance table, canonical evaluation of the cut-off band and of the minimum
candidates only) against the canonical table (GPANN_ROUTER_BOUND_TABLE=0) on randomly drawn trees, data scales / offsets, duplicates, budgets,
batch sizes and vote counts — Query (min_dist bits, probe orders) and FrequencyQuery (votes).  Both run in this process on the same handle;
every difference is printed.    python tools/stress_bounded_router.py [seconds=240] [seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pygpann as gp
import pyoracle

limit = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = pyoracle.Oracle()
t_end = time.time() + limit
n_cfg = n_calls = n_table = bad = 0
tot = dict(table_band_nodes=0, table_exact=0, table_band_overflow=0, table_queries=0, heap_redo=0)
while time.time() < t_end:
    d = int(rng.choice([4, 8, 16, 24, 32, 48, 64, 96, 100, 128, 160, 200, 256]))
    S = int(rng.integers(2, 41))
    n = int(rng.integers(6000, 50000))
    metric = int(rng.integers(0, 2))
    nc = int(rng.choice([4, 6, 8, 16, 32, 64]))
    min_cluster = int(rng.choice([8, 12, 30, 100]))
    scale = float(10.0 ** rng.uniform(-3, 3))
    offset = (rng.standard_normal(d) * float(rng.choice([0.0, 0.0, 1.0, 30.0]))).astype(np.float32)
    centres = rng.standard_normal((int(rng.integers(3, 64)), d)).astype(np.float32)
    pts = centres[rng.integers(0, len(centres), n)] + float(rng.choice([0.05, 0.35, 1.0])) * rng.standard_normal((n, d)).astype(np.float32)
    if rng.random() < 0.3:   # duplicates: equal keys everywhere
        pts[rng.integers(0, n, n // 3)] = pts[rng.integers(0, n, n // 3)]
    pts = ((pts + offset) * scale).astype(np.float32)
    clusters = [np.flatnonzero(a).astype(np.uint32) for a in (rng.integers(0, S, n)[None, :] == np.arange(S)[:, None])]
    if any(len(c) == 0 for c in clusters):
        continue
    tree = orc.kmtr_train(pts, clusters, num_centroids=nc, min_cluster_size=min_cluster, budget=int(rng.integers(n // 20 + S * nc, n // 2 + S * nc)), metric=metric, nthreads=8)
    f = tree.flat()
    C = int(f["node_cent_first"][-1])
    roots = f["shard_root"]
    sizes = (f["node_cent_first"][roots + 1] - f["node_cent_first"][roots]).astype(np.int64)
    lo_budget = int(sizes.sum() - sizes.min() + 1)
    if C <= lo_budget + 2:
        continue
    nq = int(rng.choice([1, 7, 33, 200, 513]))
    queries = pts[rng.integers(0, n, nq)] + (float(rng.choice([0.0, 0.3, 3.0])) * scale) * rng.standard_normal((nq, d)).astype(np.float32)
    if rng.random() < 0.3:   # queries ON centroids: zero distances
        take = rng.integers(0, C, max(1, nq // 3))
        queries[:len(take)] = f["centroids"].reshape(C, d)[take][:nq]
    queries = np.ascontiguousarray(queries, np.float32)
    router = gp.Router(f, metric, 0)
    n_cfg += 1
    for budget in sorted(set(int(x) for x in [lo_budget, lo_budget + 1, C - 1, C, C + 5, *rng.integers(lo_budget, C + 1, 4)])):
        before = router.stats()
        router.set_option("GPANN_ROUTER_BOUND_TABLE", None)
        p1, m1, _ = router.query_batch(queries, budget)
        st = router.stats()
        used = st["table_blocks"] > before["table_blocks"]
        router.set_option("GPANN_ROUTER_BOUND_TABLE", "0")
        p0, m0, _ = router.query_batch(queries, budget)
        n_calls += 1
        n_table += int(used)
        if not (np.array_equal(m1.view(np.uint32), m0.view(np.uint32)) and np.array_equal(p1, p0)):
            bad += 1
            q = np.flatnonzero((m1.view(np.uint32) != m0.view(np.uint32)).any(axis=1))
            print(f"DIFF Query: d {d} S {S} n {n} metric {metric} nc {nc} min_cluster {min_cluster} scale {scale:.3g} C {C} budget {budget} nq {nq}: {len(q)} queries, first {q[:3]}", flush=True)
        if (used or budget > C) and rng.random() < 0.5:
            votes = int(rng.choice([1, 7, 50, 500, 2048]))
            router.set_option("GPANN_ROUTER_BOUND_TABLE", None)
            a1 = router.frequency_query_batch(queries, budget, votes)
            router.set_option("GPANN_ROUTER_BOUND_TABLE", "0")
            a0 = router.frequency_query_batch(queries, budget, votes)
            n_calls += 1
            if not all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a1[:4], a0[:4])):
                bad += 1
                print(f"DIFF FrequencyQuery: d {d} S {S} n {n} metric {metric} nc {nc} scale {scale:.3g} C {C} budget {budget} nq {nq} votes {votes}", flush=True)
    router.set_option("GPANN_ROUTER_BOUND_TABLE", None)
    st = router.stats()
    for k_ in tot:
        tot[k_] += st[k_]
    assert st["verify_failures"] == 0
    router.close()
print(f"{n_cfg} random trees, {n_calls} compared calls ({n_table} Query calls through the bounded replay), {bad} differences; "
      f"{tot['table_queries']} queries on the bounded path: band {tot['table_band_nodes'] / max(1, tot['table_queries']):.1f} nodes and "
      f"{tot['table_exact'] / max(1, tot['table_queries']):.1f} canonical evaluations per query, {tot['table_band_overflow']} band overflows, {tot['heap_redo']} queries to the real queue")
sys.exit(1 if bad else 0)
