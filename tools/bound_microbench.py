#!/usr/bin/env python3
"""RouterBoundKernel alone: 10 000 queries against a flat tree of S roots (visit-everything budget), min_dist checked against numpy on a
sample; prints the bound launches' HIP-event time and the matrix-core rate (gpann_router_stats.last_bound_ms / last_bound_flops).
    python tools/bound_microbench.py [C=19200] [d=128] [S=40] [nq=10000] [metric=0]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gp-ann_amd"))
import pygpann as gp

C_, d, S, nq, metric = [int(x) for x in (sys.argv[1:] + [None] * 5)[:5] if x is not None] + [19200, 128, 40, 10000, 0][len(sys.argv) - 1:]
rng = np.random.default_rng(7)
centers = rng.standard_normal((S, d)).astype(np.float32) * 2
per = C_ // S
cent = (np.repeat(centers, per, axis=0) + rng.standard_normal((S * per, d)).astype(np.float32) * 0.5).astype(np.float32)
flat = dict(S=S, d=d, node_shard=np.arange(S, dtype=np.int32), node_cent_first=(np.arange(S + 1) * per).astype(np.uint32),
            node_child_first=np.zeros(S, np.uint32), node_child_count=np.zeros(S, np.uint32), shard_root=np.arange(S, dtype=np.uint32), centroids=cent)
router = gp.Router(flat, metric, 0)
queries = (centers[rng.integers(0, S, nq)] + rng.standard_normal((nq, d)).astype(np.float32) * 0.7).astype(np.float32)
budget = S * per + 1
probes, md, _ = router.query_batch(queries, budget)
# spot check against numpy (float64 reference of the minimiser; the product's value is the canonical float of that pair)
for q in range(0, nq, 997):
    dist = ((cent.astype(np.float64) - queries[q].astype(np.float64)) ** 2).sum(1) if metric == 0 else 1.0 - cent.astype(np.float64) @ queries[q].astype(np.float64)
    want = dist.reshape(S, per).min(1)
    assert np.allclose(md[q], want, rtol=1e-4, atol=1e-4), (q, md[q][:4], want[:4])
ms = []
for _ in range(8):
    router.query_batch(queries, budget)
    st = router.stats()
    ms.append(st["last_bound_ms"])
ms = np.array(ms[2:])
print(f"C {S * per} d {d} S {S} nq {nq} metric {metric}: bound launches {ms.mean():.3f} ms (min {ms.min():.3f}), "
      f"{st['last_bound_flops'] / ms.mean() / 1e9:.1f} TFLOP/s = {st['last_bound_flops'] / ms.mean() / 1e9 / 2500:.3f} of the bf16 peak; selfcheck {st['bound_selfcheck']}")
