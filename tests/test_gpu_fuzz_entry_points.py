"""GPU: randomised small configurations of the entry points beside gpann_scan_buckets — gpann_scan_merged (InvertedIndex::Query, src/inverted_index.h:62-75),
gpann_ground_truth over a point array and over a store (ComputeGroundTruth, src/recall.h:25-39), gpann_shard_search_hits (RunInShardSearches' hit lists,
src/shard_searches.cpp) and gpann_merge_dedup (DedupNeighbors, small_scale_queries.cpp:15-23) — against the CPU oracle: odd dimensions, one query, k larger
than a shard, overlapping shards (ids repeat across probes), empty and three-row shards, skipped probes, both metrics, both scan modes.
Seeds are fixed: a failure names its configuration.  GPANN_FUZZ_SEEDS=a:b replaces the suite's range for a longer run."""
import os

import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu
NO_ID = 0xFFFFFFFF


def _seeds():
    a, _, b = os.environ.get("GPANN_FUZZ_SEEDS", "0:40").partition(":")
    return range(int(a), int(b))


def _bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("seed", _seeds())
def test_random_entry_point_configuration(orc, gp, seed, monkeypatch):
    rng = np.random.default_rng(9000 + seed)
    d = int(rng.choice([3, 17, 40, 64, 96, 100, 128, 200, 260]))
    n = int(rng.integers(200, 12000))
    S = int(rng.integers(1, 8))
    nq = int(rng.choice([1, 2, 33, 100]))
    k = int(rng.choice([1, 5, 10, 33]))
    metric = int(rng.integers(0, 2))
    monkeypatch.setenv("GPANN_CELL_ROWS", str(int(rng.choice([64, 300, 1500]))))
    if rng.integers(0, 2):
        monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)
    pts = make_mixture(orc, n, d, n_centers=int(rng.integers(1, 8)), sigma=float(rng.choice([0.05, 0.35, 1.0])))
    if rng.integers(0, 3) == 0:
        pts += np.float32(rng.normal(0, 30))
    if rng.integers(0, 4) == 0:
        pts[: n // 4] = pts[0]                                     # masses of equal distances: the (distance, id) order decides
    clusters = [np.asarray(c, np.uint32) for c in random_clusters(rng, n, S, overlap=float(rng.choice([0.0, 0.25])))]
    if S > 2:
        clusters[int(rng.integers(0, S))] = np.zeros(0, np.uint32)
        clusters[int(rng.integers(0, S))] = np.arange(3, dtype=np.uint32)
    queries = make_mixture(orc, nq, d, seed=557, n_centers=3)
    queries[0] = pts[0]
    cfg = dict(seed=seed, d=d, n=n, S=S, nq=nq, k=k, metric=metric)

    # ---- ComputeGroundTruth: point array, then the same rows as a one-shard store
    wd, wi = orc.ground_truth(pts, queries, k, metric, nthreads=4)
    ids, dist, _ = gp.ground_truth(pts, queries, k, metric)
    assert np.array_equal(ids, wi) and np.array_equal(_bits(dist), _bits(wd)), ("ground truth", cfg)
    kth = wd[:, min(k, n) - 1].copy()

    ivf = orc.ivf_build(pts, clusters)
    shards = gp.Shards(pts, clusters, metric, 0)
    try:
        # ---- InvertedIndex::Query: one TopN over the probed shards (copies of an id in two shards both stay), both scan modes
        nprobe = int(rng.integers(1, S + 1))
        probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
        for mode in (1, 0):
            shards.set_scan_mode(mode)
            mi, md, _ = shards.scan_merged(queries, k, probes, nprobe)
            for q in range(nq):
                qd, qi = orc.ivf_query(ivf, queries[q], k, probes[q], nprobe, metric)   # (TopN::Take: as many entries as there were rows, at most k)
                c = len(qi)
                assert np.array_equal(mi[q, :c], qi) and np.array_equal(_bits(md[q, :c]), _bits(qd)), ("scan_merged", cfg, mode, q)
                assert (mi[q, c:] == NO_ID).all() and np.isinf(md[q, c:]).all(), ("scan_merged: unused slots", cfg, mode, q)

        # ---- per-slot lists (some probes skipped) -> DedupNeighbors
        sp = probes[:, :nprobe].copy()
        sp[rng.random(sp.shape) < 0.15] = -1
        shards.set_scan_mode(1)
        bi, bd, _ = shards.scan_buckets(queries, k, sp, nprobe)
        od, oi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, sp, nprobe, metric, nthreads=4)
        assert np.array_equal(bi, oi) and np.array_equal(_bits(bd), _bits(od)), ("scan_buckets", cfg)
        gd, gi, gc = gp.merge_dedup(bd, bi, nprobe, k)
        flat_d, flat_i = od.reshape(nq, -1), oi.reshape(nq, -1)
        keep = flat_i != NO_ID
        offs = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.uint64)
        xd, xi, xc = orc.dedup_neighbors(flat_d[keep], flat_i[keep], offs, k)
        assert np.array_equal(gc, xc), ("merge_dedup count", cfg)
        for q in range(nq):
            c = int(xc[q])
            assert np.array_equal(gi[q, :c], xi[q, :c]) and np.array_equal(_bits(gd[q, :c]), _bits(xd[q, :c])), ("merge_dedup", cfg, q)

        # ---- RunInShardSearches' hits: per shard the rows of its top k that are no farther than the query's k-th ground-truth distance
        hits, cnt, _ = shards.shard_search_hits(queries, k, kth)
        allp = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
        ad, ai, _ = orc.ivf_query_bucket_batch(ivf, queries, k, allp, S, metric, nthreads=4)
        for b in range(S):
            for q in range(nq):
                want = [int(ai[q, b, x]) for x in range(k) if ai[q, b, x] != NO_ID and ad[q, b, x] <= kth[q]]
                assert cnt[b, q] == len(want), ("hits count", cfg, b, q)
                assert list(hits[b, q, :len(want)]) == want[::-1], ("hits", cfg, b, q)

        # ---- ground truth over a store (the contract of include/gpann.h: ONE shard with ids 0 .. n-1 in order — ties are then broken by id as ComputeGroundTruth does)
        if seed % 2 == 0:
            store = gp.Shards(pts, [np.arange(n, dtype=np.uint32)], metric, 0)
            try:
                ids2, dist2, _ = gp.ground_truth(None, queries, k, metric, store=store)
                assert np.array_equal(ids2, wi) and np.array_equal(_bits(dist2), _bits(wd)), ("ground truth over a store", cfg)
            finally:
                store.close()
            if seed % 8 == 0:      # a store in another row order is refused (it would break ties by row, not by id), not answered differently
                shuffled = gp.Shards(pts, [rng.permutation(n).astype(np.uint32)], metric, 0)
                try:
                    with pytest.raises(gp.GpannError):
                        gp.ground_truth(None, queries, k, metric, store=shuffled)
                finally:
                    shuffled.close()
    finally:
        shards.close()
