"""GPU: the route -> scan -> merge path end to end at the shapes of BASELINE.json's configurations: C1 100k x 128 /
1k queries / 4 shards, C2 1M x 128 / 10k queries / 8 shards and C3 10M x 96 / 10k queries / 16 shards / nprobe 4 at FULL size; the
multi-GPU configurations C4 (200-dim inner product, 40 shards) and C5 (100-dim, 60 overlapping shards) at their shapes with fewer rows.
Shards are the reference's FlatKMeans partition (gpann_kmeans), the router is trained on the GPU.  Checks: the whole pipeline
against the CPU oracle on a sample of the queries (bit-exact probes, ids, distances), and for ALL queries the size-independent
properties — filter mode == exact mode bit for bit, per-slot lists sorted ascending by (distance, row), merged lists free of
duplicates and sorted, recall against the exact ground truth consistent with the oracle's recall arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mixture_host(orc, n, d, seed, centers=64):
    return orc.synth_rows(seed, 7, centers, 0.35, 0, n, d, nthreads=16)


def _pipeline(gp, shards, router, queries, k, nprobe, budget):
    probes, min_dist, _ = router.query_batch(queries, budget)
    ids, dist, _ = shards.scan_buckets(queries, k, probes[:, :nprobe].copy(), nprobe)
    md, mi, mc = gp.merge_dedup(dist, ids, nprobe, k)
    return probes, min_dist, ids, dist, md, mi, mc


def _check_properties(ids, dist, md, mi, mc, k):
    # per slot: ascending distances, unused slots (+inf, NO_ID) at the end
    assert (np.diff(dist, axis=2) >= 0).all() or np.isinf(dist).any()
    fin = np.isfinite(dist)
    assert ((ids == 0xFFFFFFFF) == ~fin).all()
    d2 = np.where(fin, dist, np.float32(np.inf))
    assert (d2[:, :, 1:] >= d2[:, :, :-1]).all()
    # merged: ascending, no duplicate ids among the first mc entries
    for q in range(0, len(mi), max(1, len(mi) // 500)):
        c = int(mc[q])
        assert len(set(mi[q, :c].tolist())) == c
        assert (np.diff(md[q, :c]) >= 0).all()


def _oracle_pipeline(orc, tree_flat, metric, pts, clusters, queries, k, nprobe, budget):
    tree = orc.kmtr_from_flat(tree_flat, metric)
    probes, min_dist = tree.query_batch(queries, budget, nthreads=16)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes[:, :nprobe].copy(), nprobe, metric, nthreads=16)
    nq = len(queries)
    flat_d, flat_i = wd.reshape(nq, -1), wi.reshape(nq, -1)
    keep = flat_i != 0xFFFFFFFF
    offs = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.uint64)
    md, mi, mc = orc.dedup_neighbors(flat_d[keep], flat_i[keep], offs, k)
    return probes, min_dist, wi, wd, md, mi, mc


def _run_config(orc, gp, n, d, nq, S, k, nprobe, metric, index_budget, search_budget, sample, overlap=0.0, centers=64, filtered=True):
    pts = _mixture_host(orc, n, d, 555, centers)
    queries = _mixture_host(orc, nq, d, 556, centers)
    # shards: FlatKMeans (partition.cpp:23-26) on the device
    plain = gp.Shards(pts, [np.arange(n, dtype=np.uint32)], metric=metric, plain=True)
    part, _ = plain.kmeans(S, 555)
    plain.close()
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(int(part.max()) + 1)]
    if overlap > 0:   # C5: every shard also gets a share of foreign points (ids repeat across shards)
        rng = np.random.default_rng(5)
        extra = rng.choice(n, int(overlap * n), replace=False).astype(np.uint32)
        tgt = rng.integers(0, len(clusters), size=len(extra))
        clusters = [np.concatenate([c, np.setdiff1d(extra[tgt == b], c)]).astype(np.uint32) for b, c in enumerate(clusters)]
    shards = gp.Shards(pts, clusters, metric=metric)
    router = gp.Router.train(shards, n, 64, 350, index_budget)
    shards.set_scan_mode(1)
    got = _pipeline(gp, shards, router, queries, k, nprobe, search_budget)
    st = shards.stats()
    assert bool(st["filtered"]) == filtered
    shards.set_scan_mode(0)
    exact = _pipeline(gp, shards, router, queries, k, nprobe, search_budget)
    for a, b in zip(got, exact):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    _check_properties(*got[2:], k)
    if overlap > 0:
        # overlapping shards: the same id does come back from several probes, and the merge keeps one copy — per query the merged count is
        # the number of DISTINCT ids among its slots (capped at k), which is smaller than the number of slot entries where ids repeat
        ids_all = got[2].reshape(nq, -1)
        srt = np.sort(ids_all, axis=1)
        valid = srt != 0xFFFFFFFF
        distinct = valid[:, 0].astype(np.int64) + np.count_nonzero((srt[:, 1:] != srt[:, :-1]) & valid[:, 1:], axis=1)
        entries = np.count_nonzero(valid, axis=1)
        assert np.array_equal(got[6].astype(np.int64), np.minimum(distinct, k))
        repeated = distinct < entries
        assert repeated.mean() > 0.2, repeated.mean()          # ids repeat for a good share of the queries (20 % of the points live in two shards)
        few = repeated & (distinct < k)
        assert np.all(got[6][few] < np.minimum(entries[few], k))   # ... and where fewer than k distinct ids came back the merged list is SHORTER than the slot entries
    # recall against the exact ground truth (recall.h:139-150 arithmetic: hits = merged distances <= k-th ground-truth distance)
    gt_ids, gt_dist, _ = gp.ground_truth(pts, queries, k, metric)
    kth = gt_dist[:, k - 1]
    recall = float(np.count_nonzero(got[4] <= kth[:, None])) / (nq * k)
    assert 0.0 < recall <= 1.0
    # the oracle on a sample of the queries: the whole pipeline, bit for bit
    sub = np.linspace(0, nq - 1, sample).astype(np.int64)
    want = _oracle_pipeline(orc, router.export(), metric, pts, clusters, queries[sub], k, nprobe, search_budget)
    names = ("probes", "min_dist", "slot ids", "slot dist", "merged dist", "merged ids", "merged count")
    for name, a, b in zip(names, got, want):
        a = a[sub]
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), name
    o_recall = float(np.count_nonzero(want[4] <= kth[sub][:, None])) / (len(sub) * k)
    g_recall = float(np.count_nonzero(got[4][sub] <= kth[sub][:, None])) / (len(sub) * k)
    assert abs(o_recall - g_recall) <= 1e-6
    # ... and the recall over ALL queries is the oracle's recall on its sample up to the sampling error of that sample (4 sigma of a
    # mean of `sample` per-query recalls; the per-query values lie in [0, 1])
    per_query = np.count_nonzero(want[4] <= kth[sub][:, None], axis=1) / k
    sigma = float(per_query.std(ddof=1)) / np.sqrt(len(sub)) if len(sub) > 1 else 1.0
    assert abs(recall - o_recall) <= 4.0 * sigma + 1e-6 or len(sub) == nq, (recall, o_recall, sigma)
    if len(sub) == nq:
        assert abs(recall - o_recall) <= 1e-6
    shards.close()
    router.close()
    return recall


def test_c1_100k_x_128(orc, gp, monkeypatch):
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)   # the shipped defaults for a store of this size (conftest.py switches the small-shard rule off)
    _run_config(orc, gp, 100_000, 128, 1000, 4, 10, 2, 0, 5000, 5000, sample=1000)   # every query against the oracle


def test_c2_sift1m_shape(orc, gp, monkeypatch):
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)   # the shipped defaults: 125k rows per shard = small shards
    r = _run_config(orc, gp, 1_000_000, 128, 10_000, 8, 10, 2, 0, 20_000, 50_000, sample=200)
    assert r > 0.9


def test_c3_deep10m_shape(orc, gp):
    r = _run_config(orc, gp, 10_000_000, 96, 10_000, 16, 10, 4, 0, 20_000, 50_000, sample=48)
    assert r > 0.9


def test_c4_shape_inner_product_200d(orc, gp):
    # 200 dimensions: the filter's 16-k-step variant (hi-plane passes, 128-task groups); inner-product router and scan
    _run_config(orc, gp, 400_000, 200, 2000, 40, 10, 3, 1, 20_000, 50_000, sample=100)


def test_c5_shape_overlapping_shards(orc, gp):
    # 60 overlapping shards: the same id can come back from several probes, the merge drops the copies
    _run_config(orc, gp, 1_200_000, 100, 4000, 60, 10, 6, 0, 50_000, 50_000, sample=100, overlap=0.2)
