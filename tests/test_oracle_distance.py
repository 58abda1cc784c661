"""CPU: the oracle's distance / TopN / scan restatement against an independent numpy float32 emulation."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters
import np_ref


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 96, 100, 128, 200])
def test_distance_matches_numpy_emulation(orc, metric, d):
    rng = np.random.default_rng(d * 7 + metric)
    pts = rng.standard_normal((257, d)).astype(np.float32) * 3
    q = rng.standard_normal(d).astype(np.float32)
    got = orc.distance_many(pts, q, metric)
    want = np_ref.dist_many(pts, q, metric)
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert orc.distance(pts[3], q, metric) == want[3]


def test_distance_is_not_fused(orc):
    # a case where fma(d1,d1,d0*d0) differs from round(round(d0*d0)+round(d1*d1))
    rng = np.random.default_rng(0)
    found = False
    for _ in range(2000):
        p = rng.standard_normal(4).astype(np.float32)
        q = rng.standard_normal(4).astype(np.float32)
        diff = (p.astype(np.float64) - q.astype(np.float64))
        strict = np_ref.l2_many(p[None], q)[0]
        fused_like = np.float32(np.sum(diff * diff))
        if strict != fused_like:
            found = True
            assert orc.distance(p, q, 0) == strict
    assert found


def test_topn_semantics(orc):
    rng = np.random.default_rng(1)
    d = rng.integers(0, 5, size=200).astype(np.float32)      # many ties
    ids = rng.permutation(200).astype(np.uint32)
    for n in (1, 3, 10, 64, 200, 300):
        gd, gi = orc.topn(d, ids, n)
        wd, wi = np_ref.topk_lex(d, ids, n)
        assert np.array_equal(gd, wd) and np.array_equal(gi, wi)


@pytest.mark.parametrize("metric", [0, 1])
def test_query_bucket_and_query(orc, metric):
    rng = np.random.default_rng(2)
    n, d, S, k = 3000, 24, 5, 10
    pts = make_mixture(orc, n, d)
    pts[100:140] = pts[100]     # duplicates -> distance ties broken by row position
    clusters = random_clusters(rng, n, S, overlap=0.2)
    ivf = orc.ivf_build(pts, clusters)
    clustered, offsets, perm = ivf
    assert offsets[-1] == sum(len(c) for c in clusters)
    for b in range(S):
        assert np.array_equal(perm[offsets[b]:offsets[b + 1]], clusters[b])
        assert np.array_equal(clustered[offsets[b]:offsets[b + 1]], pts[clusters[b]])
    queries = make_mixture(orc, 20, d, seed=556)
    queries[0] = pts[100]
    for qi in range(20):
        Q = queries[qi]
        for b in range(S):
            gd, gi, grow = orc.ivf_query_bucket(ivf, Q, k, b, metric)
            rows = np.arange(offsets[b], offsets[b + 1], dtype=np.uint32)
            dist = np_ref.dist_many(clustered[rows], Q, metric)
            wd, wrow = np_ref.topk_lex(dist, rows, k)
            assert np.array_equal(gd, wd) and np.array_equal(grow, wrow) and np.array_equal(gi, perm[wrow])
        probes = rng.permutation(S).astype(np.int32)
        for nprobe in (1, 3, S):
            gd, gi = orc.ivf_query(ivf, Q, k, probes, nprobe, metric)
            rows = np.concatenate([np.arange(offsets[b], offsets[b + 1], dtype=np.uint32) for b in probes[:nprobe]])
            dist = np_ref.dist_many(clustered[rows], Q, metric)
            wd, wrow = np_ref.topk_lex(dist, rows, k)
            assert np.array_equal(gd, wd) and np.array_equal(gi, perm[wrow])


def test_ground_truth_and_recall(orc):
    n, d, k = 2000, 16, 10
    pts = make_mixture(orc, n, d)
    queries = make_mixture(orc, 30, d, seed=556)
    gd, gi = orc.ground_truth(pts, queries, k, 0, nthreads=2)
    for q in range(30):
        dist = np_ref.l2_many(pts, queries[q])
        wd, wi = np_ref.topk_lex(dist, np.arange(n, dtype=np.uint32), k)
        assert np.array_equal(gd[q], wd) and np.array_equal(gi[q], wi)
    kth = orc.distance_to_kth(pts, queries, k, 0)
    assert np.array_equal(kth, gd[:, k - 1])
    # ConvertGroundTruthToDistanceToKthNeighbor: distances get recomputed and re-sorted
    bad = gd.copy()
    bad[:, 0] += 5.0
    kth2, fixed_d, fixed_i, wrong = orc.convert_ground_truth(bad, gi, k, pts, queries, 0)
    assert wrong == 0 and np.array_equal(kth2, kth) and np.array_equal(fixed_d, gd)
    swapped = gi.copy()
    swapped[:, [0, k - 1]] = swapped[:, [k - 1, 0]]
    kth3, _, _, wrong3 = orc.convert_ground_truth(gd, swapped, k, pts, queries, 0)
    assert wrong3 > 0 and np.array_equal(kth3, kth)
    # Recall = #(dist <= kth) / (nq * k)
    offsets = np.arange(31, dtype=np.uint64) * k
    assert orc.recall(gd.ravel(), offsets, kth, k) == 1.0
    half = gd.copy()
    half[:, k // 2:] = np.inf
    assert orc.recall(half.ravel(), offsets, kth, k) == 0.5


def test_dedup_neighbors(orc):
    d = np.array([3.0, 1.0, 3.0, 2.0, 1.0, 9.0], np.float32)
    i = np.array([7, 4, 7, 5, 4, 1], np.uint32)
    od, oi, oc = orc.dedup_neighbors(d, i, np.array([0, 6], np.uint64), 3)
    assert oc[0] == 3 and list(oi[0]) == [4, 5, 7] and list(od[0]) == [1.0, 2.0, 3.0]
    od, oi, oc = orc.dedup_neighbors(d, i, np.array([0, 6], np.uint64), 10)
    assert oc[0] == 4 and list(oi[0][:4]) == [4, 5, 7, 1] and oi[0][4] == 0xFFFFFFFF and np.isinf(od[0][4])
