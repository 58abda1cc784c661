"""GPU parity of the MFMA-filtered scan mode (gp-ann_amd/csrc/scan_filter.h): same bit-exact ids / distances as the plain exact
scan and the CPU oracle — the filter only chooses which rows receive the canonical evaluation."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu


def _check(orc, gp, pts, clusters, queries, probes, nprobe, k, metric, expect_filtered=True):
    shards = gp.Shards(pts, clusters, metric, 0)
    shards.set_scan_mode(1)
    ids, dist, rows, _ = shards.scan_buckets(queries, k, probes, nprobe, want_rows=True)
    st = shards.stats()
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, metric, nthreads=4)
    assert np.array_equal(ids, wi)
    assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
    if expect_filtered is not None:
        assert st["filtered"] == (1 if expect_filtered else 0), st
    shards.set_scan_mode(0)
    ids0, dist0, _ = shards.scan_buckets(queries, k, probes, nprobe)
    assert np.array_equal(ids0, ids) and np.array_equal(dist0.view(np.uint32), dist.view(np.uint32))
    shards.close()
    return st


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [128, 96, 100, 30, 7])
def test_filtered_scan_matches_oracle(orc, gp, metric, d):
    rng = np.random.default_rng(d * 3 + metric)
    n, S, nq, k = 40000, 3, 300, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S, overlap=0.15)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    st = _check(orc, gp, pts, clusters, queries, probes, 2, k, metric)
    assert st["filter_pairs"] > 0


def test_filtered_scan_two_passes_and_many_tasks(orc, gp):
    """a 200k-row shard (several cells) and more than 256 tasks per shard (several task groups)"""
    rng = np.random.default_rng(17)
    n, d, nq, k = 200000, 32, 600, 10
    pts = make_mixture(orc, n, d, n_centers=40)
    clusters = [rng.permutation(n).astype(np.uint32)]
    queries = make_mixture(orc, nq, d, seed=556, n_centers=40)
    probes = np.zeros((nq, 1), np.int32)
    _check(orc, gp, pts, clusters, queries, probes, 1, k, 0)


@pytest.mark.parametrize("k", [1, 7, 64])
def test_filtered_scan_k_values(orc, gp, k):
    rng = np.random.default_rng(k)
    n, d, S, nq = 30000, 64, 2, 130
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    _check(orc, gp, pts, clusters, queries, probes, S, k, 0)


def test_filtered_scan_far_queries_and_scaled_data(orc, gp):
    """large norms / tiny gaps: the error bound has to hold when |p|, |q| >> distance"""
    rng = np.random.default_rng(23)
    n, d, nq, k = 30000, 48, 100, 10
    base = make_mixture(orc, n, d, n_centers=8, sigma=0.01)
    pts = (base * np.float32(37.0) + np.float32(900.0)).astype(np.float32)       # |p| ~ 6000, neighbours within ~1
    queries = pts[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.05)
    clusters = [rng.permutation(n).astype(np.uint32)]
    probes = np.zeros((nq, 1), np.int32)
    for metric in (0, 1):
        _check(orc, gp, pts, clusters, queries, probes, 1, k, metric, expect_filtered=None)


def test_filtered_scan_duplicates_overflow_falls_back(orc, gp):
    """tens of thousands of identical rows tie with the k-th distance: every one is a candidate, buckets overflow, the call
    falls back to the plain exact scan and still returns the reference's answer (ties broken by row)"""
    rng = np.random.default_rng(29)
    n, d, nq, k = 30000, 16, 90, 10
    pts = make_mixture(orc, n, d, n_centers=4)
    pts[2000:] = pts[2000]
    clusters = [np.arange(n, dtype=np.uint32)]
    queries = make_mixture(orc, nq, d, seed=556, n_centers=4)
    queries[:40] = pts[2000]          # for these every duplicate ties at distance 0: 40 buckets overflow -> whole-call fallback
    probes = np.zeros((nq, 1), np.int32)
    st = _check(orc, gp, pts, clusters, queries, probes, 1, k, 0, expect_filtered=None)
    assert st["filtered"] == 0
    queries[1:40] = queries[41:80][:39] if nq >= 80 else queries[40:41]   # only one flooding query: a DIRTY task, its shard is redone exactly on the device
    st = _check(orc, gp, pts, clusters, queries, probes, 1, k, 0, expect_filtered=None)
    assert st["filtered"] == 1 and st["filter_dirty_tasks"] >= 1, st


def test_filtered_ground_truth_and_merged(orc, gp):
    rng = np.random.default_rng(31)
    n, d, S, nq, k = 50000, 40, 4, 150, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S, overlap=0.2)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(1)
    ivf = orc.ivf_build(pts, clusters)
    ids, dist, _ = shards.scan_merged(queries, k, probes, 3)
    for q in range(0, nq, 7):
        wd, wi = orc.ivf_query(ivf, queries[q], k, probes[q], 3, 0)
        assert np.array_equal(ids[q], wi) and np.array_equal(dist[q].view(np.uint32), wd.view(np.uint32))
    assert shards.stats()["filtered"] == 1


def test_filtered_scan_single_bucket_overflow_is_rescanned_in_kernel(orc, gp, monkeypatch):
    """a tiny candidate cap forces a few buckets to overflow: those tasks become dirty and their shards are evaluated exactly, in segments
    spread over many wavefronts (RescoreKernel RANGE mode + DirtyMergeKernel), after the last pass"""
    rng = np.random.default_rng(37)
    n, d, nq, k = 30000, 24, 3000, 10
    pts = make_mixture(orc, n, d, n_centers=16)
    clusters = [rng.permutation(n).astype(np.uint32)]
    queries = make_mixture(orc, nq, d, seed=556, n_centers=16)
    probes = np.zeros((nq, 1), np.int32)
    monkeypatch.setenv("GPANN_FILTER_CAND_CAP", "200")     # expected fill ~ k * 26000 / 4096 = 63 + tail
    st = _check(orc, gp, pts, clusters, queries, probes, 1, k, 0, expect_filtered=None)
    assert st["filtered"] == 0 or st["filter_dirty_tasks"] >= 1, st
    # several shards of different sizes, dirty tasks in each, L2 and inner product, merge bound on: the redone lists obey the same contract
    S = 4
    clusters = random_clusters(rng, n, S, overlap=0.2)
    probes = np.stack([rng.permutation(S)[:3] for _ in range(nq)]).astype(np.int32)
    monkeypatch.setenv("GPANN_FILTER_CAND_CAP", "120")
    for metric in (0, 1):
        st = _check(orc, gp, pts, clusters, queries[:700], probes[:700], 3, k, metric, expect_filtered=None)


def test_filter_without_precomputed_planes(orc, gp, monkeypatch):
    """GPANN_FILTER_PLANES=0: a store without bf16 planes / cells offers only the plain exact scan; asking for filter mode
    must silently give the exact answer"""
    rng = np.random.default_rng(41)
    n, d, S, nq, k = 30000, 100, 2, 140, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    monkeypatch.setenv("GPANN_FILTER_PLANES", "0")
    for metric in (0, 1):
        st = _check(orc, gp, pts, clusters, queries, probes, S, k, metric, expect_filtered=False)
        assert st["filter_pairs"] == 0


@pytest.mark.parametrize("parts,split_min", [("1", "192"), ("8", "64"), ("3", "100")])
def test_split_buckets_give_the_same_lists(orc, gp, monkeypatch, parts, split_min):
    """heavy candidate buckets are re-scored by several wavefronts whose lists the last one merges (RescoreArgs::parts): whatever the
    split, the lists are the oracle's — one blob (no pruning: hundreds of candidates per task), duplicates of the query's neighbours
    (equal distances across the pieces), L2 and inner product"""
    monkeypatch.setenv("GPANN_RESCORE_PARTS", parts)
    monkeypatch.setenv("GPANN_RESCORE_SPLIT_MIN", split_min)
    monkeypatch.setenv("GPANN_CELL_ROWS", "2000")
    rng = np.random.default_rng(53)
    n, d, nq, k = 60000, 32, 400, 10
    pts = rng.standard_normal((n, d)).astype(np.float32)
    pts[100:130] = pts[100]
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    queries[0] = pts[100]
    clusters = [rng.permutation(n).astype(np.uint32)]
    probes = np.zeros((nq, 1), np.int32)
    for metric in (0, 1):
        st = _check(orc, gp, pts, clusters, queries, probes, 1, k, metric, expect_filtered=None)
        assert st["filtered"] == 1 and st["filter_candidates"] > 64 * nq, st   # buckets heavy enough to be split
