"""CPU: the host mirror's k-means shard builders (gp-ann_amd/host/partition_host.cpp; reference src/partitioning.cpp:12-146,
src/overlapping_partitioning.cpp:143-235) in host mode against the oracle's restatement (oracle/partitioners.py): partitions and
overlapping clusters must be identical.  The GPU mode of the same functions is checked in test_gpu_partition.py."""
import numpy as np
import pytest

from conftest import make_mixture


@pytest.fixture(scope="module")
def host():
    import pygpann
    pygpann.build()
    from pygpann import host as h
    h.lib()
    return h


@pytest.fixture(scope="module")
def parts():
    import partitioners
    return partitioners


def skewed_points(orc, n, d, seed=555):
    """a mixture with one heavy component so that flat k-means leaves overloaded clusters"""
    a = make_mixture(orc, n // 2, d, seed=seed, n_centers=1, sigma=0.5)
    b = make_mixture(orc, n - n // 2, d, seed=seed + 1, n_centers=12)
    return np.concatenate([a, b])


@pytest.mark.parametrize("metric", [0, 1])
def test_kmeans_partitioning_matches_oracle(orc, host, parts, metric):
    # (with the inner-product metric a single dense blob may never split — the reference recurses forever there — so MIPS gets the plain mixture)
    pts = skewed_points(orc, 6000, 16) if metric == 0 else make_mixture(orc, 6000, 16, n_centers=14)
    for k, eps in ((6, 0.05), (11, 0.3)):
        got = host.partition(host.PART_KMEANS, pts, k=k, eps=eps, mips=bool(metric), threads=4)
        want = parts.kmeans_partitioning(orc, pts, k, eps, metric)
        assert np.array_equal(got, want)
        assert np.bincount(got).max() <= int(len(pts) * (1 + eps) / k)     # the recursion ends with every part inside the bound
        if metric == 0:
            assert got.max() + 1 > k                                         # ... and it had to split something


def test_recursive_derives_cluster_count(orc, host, parts):
    pts = make_mixture(orc, 3000, 8, n_centers=5)
    got = host.partition(host.PART_RECURSIVE, pts, k=-1, max_cluster_size=700, threads=4)
    want = parts.recursive_kmeans_partitioning(orc, pts, 700)
    assert np.array_equal(got, want) and np.bincount(got).max() <= 700


@pytest.mark.parametrize("metric", [0, 1])
def test_rebalancing_matches_oracle(orc, host, parts, metric):
    pts = skewed_points(orc, 5000, 12, seed=9)
    k = 8
    cap = int(1.05 * len(pts) / k)
    got = host.partition(host.PART_REBALANCING, pts, k=k, max_cluster_size=cap, mips=bool(metric), threads=4)
    want = parts.rebalancing_kmeans_partitioning(orc, pts, cap, k, metric)
    assert np.array_equal(got, want)
    assert np.bincount(got).max() <= cap
    flat, _ = orc.kmeans(pts, k, 555, metric)
    assert (flat != got).any()   # points were moved


@pytest.mark.parametrize("metric,overlap", [(0, 0.2), (1, 0.1), (0, 0.0)])
def test_overlap_spann_matches_oracle(orc, host, parts, metric, overlap):
    pts = make_mixture(orc, 4000, 12, n_centers=9)
    k = 6
    base = host.partition(host.PART_REBALANCING, pts, k=k, max_cluster_size=int(1.05 * len(pts) / k), mips=bool(metric), threads=4)
    got = host.overlap_spann(pts, base, k, 0.25, overlap, mips=bool(metric), threads=4)
    want = parts.overlapping_kmeans_partitioning_spann(orc, pts, base, k, 0.25, overlap, metric)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, np.asarray(w, np.uint32))
    extra = sum(len(c) for c in got) - len(pts)
    cap = int(1.25 * len(pts) / k)
    assert max(len(c) for c in got) <= cap
    if overlap > 0:
        assert extra == int(overlap * len(pts))
        assert all(len(set(c.tolist())) == len(c) for c in got)   # no point twice in a cluster
    else:
        assert extra > 0   # the reference's unsigned counter never reaches zero: every admissible rating is taken
