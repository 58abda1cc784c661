"""CPU: the product's host-side trainer (gp-ann_amd/host, KMeansTreeRouter::Train / KMeans / RandomSample / distance) against
the oracle's independent restatement — trees, partitions and distances must be bit-identical."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters


@pytest.fixture(scope="module")
def host():
    import pygpann
    pygpann.build()
    from pygpann import host as h
    h.lib()
    return h


@pytest.mark.parametrize("metric", [0, 1])
def test_host_distance_matches_oracle(orc, host, metric):
    rng = np.random.default_rng(3)
    for d in (1, 3, 4, 7, 96, 100, 128, 200):
        p = rng.standard_normal(d).astype(np.float32)
        q = rng.standard_normal(d).astype(np.float32)
        assert host.distance(p, q, mips=bool(metric)) == orc.distance(p, q, metric)


@pytest.mark.parametrize("metric", [0, 1])
def test_host_kmeans_matches_oracle(orc, host, metric):
    pts = make_mixture(orc, 5000, 20, n_centers=12)
    for k, seed in ((8, 555), (40, 7), (300, 1)):
        hp, hc = host.kmeans(pts, k, seed, mips=bool(metric), threads=4)
        op, oc = orc.kmeans(pts, k, seed, metric, nthreads=4)
        assert np.array_equal(hp, op)
        assert hc.shape == oc.shape and np.array_equal(hc.view(np.uint32), oc.view(np.uint32))
    # fewer points than samples: RandomSample repeats row 0, duplicates become empty clusters and are removed
    hp, hc = host.kmeans(pts[:10], 16, 3, mips=bool(metric))
    op, oc = orc.kmeans(pts[:10], 16, 3, metric)
    assert np.array_equal(hp, op) and np.array_equal(hc.view(np.uint32), oc.view(np.uint32)) and len(hc) <= 10


@pytest.mark.parametrize("metric,num_centroids,min_cluster,budget", [(0, 16, 40, 1500), (1, 8, 30, 900), (0, 64, 350, 400), (0, 4, 10, 5000)])
def test_host_trainer_matches_oracle(orc, host, metric, num_centroids, min_cluster, budget):
    rng = np.random.default_rng(budget)
    n, d, S = 9000, 24, 5
    pts = make_mixture(orc, n, d, n_centers=20)
    clusters = random_clusters(rng, n, S, overlap=0.1)
    f = host.train_router(pts, clusters, num_centroids, min_cluster, budget, mips=bool(metric), threads=4)
    o = orc.kmtr_train(pts, clusters, num_centroids, min_cluster, budget, metric, nthreads=4).flat()
    for key in ("node_shard", "node_cent_first", "node_child_first", "node_child_count", "shard_root"):
        assert np.array_equal(f[key], o[key]), key
    assert np.array_equal(f["centroids"].view(np.uint32), o["centroids"].view(np.uint32))
    # tree invariants of the flat layout
    nc = np.diff(f["node_cent_first"])
    assert (f["node_child_count"] <= nc).all() and (nc >= 1).all() and nc.max() <= max(2, num_centroids)
    assert len(f["node_shard"]) >= S and sorted(f["shard_root"]) == list(f["shard_root"])
