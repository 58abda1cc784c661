"""GPU parity: KMeansTreeRouter::Train on the GPU (gpann_router_train, level-synchronous batched Lloyd) against the CPU
oracle's restatement of kmeans.cpp / kmeans_tree_router.cpp — node structure identical, centroids bit-exact."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu

KEYS = ("node_shard", "node_cent_first", "node_child_first", "node_child_count", "shard_root")


def _same_tree(f, o):
    for key in KEYS:
        assert np.array_equal(np.asarray(f[key]), np.asarray(o[key])), key
    same = np.array_equal(np.asarray(f["centroids"], np.float32).view(np.uint32), np.asarray(o["centroids"], np.float32).view(np.uint32))
    assert same, "centroid bits differ"


def _train_both(orc, gp, pts, clusters, metric, num_centroids, min_cluster, budget):
    shards = gp.Shards(pts, clusters, metric=metric)
    router = gp.Router.train(shards, len(pts), num_centroids, min_cluster, budget)
    f = router.export()
    o = orc.kmtr_train(pts, clusters, num_centroids, min_cluster, budget, metric, nthreads=8).flat()
    return shards, router, f, o


@pytest.mark.parametrize("metric,d,num_centroids,min_cluster,budget", [
    (0, 24, 16, 40, 1500),      # three levels
    (1, 24, 8, 30, 900),        # IP: normalised sums, rescaled centroids
    (0, 128, 64, 350, 400),     # budget smaller than one node per shard allows: roots only
    (0, 7, 4, 10, 5000),        # deep narrow tree, d % 4 != 0
    (1, 100, 70, 100, 3000),    # more than one centroid tile
    (0, 3, 130, 25, 2000),
])
def test_device_trainer_matches_oracle(orc, gp, metric, d, num_centroids, min_cluster, budget):
    rng = np.random.default_rng(budget)
    n, S = 9000, 5
    pts = make_mixture(orc, n, d, n_centers=20)
    clusters = random_clusters(rng, n, S, overlap=0.1)
    _, router, f, o = _train_both(orc, gp, pts, clusters, metric, num_centroids, min_cluster, budget)
    _same_tree(f, o)
    # and the trained handle routes like the oracle's tree
    queries = make_mixture(orc, 64, d, seed=556, n_centers=20)
    tree = orc.kmtr_train(pts, clusters, num_centroids, min_cluster, budget, metric, nthreads=8)
    probes, md, _ = router.query_batch(queries, 700)
    wp, wm = tree.query_batch(queries, 700, nthreads=4)
    assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)


def test_device_trainer_duplicates_and_tiny_shards(orc, gp):
    """fewer distinct points than samples: duplicate centroids lose every point to the first copy (strict '<') and are
    removed as empty clusters (kmeans.cpp:29-50); a shard smaller than the sample repeats row 0 (kmeans.cpp:207-226)"""
    rng = np.random.default_rng(11)
    d = 12
    base = make_mixture(orc, 40, d, n_centers=5)
    pts = base[rng.integers(0, 40, size=5000)]
    clusters = [np.arange(0, 2000, dtype=np.uint32), np.arange(2000, 4990, dtype=np.uint32), np.arange(4990, 5000, dtype=np.uint32)]
    _, _, f, o = _train_both(orc, gp, pts, clusters, 0, 64, 5, 3000)
    _same_tree(f, o)
    assert np.diff(f["node_cent_first"]).min() < 64


def test_device_trainer_larger_shards(orc, gp):
    """several counting chunks and SumKernel batches per cluster; sums over tens of thousands of members in point order"""
    rng = np.random.default_rng(5)
    n, d, S = 150000, 32, 3
    pts = make_mixture(orc, n, d, n_centers=64)
    clusters = random_clusters(rng, n, S)
    _, _, f, o = _train_both(orc, gp, pts, clusters, 0, 32, 350, 2000)
    _same_tree(f, o)


def test_device_trainer_matches_host_mirror(orc, gp):
    from pygpann import host
    rng = np.random.default_rng(9)
    n, d, S = 20000, 48, 4
    pts = make_mixture(orc, n, d, n_centers=30)
    clusters = random_clusters(rng, n, S)
    shards = gp.Shards(pts, clusters, metric=0)
    f = gp.Router.train(shards, n, 32, 100, 1200).export()
    h = host.train_router(pts, clusters, 32, 100, 1200, mips=False, threads=8)
    _same_tree(f, h)


@pytest.mark.parametrize("metric", [0, 1])
def test_device_trainer_beyond_reference_block_size(orc, gp, metric):
    """n > 5,000,000 rows in one node: the reference sums per 5M-row slice and adds the slice partials with atomics
    (kmeans.cpp:151-204); oracle and GPU both take the partials in slice order"""
    n, d = 5300000, 4
    pts = orc.synth_rows(555, 7, 9, 0.35, 0, n, d, nthreads=8)
    clusters = [np.arange(n, dtype=np.uint32)]
    _, _, f, o = _train_both(orc, gp, pts, clusters, metric, 4, 350, 4)
    _same_tree(f, o)
    assert len(f["node_shard"]) == 1


def test_device_trainer_per_rank_parts_merge_to_full_tree(orc, gp):
    """shard-parallel training: every 'rank' trains over a store holding only its own shards (the others are empty and get
    a root without centroids); merging the parts gives the tree trained over all shards"""
    import multirank_ref
    from pygpann import comm as gcomm
    rng = np.random.default_rng(21)
    n, d, S = 16000, 20, 5
    pts = make_mixture(orc, n, d, n_centers=25)
    clusters = random_clusters(rng, n, S)
    o = orc.kmtr_train(pts, clusters, 16, 60, 1000, 0, nthreads=8).flat()
    owner, _ = gcomm.lpt_owner(np.asarray([len(c) for c in clusters]), 2)
    parts = []
    for rank in range(2):
        mine = [c if owner[b] == rank else np.zeros(0, np.uint32) for b, c in enumerate(clusters)]
        shards = gp.Shards(pts, mine, metric=0)
        parts.append(gp.Router.train(shards, n, 16, 60, 1000).export())
    _same_tree(multirank_ref.merge_trees(parts, owner), o)
