"""GPU: randomised small configurations of the router path against the CPU oracle — KMeansTreeRouter::Train on the GPU (gpann_router_train: the tree's
structure identical, centroids bit-exact; src/kmeans_tree_router.cpp:9-91, src/kmeans.cpp), KMeansTreeRouter::Query at budgets below, around and above the
tree's size (the walk, the distance table with its replays, the matrix-core bound; :93-130) on the trained handle AND on a handle created from the oracle's
tree, and FrequencyQuery (:132-172) with random vote counts: odd dimensions, both metrics, few or many centroids per node, deep narrow trees, shards smaller
than a sample, duplicated points (equal keys everywhere).  The session's GPANN_ROUTER_VERIFY / GPANN_ROUTER_STRICT stay armed (conftest.py).
Seeds are fixed: a failure names its configuration.  GPANN_FUZZ_SEEDS=a:b replaces the suite's range for a longer run."""
import os

import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu
KEYS = ("node_shard", "node_cent_first", "node_child_first", "node_child_count", "shard_root")


def _seeds():
    a, _, b = os.environ.get("GPANN_FUZZ_SEEDS", "0:30").partition(":")
    return range(int(a), int(b))


@pytest.mark.parametrize("seed", _seeds())
def test_random_router_configuration(orc, gp, seed):
    rng = np.random.default_rng(13000 + seed)
    d = int(rng.choice([3, 7, 16, 24, 33, 64, 100, 128, 200]))
    n = int(rng.integers(1500, 14000))
    S = int(rng.integers(1, 9))
    metric = int(rng.integers(0, 2))
    num_centroids = int(rng.choice([4, 5, 8, 16, 64, 130]))
    min_cluster = int(rng.choice([5, 10, 40, 350]))
    budget_train = int(rng.choice([300, 900, 3000, 8000]))
    nq = int(rng.choice([1, 2, 33, 150]))
    if rng.integers(0, 4) == 0:                                     # few distinct points: duplicate centroids, empty clusters, equal keys in the walk
        base = make_mixture(orc, int(rng.integers(20, 80)), d, n_centers=4)
        pts = np.ascontiguousarray(base[rng.integers(0, len(base), size=n)])
        queries = np.ascontiguousarray(base[rng.integers(0, len(base), size=nq)])
    else:
        pts = make_mixture(orc, n, d, n_centers=int(rng.integers(1, 24)), sigma=float(rng.choice([0.05, 0.35, 1.0])))
        queries = make_mixture(orc, nq, d, seed=561, n_centers=5)
    if rng.integers(0, 4) == 0:
        pts += np.float32(rng.normal(0, 20))
        queries += np.float32(rng.normal(0, 20))
    clusters = [np.asarray(c, np.uint32) for c in random_clusters(rng, n, S, overlap=float(rng.choice([0.0, 0.1])))]
    if S > 2:
        clusters[int(rng.integers(0, S))] = np.arange(7, dtype=np.uint32)      # a shard smaller than one node's sample
    cfg = dict(seed=seed, d=d, n=n, S=S, metric=metric, num_centroids=num_centroids, min_cluster=min_cluster, budget_train=budget_train, nq=nq)

    tree = orc.kmtr_train(pts, clusters, num_centroids, min_cluster, budget_train, metric, nthreads=4)
    o = tree.flat()
    store = gp.Shards(pts, clusters, metric=metric, plain=True)
    trained = gp.Router.train(store, n, num_centroids, min_cluster, budget_train)
    created = gp.Router(o, metric)
    try:
        f = trained.export()
        for key in KEYS:
            assert np.array_equal(np.asarray(f[key]), np.asarray(o[key])), (key, cfg)
        assert np.array_equal(np.asarray(f["centroids"], np.float32).view(np.uint32), np.asarray(o["centroids"], np.float32).view(np.uint32)), ("centroid bits", cfg)
        C = int(o["node_cent_first"][-1])
        budgets = sorted({max(1, C // 5), max(1, C // 2), max(1, C - 1), C, C + 17, 1 << 30, int(rng.integers(1, 2 * C + 2))})
        for budget in budgets:
            wp, wm = tree.query_batch(queries, budget, nthreads=4)
            for name, router in (("trained", trained), ("created", created)):
                p, md, _ = router.query_batch(queries, budget)
                assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)), ("min_dist", name, budget, cfg)
                assert np.array_equal(p, wp), ("probes", name, budget, cfg)
        votes = int(rng.choice([1, 5, 50, 400]))
        for budget in (max(1, C // 2), C + 17):
            md, nd, ns, nc, _ = created.frequency_query_batch(queries, budget, votes)
            for q in range(0, nq, max(1, nq // 12)):
                wmd, wnd, wns = tree.frequency_query(queries[q], budget, votes)
                assert np.array_equal(md[q].view(np.uint32), wmd.view(np.uint32)), ("frequency min_dist", budget, votes, q, cfg)
                assert nc[q] == len(wnd) and np.array_equal(nd[q, :nc[q]].view(np.uint32), wnd.view(np.uint32)) and np.array_equal(ns[q, :nc[q]], wns), \
                    ("frequency votes", budget, votes, q, cfg)
        for router in (trained, created):
            st = router.stats()
            assert st["bound_selfcheck"] == 0 and st["verify_failures"] == 0, (st, cfg)
    finally:
        trained.close()
        created.close()
        store.close()
