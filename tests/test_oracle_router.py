"""CPU: properties of the oracle's KMeansTreeRouter restatement (src/kmeans_tree_router.cpp) and of the golden vectors."""
import os

import numpy as np
import pytest

from conftest import ROOT, make_mixture, random_clusters
import np_ref

FLT_MAX = np.finfo(np.float32).max


def _setup(orc, metric=0):
    rng = np.random.default_rng(12)
    pts = make_mixture(orc, 8000, 16, n_centers=20)
    clusters = random_clusters(rng, 8000, 6)
    tree = orc.kmtr_train(pts, clusters, num_centroids=16, min_cluster_size=30, budget=1200, metric=metric, nthreads=4)
    queries = make_mixture(orc, 40, 16, seed=556, n_centers=20)
    return pts, clusters, tree, queries


@pytest.mark.parametrize("metric", [0, 1])
def test_exhaustive_budget_gives_min_over_all_centroids(orc, metric):
    pts, clusters, tree, queries = _setup(orc, metric)
    f = tree.flat()
    shard_of = tree.extract_partition()
    assert len(shard_of) == tree.n_centroids and set(shard_of) == set(range(6))
    probes, md = tree.query_batch(queries, 10 ** 8)
    for q in range(len(queries)):
        dist = np_ref.dist_many(f["centroids"], queries[q], metric)
        want = np.array([dist[shard_of == b].min() for b in range(6)], np.float32)
        assert np.array_equal(md[q], want)
        assert np.all(np.diff(md[q][probes[q]]) >= 0) and sorted(probes[q]) == list(range(6))


def test_budget_semantics(orc):
    pts, clusters, tree, queries = _setup(orc)
    f = tree.flat()
    root_sizes = np.diff(f["node_cent_first"])[f["shard_root"]]
    # budget <= 0: nothing is evaluated (the loop tests budget > 0 before the first pop)
    for budget in (0, -3):
        probes, md = tree.query_batch(queries, budget)
        assert (md == FLT_MAX).all()
    # budget 1: exactly one root is opened completely (budget may go negative after a pop)
    probes, md = tree.query_batch(queries, 1)
    assert ((md != FLT_MAX).sum(axis=1) == 1).all()
    # enough for all roots but one centroid short of the last: all but one root get opened ... when sizes are equal
    total = int(root_sizes.sum())
    probes, md = tree.query_batch(queries, total - int(root_sizes.min()) + 1)
    assert ((md != FLT_MAX).sum(axis=1) == 6).all()
    # monotone: a larger budget never increases min_dist
    prev = None
    for budget in (10, 50, 200, 1000, 100000):
        _, md = tree.query_batch(queries, budget)
        if prev is not None:
            assert (md <= prev).all()
        prev = md


def test_frequency_query_consistency(orc):
    pts, clusters, tree, queries = _setup(orc)
    for q in range(5):
        md, nd, ns = tree.frequency_query(queries[q], 300, 50)
        _, md2 = tree.query(queries[q], 300)
        assert np.array_equal(md, md2)                      # same walk
        assert len(nd) <= 50 and np.all(np.diff(nd) >= 0)
        for b in set(ns):
            assert nd[ns == b].min() == md[b] or md[b] <= nd[ns == b].min()
        probes = orc.frequency_probes(md, ns, 20)
        votes = np.bincount(ns[:20], minlength=6)
        assert probes[0] == int(np.argmax(votes)) and sorted(probes) == list(range(6))
        rest = probes[1:]
        assert np.all(np.diff(md[rest]) >= 0)


def test_golden_hotpath_vectors_replay_on_the_oracle(orc):
    """the committed hot-path fixtures are what the current oracle produces (guards against silent oracle drift)"""
    gold = os.path.join(ROOT, "tests", "golden")
    for name in ("l2_d128", "l2_d100", "ip_d200", "l2_d7"):
        z = np.load(os.path.join(gold, f"hotpath_{name}.npz"))
        n, d, k, metric = int(z["n"]), int(z["d"]), int(z["k"]), int(z["metric"])
        pts = orc.synth_rows(555, 7, 16, 0.35, 0, n, d)
        a, b = z["tie_rows"]
        pts[a:b] = pts[a]
        queries = orc.synth_rows(556, 7, 16, 0.35, 0, 24, d)
        queries[0] = pts[a]
        offs = z["cluster_offsets"]
        clusters = [z["cluster_ids"][int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
        ivf = orc.ivf_build(pts, clusters)
        sd, si, _ = orc.ivf_query_bucket_batch(ivf, queries, k, z["probes"], 3, metric)
        assert np.array_equal(si, z["scan_ids"]) and np.array_equal(sd.view(np.uint32), z["scan_dist"].view(np.uint32))
        gd, gi = orc.ground_truth(pts, queries, k, metric)
        assert np.array_equal(gi, z["gt_ids"]) and np.array_equal(gd.view(np.uint32), z["gt_dist"].view(np.uint32))
        flat = {key[5:]: z[key] for key in z.files if key.startswith("tree_")}
        flat["d"], flat["S"] = int(flat["d"]), int(flat["S"])
        tree = orc.kmtr_from_flat(flat, metric)
        rp, rm = tree.query_batch(queries, int(z["router_budget"][0]))
        assert np.array_equal(rp, z["router_probes"]) and np.array_equal(rm.view(np.uint32), z["router_min_dist"].view(np.uint32))
        rp, rm = tree.query_batch(queries, int(z["router_budget"][1]))
        assert np.array_equal(rp, z["router_probes_small"]) and np.array_equal(rm.view(np.uint32), z["router_min_dist_small"].view(np.uint32))
