"""CPU: second, independent check of the surfaces the reference cannot pin here: the C++ oracle's InvertedIndex::QueryBucket
and KMeansTreeRouter::Query / FrequencyQuery against the numpy binary32 emulations of tests/np_ref.py (own distance sequence, own
lexicographic selection, own transcription of libstdc++'s heap, and Python's heapq as a third opinion where no keys are equal)."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters
import np_ref

FLT_MAX = np.finfo(np.float32).max


@pytest.mark.parametrize("metric,d", [(0, 24), (1, 24), (0, 7), (0, 130)])
def test_query_bucket_against_numpy_emulation(orc, metric, d):
    rng = np.random.default_rng(5)
    n, S, k = 2500, 4, 10
    pts = make_mixture(orc, n, d)
    pts[40:70] = pts[40]                      # equal distances: the row index decides
    clusters = random_clusters(rng, n, S, overlap=0.15)
    clustered, offsets, perm = orc.ivf_build(pts, clusters)
    queries = make_mixture(orc, 12, d, seed=556)
    queries[0] = pts[40]
    for q in queries:
        for b in range(S):
            gd, gi, grow = orc.ivf_query_bucket((clustered, offsets, perm), q, k, b, metric)
            wd, wi, wrow = np_ref.query_bucket(clustered, offsets, perm, q, k, b, metric)
            assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32)) and np.array_equal(gi, wi) and np.array_equal(grow, wrow)


def _tree(orc, metric, d=16, n=9000, S=6, num_centroids=16, min_cluster_size=30, budget=1500, seed=12):
    rng = np.random.default_rng(seed)
    pts = make_mixture(orc, n, d, n_centers=20)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=num_centroids, min_cluster_size=min_cluster_size, budget=budget, metric=metric, nthreads=4)
    return tree, make_mixture(orc, 30, d, seed=556, n_centers=20)


@pytest.mark.parametrize("metric", [0, 1])
def test_router_walk_against_numpy_emulation(orc, metric):
    tree, queries = _tree(orc, metric)
    flat = tree.flat()
    assert flat["node_child_count"].max() > 0 and len(flat["node_shard"]) > 6 * 10       # a real tree, several levels
    root_total = int(np.diff(flat["node_cent_first"])[flat["shard_root"]].sum())
    for budget in (1, 17, root_total - 5, root_total, root_total + 40, 400, 900, 10 ** 7):
        probes, md = tree.query_batch(queries, budget, nthreads=2)
        for qi, q in enumerate(queries):
            want, _ = np_ref.kmtr_query(flat, q, budget, metric)                         # libstdc++'s heap, transcribed
            assert np.array_equal(md[qi].view(np.uint32), want.view(np.uint32)), (budget, qi)
            if len(set(want.tolist())) == len(want):                                     # no equal min_dist: the order is unambiguous
                assert np.array_equal(probes[qi], np_ref.probe_order(want))
            if budget >= root_total:                                                     # all roots open: a different heap must agree ...
                alt, equal = np_ref.kmtr_query(flat, q, budget, metric, heap="heapq")
                if not equal:                                                            # ... unless two equal keys met in the queue
                    assert np.array_equal(alt.view(np.uint32), want.view(np.uint32))


def test_router_walk_with_duplicate_centroids(orc):
    """equal keys in the queue (duplicated points -> duplicate centroids): only libstdc++'s own heap order reproduces the walk"""
    rng = np.random.default_rng(3)
    n, d, S = 6000, 12, 4
    pts = make_mixture(orc, n, d, n_centers=10)
    pts[::3] = pts[0]                                  # a third of the points coincide
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=12, min_cluster_size=25, budget=900, metric=0, nthreads=4)
    flat = tree.flat()
    queries = make_mixture(orc, 20, d, seed=556, n_centers=10)
    for budget in (30, 100, 333):
        _, md = tree.query_batch(queries, budget, nthreads=2)
        for qi, q in enumerate(queries):
            want, _ = np_ref.kmtr_query(flat, q, budget, 0)
            assert np.array_equal(md[qi].view(np.uint32), want.view(np.uint32))


def test_frequency_query_against_numpy_emulation(orc):
    tree, queries = _tree(orc, 0)
    flat = tree.flat()
    for q in queries[:8]:
        for budget, votes in ((300, 40), (10 ** 6, 500)):
            md, nd, ns = tree.frequency_query(q, budget, votes)
            rec = []
            want_md, _ = np_ref.kmtr_query(flat, q, budget, 0, record=rec)
            assert np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
            # TopN(votes) of (dist, shard) pairs, lexicographic, ascending (topn.h + FrequencyQueryData::Init)
            rec.sort(key=lambda t: (t[0], t[1]))
            keep = rec[:votes]
            assert len(nd) == len(keep)
            assert np.array_equal(nd, np.array([t[0] for t in keep], np.float32)) and list(ns) == [t[1] for t in keep]
            # FrequencyQueryData::Query (kmeans_tree_router.h:23-44)
            for v in (5, 20, votes):
                probes = orc.frequency_probes(md, ns, v)
                cnt = np.bincount(np.asarray(ns[:min(v, len(ns))], np.int64), minlength=len(md))
                assert probes[0] == int(np.argmax(cnt))
                rest = [b for b in probes[1:]]
                assert sorted(rest + [probes[0]]) == list(range(len(md))) and all(md[rest[i]] <= md[rest[i + 1]] for i in range(len(rest) - 1))
