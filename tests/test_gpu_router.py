"""GPU parity: KMeansTreeRouter::Query on the GPU against the CPU oracle — min_dist bit-exact, probes identical."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu


def _tree(orc, n, d, S, metric, num_centroids, min_cluster, budget, seed=1):
    rng = np.random.default_rng(seed)
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=num_centroids, min_cluster_size=min_cluster, budget=budget, metric=metric, nthreads=4)
    return pts, clusters, tree


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d,num_centroids", [(32, 16), (100, 8), (7, 64)])
def test_router_matches_oracle(orc, gp, metric, d, num_centroids):
    pts, clusters, tree = _tree(orc, 12000, d, 5, metric, num_centroids, 40, 1500)
    assert tree.n_nodes > 5
    queries = make_mixture(orc, 200, d, seed=556, n_centers=24)
    router = gp.Router(tree.flat(), metric, 0)
    for budget in (1, 37, 300, 100000):
        probes, md, _ = router.query_batch(queries, budget)
        wp, wm = tree.query_batch(queries, budget, nthreads=4)
        assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)), budget
        assert np.array_equal(probes, wp), budget


def test_router_unvisited_shards_and_zero_budget(orc, gp):
    pts, clusters, tree = _tree(orc, 8000, 16, 8, 0, 16, 30, 2000)
    queries = make_mixture(orc, 64, 16, seed=556, n_centers=24)
    router = gp.Router(tree.flat(), 0, 0)
    FLT_MAX = np.finfo(np.float32).max
    # budget too small to open every root: unopened shards keep FLT_MAX and the order among them is libstdc++'s
    probes, md, _ = router.query_batch(queries, 20)
    wp, wm = tree.query_batch(queries, 20, nthreads=2)
    assert (md == FLT_MAX).any()
    assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)
    for budget in (0, -5):
        probes, md, _ = router.query_batch(queries, budget)
        wp, wm = tree.query_batch(queries, budget)
        assert (md == FLT_MAX).all() and np.array_equal(probes, wp)


def test_router_duplicate_centroids_tie_order(orc, gp):
    """equal priority-queue keys: the pop order must be libstdc++'s (reference kmeans_tree_router.cpp:94-100)"""
    n, d, S = 6000, 8, 4
    rng = np.random.default_rng(4)
    base = make_mixture(orc, 50, d, n_centers=4)
    pts = base[rng.integers(0, 50, size=n)]          # only 50 distinct points -> duplicated centroids, equal distances
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=16, min_cluster_size=20, budget=1200, metric=0, nthreads=4)
    queries = base[rng.integers(0, 50, size=100)]
    router = gp.Router(tree.flat(), 0, 0)
    for budget in (30, 100, 400):
        probes, md, _ = router.query_batch(queries, budget)
        wp, wm = tree.query_batch(queries, budget)
        assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)


def test_router_wide_nodes(orc, gp):
    """nodes with more than 64 centroids (the reference also uses 128, routes.cpp:130-157)"""
    pts, clusters, tree = _tree(orc, 20000, 24, 3, 0, 128, 60, 3000)
    f = tree.flat()
    assert np.max(np.diff(f["node_cent_first"])) > 64
    queries = make_mixture(orc, 80, 24, seed=556, n_centers=24)
    router = gp.Router(f, 0, 0)
    for budget in (200, 5000):
        probes, md, _ = router.query_batch(queries, budget)
        wp, wm = tree.query_batch(queries, budget, nthreads=4)
        assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)


def test_end_to_end_route_scan_merge_recall(orc, gp):
    """small_scale_queries.cpp:93-165 end to end: route -> QueryBucket per probe -> DedupNeighbors -> Recall"""
    n, d, S, nq, k = 20000, 32, 6, 200, 10
    pts = make_mixture(orc, n, d, n_centers=32)
    part, _ = orc.kmeans(pts, S, 555, 0, nthreads=4)
    clusters = orc.partition_to_clusters(part)
    tree = orc.kmtr_train(pts, clusters, num_centroids=32, min_cluster_size=200, budget=2000, metric=0, nthreads=4)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=32)
    router = gp.Router(tree.flat(), 0, 0)
    probes, _, _ = router.query_batch(queries, 5000)
    wp, _ = tree.query_batch(queries, 5000, nthreads=4)
    assert np.array_equal(probes, wp)
    shards = gp.Shards(pts, clusters, 0, 0)
    ids, dist, _ = shards.scan_buckets(queries, k, probes, S)
    gt_ids, gt_dist, _ = gp.ground_truth(pts, queries, k, 0)
    kth = gt_dist[:, k - 1]
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, wp, S, 0, nthreads=4)
    prev = 0.0
    for nprobe in range(1, S + 1):
        md, mi, mc = gp.merge_dedup(dist, ids, nprobe, k)
        # oracle merge over the same prefix
        offs = np.arange(nq + 1, dtype=np.uint64) * (nprobe * k)
        od, oi, oc = orc.dedup_neighbors(wd[:, :nprobe].reshape(-1), wi[:, :nprobe].reshape(-1), offs, k)
        assert np.array_equal(mi, oi) and np.array_equal(md.view(np.uint32), od.view(np.uint32)) and np.array_equal(mc, oc)
        rec = orc.recall(md.reshape(-1), np.arange(nq + 1, dtype=np.uint64) * k, kth, k)
        assert rec >= prev - 1e-12
        prev = rec
    assert abs(prev - 1.0) < 1e-6        # all shards probed: exact


@pytest.mark.parametrize("metric", [0, 1])
def test_router_deep_tree_threshold_replay(orc, gp, metric, monkeypatch):
    """multi-level trees with budgets that stop the walk in the middle of the tree: the queue-free threshold replay (dense
    distances -> path maxima -> radix select of the cut-off key -> cut-off group) must visit exactly the nodes the
    reference's priority queue pops"""
    pts, clusters, tree = _tree(orc, 30000, 16, 6, metric, 6, 12, 4000, seed=3)
    f = tree.flat()
    C = int(f["node_cent_first"][-1])
    depth_nodes = int((f["node_child_count"] > 0).sum())
    assert depth_nodes > 6 + 10 and C > 1500          # inner nodes below the roots
    queries = make_mixture(orc, 300, 16, seed=556, n_centers=24)
    router = gp.Router(f, metric, 0)
    for mode in ("", "heap"):                          # default (threshold replay) and the real queue on the same table
        router.set_option("GPANN_ROUTER_REPLAY", mode)
        for budget in (C // 4 + 1, C // 3, C // 2, (3 * C) // 4, C - 1, C, C + 50):
            probes, md, _ = router.query_batch(queries, budget)
            wp, wm = tree.query_batch(queries, budget, nthreads=8)
            assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)), (mode, budget)
            assert np.array_equal(probes, wp), (mode, budget)
    router.set_option("GPANN_ROUTER_REPLAY", None)
    # FrequencyQuery over the same walk
    budget = C // 2
    md, nd, ns, nc, _ = router.frequency_query_batch(queries, budget, 50)
    for q in range(0, len(queries), 7):
        wmd, wnd, wns = tree.frequency_query(queries[q], budget, 50)
        assert np.array_equal(md[q].view(np.uint32), wmd.view(np.uint32))
        assert nc[q] == len(wnd) and np.array_equal(nd[q, :nc[q]].view(np.uint32), wnd.view(np.uint32)) and np.array_equal(ns[q, :nc[q]], wns)


def test_router_deep_tree_with_duplicates(orc, gp):
    """equal keys everywhere (50 distinct points): the threshold replay must hand every ambiguous query to the real queue"""
    n, d, S = 12000, 8, 4
    rng = np.random.default_rng(8)
    base = make_mixture(orc, 50, d, n_centers=4)
    pts = base[rng.integers(0, 50, size=n)] + (rng.integers(0, 3, size=(n, 1)) * np.float32(0.25)).astype(np.float32)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=5, min_cluster_size=10, budget=3000, metric=0, nthreads=4)
    f = tree.flat()
    C = int(f["node_cent_first"][-1])
    queries = base[rng.integers(0, 50, size=150)]
    router = gp.Router(f, 0, 0)
    for budget in (C // 3, C // 2, C - 3):
        probes, md, _ = router.query_batch(queries, budget)
        wp, wm = tree.query_batch(queries, budget, nthreads=4)
        assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp), budget


def test_frequency_query_vote_selection_ties_and_short_lists(orc, gp):
    """device-side TopN of the votes: equal (distance, shard) pairs, fewer evaluated centroids than votes asked for"""
    n, d, S = 8000, 8, 4
    rng = np.random.default_rng(12)
    base = make_mixture(orc, 40, d, n_centers=4)
    pts = base[rng.integers(0, 40, size=n)]
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=16, min_cluster_size=20, budget=800, metric=0, nthreads=4)
    queries = base[rng.integers(0, 40, size=60)]
    router = gp.Router(tree.flat(), 0, 0)
    for budget, votes in ((60, 500), (100000, 37), (100000, 2048)):
        md, nd, ns, nc, _ = router.frequency_query_batch(queries, budget, votes)
        for q in range(len(queries)):
            wmd, wnd, wns = tree.frequency_query(queries[q], budget, votes)
            assert np.array_equal(md[q].view(np.uint32), wmd.view(np.uint32))
            assert nc[q] == len(wnd), (budget, votes, q)
            assert np.array_equal(nd[q, :nc[q]].view(np.uint32), wnd.view(np.uint32)) and np.array_equal(ns[q, :nc[q]], wns)


def test_router_bound_self_check_falls_back_to_canonical_evaluation(orc, gp, monkeypatch, capfd):
    """the matrix-core bound path re-evaluates every pair canonically when its self-check fires (here: forced) — same results"""
    rng = np.random.default_rng(5)
    n, d, S = 12000, 32, 5
    pts = make_mixture(orc, n, d, n_centers=20)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, 300, d, seed=556, n_centers=20)
    tree = orc.kmtr_train(pts, clusters, 64, 350, 20000, 0, nthreads=4)
    want_p, want_md = tree.query_batch(queries, 50000, nthreads=4)
    router = gp.Router(tree.flat(), 0)
    router.set_option("GPANN_ROUTER_FORCE_RECHECK", "1")
    p, md, _ = router.query_batch(queries, 50000)
    assert np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
    assert "bound path self-check" in capfd.readouterr().err
    st = router.stats()
    assert st["bound_calls"] >= 1 and st["bound_selfcheck"] >= 1 and st["verify_failures"] == 0


def test_bound_path_is_verified_and_counted(orc, gp):
    """the session runs with GPANN_ROUTER_VERIFY (conftest): every bound block is compared with the canonical evaluation inside the library;
    the handle's counters say that it happened and that nothing differed"""
    rng = np.random.default_rng(8)
    n, d, S = 30000, 128, 9
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, 64, 350, 9000, 0, nthreads=4)
    queries = make_mixture(orc, 777, d, seed=556, n_centers=24)     # an odd count: partly filled query tiles, odd nq * S
    want_p, want_md = tree.query_batch(queries, 10 ** 6, nthreads=4)
    router = gp.Router(tree.flat(), 0)
    for _ in range(3):
        p, md, _ = router.query_batch(queries, 10 ** 6)
        assert np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
    st = router.stats()
    assert st["bound_calls"] == 3 and st["verified"] == 3 and st["verify_failures"] == 0 and st["bound_selfcheck"] == 0 and st["bound_overflow"] == 0
    assert st["last_bound_ms"] > 0 and st["last_bound_flops"] > 0


def test_router_create_rejects_shards_out_of_order(orc, gp):
    """the kernels rely on ExtractPoints order (nodes grouped per shard, shards ascending, the root first): anything else is refused"""
    rng = np.random.default_rng(9)
    pts = make_mixture(orc, 6000, 16, n_centers=8)
    clusters = random_clusters(rng, 6000, 3)
    flat = orc.kmtr_train(pts, clusters, 16, 50, 600, 0, nthreads=2).flat()
    bad = dict(flat)
    bad["node_shard"] = flat["node_shard"].copy()
    bad["node_shard"][-1] = 0                       # the last node claims shard 0: shards no longer ascending
    with pytest.raises(gp.GpannError):
        gp.Router(bad, 0)
    bad = dict(flat)
    bad["shard_root"] = flat["shard_root"].copy()
    bad["shard_root"][1] += 1                       # not the first node of its shard
    with pytest.raises(gp.GpannError):
        gp.Router(bad, 0)


def test_bound_path_without_verify_defers_its_counter_check(orc, gp, monkeypatch):
    """production setting (no GPANN_ROUTER_VERIFY): the bound block's counters are looked at after the call's single synchronisation"""
    monkeypatch.delenv("GPANN_ROUTER_VERIFY", raising=False)
    rng = np.random.default_rng(10)
    n, d, S = 30000, 96, 7
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, 64, 350, 9000, 0, nthreads=4)
    queries = make_mixture(orc, 1500, d, seed=556, n_centers=24)
    want_p, want_md = tree.query_batch(queries, 10 ** 6, nthreads=4)
    router = gp.Router(tree.flat(), 0)
    for want_min in (True, False):
        p, md, _ = router.query_batch(queries, 10 ** 6, want_min_dist=want_min)
        assert np.array_equal(p, want_p)
        if want_min:
            assert np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
    st = router.stats()
    assert st["bound_calls"] == 2 and st["verified"] == 0 and st["bound_selfcheck"] == 0 and st["bound_overflow"] == 0 and st["last_bound_ms"] > 0


def test_query_batch_top_returns_the_first_columns(orc, gp):
    """gpann_router_query_batch_top: only the first n_top shards of every order come back — the same shards, in the same order, as the
    full call returns, including for queries whose order the host has to settle (equal min_dist values: duplicated centroids)"""
    rng = np.random.default_rng(11)
    n, d, S = 20000, 32, 9
    pts = make_mixture(orc, n, d, n_centers=16)
    pts[::4] = pts[0]                                   # duplicates -> duplicate centroids -> equal min_dist values for some queries
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, 32, 100, 3000, 0, nthreads=4)
    queries = make_mixture(orc, 400, d, seed=556, n_centers=16)
    queries[:20] = pts[0]
    router = gp.Router(tree.flat(), 0)
    for budget in (10 ** 6, 700):
        full, md, _ = router.query_batch(queries, budget)
        want_p, want_md = tree.query_batch(queries, budget, nthreads=4)
        assert np.array_equal(full, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
        for n_top in (1, 3, S):
            top, _, _ = router.query_batch(queries, budget, n_top=n_top, want_min_dist=False)
            assert top.shape == (400, n_top) and np.array_equal(top, full[:, :n_top])


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [130, 200, 256])
def test_bound_path_beyond_128_dimensions(orc, gp, metric, d):
    """129 - 256 dimensions: the router's matrix-core bound with 16 k-step centroid tiles and 4 query tiles per workgroup (C4's
    200-dimensional inner-product router runs through it); every minimum verified against the canonical evaluation (conftest)"""
    rng = np.random.default_rng(d + metric)
    n, S = 24000, 7
    pts = make_mixture(orc, n, d, n_centers=20)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, 64, 200, 5000, metric, nthreads=4)
    queries = make_mixture(orc, 333, d, seed=556, n_centers=20)          # 333: partly filled query tiles and groups
    want_p, want_md = tree.query_batch(queries, 10 ** 6, nthreads=4)
    router = gp.Router(tree.flat(), metric)
    p, md, _ = router.query_batch(queries, 10 ** 6)
    assert np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
    st = router.stats()
    assert st["bound_calls"] == 1 and st["verified"] == 1 and st["verify_failures"] == 0 and st["bound_overflow"] == 0 and st["bound_selfcheck"] == 0


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d,S,num_centroids", [(32, 5, 16), (100, 60, 8), (128, 40, 64), (200, 7, 32)])
def test_shipping_configuration_of_the_bound_path(orc, gp, metric, d, S, num_centroids, monkeypatch):
    """What ships is the bound path WITHOUT GPANN_ROUTER_VERIFY (the session arms it, conftest.py): only the always-on checks run — pass 1
    left a maximum and a finite threshold for every (query, shard-with-centroids), pass 2 a minimum — and must stay silent (STRICT stays
    armed: a fired check would fail the call).  Same parity bar as the verified runs, first call of a fresh handle included."""
    monkeypatch.delenv("GPANN_ROUTER_VERIFY", raising=False)
    pts, clusters, tree = _tree(orc, 30000, d, S, metric, num_centroids, 40, 6000, seed=d + metric)
    queries = make_mixture(orc, 700, d, seed=556, n_centers=24)
    want_p, want_md = tree.query_batch(queries, 10 ** 6, nthreads=4)
    for fresh in range(2):
        router = gp.Router(tree.flat(), metric, 0)
        for _ in range(2):
            p, md, _ = router.query_batch(queries, 10 ** 6)          # budget covers the tree: the matrix-core bound path
            assert np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
            assert np.array_equal(p, want_p)
        st = router.stats()
        assert st["bound_calls"] == 2 and st["verified"] == 0, st
        assert st["bound_selfcheck"] == 0 and st["bound_overflow"] == 0 and st["verify_failures"] == 0, st
        router.close()


def test_bound_evidence_dump_and_determinism_probe(orc, gp, monkeypatch, tmp_path):
    """The diagnostics of the bound path: GPANN_ROUTER_PROBE runs every block twice and compares every bit; an event would
    leave BoundEvidence — text + a binary dump of both runs' arrays and the canonical minima.  GPANN_ROUTER_FORCE_EVIDENCE (a test hook) writes
    that evidence for a healthy block: the dump must parse, its two runs must agree bit for bit and equal the canonical minima."""
    import glob
    import struct
    report = str(tmp_path / "verify_report.txt")
    monkeypatch.setenv("GPANN_VERIFY_REPORT", report)
    monkeypatch.setenv("GPANN_ROUTER_PROBE", "1")
    monkeypatch.setenv("GPANN_ROUTER_FORCE_EVIDENCE", "1")
    d, S, metric = 32, 5, 0
    pts, clusters, tree = _tree(orc, 20000, d, S, metric, 16, 40, 4000, seed=91)
    queries = make_mixture(orc, 200, d, seed=556, n_centers=24)       # the shape of the reference driver's call in the drop-in test
    want_p, want_md = tree.query_batch(queries, 10 ** 6, nthreads=4)
    router = gp.Router(tree.flat(), metric, 0)
    p, md, _ = router.query_batch(queries, 10 ** 6)
    assert np.array_equal(md.view(np.uint32), want_md.view(np.uint32)) and np.array_equal(p, want_p)
    st = router.stats()
    assert st["probed"] >= 1 and st["probe_failures"] == 0 and st["verify_failures"] == 0 and st["bound_selfcheck"] == 0, st
    text = open(report).read()
    assert "EVIDENCE" in text and "forced" in text and " 0 of " in text, text[:600]
    dumps = glob.glob(report + ".*.bin")
    assert len(dumps) >= 1
    raw = open(dumps[0], "rb").read()
    assert raw[:8] == b"GPANNBD1"
    nb, S_, d_, ld, n_seg, cand_cap, metric_, plane_ks = struct.unpack_from("<8I", raw, 8)
    (C,) = struct.unpack_from("<Q", raw, 40)
    counters1 = struct.unpack_from("<8I", raw, 48)
    counters2 = struct.unpack_from("<8I", raw, 80)
    n_cand1, n_cand2 = struct.unpack_from("<2I", raw, 112)
    assert (nb, S_, d_, metric_) == (200, S, d, metric) and C == tree.flat()["centroids"].shape[0]
    off = 120
    nb_s = nb * S_

    def take(n_words, dtype):
        nonlocal off
        a = np.frombuffer(raw, dtype, n_words, off)
        off += 4 * n_words
        return a
    runs = []
    for n_cand in (n_cand1, n_cand2):
        max_bits, thr, min_bits, q = take(nb_s, np.uint32), take(nb_s, np.uint32), take(nb_s, np.uint32), take(nb * ld, np.float32)
        cand = take(2 * n_cand, np.uint32).reshape(-1, 2)
        runs.append((max_bits, thr, min_bits, q, cand))
    want = take(nb_s, np.uint32)
    assert off == len(raw)
    for a, b in zip(runs[0][:4], runs[1][:4]):
        assert np.array_equal(a, b)                                   # maxima, thresholds, minima, staged queries: bit-equal between the runs
    def cand_set(c):
        v = c[c[:, 0] != 0xFFFFFFFF]
        return np.sort(v[:, 0].astype(np.uint64) << np.uint64(32) | v[:, 1].astype(np.uint64))
    assert np.array_equal(cand_set(runs[0][4]), cand_set(runs[1][4]))   # the candidate SETS (slot order is not defined)
    assert np.array_equal(runs[0][2], want)                           # the minima equal the canonical evaluation's (ordered bits)
    assert counters1[1] == 0 and counters1[4] == 0 and counters2[1] == 0
    assert np.array_equal(runs[0][3].reshape(nb, ld)[:, :d], queries)
    router.close()


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d,S,num_centroids,n", [(16, 6, 6, 30000), (100, 9, 8, 40000), (128, 12, 16, 60000), (200, 5, 6, 24000)])
def test_bounded_table_replay_matches_the_canonical_table(orc, gp, metric, d, S, num_centroids, n, monkeypatch):
    """search budgets that stop the walk inside a multi-level tree: the distance table comes from the matrix cores (RouterBoundKernel pass 3)
    and only the cut-off band and the per-shard minimum candidates get the canonical evaluation (RouterBoundedReplayKernel) — min_dist bits
    and probes must equal the oracle's and those of the canonical table; a band that does not fit goes to the real queue on a canonical row"""
    pts, clusters, tree = _tree(orc, n, d, S, metric, num_centroids, 12, n // 6, seed=5)
    f = tree.flat()
    C = int(f["node_cent_first"][-1])
    assert int((f["node_child_count"] > 0).sum()) > S + 5 and C > 1200
    queries = make_mixture(orc, 257, d, seed=557, n_centers=24)
    router = gp.Router(f, metric, 0)
    depth = np.zeros(len(f["node_shard"]), np.int64)   # (nodes are numbered parents first)
    for u in range(len(depth)):
        depth[f["node_child_first"][u]:f["node_child_first"][u] + f["node_child_count"][u]] = depth[u] + 1
    expect_table = depth.max() <= 6                      # kPathMax: deeper trees (inner-product k-means can be very lopsided) keep the queue
    assert expect_table or metric == 1
    for budget in (C // 5, C // 3, C // 2, (3 * C) // 4, C - 1):
        wp, wm = tree.query_batch(queries, budget, nthreads=8)
        before = router.stats()
        probes, md, _ = router.query_batch(queries, budget)
        st = router.stats()
        if expect_table:
            assert st["table_blocks"] == before["table_blocks"] + 1 and st["table_queries"] == before["table_queries"] + len(queries), st
            assert st["table_band_overflow"] == before["table_band_overflow"], st
            assert st["table_exact"] > before["table_exact"]
        assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)), budget
        assert np.array_equal(probes, wp), budget
        router.set_option("GPANN_ROUTER_BOUND_TABLE", "0")
        p0, m0, _ = router.query_batch(queries, budget)
        assert router.stats()["table_blocks"] == st["table_blocks"]
        router.set_option("GPANN_ROUTER_BOUND_TABLE", None)
        assert np.array_equal(m0.view(np.uint32), md.view(np.uint32)) and np.array_equal(p0, probes), budget
    if not expect_table:
        return
    # FrequencyQuery through the same path: only the pairs within the error of the num_voting-th smallest table value are evaluated
    # (C // 5, 2048): fewer visited centroids than votes; C + 50: the budget covers the tree — every centroid is visited, no tree logic at all
    for budget, votes in ((C // 2, 500), (C // 3, 7), (C // 5, 2048), (C + 50, 500), (C + 50, 3)):
        before = router.stats()
        md, nd, ns, nc, _ = router.frequency_query_batch(queries, budget, votes)
        st = router.stats()
        assert st["table_blocks"] == before["table_blocks"] + 1, (budget, votes, st)
        for q in range(0, len(queries), 5):
            wmd, wnd, wns = tree.frequency_query(queries[q], budget, votes)
            assert np.array_equal(md[q].view(np.uint32), wmd.view(np.uint32)), (budget, votes, q)
            assert nc[q] == len(wnd) and np.array_equal(nd[q, :nc[q]].view(np.uint32), wnd.view(np.uint32)) and np.array_equal(ns[q, :nc[q]], wns), (budget, votes, q)
    # ... with room for few records per query (what the full-size trees get: the batch in one block): enough room, and too little (the call is repeated)
    for cap, votes in (("600", 500), ("56", 50), ("8", 7)):
        router.set_option("GPANN_ROUTER_FREQ_SMALL", cap)
        md, nd, ns, nc, _ = router.frequency_query_batch(queries, C // 2, votes)
        for q in range(0, len(queries), 7):
            wmd, wnd, wns = tree.frequency_query(queries[q], C // 2, votes)
            assert np.array_equal(md[q].view(np.uint32), wmd.view(np.uint32)), (cap, votes, q)
            assert nc[q] == len(wnd) and np.array_equal(nd[q, :nc[q]].view(np.uint32), wnd.view(np.uint32)) and np.array_equal(ns[q, :nc[q]], wns), (cap, votes, q)
    router.set_option("GPANN_ROUTER_FREQ_SMALL", None)
    # a 30 times wider band (the error bound inflated): many more canonical evaluations, the same bits
    budget = C // 2
    wp, wm = tree.query_batch(queries, budget, nthreads=8)
    router.set_option("GPANN_ROUTER_EPS_SCALE", "30")
    before = router.stats()
    probes, md, _ = router.query_batch(queries, budget)
    st = router.stats()
    assert st["table_band_nodes"] - before["table_band_nodes"] > 3 * before["table_band_nodes"] / 5 / 2, (before, st)
    assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)
    # ... and with a tiny band list on top: overflowing queries are marked and replayed on canonical rows — same bits
    router.set_option("GPANN_ROUTER_EPS_SCALE", "300")
    router.set_option("GPANN_ROUTER_BAND_CAP", "96")
    before = router.stats()
    probes, md, _ = router.query_batch(queries, budget)
    st = router.stats()
    assert st["table_band_overflow"] > before["table_band_overflow"], st
    assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)
    # ... and with the device queue instead of the host's
    router.set_option("GPANN_ROUTER_HOST_REDO", "0")
    probes, md, _ = router.query_batch(queries, budget)
    assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp)


def test_bounded_table_replay_with_duplicate_points(orc, gp):
    """equal keys everywhere: every ambiguous query of the bounded replay must reach the real queue with canonical values"""
    n, d, S = 12000, 8, 4
    rng = np.random.default_rng(8)
    base = make_mixture(orc, 50, d, n_centers=4)
    pts = base[rng.integers(0, 50, size=n)] + (rng.integers(0, 3, size=(n, 1)) * np.float32(0.25)).astype(np.float32)
    clusters = random_clusters(rng, n, S)
    tree = orc.kmtr_train(pts, clusters, num_centroids=5, min_cluster_size=10, budget=3000, metric=0, nthreads=4)
    f = tree.flat()
    C = int(f["node_cent_first"][-1])
    queries = np.concatenate([base[rng.integers(0, 50, size=150)], make_mixture(orc, 50, d, seed=9, n_centers=4)])
    router = gp.Router(f, 0, 0)
    for budget in (C // 3, C // 2, C - 3):
        probes, md, _ = router.query_batch(queries, budget)
        wp, wm = tree.query_batch(queries, budget, nthreads=4)
        assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)) and np.array_equal(probes, wp), budget
    assert router.stats()["table_blocks"] == 3
    # FrequencyQuery on the same tree of equal keys: inside the tree and with a budget that covers it (votes among masses of equal distances)
    for budget, votes in ((C // 2, 40), (C + 10, 40), (C + 10, 500)):
        md, nd, ns, nc, _ = router.frequency_query_batch(queries, budget, votes)
        for q in range(0, len(queries), 9):
            wmd, wnd, wns = tree.frequency_query(queries[q], budget, votes)
            assert np.array_equal(md[q].view(np.uint32), wmd.view(np.uint32)), (budget, votes, q)
            assert nc[q] == len(wnd) and np.array_equal(nd[q, :nc[q]].view(np.uint32), wnd.view(np.uint32)) and np.array_equal(ns[q, :nc[q]], wns), (budget, votes, q)
    assert router.stats()["table_blocks"] == 6


@pytest.mark.parametrize("metric", [0, 1])
def test_bounded_table_replay_budget_and_batch_edges(orc, gp, metric):
    """the bounded replay at its edges: budgets around the roots' total (nothing / almost nothing below the roots is visited), budgets C - 1 and C
    (the last one visits everything: no cut-off), batches of 1, 31, 33 and 300 queries (partly filled query tiles), one query twice in a batch"""
    pts, clusters, tree = _tree(orc, 36000, 48, 7, metric, 8, 12, 6000, seed=11)
    f = tree.flat()
    C = int(f["node_cent_first"][-1])
    roots = f["shard_root"]
    root_sizes = (f["node_cent_first"][roots + 1] - f["node_cent_first"][roots]).astype(np.int64)
    root_total, root_min = int(root_sizes.sum()), int(root_sizes.min())
    queries = make_mixture(orc, 300, 48, seed=558, n_centers=24)
    queries[7] = queries[3]
    router = gp.Router(f, metric, 0)
    for budget in (root_total - root_min + 1, root_total, root_total + 1, root_total + 2, root_total + 40, C - 1, C):
        for nq in (1, 31, 33, 300):
            probes, md, _ = router.query_batch(queries[:nq], budget)
            wp, wm = tree.query_batch(queries[:nq], budget, nthreads=4)
            assert np.array_equal(md.view(np.uint32), wm.view(np.uint32)), (budget, nq)
            assert np.array_equal(probes, wp), (budget, nq)
    st = router.stats()
    assert st["verify_failures"] == 0 and st["bound_selfcheck"] == 0, st
