"""GPU: the pass structure of the filtered scan on stores with SMALL shards (round 6).

Stores whose shards average fewer than GPANN_FILTER_SMALL_SHARD_ROWS rows (default 262144) run without the shard-sample and the head pass — the two launches
that read both bf16 planes (`gpann_scan_stats.filter_two_plane_rows`) — unless GPANN_FILTER_SHARD_SAMPLE / GPANN_FILTER_HEAD_ROWS ask for them, and cells go
down to 512 rows (csrc/scan.hip RunCellFilteredScan, csrc/shards.hip DefaultCellRows: rows per shard / 256; measured on the C2 shape: profiles/r06/sweeps/small_store_passes.txt).
Whatever the pass structure, ids and distances are the oracle's bits.  tests/conftest.py switches the rule off for the session, so every test here sets
it explicitly.  The rule also looks at how well the cells pruned in the store's previous call (GPANN_FILTER_SMALL_SHARD_MAX_SHARE)."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu


def _want(orc, pts, clusters, queries, k, probes, nprobe, metric):
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, metric, nthreads=4)
    return wd, wi


def _scan(shards, queries, k, probes, nprobe, wd, wi, tag):
    ids, dist, _ = shards.scan_buckets(queries, k, probes, nprobe)
    assert np.array_equal(ids, wi), tag
    assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32)), tag
    st = shards.stats()
    assert st["filtered"] == 1, (tag, st)
    return st


@pytest.mark.parametrize("metric,d", [(0, 128), (0, 64), (1, 96), (0, 200), (1, 200)])
def test_small_shards_skip_the_two_plane_passes_by_default(orc, gp, monkeypatch, metric, d):
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)   # the library's default rule
    monkeypatch.setenv("GPANN_FILTER_SMALL_SHARD_MAX_SHARE", "100000")    # (this test is about the size rule alone: whatever the previous call's pruning was)
    rng = np.random.default_rng(900 + d + metric)
    n, S, nq, k, nprobe = 120000, 4, 500, 10, 2
    pts = make_mixture(orc, n, d, n_centers=48)
    clusters = random_clusters(rng, n, S, overlap=0.1)
    queries = make_mixture(orc, nq, d, seed=557, n_centers=48)
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    wd, wi = _want(orc, pts, clusters, queries, k, probes, nprobe, metric)
    shards = gp.Shards(pts, clusters, metric, 0)
    shards.set_scan_mode(1)
    st = _scan(shards, queries, k, probes, nprobe, wd, wi, "default")
    assert st["filter_two_plane_rows"] == 0, st            # neither the shard-sample nor the head pass ran
    assert st["filter_rows"] > 0 and st["filter_seed_rows"] > 0
    cand_small = st["filter_candidates"]
    # the rule is an option of the handle: 0 = never -> the five passes of the large stores, same bits
    shards.set_option("FILTER_SMALL_SHARD_ROWS", "0")
    st = _scan(shards, queries, k, probes, nprobe, wd, wi, "rule off")
    assert st["filter_two_plane_rows"] > 0 or d > 128, st   # (beyond 128 dimensions every pass reads the hi plane only: nothing counts as two-plane rows)
    assert st["filter_candidates"] <= cand_small * 4 + 1000   # (the extra passes tighten the thresholds; no claim beyond sanity)
    # ... and with the rule on, each pass can be asked for by its own knob
    shards.set_option("FILTER_SMALL_SHARD_ROWS", None)
    shards.set_option("FILTER_HEAD_ROWS", "512")
    st_head = _scan(shards, queries, k, probes, nprobe, wd, wi, "head only")
    shards.set_option("FILTER_HEAD_ROWS", None)
    shards.set_option("FILTER_SHARD_SAMPLE", "1")
    st_ss = _scan(shards, queries, k, probes, nprobe, wd, wi, "shard sample only")
    if d <= 128:
        assert st_head["filter_two_plane_rows"] > 0 and st_ss["filter_two_plane_rows"] > 0
    shards.close()


def test_the_rule_is_about_the_average_shard(orc, gp, monkeypatch):
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)
    monkeypatch.setenv("GPANN_FILTER_SMALL_SHARD_MAX_SHARE", "100000")    # (this test is about the size rule alone: whatever the previous call's pruning was)
    rng = np.random.default_rng(77)
    n, S, nq, k, d = 60000, 3, 200, 10, 64
    pts = make_mixture(orc, n, d, n_centers=16)
    part = rng.integers(0, S, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]   # a partition: rows = n exactly
    queries = make_mixture(orc, nq, d, seed=558, n_centers=16)
    probes = np.stack([rng.permutation(S)[:2] for _ in range(nq)]).astype(np.int32)
    wd, wi = _want(orc, pts, clusters, queries, k, probes, 2, 0)
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(1)
    shards.set_option("FILTER_SMALL_SHARD_ROWS", str(n // S))        # rows / S = 20000 is NOT below 20000: the large-store passes
    assert _scan(shards, queries, k, probes, 2, wd, wi, "at the bound")["filter_two_plane_rows"] > 0
    shards.set_option("FILTER_SMALL_SHARD_ROWS", str(n // S + 1))    # ... below 20001: small
    assert _scan(shards, queries, k, probes, 2, wd, wi, "below the bound")["filter_two_plane_rows"] == 0
    shards.close()


def test_the_rule_holds_only_while_the_cells_prune(orc, gp, monkeypatch):
    """a store's first filtered call takes the rule; from then on it applies while the previous call sent at most GPANN_FILTER_SMALL_SHARD_MAX_SHARE per cent of
    the (task, row) pairs it was asked for to the matrix cores: a clustered cloud keeps it, one blob — every row reaches the matrix cores — gets the two passes
    back from its second call on (their tighter thresholds save more re-scoring than their launches cost: profiles/r06/sweeps/small_store_passes.txt)"""
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_MAX_SHARE", raising=False)
    rng = np.random.default_rng(80)
    n, S, nq, k, d = 160000, 2, 600, 10, 64
    for centers, sigma, keeps in ((160, 0.2, True), (1, 1.0, False)):
        pts = make_mixture(orc, n, d, n_centers=centers, sigma=sigma)
        part = rng.integers(0, S, size=n)
        clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
        queries = make_mixture(orc, nq, d, seed=562, n_centers=centers, sigma=sigma)
        probes = np.stack([rng.permutation(S)[:1] for _ in range(nq)]).astype(np.int32)
        wd, wi = _want(orc, pts, clusters, queries, k, probes, 1, 0)
        shards = gp.Shards(pts, clusters, 0, 0)
        shards.set_scan_mode(1)
        st = _scan(shards, queries, k, probes, 1, wd, wi, ("first call", centers))
        assert st["filter_two_plane_rows"] == 0, st            # no history: the rule
        share = st["filter_pairs"] / st["pairs"]
        assert (share <= 0.5) == keeps, (centers, share)          # (what the test's two clouds are chosen for)
        for rep in range(2):
            st = _scan(shards, queries, k, probes, 1, wd, wi, ("later call", centers, rep))
            assert (st["filter_two_plane_rows"] == 0) == keeps, (centers, rep, st)
        shards.set_option("FILTER_SMALL_SHARD_MAX_SHARE", "100000")   # the share is an option of the handle like the size
        assert _scan(shards, queries, k, probes, 1, wd, wi, ("share lifted", centers))["filter_two_plane_rows"] == 0
        shards.close()


def test_small_shards_without_structure_and_with_overflowing_buckets(orc, gp, monkeypatch):
    """no cluster structure: the cells prune nothing and, without the passes that tighten the thresholds first, the sample pass meets looser ones — more
    candidates, with tiny buckets dirty tasks or the whole-call fallback; the lists are the oracle's either way"""
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)
    rng = np.random.default_rng(78)
    n, S, nq, k, d = 90000, 3, 300, 10, 128
    pts = make_mixture(orc, n, d, n_centers=1, sigma=1.0)
    pts[: n // 5] = pts[0]                                     # masses of equal distances
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=559, n_centers=1, sigma=1.0)
    queries[0] = pts[0]
    probes = np.stack([rng.permutation(S)[:2] for _ in range(nq)]).astype(np.int32)
    wd, wi = _want(orc, pts, clusters, queries, k, probes, 2, 0)
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(1)
    for cap in (None, "64"):
        shards.set_option("FILTER_CAND_CAP", cap)
        ids, dist, _ = shards.scan_buckets(queries, k, probes, 2)
        assert np.array_equal(ids, wi), cap
        assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32)), cap
    shards.close()


def test_default_cells_of_a_small_store_go_down_to_512_rows(orc, gp, monkeypatch):
    """125k rows per shard / 256 = 488 -> the 512-row floor (until round 6: 2048): on a mixture whose components are ~1000 rows the smaller cells are what lets the
    ball test prune — fewer (task, row) pairs reach the matrix cores than with cells of 2048 rows, same lists"""
    monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)
    monkeypatch.delenv("GPANN_CELL_ROWS", raising=False)
    rng = np.random.default_rng(79)
    n, S, nq, k, d = 250000, 2, 400, 10, 64
    pts = make_mixture(orc, n, d, n_centers=256)               # ~977 rows per component
    part = rng.integers(0, S, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
    queries = make_mixture(orc, nq, d, seed=560, n_centers=256)
    probes = np.stack([rng.permutation(S)[:1] for _ in range(nq)]).astype(np.int32)
    wd, wi = _want(orc, pts, clusters, queries, k, probes, 1, 0)
    pairs = {}
    for cell_rows in (None, "2048"):
        if cell_rows is None:
            monkeypatch.delenv("GPANN_CELL_ROWS", raising=False)
        else:
            monkeypatch.setenv("GPANN_CELL_ROWS", cell_rows)
        shards = gp.Shards(pts, clusters, 0, 0)
        shards.set_scan_mode(1)
        pairs[cell_rows] = _scan(shards, queries, k, probes, 1, wd, wi, cell_rows)["filter_pairs"]
        shards.close()
    assert pairs[None] < pairs["2048"], pairs
