"""GPU parity of GPANN_SCAN_FOR_MERGE (include/gpann.h): per-slot lists cut at the query's merge bound.  Contract: slot j of query q is
the reference's QueryBucket list (src/inverted_index.h:62-75) truncated after its last entry with distance <= tau_q, tau_q = min over
the query's slots of the slot's k-th distance; the merge over the slots (DedupNeighbors, small_scale_queries.cpp:15-23) is bit-identical
to the merge over the full lists.  Checked against the CPU oracle's full lists, in exact and in filter mode."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu

NO_ID = 0xFFFFFFFF


def _truncate(wd, wi):
    """the contract, on the oracle's full lists [nq][nprobe][k]"""
    tau = wd[:, :, -1].min(axis=1)                      # +inf where a slot is short / unused
    cut = ~(wd <= tau[:, None, None])
    td, ti = wd.copy(), wi.copy()
    td[cut] = np.inf
    ti[cut] = NO_ID
    return td, ti, int((cut & (wi != NO_ID)).sum())


def _merged(orc, dist, ids, k):
    nq = dist.shape[0]
    fd, fi = dist.reshape(nq, -1), ids.reshape(nq, -1)
    keep = fi != NO_ID
    offs = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.uint64)
    return orc.dedup_neighbors(fd[keep], fi[keep], offs, k)


def _check(orc, gp, pts, clusters, queries, probes, nprobe, k, metric, modes=(0, 1), check_filtered=True):
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, metric, nthreads=4)
    td, ti, n_cut = _truncate(wd, wi)
    want_merged = _merged(orc, wd, wi, k)                # over the FULL lists: what the reference returns
    shards = gp.Shards(pts, clusters, metric, 0)
    stats = {}
    for mode in modes:
        shards.set_scan_mode(mode)
        ids, dist, rows, _ = shards.scan_buckets(queries, k, probes, nprobe, want_rows=True, for_merge=True)
        stats[mode] = shards.stats()
        assert not check_filtered or stats[mode]["filtered"] == mode
        assert np.array_equal(ids, ti), f"mode {mode}"
        assert np.array_equal(dist.view(np.uint32), td.view(np.uint32)), f"mode {mode}"
        assert np.array_equal(rows == NO_ID, ids == NO_ID)
        got_merged = _merged(orc, dist, ids, k)
        for a, b in zip(got_merged, want_merged):
            a, b = np.asarray(a), np.asarray(b)
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
        # the product's own merge (gpann_merge_dedup) over the cut lists against the same merge over the oracle's full lists
        md, mi, mc = gp.merge_dedup(dist, ids, nprobe, k)
        fd, fi, fc = gp.merge_dedup(wd, wi, nprobe, k)
        assert np.array_equal(mi, fi) and np.array_equal(md.view(np.uint32), fd.view(np.uint32)) and np.array_equal(mc, fc)
    shards.close()
    return n_cut, stats


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [100, 128, 40])
def test_for_merge_lists_are_the_reference_lists_cut_at_the_merge_bound(orc, gp, metric, d, monkeypatch):
    monkeypatch.setenv("GPANN_CELL_ROWS", "1500")
    rng = np.random.default_rng(d + 17 * metric)
    n, S, nq, k, nprobe = 50000, 6, 500, 10, 4
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S, overlap=0.2)        # ids repeat across shards
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    probes[::7, 2] = -1                                        # skipped slots
    n_cut, stats = _check(orc, gp, pts, clusters, queries, probes, nprobe, k, metric)
    assert n_cut > 0                                           # the bound did cut something
    # the bound prunes work in the filtered pipeline: fewer (query, row) pairs than without it
    shards = gp.Shards(pts, clusters, metric, 0)
    shards.set_scan_mode(1)
    shards.scan_buckets(queries, k, probes, nprobe)
    full_pairs = shards.stats()["filter_pairs"]
    shards.close()
    # (not exactly monotone from run to run: the tasks of a shard are grouped into 32-task query tiles by a device-side counting sort whose
    # order inside a cell is not deterministic, and a tile runs on a cell as soon as one of its tasks needs it)
    assert stats[1]["filter_pairs"] <= 1.05 * full_pairs


def test_for_merge_with_ties_short_shards_and_one_probe(orc, gp, monkeypatch):
    monkeypatch.setenv("GPANN_CELL_ROWS", "1500")
    rng = np.random.default_rng(4)
    n, d, S, nq, k = 20000, 32, 5, 200, 10
    pts = make_mixture(orc, n, d)
    pts[500:900] = pts[500]                # 400 identical points: masses of equal distances at the bound
    part = rng.integers(0, 3, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(3)] + [np.array([5, 6, 7], np.uint32), np.zeros(0, np.uint32)]
    queries = make_mixture(orc, nq, d, seed=556)
    queries[0] = pts[500]
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    _check(orc, gp, pts, clusters, queries, probes, 4, k, 0)
    # one probe: nothing can be cut, the lists are the plain QueryBucket lists
    n_cut, _ = _check(orc, gp, pts, clusters, queries, probes, 1, k, 0)
    assert n_cut == 0


@pytest.mark.parametrize("metric", [0, 1])
def test_scan_merged_in_filter_mode_matches_inverted_index_query(orc, gp, metric, monkeypatch):
    """gpann_scan_merged = InvertedIndex::Query (one TopN over the probed shards, no dedupe) runs with the merge bound internally"""
    monkeypatch.setenv("GPANN_CELL_ROWS", "1500")
    rng = np.random.default_rng(8)
    n, d, S, nq, k = 40000, 48, 6, 64, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S, overlap=0.25)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    shards = gp.Shards(pts, clusters, metric, 0)
    ivf = orc.ivf_build(pts, clusters)
    for mode in (0, 1):
        shards.set_scan_mode(mode)
        for nprobe in (1, 3, S):
            ids, dist, _ = shards.scan_merged(queries, k, probes, nprobe)
            for q in range(nq):
                wd, wi = orc.ivf_query(ivf, queries[q], k, probes[q], nprobe, metric)
                assert np.array_equal(ids[q], wi) and np.array_equal(dist[q].view(np.uint32), wd.view(np.uint32))
    shards.close()


@pytest.mark.parametrize("metric", [0, 1])
def test_for_merge_with_skipped_first_slots_and_dirty_tasks(orc, gp, metric, monkeypatch):
    """The corners of the filtered pipeline under the merge bound: queries whose FIRST slot is skipped or is a shard with fewer than k rows
    (only the first slot of a query is seeded — these queries get their bound from the catch-up seed launch), tiny candidate buckets
    (tasks go dirty and are redone exactly; their lists obey the same cut), split buckets."""
    monkeypatch.setenv("GPANN_CELL_ROWS", "1500")
    monkeypatch.setenv("GPANN_FILTER_CAND_CAP", "150")
    monkeypatch.setenv("GPANN_RESCORE_SPLIT_MIN", "64")
    rng = np.random.default_rng(31 + metric)
    n, d, nq, k, nprobe = 40000, 64, 600, 10, 3
    pts = make_mixture(orc, n, d, n_centers=20)
    part = rng.integers(0, 4, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(4)] + [np.array([11, 12, 13, 14], np.uint32)]
    S = len(clusters)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=20)
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    probes[::5, 0] = -1                       # no first slot at all
    probes[1::5, 0] = 4                       # the first slot is the 4-row shard: its list never fills, it sets no bound
    _, st = _check(orc, gp, pts, clusters, queries, probes, nprobe, k, metric, modes=(1,), check_filtered=False)
    assert st[1]["filtered"] == 0 or st[1]["filter_dirty_tasks"] >= 1, st[1]   # (the small buckets did overflow)
