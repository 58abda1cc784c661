"""GPU parity: the HIP shard scan (through the C-ABI) against the CPU oracle — ids and distances bit-exact."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu


def _check_buckets(orc, gp, pts, clusters, queries, probes, nprobe, k, metric):
    shards = gp.Shards(pts, clusters, metric, 0)
    ids, dist, rows, _ = shards.scan_buckets(queries, k, probes, nprobe, want_rows=True)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, wc = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, metric, nthreads=4)
    assert np.array_equal(ids, wi)
    assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
    # rows are the reference's TopN keys: permutation[row] == id
    _, perm, _ = shards.export()
    valid = ids != gp.NO_ID
    assert np.array_equal(perm[rows[valid]], ids[valid])
    shards.close()
    return ids, dist


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [128, 96, 100, 200, 7, 30])
def test_scan_buckets_matches_oracle(orc, gp, metric, d):
    rng = np.random.default_rng(d + metric)
    n, S, nq, k = 6000, 5, 150, 10
    pts = make_mixture(orc, n, d)
    clusters = random_clusters(rng, n, S, overlap=0.2)
    queries = make_mixture(orc, nq, d, seed=556)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    _check_buckets(orc, gp, pts, clusters, queries, probes, 3, k, metric)


def test_scan_ties_broken_by_row(orc, gp):
    rng = np.random.default_rng(3)
    n, d, S, nq, k = 4000, 32, 3, 70, 10
    pts = make_mixture(orc, n, d)
    pts[500:900] = pts[500]            # 400 identical points: every distance to them ties
    pts[1500:1600, :] = 0.0
    clusters = random_clusters(rng, n, S, overlap=0.3)
    queries = make_mixture(orc, nq, d, seed=556)
    queries[0] = pts[500]
    queries[1] = 0.0
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    for metric in (0, 1):
        _check_buckets(orc, gp, pts, clusters, queries, probes, S, k, metric)


@pytest.mark.parametrize("k", [1, 2, 10, 33, 64])
def test_scan_k_values(orc, gp, k):
    rng = np.random.default_rng(k)
    n, d, S, nq = 3000, 64, 4, 90
    pts = make_mixture(orc, n, d)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556)
    probes = np.stack([rng.permutation(S)[:2] for _ in range(nq)]).astype(np.int32)
    _check_buckets(orc, gp, pts, clusters, queries, probes, 2, k, 0)


def test_scan_edge_cases(orc, gp):
    rng = np.random.default_rng(5)
    n, d, S, k = 1000, 16, 6, 10
    pts = make_mixture(orc, n, d)
    # shard 5 is empty, shard 4 has 3 points (< k), negative probes are skipped, ragged probe lists
    part = rng.integers(0, 4, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(4)] + [np.array([5, 6, 7], np.uint32), np.zeros(0, np.uint32)]
    queries = make_mixture(orc, 33, d, seed=556)
    probes = rng.integers(-1, S, size=(33, 4)).astype(np.int32)
    ids, dist = _check_buckets(orc, gp, pts, clusters, queries, probes, 4, k, 0)
    q, j = np.nonzero(probes == 5)
    assert np.all(ids[q, j] == gp.NO_ID) and np.all(np.isinf(dist[q, j]))
    q, j = np.nonzero(probes == 4)
    assert np.all(ids[q, j, 3:] == gp.NO_ID) and np.all(ids[q, j, :3] != gp.NO_ID)
    # a single query, a single probe
    _check_buckets(orc, gp, pts, clusters, queries[:1], probes[:1, :1].clip(0), 1, k, 0)


def test_scan_k_too_large_is_an_error(orc, gp):
    pts = make_mixture(orc, 100, 8)
    shards = gp.Shards(pts, [np.arange(100, dtype=np.uint32)], 0, 0)
    with pytest.raises(gp.GpannError) as e:
        shards.scan_buckets(pts[:2], 65, np.zeros((2, 1), np.int32))
    assert e.value.status == gp.ERR_UNSUPPORTED
    with pytest.raises(gp.GpannError):
        shards.scan_buckets(pts[:2], 10, np.full((2, 1), 3, np.int32))   # shard id out of range


@pytest.mark.parametrize("metric", [0, 1])
def test_scan_merged_matches_inverted_index_query(orc, gp, metric):
    rng = np.random.default_rng(8)
    n, d, S, nq, k = 5000, 48, 6, 64, 10
    pts = make_mixture(orc, n, d)
    clusters = random_clusters(rng, n, S, overlap=0.25)
    queries = make_mixture(orc, nq, d, seed=556)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    shards = gp.Shards(pts, clusters, metric, 0)
    ivf = orc.ivf_build(pts, clusters)
    for nprobe in (1, 3, S):
        ids, dist, _ = shards.scan_merged(queries, k, probes, nprobe)
        for q in range(nq):
            wd, wi = orc.ivf_query(ivf, queries[q], k, probes[q], nprobe, metric)
            assert np.array_equal(ids[q], wi) and np.array_equal(dist[q].view(np.uint32), wd.view(np.uint32))


def test_ground_truth_and_shard_search_hits(orc, gp):
    rng = np.random.default_rng(9)
    n, d, S, nq, k = 8000, 40, 4, 50, 10
    pts = make_mixture(orc, n, d)
    queries = make_mixture(orc, nq, d, seed=556)
    ids, dist, _ = gp.ground_truth(pts, queries, k, 0)
    wd, wi = orc.ground_truth(pts, queries, k, 0, nthreads=4)
    assert np.array_equal(ids, wi) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
    kth = wd[:, k - 1].copy()
    clusters = random_clusters(rng, n, S, overlap=0.2)
    shards = gp.Shards(pts, clusters, 0, 0)
    hits, cnt, _ = shards.shard_search_hits(queries, k, kth)
    ivf = orc.ivf_build(pts, clusters)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    bd, bi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, S, 0, nthreads=4)
    for b in range(S):
        for q in range(nq):
            keep = [int(bi[q, b, x]) for x in range(k) if bi[q, b, x] != 0xFFFFFFFF and bd[q, b, x] <= kth[q]]
            assert cnt[b, q] == len(keep)
            assert list(hits[b, q, :len(keep)]) == keep[::-1]      # descending distance, as the reference's heap pops
    # every true neighbour is found in some shard: union of hits over shards has >= k ids
    for q in range(nq):
        u = set()
        for b in range(S):
            u.update(int(x) for x in hits[b, q, :cnt[b, q]])
        assert len(u) >= k


def test_larger_segmented_scan(orc, gp):
    """many segments per shard and several query tiles: exercises the partial-list merge"""
    rng = np.random.default_rng(11)
    n, d, S, nq, k = 60000, 32, 2, 300, 10
    pts = make_mixture(orc, n, d)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    _check_buckets(orc, gp, pts, clusters, queries, probes, S, k, 0)


def test_back_to_back_untimed_device_calls_do_not_share_staged_tasks(orc, gp):
    """Two untimed exact scans with device output return without synchronising; the second call must not overwrite the pinned staging
    bytes (task arrays, unit lists) the first one is still uploading from (PinnedStage reuse)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")   # the runtime libgpann_hip.so is linked against (device buffers and a stream without torch)

    def ok(rc):
        assert rc == 0, f"hip error {rc}"

    def dev_alloc(nbytes):
        p = C.c_void_p()
        ok(hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)))
        return p

    rng = np.random.default_rng(21)
    n, d, S, nq, k = 40000, 64, 6, 2000, 10
    pts = make_mixture(orc, n, d)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556)
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(0)
    ivf = orc.ivf_build(pts, clusters)
    q_dev = dev_alloc(queries.nbytes)
    ok(hip.hipMemcpy(q_dev, queries.ctypes.data_as(C.c_void_p), C.c_size_t(queries.nbytes), 1))
    st = C.c_void_p()
    ok(hip.hipStreamCreate(C.byref(st)))
    probes = [np.stack([rng.permutation(S)[:2] for _ in range(nq)]).astype(np.int32) for _ in range(3)]
    nb = nq * 2 * k * 4
    outs = [(dev_alloc(nb), dev_alloc(nb)) for _ in range(3)]
    ok(hip.hipDeviceSynchronize())
    for pr, (oi, od) in zip(probes, outs):   # no timing, no synchronisation in between
        shards.scan_buckets_device(q_dev.value, nq, k, pr, 2, oi.value, od.value, st.value, timed=False)
    ok(hip.hipStreamSynchronize(st))
    for pr, (oi, od) in zip(probes, outs):
        ids = np.empty((nq, 2, k), np.uint32)
        dist = np.empty((nq, 2, k), np.float32)
        ok(hip.hipMemcpy(ids.ctypes.data_as(C.c_void_p), oi, C.c_size_t(nb), 2))
        ok(hip.hipMemcpy(dist.ctypes.data_as(C.c_void_p), od, C.c_size_t(nb), 2))
        wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, pr, 2, 0, nthreads=4)
        assert np.array_equal(ids, wi)
        assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
    shards.close()
    for a, b in outs:
        hip.hipFree(a)
        hip.hipFree(b)
    hip.hipFree(q_dev)
    hip.hipStreamDestroy(st)


def test_scan_stats_sized_for_callers_of_another_abi_revision(orc, gp):
    """gpann_scan_last_stats_sized: gpann_scan_stats only ever grows at its end, so a caller built against an older (shorter) or newer (longer) header
    passes ITS sizeof and gets the fields both sides know — nothing written past its buffer, the unknown tail zeroed; gpann_abi_version() = the header's"""
    import ctypes as C
    rng = np.random.default_rng(5)
    n, d, S, nq, k = 3000, 32, 2, 40, 5
    pts = make_mixture(orc, n, d)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.scan_buckets(queries, k, probes, S)
    full = shards.stats()
    L = gp.lib()
    assert L.gpann_abi_version() >= 6
    size = C.sizeof(gp.ScanStats)
    # an older caller: knows the struct up to filter_candidates (56 bytes); a guard word behind its buffer must stay untouched
    buf = (C.c_uint8 * 64)(*([0xAB] * 64))
    assert L.gpann_scan_last_stats_sized(shards.h, buf, C.c_size_t(56)) == 0
    old = np.frombuffer(bytes(buf), dtype=np.uint64)
    assert old[0] == full["pairs"] and old[2] == full["tasks"] and old[6] == full["filter_candidates"] and old[7] == 0xABABABABABABABAB
    # a newer caller: a longer struct — the tail it knows and this library does not is zeroed
    big = (C.c_uint8 * (size + 32))(*([0xCD] * (size + 32)))
    assert L.gpann_scan_last_stats_sized(shards.h, big, C.c_size_t(size + 32)) == 0
    raw = bytes(big)
    st = gp.ScanStats.from_buffer_copy(raw[:size])
    assert st.pairs == full["pairs"] and st.tasks == nq * S and raw[size:] == bytes(32)
    shards.close()
