"""The reference's callers run Query / QueryBucket from many threads at once, one query per call (routes.cpp:194-200, shard_searches.cpp:48).
A handle owns one set of device workspaces: concurrent calls on it take turns (include/gpann.h, "Threads") and return what serial calls return."""
import threading

import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu


def test_concurrent_calls_on_one_handle_return_the_serial_results(orc, gp):
    rng = np.random.default_rng(31)
    n, d, S, nq, k = 30000, 32, 4, 240, 10
    pts = make_mixture(orc, n, d, n_centers=16)
    clusters = random_clusters(rng, n, S, overlap=0.1)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=16)
    tree = orc.kmtr_train(pts, clusters, 32, 200, 4000, 0, nthreads=4)
    router = gp.Router(tree.flat(), 0)
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(1)
    # the serial answers, slice by slice (slices of different sizes, down to the reference's one query per call)
    slices = [(0, 1), (1, 3), (3, 40), (40, 41), (41, 120), (120, 200), (200, 239), (239, 240)]
    want = []
    for lo, hi in slices:
        p, md, _ = router.query_batch(queries[lo:hi], 10 ** 6)
        ids, dist, _ = shards.scan_buckets(queries[lo:hi], k, p, 2)
        want.append((p.copy(), md.copy(), ids.copy(), dist.copy()))
    errors = []

    def worker(i):
        try:
            lo, hi = slices[i]
            for _ in range(6):
                p, md, _ = router.query_batch(queries[lo:hi], 10 ** 6)
                ids, dist, _ = shards.scan_buckets(queries[lo:hi], k, p, 2)
                wp, wmd, wi, wd = want[i]
                assert np.array_equal(p, wp) and np.array_equal(md.view(np.uint32), wmd.view(np.uint32))
                assert np.array_equal(ids, wi) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
        except BaseException as e:   # (collected: an assertion in a thread would otherwise vanish)
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(slices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    st = router.stats()
    assert st["verify_failures"] == 0 and st["bound_selfcheck"] == 0
    shards.close()


def test_one_query_calls_from_many_threads_are_coalesced(orc, gp):
    """The reference's callers issue KMeansTreeRouter::Query / InvertedIndex::QueryBucket one query per call from up to 32 threads
    (routes.cpp:194-200, shard_searches.cpp:48).  The host mirror coalesces concurrent calls into device batches (host_util.h Coalescer):
    same results as the calls one by one and as the batch entry points, far fewer device batches than calls."""
    from pygpann import host
    rng = np.random.default_rng(41)
    n, d, S, nq, k = 30000, 32, 4, 1500, 10
    pts = make_mixture(orc, n, d, n_centers=16)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=16)
    tree = orc.kmtr_train(pts, clusters, 32, 200, 4000, 0, nthreads=4)
    flat = tree.flat()
    router = gp.Router(flat, 0)
    want_p, _, _ = router.query_batch(queries, 10 ** 6)
    p_on, s_on, b_on = host.router_query_threads(flat, queries, 10 ** 6, 32, True)
    p_off, s_off, b_off = host.router_query_threads(flat, queries[:300], 10 ** 6, 32, False)
    assert np.array_equal(p_on, want_p) and np.array_equal(p_off, want_p[:300])
    assert b_off == 300 and b_on < nq / 4, (b_on, b_off)          # (32 callers in flight: batches of up to 32)
    rate_on, rate_off = nq / s_on, 300 / s_off
    print(f"router: {rate_on:.0f} calls/s coalesced ({nq / b_on:.1f} per batch) vs {rate_off:.0f} one by one: {rate_on / rate_off:.1f}x")
    assert rate_on > 2.0 * rate_off
    # QueryBucket: every query against its nearest shard
    buckets = want_p[:, 0].astype(np.int32)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, want_p, 1, 0, nthreads=4)
    i_on, d_on, s_on, b_on = host.index_query_bucket_threads(pts, clusters, queries, k, buckets, 32, True)
    i_off, d_off, s_off, b_off = host.index_query_bucket_threads(pts, clusters, queries[:300], k, buckets[:300], 32, False)
    assert np.array_equal(i_on, wi[:, 0]) and np.array_equal(d_on.view(np.uint32), wd[:, 0].view(np.uint32))
    assert np.array_equal(i_off, wi[:300, 0]) and np.array_equal(d_off.view(np.uint32), wd[:300, 0].view(np.uint32))
    assert b_on < nq / 4
    rate_on, rate_off = nq / s_on, 300 / s_off
    print(f"QueryBucket: {rate_on:.0f} calls/s coalesced ({nq / b_on:.1f} per batch) vs {rate_off:.0f} one by one: {rate_on / rate_off:.1f}x")
    assert rate_on > 2.0 * rate_off


def test_a_bad_request_fails_its_caller_alone(orc, gp):
    """with the coalescing queue on, one caller's bad argument (a bucket that does not exist) used to fail every thread that shared its
    device batch.  A failed group is now retried request by request (index_host.cpp RunScanGroup): the offender gets the exception, everybody
    else the reference's answer."""
    from pygpann import host
    rng = np.random.default_rng(43)
    n, d, S, nq, k = 20000, 24, 3, 400, 10
    pts = make_mixture(orc, n, d, n_centers=12)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=12)
    buckets = rng.integers(0, S, nq).astype(np.int32)
    bad = np.array([7, 8, 150, 399])
    buckets_bad = buckets.copy()
    buckets_bad[bad] = S + 5                      # no such shard
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, buckets.reshape(-1, 1), 1, 0, nthreads=4)
    ids, dist, _, batches, failed = host.index_query_bucket_threads(pts, clusters, queries, k, buckets_bad, 32, True, tolerate_failures=True)
    assert batches < nq                            # the queue did coalesce
    assert np.array_equal(np.flatnonzero(failed), bad)
    ok = failed == 0
    assert np.array_equal(ids[ok], wi[ok, 0]) and np.array_equal(dist[ok].view(np.uint32), wd[ok, 0].view(np.uint32))
