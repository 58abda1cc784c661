"""GPU: randomised MID-SIZE configurations of the filtered scan (50k - 600k rows, hundreds to thousands of queries) — the sizes at which a shard has several task
groups, cells have many row tiles, work items are cut at GPANN_FILTER_SEG_TILES, buckets are split over wavefronts and the sample pass's share is derived from
the bucket size; the small fuzz (tests/test_gpu_fuzz.py, at most 30k rows and 257 queries) does not reach them and the C1 - C5 shapes are five fixed points.
Every configuration: filter mode == exact mode for ALL queries bit for bit (ids, distances), and both == the CPU oracle on a sample of the queries;
deferred + overlapped calls on half of the seeds.  Seeds are fixed: a failure names its configuration.  GPANN_FUZZ_SEEDS=a:b replaces the suite's range."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NO_ID = 0xFFFFFFFF


def _seeds():
    a, _, b = os.environ.get("GPANN_FUZZ_SEEDS", "0:10").partition(":")
    return range(int(a), int(b))


def _cut(wd, wi):
    tau = wd[:, :, -1].min(axis=1)
    cut = ~(wd <= tau[:, None, None])
    wd, wi = wd.copy(), wi.copy()
    wd[cut] = np.inf
    wi[cut] = NO_ID
    return wd, wi


@pytest.mark.parametrize("seed", _seeds())
def test_random_midsize_configuration(orc, gp, seed, monkeypatch):
    rng = np.random.default_rng(21000 + seed)
    d = int(rng.choice([32, 64, 96, 100, 128, 200]))
    n = int(rng.integers(50_000, min(600_000, 40_000_000 // d)))
    S = int(rng.integers(1, 13))
    nq = int(rng.choice([300, 1000, 3000]))
    k = int(rng.choice([1, 10, 50]))
    nprobe = int(rng.integers(1, min(S, 4) + 1))
    metric = int(rng.integers(0, 2))
    for_merge = bool(rng.integers(0, 2))
    centers = int(rng.choice([1, 8, 64, 200]))
    sigma = float(rng.choice([0.05, 0.35, 1.0]))
    cell_rows = rng.choice(["", "512", "2000", "8000"])
    if cell_rows:
        monkeypatch.setenv("GPANN_CELL_ROWS", str(cell_rows))
    if rng.integers(0, 2):
        monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)   # the library's default rule (these shards are mostly "small")
    seg = rng.choice(["", "8", "64"])
    if seg:
        monkeypatch.setenv("GPANN_FILTER_SEG_TILES", str(seg))
    if rng.integers(0, 6) == 0:
        monkeypatch.setenv("GPANN_FILTER_CAND_CAP", "256")
    pts = orc.synth_rows(555 + seed, 7, centers, sigma, 0, n, d, nthreads=8)
    queries = orc.synth_rows(556, 7, centers, sigma, 0, nq, d, nthreads=8)
    if rng.integers(0, 4) == 0:
        off = np.float32(rng.normal(0, 40))
        pts += off
        queries += off
    # shards: a skewed random partition (sizes differ by an order of magnitude), optionally with a share of the points in a second shard
    w = rng.random(S) ** 2 + 0.02
    part = rng.choice(S, size=n, p=w / w.sum())
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
    if rng.integers(0, 3) == 0 and S > 1:
        extra = rng.choice(n, n // 5, replace=False).astype(np.uint32)
        tgt = rng.integers(0, S, size=len(extra))
        clusters = [np.concatenate([c, np.setdiff1d(extra[tgt == b], c)]).astype(np.uint32) for b, c in enumerate(clusters)]
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    probes[rng.random(probes.shape) < 0.05] = -1
    cfg = dict(seed=seed, d=d, n=n, S=S, nq=nq, k=k, nprobe=nprobe, metric=metric, for_merge=for_merge, centers=centers, sigma=sigma, cell_rows=str(cell_rows), seg=str(seg))

    shards = gp.Shards(pts, clusters, metric, 0)
    try:
        shards.set_scan_mode(1)
        ids, dist, _ = shards.scan_buckets(queries, k, probes, nprobe, for_merge=for_merge)
        st = shards.stats()
        shards.set_scan_mode(0)
        ids0, dist0, _ = shards.scan_buckets(queries, k, probes, nprobe, for_merge=for_merge)
        assert np.array_equal(ids, ids0), ("filter vs exact: ids", cfg, st)
        assert np.array_equal(dist.view(np.uint32), dist0.view(np.uint32)), ("filter vs exact: distances", cfg, st)
        # the oracle on a sample of the queries (the merge bound is a property of a query's own slots: a subset of the queries keeps it)
        sub = np.sort(rng.choice(nq, size=48, replace=False))
        ivf = orc.ivf_build(pts, clusters)
        wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries[sub], k, probes[sub], nprobe, metric, nthreads=16)
        if for_merge:
            wd, wi = _cut(wd, wi)
        assert np.array_equal(ids[sub], wi), ("oracle: ids", cfg)
        assert np.array_equal(dist[sub].view(np.uint32), wd.view(np.uint32)), ("oracle: distances", cfg)
        if seed % 2 == 0:
            # the same batch twice through deferred + overlapped calls (both in flight), host outputs
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            shards.set_scan_mode(1)
            qd = C.c_void_p()
            assert hip.hipMalloc(C.byref(qd), C.c_size_t(queries.nbytes)) == 0
            assert hip.hipMemcpy(qd, queries.ctypes.data_as(C.c_void_p), C.c_size_t(queries.nbytes), 1) == 0
            stream = C.c_void_p()
            assert hip.hipStreamCreate(C.byref(stream)) == 0
            words = ids.size
            pinned = []                                              # (the lists reach the host through a copy kernel: pinned arrays)
            for _ in range(4):
                h = C.c_void_p()
                assert hip.hipHostMalloc(C.byref(h), C.c_size_t(words * 4), 0) == 0
                pinned.append(h)

            def view(h):
                return np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_uint32)), shape=(words,)).reshape(ids.shape)
            try:
                for h in pinned:
                    view(h)[:] = 7
                for j in range(2):
                    shards.scan_buckets_device(qd.value, nq, k, probes, nprobe, pinned[2 * j].value, pinned[2 * j + 1].value, stream.value, defer=True, host_out=True,
                                               for_merge=for_merge, overlap=True)
                shards.finish()
                shards.finish()
                for j in range(2):
                    assert np.array_equal(view(pinned[2 * j]), ids0), ("overlapped: ids", cfg)
                    assert np.array_equal(view(pinned[2 * j + 1]), dist0.view(np.uint32)), ("overlapped: distances", cfg)
            finally:
                hip.hipStreamDestroy(stream)
                hip.hipFree(qd)
                for h in pinned:
                    hip.hipHostFree(h)
    finally:
        shards.close()
