"""GPU: randomised small configurations of the filtered scan against the CPU oracle — odd dimensions (incl. d > 256: exact fallback), one
task, one shard, k up to 64, every probe skipped, tiny and empty shards, both metrics, with and without the merge bound, cells of a few
hundred rows.  Seeds are fixed: a failure names its configuration."""
import numpy as np
import pytest

from conftest import make_mixture

pytestmark = pytest.mark.gpu
NO_ID = 0xFFFFFFFF


def _cut(wd, wi):
    tau = wd[:, :, -1].min(axis=1)
    cut = ~(wd <= tau[:, None, None])
    wd, wi = wd.copy(), wi.copy()
    wd[cut] = np.inf
    wi[cut] = NO_ID
    return wd, wi


def _seeds():
    """96 seeds in the suite; GPANN_FUZZ_SEEDS=a:b (a tool's longer run, e.g. profiles/r06/NOTES.md §5: 96:800) replaces the range"""
    import os
    a, _, b = os.environ.get("GPANN_FUZZ_SEEDS", "0:96").partition(":")
    return range(int(a), int(b))


@pytest.mark.parametrize("seed", _seeds())
def test_random_configuration(orc, gp, seed, monkeypatch):
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([3, 8, 17, 33, 64, 65, 96, 100, 127, 128, 129, 200, 256, 260]))
    n = int(rng.integers(300, 30000))
    S = int(rng.integers(1, 10))
    nq = int(rng.choice([1, 2, 31, 33, 100, 257]))
    k = int(rng.choice([1, 5, 10, 33, 64]))
    nprobe = int(rng.integers(1, S + 1))
    metric = int(rng.integers(0, 2))
    for_merge = bool(rng.integers(0, 2))
    monkeypatch.setenv("GPANN_CELL_ROWS", str(int(rng.choice([64, 300, 1500]))))
    if rng.integers(0, 4) == 0:
        monkeypatch.setenv("GPANN_FILTER_CAND_CAP", "64")          # buckets overflow: dirty tasks / whole-call fallback
    if seed % 2 == 1:
        monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)   # the library's default for stores of this size: no shard-sample, no head pass
    pts = make_mixture(orc, n, d, n_centers=int(rng.integers(1, 12)), sigma=float(rng.choice([0.05, 0.35, 1.0])))
    if rng.integers(0, 3) == 0:
        pts += np.float32(rng.normal(0, 50))                       # far from the origin
    if rng.integers(0, 3) == 0:
        pts[: n // 3] = pts[0]                                     # masses of duplicates
    part = rng.integers(0, S, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
    if S > 2:
        clusters[int(rng.integers(0, S))] = np.zeros(0, np.uint32)            # an empty shard
        clusters[int(rng.integers(0, S))] = np.arange(3, dtype=np.uint32)     # a three-row shard (ids repeat across shards)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=4)
    queries[0] = pts[0]
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    probes[rng.random(probes.shape) < 0.1] = -1
    if seed % 7 == 0:
        probes[:] = -1                                             # nothing to scan
    cfg = dict(seed=seed, d=d, n=n, S=S, nq=nq, k=k, nprobe=nprobe, metric=metric, for_merge=for_merge)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, metric, nthreads=4)
    if for_merge:
        wd, wi = _cut(wd, wi)
    shards = gp.Shards(pts, clusters, metric, 0)
    for mode in (1, 0):
        shards.set_scan_mode(mode)
        ids, dist, _ = shards.scan_buckets(queries, k, probes, nprobe, for_merge=for_merge)
        assert np.array_equal(ids, wi), (cfg, mode)
        assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32)), (cfg, mode)
    shards.close()
