"""GPU: randomised small configurations through the DEFERRED scan path (GPANN_SCAN_DEFER, with and without GPANN_SCAN_OVERLAP) against the CPU oracle —
three batches of one store in flight the way bench.py and ProcessQueryStream issue them (two enqueued, finish, a third enqueued, finish, finish), each batch
with its own queries and probe lists (skipped slots, batches that probe nothing, batches whose tasks all sit on an empty or a three-row shard), odd
dimensions, k up to 33, both metrics, with and without the merge bound, cells of a few hundred rows, the small-shard rule on and off, rest-pass workgroups
persistent or leaving after two items.  The blocking path has tests/test_gpu_fuzz.py; this one is about call slots, streams and per-slot workspaces.
Seeds are fixed: a failure names its configuration.  GPANN_FUZZ_SEEDS=a:b replaces the suite's range for a longer run."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import make_mixture

pytestmark = pytest.mark.gpu
NO_ID = 0xFFFFFFFF


def _seeds():
    a, _, b = os.environ.get("GPANN_FUZZ_SEEDS", "0:40").partition(":")
    return range(int(a), int(b))


def _cut(wd, wi):
    tau = wd[:, :, -1].min(axis=1)
    cut = ~(wd <= tau[:, None, None])
    wd, wi = wd.copy(), wi.copy()
    wd[cut] = np.inf
    wi[cut] = NO_ID
    return wd, wi


@pytest.mark.parametrize("seed", _seeds())
def test_random_deferred_configuration(orc, gp, seed, monkeypatch):
    hip = C.CDLL("libamdhip64.so")   # the runtime libgpann_hip.so is linked against (device buffers, pinned arrays and a stream without torch)

    def ok(rc):
        assert rc == 0, f"hip error {rc}"
    rng = np.random.default_rng(5000 + seed)
    d = int(rng.choice([8, 33, 64, 100, 128, 200]))
    n = int(rng.integers(300, 30000))
    S = int(rng.integers(1, 7))
    nq = int(rng.choice([1, 2, 33, 100, 257]))
    k = int(rng.choice([1, 10, 33]))
    nprobe = int(rng.integers(1, S + 1))
    metric = int(rng.integers(0, 2))
    for_merge = bool(rng.integers(0, 2))
    overlap = bool(rng.integers(0, 2))
    mode = int(rng.integers(0, 4) != 0)                              # one in four: the exact scan behind the same call slots
    monkeypatch.setenv("GPANN_CELL_ROWS", str(int(rng.choice([64, 300, 1500]))))
    monkeypatch.setenv("GPANN_OVERLAP_REST_ITEMS", str(int(rng.choice([0, 2]))))
    if rng.integers(0, 2):
        monkeypatch.delenv("GPANN_FILTER_SMALL_SHARD_ROWS", raising=False)   # the library's default for stores of this size: no shard-sample, no head pass
    if rng.integers(0, 5) == 0:
        monkeypatch.setenv("GPANN_FILTER_CAND_CAP", "64")            # buckets overflow: dirty tasks / the call redone exactly at its finish
    pts = make_mixture(orc, n, d, n_centers=int(rng.integers(1, 8)), sigma=float(rng.choice([0.05, 0.35, 1.0])))
    if rng.integers(0, 3) == 0:
        pts += np.float32(rng.normal(0, 50))
    part = rng.integers(0, S, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
    if S > 2:
        clusters[int(rng.integers(0, S))] = np.zeros(0, np.uint32)            # an empty shard
        clusters[int(rng.integers(0, S))] = np.arange(3, dtype=np.uint32)     # a three-row shard
    cfg = dict(seed=seed, d=d, n=n, S=S, nq=nq, k=k, nprobe=nprobe, metric=metric, for_merge=for_merge, overlap=overlap, mode=mode)
    ivf = orc.ivf_build(pts, clusters)
    batches, probes, want = [], [], []
    for b in range(3):
        q = make_mixture(orc, nq, d, seed=600 + b, n_centers=3)
        p = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
        p[rng.random(p.shape) < 0.1] = -1
        kind = int(rng.integers(0, 6))
        if kind == 0:
            p[:] = -1                                                # a batch with nothing to scan between two that have
        elif kind == 1 and S > 2:
            small = [i for i, c in enumerate(clusters) if len(c) <= 3]
            p[:] = -1
            p[:, 0] = small[0]                                       # every task on a shard without cells
        wd, wi, _ = orc.ivf_query_bucket_batch(ivf, q, k, p, nprobe, metric, nthreads=4)
        if for_merge:
            wd, wi = _cut(wd, wi)
        batches.append(q)
        probes.append(p)
        want.append((wd, wi))
    shards = gp.Shards(pts, clusters, metric, 0)
    shards.set_scan_mode(mode)
    st = C.c_void_p()
    ok(hip.hipStreamCreate(C.byref(st)))
    q_dev, h_ids, h_dist = [], [], []
    words = nq * nprobe * k
    for q in batches:
        p = C.c_void_p()
        ok(hip.hipMalloc(C.byref(p), C.c_size_t(q.nbytes)))
        ok(hip.hipMemcpy(p, q.ctypes.data_as(C.c_void_p), C.c_size_t(q.nbytes), 1))
        q_dev.append(p)
        for lst in (h_ids, h_dist):
            h = C.c_void_p()
            ok(hip.hipHostMalloc(C.byref(h), C.c_size_t(words * 4), 0))
            lst.append(h)
    ok(hip.hipDeviceSynchronize())

    def view(h, dtype):
        return np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_uint32)), shape=(words,)).view(dtype).reshape(nq, nprobe, k)

    def begin(b):
        shards.scan_buckets_device(q_dev[b].value, nq, k, probes[b], nprobe, h_ids[b].value, h_dist[b].value, st.value, defer=True, host_out=True,
                                   for_merge=for_merge, overlap=overlap)
    try:
        for rep in range(2):                                         # twice: the second round re-uses every call slot's workspaces
            begin(0)
            begin(1)
            shards.finish()
            begin(2)
            shards.finish()
            shards.finish()
            for b in range(3):
                wd, wi = want[b]
                assert np.array_equal(view(h_ids[b], np.uint32), wi), (cfg, rep, b)
                assert np.array_equal(view(h_dist[b], np.uint32), wd.view(np.uint32)), (cfg, rep, b)
                view(h_ids[b], np.uint32)[:] = 0x12345678            # (stale results must not pass for fresh ones)
                view(h_dist[b], np.uint32)[:] = 0
    finally:
        shards.close()
        for p in q_dev:
            hip.hipFree(p)
        for h in h_ids + h_dist:
            hip.hipHostFree(h)
        hip.hipStreamDestroy(st)
