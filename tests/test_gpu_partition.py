"""GPU: shard synthesis.  gpann_kmeans (RandomSample + KMeans on the device, csrc/kmeans.hip) and the k-means shard
builders of the host mirror in device mode (gp-ann_amd/host/partition_host.cpp; reference src/partitioning.cpp:12-146,
src/overlapping_partitioning.cpp:143-235, partition.cpp:36-144) against the CPU oracle (oracle.cpp k-means / trainer,
oracle/partitioners.py control flow): identical partitions, centroids bit for bit, identical overlapping clusters."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_mixture
from test_gpu_dropin import write_fbin
from test_partitioners import skewed_points

pytestmark = pytest.mark.gpu

PARTITION = os.path.join(ROOT, "gp-ann_amd", "bin", "Partition")


@pytest.fixture(scope="module")
def host():
    from pygpann import host as h
    h.lib()
    return h


@pytest.fixture(scope="module")
def parts():
    import partitioners
    return partitioners


def read_clusters(path):
    with open(path) as f:
        return [np.array(line.split(), np.uint32) for line in f]


@pytest.mark.parametrize("metric", [0, 1])
def test_device_kmeans_matches_oracle(orc, gp, metric):
    for n, d, k, seed in ((20000, 32, 40, 555), (5000, 7, 300, 3), (70000, 128, 16, 555), (9, 5, 16, 1)):
        pts = make_mixture(orc, n, d, n_centers=13, seed=seed)
        store = gp.Shards(pts, [np.arange(n, dtype=np.uint32)], metric=metric, plain=True)
        got_p, got_c = store.kmeans(k, seed)
        store.close()
        want_p, want_c = orc.kmeans(pts, k, seed, metric, nthreads=8)
        assert np.array_equal(got_p, want_p), (n, d, k)
        assert got_c.shape == want_c.shape and np.array_equal(got_c.view(np.uint32), want_c.view(np.uint32)), (n, d, k)


def test_plain_store_scans_exactly(orc, gp):
    """GPANN_STORE_PLAIN drops the filter support data; scans on such a store are the plain exact scan"""
    pts = make_mixture(orc, 30000, 24)
    q = make_mixture(orc, 64, 24, seed=556)
    clusters = [np.arange(0, 30000, 2, dtype=np.uint32), np.arange(1, 30000, 2, dtype=np.uint32)]
    plain, full = gp.Shards(pts, clusters, plain=True), gp.Shards(pts, clusters)
    probes = np.tile(np.array([1, 0], np.int32), (64, 1))
    a, b = plain.scan_buckets(q, 10, probes), full.scan_buckets(q, 10, probes)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert plain.stats()["filtered"] == 0


@pytest.mark.parametrize("metric", [0, 1])
def test_partitioners_on_device_match_oracle(orc, host, parts, metric):
    pts = skewed_points(orc, 6000, 16) if metric == 0 else make_mixture(orc, 6000, 16, n_centers=14)
    got = host.partition(host.PART_KMEANS, pts, k=6, eps=0.05, mips=bool(metric), device=True)
    assert np.array_equal(got, parts.kmeans_partitioning(orc, pts, 6, 0.05, metric))
    cap = int(1.05 * len(pts) / 8)
    got = host.partition(host.PART_REBALANCING, pts, k=8, max_cluster_size=cap, mips=bool(metric), device=True)
    assert np.array_equal(got, parts.rebalancing_kmeans_partitioning(orc, pts, cap, 8, metric))
    assert np.bincount(got).max() <= cap


@pytest.mark.parametrize("metric,overlap", [(0, 0.2), (1, 0.1)])
def test_overlap_on_device_matches_oracle(orc, host, parts, metric, overlap):
    pts = make_mixture(orc, 4000, 12, n_centers=9)
    k = 6
    base = host.partition(host.PART_REBALANCING, pts, k=k, max_cluster_size=int(1.05 * len(pts) / k), mips=bool(metric), device=True)
    got = host.overlap_spann(pts, base, k, 0.25, overlap, mips=bool(metric), device=True)
    want = parts.overlapping_kmeans_partitioning_spann(orc, pts, base, k, 0.25, overlap, metric)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, np.asarray(w, np.uint32))
    assert sum(len(c) for c in got) - len(pts) == int(overlap * len(pts))


def test_partition_driver(orc, parts, tmp_path):
    """argv, output names and formats of the reference's Partition driver (partition.cpp:36-144)"""
    pts = skewed_points(orc, 5000, 10)
    inp, out = str(tmp_path / "base.fbin"), str(tmp_path / "base.partition")
    write_fbin(inp, pts)

    def run(*args):
        r = subprocess.run([PARTITION, inp, out, *args], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout

    run("7", "KMeans", "default")
    want = parts.kmeans_partitioning(orc, pts, 7, 0.05)
    got = read_clusters(out + ".k=7.KMeans")
    assert len(got) == want.max() + 1
    for b, ids in enumerate(got):
        assert np.array_equal(ids, np.flatnonzero(want == b))

    run("7", "RKM", "default")
    want = parts.rebalancing_kmeans_partitioning(orc, pts, int(1.05 * len(pts) / 7), 7)
    got = read_clusters(out + ".k=7.RKM")
    for b, ids in enumerate(got):   # ConvertPartitionToClusters: ascending ids
        assert np.array_equal(ids, np.flatnonzero(want == b))

    run("7", "FlatKMeans", "default")
    want, _ = orc.kmeans(pts, 7, 555, 0)
    for b, ids in enumerate(read_clusters(out + ".k=7.FlatKMeans")):
        assert np.array_equal(ids, np.flatnonzero(want == b))

    run("5", "ORKM", "default", "0.2")
    cap = int(1.05 * len(pts) / 5)
    rkm = parts.rebalancing_kmeans_partitioning(orc, pts, cap, 6)   # ceil(5 * 1.2) clusters
    want = parts.overlapping_kmeans_partitioning_spann(orc, pts, rkm, 5, 0.05, 0.2)
    got = read_clusters(out + ".k=5.ORKM.o=0.2")
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert np.array_equal(g, np.asarray(w, np.uint32))

    run("4", "Random", "default")
    part = np.loadtxt(out + ".k=4.Random", dtype=np.int64)
    assert len(part) == 5000 and (np.bincount(part) == 1250).all() and (part[:50] != np.sort(part[:50])).any()

    r = subprocess.run([PARTITION, inp, out, "4", "GP", "default"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "Unsupported partitioning method" in r.stdout
