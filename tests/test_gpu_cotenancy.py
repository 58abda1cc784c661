"""GPU: the library beside ANOTHER process's int8 matrix wavefronts (profiles/r06/NOTES.md §2).

On MI355X the results of packed fp32 instructions go wrong in lanes 48 - 63 while another kernel — of any process — issues v_mfma_i32_32x32x32_i8 on the same
SIMD: with bin/MatrixNeighbour (tools/microbench/matrix_neighbour.hip: one 128-register int8 wavefront per SIMD on half of the compute units) running, round
5's library failed 186 of 285 GPU tests.  The library is therefore built without packed fp32 instructions (tests/test_capi_exports.py checks the code objects);
these tests run its main paths beside that neighbour and expect the oracle's bits, as always."""
import os
import subprocess
import time

import numpy as np
import pytest

from conftest import ROOT, make_mixture, random_clusters

pytestmark = pytest.mark.gpu

NEIGHBOUR = os.path.join(ROOT, "gp-ann_amd", "bin", "MatrixNeighbour")


@pytest.fixture(scope="module")
def int8_neighbour():
    """the neighbour process for the duration of the module: int8 (kind 0), 256-thread workgroups, on half of the compute units (the library's matrix-core
    kernels need whole compute units: on a chip whose every CU holds a neighbour workgroup they would wait for its next launch boundary)"""
    assert os.path.exists(NEIGHBOUR), f"{NEIGHBOUR} not built (make -C gp-ann_amd)"
    proc = subprocess.Popen([NEIGHBOUR, "90", "0", "256", "128"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    time.sleep(2.0)
    assert proc.poll() is None, proc.stdout.read().decode()
    yield proc
    alive = proc.poll() is None      # (still there when the tests are done: they DID run beside it)
    proc.terminate()
    try:
        proc.wait(timeout=30)
    except subprocess.TimeoutExpired:
        proc.kill()
    assert alive, "the neighbour ended before the tests did: they did not run beside it"


@pytest.mark.parametrize("d", [128, 64])
def test_exact_and_filtered_scan_beside_an_int8_neighbour(orc, gp, int8_neighbour, d):
    rng = np.random.default_rng(40 + d)
    n, S, nq, k, nprobe = 60000, 4, 400, 10, 2
    pts = make_mixture(orc, n, d, n_centers=32)
    clusters = random_clusters(rng, n, S, overlap=0.1)
    queries = make_mixture(orc, nq, d, seed=777, n_centers=32)
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, 0, nthreads=4)
    shards = gp.Shards(pts, clusters, 0, 0)
    for mode in (0, 1, 0, 1):          # exact, filtered, and once more: the neighbour's launches come and go
        shards.set_scan_mode(mode)
        ids, dist, _ = shards.scan_buckets(queries, k, probes, nprobe)
        assert np.array_equal(ids, wi), (d, mode)
        assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32)), (d, mode)
        if mode == 1:
            assert shards.stats()["filtered"] == 1
    shards.close()


def test_router_and_trainer_beside_an_int8_neighbour(orc, gp, int8_neighbour):
    n, d, nq, S = 30000, 64, 500, 4
    pts = orc.synth_rows(555, 7, 32, 0.35, 0, n, d, nthreads=4)
    queries = orc.synth_rows(556, 7, 32, 0.35, 0, nq, d, nthreads=4)
    part, _ = orc.kmeans(pts, S, 555, 0, nthreads=4)
    clusters = orc.partition_to_clusters(part)
    tree = orc.kmtr_train(pts, clusters, num_centroids=32, min_cluster_size=100, budget=4000, metric=0, nthreads=4)
    want_p, want_md = tree.query_batch(queries, 1 << 30, nthreads=4)
    store = gp.Shards(pts, clusters, plain=True)
    dev = gp.Router.train(store, n, 32, 100, 4000)          # KMeansTreeRouter::Train on the GPU (k-means kernels: fp32 tiles)
    flat_dev, flat_cpu = dev.export(), tree.flat()
    assert np.array_equal(flat_dev["centroids"].view(np.uint32), flat_cpu["centroids"].view(np.uint32))
    for router in (dev, gp.Router(flat_cpu, 0)):
        for budget in (1 << 30, 1500):                        # matrix-core bound path / inside the tree: table + bounded replay or walk
            want_pb, want_mdb = (want_p, want_md) if budget == 1 << 30 else tree.query_batch(queries, budget, nthreads=4)
            p, md, _ = router.query_batch(queries, budget)
            assert np.array_equal(md.view(np.uint32), want_mdb.view(np.uint32)), budget
            assert np.array_equal(p, want_pb), budget
        st = router.stats()
        assert st["bound_selfcheck"] == 0 and st["verify_failures"] == 0, st
        router.close()
    store.close()
