"""GPU parity of the cell-pruned filtered scan (gp-ann_amd/csrc/cells.hip, scan_cells.h): shards cut into k-means cells whose
bounding balls let the filter skip rows; ids / distances must stay bit-identical to the CPU oracle and to the plain exact scan."""
import numpy as np
import pytest

from conftest import make_mixture, random_clusters

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _cells_on(monkeypatch):
    monkeypatch.setenv("GPANN_CELL_ROWS", "1500")     # small cells: many cells even in small test stores


def _check(orc, gp, pts, clusters, queries, probes, nprobe, k, metric, expect_filtered=True):
    shards = gp.Shards(pts, clusters, metric, 0)
    shards.set_scan_mode(1)
    ids, dist, rows, _ = shards.scan_buckets(queries, k, probes, nprobe, want_rows=True)
    st = shards.stats()
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, metric, nthreads=4)
    assert np.array_equal(ids, wi)
    assert np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
    if expect_filtered is not None:
        assert st["filtered"] == (1 if expect_filtered else 0), st
    shards.set_scan_mode(0)
    ids0, dist0, _ = shards.scan_buckets(queries, k, probes, nprobe)
    assert np.array_equal(ids0, ids) and np.array_equal(dist0.view(np.uint32), dist.view(np.uint32))
    shards.close()
    return st


@pytest.mark.parametrize("metric", [0, 1])
# every k-step count the kernels exist for (PlaneKSteps: 2, 4, 6, 7, 8, 10, 13, 16) and the dimensions that round up to them;
# > 128: 128-task groups, hi-plane passes only
@pytest.mark.parametrize("d", [128, 96, 100, 112, 30, 7, 50, 130, 160, 200, 208, 256])
def test_cells_match_oracle(orc, gp, metric, d):
    rng = np.random.default_rng(d * 5 + metric)
    n, S, nq, k = 40000, 3, 300, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S, overlap=0.15)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    st = _check(orc, gp, pts, clusters, queries, probes, 2, k, metric)
    assert st["filter_pairs"] > 0
    want_ks = next(ks for ks in (2, 4, 6, 7, 8, 10, 13, 16) if 16 * ks >= d)
    assert st["filter_plane_ksteps"] == want_ks, st       # the planes are stored with ceil(d / 16) k-steps, not 4 / 8 / 16


@pytest.mark.parametrize("lo_rows", ["all", "0", "64", None])
def test_cells_lo_plane_layouts(orc, gp, monkeypatch, lo_rows):
    """The lo planes exist only for the rows the two-plane passes read (default: the shard sample and the first 512 rows of every cell);
    GPANN_FILTER_LO_ROWS=all keeps one per tile (and allows full-precision later passes), 0 none for the cells: same bits every time."""
    if lo_rows is not None:
        monkeypatch.setenv("GPANN_FILTER_LO_ROWS", lo_rows)
    rng = np.random.default_rng(77)
    n, d, S, nq, k = 50000, 100, 3, 400, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S, overlap=0.1)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    _check(orc, gp, pts, clusters, queries, probes, 2, k, 0)
    if lo_rows == "all":
        monkeypatch.setenv("GPANN_FILTER_HI_ONLY", "0")      # all three products in every pass: needs the full lo array
        _check(orc, gp, pts, clusters, queries, probes, 2, k, 0)


def test_cells_prune_clustered_data(orc, gp):
    """well separated mixture: most (task, cell) blocks are skipped, the answer is still exact"""
    rng = np.random.default_rng(3)
    n, d, nq, k = 200000, 32, 700, 10
    pts = make_mixture(orc, n, d, n_centers=40, sigma=0.2)
    clusters = [rng.permutation(n).astype(np.uint32)]
    queries = make_mixture(orc, nq, d, seed=556, n_centers=40, sigma=0.2)
    probes = np.zeros((nq, 1), np.int32)
    st = _check(orc, gp, pts, clusters, queries, probes, 1, k, 0)
    assert st["filter_pairs"] < 0.5 * nq * n, st          # pruning happened


@pytest.mark.parametrize("k", [1, 7, 64])
def test_cells_k_values(orc, gp, k):
    rng = np.random.default_rng(k)
    n, d, S, nq = 30000, 64, 2, 130
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    _check(orc, gp, pts, clusters, queries, probes, S, k, 0)


def test_cells_uniform_data_no_pruning_possible(orc, gp):
    """one Gaussian blob: no cell can be excluded, everything is filtered, results exact"""
    rng = np.random.default_rng(8)
    n, d, nq, k = 60000, 48, 200, 10
    pts = rng.standard_normal((n, d)).astype(np.float32)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    clusters = [np.arange(n, dtype=np.uint32)]
    probes = np.zeros((nq, 1), np.int32)
    for metric in (0, 1):
        _check(orc, gp, pts, clusters, queries, probes, 1, k, metric)


def test_cells_far_queries_and_scaled_data(orc, gp):
    rng = np.random.default_rng(23)
    n, d, nq, k = 30000, 48, 100, 10
    base = make_mixture(orc, n, d, n_centers=8, sigma=0.01)
    pts = (base * np.float32(37.0) + np.float32(900.0)).astype(np.float32)
    queries = pts[rng.integers(0, n, nq)] + rng.standard_normal((nq, d)).astype(np.float32) * np.float32(0.05)
    clusters = [rng.permutation(n).astype(np.uint32)]
    probes = np.zeros((nq, 1), np.int32)
    for metric in (0, 1):
        _check(orc, gp, pts, clusters, queries, probes, 1, k, metric, expect_filtered=None)


def test_cells_duplicates_and_small_shards(orc, gp):
    """duplicate rows (ties broken by row), a shard smaller than k, an empty shard, skipped probes"""
    rng = np.random.default_rng(31)
    d, k = 16, 10
    base = make_mixture(orc, 300, d, n_centers=6)
    pts = base[rng.integers(0, 300, size=20000)]
    clusters = [np.arange(0, 12000, dtype=np.uint32), np.arange(12000, 19995, dtype=np.uint32), np.arange(19995, 20000, dtype=np.uint32),
                np.zeros(0, np.uint32)]
    queries = base[rng.integers(0, 300, size=90)] + np.float32(0.001)
    probes = np.stack([rng.permutation(4) for _ in range(90)]).astype(np.int32)
    probes[::7, 1] = -1
    _check(orc, gp, pts, clusters, queries, probes, 3, k, 0, expect_filtered=None)


def test_cells_ground_truth(orc, gp):
    n, d, nq, k = 50000, 24, 150, 10
    pts = make_mixture(orc, n, d, n_centers=16)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=16)
    ids, dist, _ = gp.ground_truth(pts, queries, k, 0)
    wd, wi = orc.ground_truth(pts, queries, k, 0, nthreads=4)
    assert np.array_equal(ids, wi) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32))


def test_cells_hi_plane_only_last_pass(orc, gp, monkeypatch):
    """GPANN_FILTER_HI_ONLY=1: the last pass multiplies only the hi plane of the points; its threshold carries the dropped term"""
    monkeypatch.setenv("GPANN_FILTER_HI_ONLY", "1")
    rng = np.random.default_rng(77)
    n, d, nq, k = 120000, 64, 400, 10
    pts = make_mixture(orc, n, d, n_centers=30)
    clusters = [rng.permutation(n).astype(np.uint32)]
    queries = make_mixture(orc, nq, d, seed=556, n_centers=30)
    probes = np.zeros((nq, 1), np.int32)
    for metric in (0, 1):
        _check(orc, gp, pts, clusters, queries, probes, 1, k, metric)


def test_cells_default_configuration_midsize(orc, gp, monkeypatch):
    """a store large enough to get cells by default (>= 1M rows, 32k-row cells, hi-plane last pass): filter mode must equal the
    plain exact scan bit for bit on every query and the CPU oracle on a sample of them"""
    monkeypatch.delenv("GPANN_CELL_ROWS")
    rng = np.random.default_rng(41)
    n, d, S, nq, k = 2_400_000, 64, 2, 320, 10
    pts = orc.synth_rows(555, 7, 96, 0.35, 0, n, d, nthreads=8)
    queries = orc.synth_rows(556, 7, 96, 0.35, 0, nq, d, nthreads=8)
    part = rng.integers(0, S, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(1)
    ids, dist, _ = shards.scan_buckets(queries, k, probes, S)
    st = shards.stats()
    assert st["filtered"] == 1 and 0 < st["filter_pairs"] < nq * n                # some (query tile, cell) blocks were skipped
    # the rest pass is accounted for on its own (bench.py's roofline block): a hi-plane launch over most of the streamed rows
    assert 0 < st["filter_rest_rows"] <= st["filter_rows"] and st["filter_rest_ms"] > 0 and st["filter_rest_hi_only"] == 1
    shards.set_scan_mode(0)
    ids0, dist0, _ = shards.scan_buckets(queries, k, probes, S)
    assert np.array_equal(ids0, ids) and np.array_equal(dist0.view(np.uint32), dist.view(np.uint32))
    shards.close()
    sub = np.arange(0, nq, 16)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries[sub], k, probes[sub], S, 0, nthreads=8)
    assert np.array_equal(ids[sub], wi) and np.array_equal(dist[sub].view(np.uint32), wd.view(np.uint32))


def test_cells_beyond_256_dimensions_use_the_exact_scan(orc, gp):
    rng = np.random.default_rng(77)
    n, d, S, nq, k = 20000, 260, 2, 100, 10
    pts = make_mixture(orc, n, d, n_centers=8)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=8)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    _check(orc, gp, pts, clusters, queries, probes, S, k, 0, expect_filtered=False)


def test_cells_200_dimensions_midsize_inner_product(orc, gp, monkeypatch):
    """the Text-to-Image shape (200-dim inner product) at a size that gets default cells: filter == exact on every query"""
    monkeypatch.delenv("GPANN_CELL_ROWS")
    rng = np.random.default_rng(43)
    n, d, S, nq, k = 1_500_000, 200, 3, 600, 10
    pts = orc.synth_rows(555, 7, 64, 0.35, 0, n, d, nthreads=16)
    queries = orc.synth_rows(556, 7, 64, 0.35, 0, nq, d, nthreads=16)
    part = rng.integers(0, S, size=n)
    clusters = [np.flatnonzero(part == b).astype(np.uint32) for b in range(S)]
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    shards = gp.Shards(pts, clusters, 1, 0)
    shards.set_scan_mode(1)
    ids, dist, _ = shards.scan_buckets(queries, k, probes, S)
    st = shards.stats()
    assert st["filtered"] == 1 and st["filter_pairs"] > 0
    shards.set_scan_mode(0)
    ids0, dist0, _ = shards.scan_buckets(queries, k, probes, S)
    assert np.array_equal(ids0, ids) and np.array_equal(dist0.view(np.uint32), dist.view(np.uint32))
    shards.close()
    sub = np.arange(0, nq, 40)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries[sub], k, probes[sub], S, 1, nthreads=16)
    assert np.array_equal(ids[sub], wi) and np.array_equal(dist[sub].view(np.uint32), wd.view(np.uint32))


@pytest.mark.parametrize("d,offset", [(128, 100.0), (96, 1000.0), (30, -250.0)])
def test_cells_on_data_far_from_the_origin(orc, gp, d, offset):
    """a cloud that sits far from the origin (SIFT-like, non-negative coordinates): squared L2 is translation invariant and the store builds
    its planes around its mean row, so the matrix-core bound stays as sharp as on centred data — the filter must still be the path that
    answers (no fallback to the exact scan), prune, and return the oracle's bits"""
    rng = np.random.default_rng(int(d))
    n, S, nq, k = 60000, 2, 300, 10
    pts = (make_mixture(orc, n, d, n_centers=24) + np.float32(offset)).astype(np.float32)
    clusters = random_clusters(rng, n, S)
    queries = (make_mixture(orc, nq, d, seed=556, n_centers=24) + np.float32(offset)).astype(np.float32)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    st = _check(orc, gp, pts, clusters, queries, probes, 2, k, 0, expect_filtered=True)
    assert 0 < st["filter_pairs"] < 0.9 * nq * 2 * (n / S), st                    # cells still prune (2 probed shards of n / S rows per query)
    assert st["filter_candidates"] < 0.05 * st["filter_pairs"], st                  # and the bound still separates


def test_router_bound_path_on_data_far_from_the_origin(orc, gp):
    """the router's matrix-core bound in the frame of the mean centroid: few candidates per (query, shard), verified minima (conftest arms
    GPANN_ROUTER_VERIFY), the oracle's probes"""
    rng = np.random.default_rng(77)
    n, d, S = 40000, 64, 6
    pts = (make_mixture(orc, n, d, n_centers=24) + np.float32(100.0)).astype(np.float32)
    clusters = random_clusters(rng, n, S)
    queries = (make_mixture(orc, 500, d, seed=556, n_centers=24) + np.float32(100.0)).astype(np.float32)
    tree = orc.kmtr_train(pts, clusters, 64, 350, 6000, 0, nthreads=4)
    want_p, want_md = tree.query_batch(queries, 10 ** 6, nthreads=4)
    router = gp.Router(tree.flat(), 0)
    p, md, _ = router.query_batch(queries, 10 ** 6)
    assert np.array_equal(p, want_p) and np.array_equal(md.view(np.uint32), want_md.view(np.uint32))
    st = router.stats()
    assert st["bound_calls"] == 1 and st["verified"] == 1 and st["verify_failures"] == 0 and st["bound_overflow"] == 0 and st["bound_selfcheck"] == 0


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("d", [128, 100, 40, 200])
def test_cells_rest_pass_on_the_8_bit_plane(orc, gp, metric, d, monkeypatch):
    """The rest pass streams the int8 plane (per-cell scaled residuals, integer matrix-core products against the queries' two int8 planes,
    thresholds per (task, cell): scan_cells.h Q8Threshold) — by default; a store built with GPANN_FILTER_Q8=0 has no such plane (and all of its
    bf16 hi tiles) and streams the bf16 hi plane.
    Both return the oracle's bits; the 8-bit pass reads fewer bytes per row and may pass more candidates."""
    rng = np.random.default_rng(d * 3 + metric)
    n, S, nq, k = 60000, 3, 500, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    if metric == 0:
        pts += np.float32(3.0)                       # not centred: the planes live in the frame of the store's mean row
    clusters = random_clusters(rng, n, S, overlap=0.1)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24) + (np.float32(3.0) if metric == 0 else np.float32(0.0))
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    st8 = _check(orc, gp, pts, clusters, queries, probes, 2, k, metric)
    assert st8["filter_rest_row_bytes"] == 32 * ((d + 31) // 32) + 4, st8
    # the sample pass runs on the 8-bit plane too (default); GPANN_FILTER_Q8_FROM_PASS=2 leaves it on the bf16 hi plane.  The cells' rows behind
    # their front are ordered by norm word (default) or left in slab order (GPANN_FILTER_SORT_ROWS=0 at store build): the same bits every way
    monkeypatch.setenv("GPANN_FILTER_Q8_FROM_PASS", "2")
    st8b = _check(orc, gp, pts, clusters, queries, probes, 2, k, metric)
    assert st8b["filter_rest_row_bytes"] == st8["filter_rest_row_bytes"] and st8b["filter_candidates"] <= st8["filter_candidates"] * 1.05 + 100, (st8, st8b)
    monkeypatch.delenv("GPANN_FILTER_Q8_FROM_PASS")
    monkeypatch.setenv("GPANN_FILTER_SORT_ROWS", "0")
    _check(orc, gp, pts, clusters, queries, probes, 2, k, metric)
    monkeypatch.delenv("GPANN_FILTER_SORT_ROWS")
    # (a store built with the 8-bit plane keeps bf16 hi tiles only where the head / sample passes reach: without the plane, all of them)
    monkeypatch.setenv("GPANN_FILTER_Q8", "0")
    st16 = _check(orc, gp, pts, clusters, queries, probes, 2, k, metric)
    ks = next(x for x in (2, 4, 6, 7, 8, 10, 13, 16) if 16 * x >= d)
    assert st16["filter_rest_row_bytes"] == 32 * ks + 4, st16
    assert abs(st8["filter_rest_rows"] - st16["filter_rest_rows"]) <= 0.1 * st16["filter_rest_rows"]   # (about) the same cells are needed either way
    assert st8["filter_candidates"] < 6 * st16["filter_candidates"] + 1000, (st8, st16)


def test_cells_8_bit_plane_extreme_values(orc, gp):
    """rows / queries with a few huge coordinates (a cell's quantisation step is set by its largest residual), duplicated rows (ties at the
    threshold), a constant dimension (zero residuals), tiny clusters: the bound must hold whatever the scale"""
    rng = np.random.default_rng(99)
    n, d, nq, k = 50000, 64, 300, 10
    pts = make_mixture(orc, n, d, n_centers=12)
    pts[:, 5] = 1.25                                  # a constant column
    pts[rng.integers(0, n, 40), rng.integers(0, d, 40)] += np.float32(500.0)   # outliers: huge residuals in their cells
    pts[1000:1400] = pts[1000]                        # duplicates
    queries = make_mixture(orc, nq, d, seed=556, n_centers=12)
    queries[:, 5] = 1.25
    queries[0] = pts[1000]
    queries[1, 3] += np.float32(400.0)
    clusters = [rng.permutation(n).astype(np.uint32)]
    probes = np.zeros((nq, 1), np.int32)
    for metric in (0, 1):
        _check(orc, gp, pts, clusters, queries, probes, 1, k, metric, expect_filtered=None)


def test_cells_all_hi_tiles_kept_allows_switching_the_rest_pass(orc, gp, monkeypatch):
    """GPANN_FILTER_HI_ROWS=all at store build keeps every bf16 hi tile beside the 8-bit plane: GPANN_FILTER_Q8 then chooses per scan"""
    monkeypatch.setenv("GPANN_FILTER_HI_ROWS", "all")
    rng = np.random.default_rng(123)
    n, d, S, nq, k = 50000, 128, 2, 300, 10
    pts = make_mixture(orc, n, d, n_centers=24)
    clusters = random_clusters(rng, n, S)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=24)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    shards = gp.Shards(pts, clusters, 0, 0)
    shards.set_scan_mode(1)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, S, 0, nthreads=4)
    for q8, row_bytes in (("1", 132), ("0", 260)):
        shards.set_option("GPANN_FILTER_Q8", q8)          # (an option of the handle, read by every call)
        assert shards.get_option("FILTER_Q8") == q8       # (names with or without the prefix)
        ids, dist, _ = shards.scan_buckets(queries, k, probes, S)
        st = shards.stats()
        assert st["filtered"] == 1 and st["filter_rest_row_bytes"] == row_bytes, st
        assert np.array_equal(ids, wi) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32))
    # the option table: a name the library does not know is an error, not a silently ignored default; NULL puts the default back; the handle no
    # longer looks at the environment (a variable set AFTER it was created changes nothing)
    with pytest.raises(gp.GpannError):
        shards.set_option("GPANN_FILTER_NOT_A_KNOB", "1")
    shards.set_option("GPANN_FILTER_Q8", None)
    assert shards.get_option("GPANN_FILTER_Q8") == ""
    monkeypatch.setenv("GPANN_FILTER_Q8", "0")
    ids, dist, _ = shards.scan_buckets(queries, k, probes, S)
    assert shards.stats()["filter_rest_row_bytes"] == 132 and np.array_equal(ids, wi)
    assert shards.get_option("GPANN_FILTER_HI_ROWS") == "all"      # (the snapshot taken when the store was created)
    shards.close()


@pytest.mark.parametrize("metric", [0, 1])
def test_cells_without_structure_and_with_huge_cells(orc, gp, metric, monkeypatch):
    """One Gaussian blob (no cell is ever skipped: a dense stream of products, the wavefronts switch to testing H.P' alone before they
    multiply L.P' — scan_cells.h quick8) in cells of ~40 000 rows (the rows behind a cell's front are ordered by norm word in several pieces of
    16 384 — cells.hip RowsByNorm), every query against every shard: the oracle's bits"""
    monkeypatch.setenv("GPANN_CELL_ROWS", "40000")
    rng = np.random.default_rng(5 + metric)
    n, d, S, nq, k = 120000, 64, 2, 1500, 10
    pts = rng.standard_normal((n, d)).astype(np.float32)
    clusters = random_clusters(rng, n, S)
    queries = rng.standard_normal((nq, d)).astype(np.float32)
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    st = _check(orc, gp, pts, clusters, queries, probes, S, k, metric, expect_filtered=True)
    assert st["filter_rest_row_bytes"] == 32 * ((d + 31) // 32) + 4, st
    assert st["filter_pairs"] > 0.5 * nq * n, st          # (nothing could be skipped)
    assert st["filter_dirty_tasks"] == 0, st
