"""GPU: the drop-in boundary end to end.

* gp-ann_amd/bin/SmallScaleQueries — this build's SmallScaleQueries driver (reference small_scale_queries.cpp:25-165, KMTR +
  brute-force rows) over the C++ host mirror;
* oracle/_ref/QueryAttribution — the REFERENCE's own run_query_attribution.cpp compiled UNCHANGED against the host mirror
  headers/libraries (oracle/Makefile `ref`; built only where /root/reference exists, the binary travels to the GPU box).
Outputs (.csv, .routes, .searches, .pareto) are checked against values computed by the CPU oracle on the same files."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_mixture

pytestmark = pytest.mark.gpu

SSQ = os.path.join(ROOT, "gp-ann_amd", "bin", "SmallScaleQueries")
QA = os.path.join(ROOT, "oracle", "_ref", "QueryAttribution")


def write_fbin(path, a):
    a = np.ascontiguousarray(a, np.float32)
    with open(path, "wb") as f:
        np.array(a.shape, np.uint32).tofile(f)
        a.tofile(f)


def write_clusters(path, clusters):
    with open(path, "w") as f:
        for c in clusters:
            f.write(" ".join(str(int(x)) for x in c) + " \n")


@pytest.fixture(scope="module")
def dataset(orc, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("dropin")
    n, d, nq, S, k = 20000, 32, 200, 5, 10
    pts = make_mixture(orc, n, d, n_centers=32)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=32)
    part, _ = orc.kmeans(pts, S, 555, 0, nthreads=4)
    clusters = orc.partition_to_clusters(part)
    files = dict(points=str(tmp / "base.fbin"), queries=str(tmp / "query.fbin"), gt=str(tmp / "gt.bin"), clusters=str(tmp / "base.partition.k=5.KMeans"))
    write_fbin(files["points"], pts)
    write_fbin(files["queries"], queries)
    write_clusters(files["clusters"], clusters)
    gd, gi = orc.ground_truth(pts, queries, k, 0, nthreads=4)
    return dict(tmp=tmp, pts=pts, queries=queries, clusters=clusters, k=k, S=S, files=files, gt_d=gd, gt_i=gi, kth=gd[:, k - 1].copy())


def test_small_scale_queries_driver(orc, dataset):
    assert os.path.exists(SSQ), "gp-ann_amd/bin/SmallScaleQueries missing: run __graft_entry__.build()"
    f, k, S = dataset["files"], dataset["k"], dataset["S"]
    out_csv = str(dataset["tmp"] / "ssq.csv")
    r = subprocess.run([SSQ, f["points"], f["queries"], f["gt"], str(k), f["clusters"], "KMeans", out_csv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # the ground-truth file the driver wrote (big-ann layout) equals the oracle's ground truth
    raw = np.fromfile(f["gt"], np.uint32)
    nq = raw[0]
    assert raw[1] == k
    assert np.array_equal(raw[2:2 + nq * k].reshape(nq, k), dataset["gt_i"])
    assert np.array_equal(raw[2 + nq * k:].view(np.float32).reshape(nq, k).view(np.uint32), dataset["gt_d"].view(np.uint32))
    # expected recall per nprobe from the oracle: Train{32,200,50000}, Query(budget 5000), QueryBucket + DedupNeighbors + Recall
    tree = orc.kmtr_train(dataset["pts"], dataset["clusters"], 32, 200, 50000, 0, nthreads=4)
    probes, _ = tree.query_batch(dataset["queries"], 5000, nthreads=4)
    ivf = orc.ivf_build(dataset["pts"], dataset["clusters"])
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, dataset["queries"], k, probes, S, 0, nthreads=4)
    lines = open(out_csv).read().strip().split("\n")
    assert lines[0] == "partitioning,routing,shard query,probes,latency,routing latency, query latency,recall"
    assert len(lines) == S + 1
    for nprobe in range(1, S + 1):
        cols = lines[nprobe].split(",")
        assert cols[:4] == ["KMeans", "KMTR", "BruteForce", str(nprobe)]
        offs = np.arange(nq + 1, dtype=np.uint64) * (nprobe * k)
        od, _, _ = orc.dedup_neighbors(wd[:, :nprobe].reshape(-1), wi[:, :nprobe].reshape(-1), offs, k)
        want = orc.recall(od.reshape(-1), np.arange(nq + 1, dtype=np.uint64) * k, dataset["kth"], k)
        assert abs(float(cols[7]) - want) < 1e-5, (nprobe, cols[7], want)


def _parse_routes(path):
    lines = open(path).read().split("\n")
    n = int(lines[0])
    pos, routes = 1, []
    for _ in range(n):
        assert lines[pos] == "R"
        head = lines[pos + 1].split()
        nq = int(head[6])
        probes = [[int(x) for x in lines[pos + 2 + q].split()] for q in range(nq)]
        routes.append(dict(algo=head[0], trainer=head[1], votes=int(head[2]), increasing=head[5], budget=int(head[7]), num_centroids=int(head[8]),
                           min_cluster=int(head[9]), probes=probes))
        pos += 2 + nq
    return routes


def _parse_searches(path):
    lines = open(path).read().split("\n")
    n = int(lines[0])
    pos, out = 1, []
    for _ in range(n):
        assert lines[pos] == "S"
        ef, S, Q = (int(x) for x in lines[pos + 1].split())
        neigh = [[[int(x) for x in lines[pos + 2 + b * Q + q].split()] for q in range(Q)] for b in range(S)]
        times = [[float(x) for x in lines[pos + 2 + S * Q + b].split()] for b in range(S)]
        out.append(dict(ef=ef, neighbors=neigh, times=times))
        pos += 2 + S * Q + S
    return out


@pytest.mark.skipif(not os.path.exists(QA), reason="oracle/_ref/QueryAttribution is built only where /root/reference is present")
def test_reference_query_attribution_driver_unchanged(orc, dataset):
    f, k, S = dataset["files"], dataset["k"], dataset["S"]
    out = str(dataset["tmp"] / "qa")
    # GPANN_ROUTER_VERIFY: inside the driver every minimum of the router's matrix-core bound path is compared with the canonical
    # evaluation of all pairs; a difference (or a fired self-check, GPANN_ROUTER_STRICT) ends the driver with an error and a report
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    env = dict(os.environ, GPANN_MAX_ROUTER_BUDGET="100000", GPANN_ROUTER_VERIFY="1", GPANN_ROUTER_STRICT="1",
               GPANN_VERIFY_REPORT=os.path.join(ROOT, "gpurun_out", "router_verify_report.txt"))
    r = subprocess.run([QA, f["points"], f["queries"], f["gt"] + ".qa", str(k), f["clusters"], out, "KMeans", str(S)], capture_output=True, text=True,
                       timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "self-check" not in r.stderr and "VERIFY" not in r.stderr, r.stderr[-3000:]
    nq = len(dataset["queries"])
    routes = _parse_routes(out + ".routes")
    assert len(routes) == 2 * 10                      # budgets 20k and 100k: 1 tree-search + 9 frequency routes each
    for budget in (20000, 100000):
        tree = orc.kmtr_train(dataset["pts"], dataset["clusters"], 64, 350, budget, 0, nthreads=4)
        want, _ = tree.query_batch(dataset["queries"], 50000, nthreads=4)
        got = [rt for rt in routes if rt["budget"] == budget and rt["algo"] == "KMeansTree"]
        assert len(got) == 1 and got[0]["num_centroids"] == 64 and got[0]["min_cluster"] == 350 and got[0]["increasing"] == "true"
        if os.environ.get("GPANN_TEST_FORCE_ROUTE_MISMATCH"):   # exercises the reporting branch below (it must end in AssertionError)
            want = want[:, ::-1].copy()
        if not np.array_equal(np.asarray(got[0]["probes"]), want):   # say which stage differs before failing
            import pygpann as gp
            shards = gp.Shards(dataset["pts"], dataset["clusters"], plain=True)
            dev = gp.Router.train(shards, len(dataset["pts"]), 64, 350, budget)
            flat, oflat = dev.export(), tree.flat()
            same_tree = all(np.array_equal(np.asarray(flat[key]).view(np.uint32) if key == "centroids" else flat[key],
                                           np.asarray(oflat[key]).view(np.uint32) if key == "centroids" else oflat[key])
                            for key in ("node_shard", "node_cent_first", "node_child_first", "node_child_count", "shard_root", "centroids"))
            pr, md, _ = dev.query_batch(dataset["queries"], 50000)
            _, omd = tree.query_batch(dataset["queries"], 50000, nthreads=4)
            bad = np.flatnonzero((np.asarray(got[0]["probes"]) != want).any(axis=1))
            report = (f"budget {budget}: driver routes differ from the oracle's for {len(bad)} of {nq} queries (first {bad[:5]}); in-process device "
                      f"training gives the oracle's tree: {same_tree}; in-process routes equal the oracle's: {np.array_equal(pr, want)}; "
                      f"in-process min_dist equal: {np.array_equal(md.view(np.uint32), omd.view(np.uint32))}; driver rows {np.asarray(got[0]['probes'])[bad[:5]].tolist()} "
                      f"oracle rows {want[bad[:5]].tolist()}; driver stdout tail: {r.stdout[-1200:]} stderr tail: {r.stderr[-800:]}")
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "dropin_route_mismatch.txt"), "a") as fh:
                fh.write(report + "\n")
            raise AssertionError(report)
        # frequency routing (kmeans_tree_router.cpp:132-172 + kmeans_tree_router.h:23-44)
        for rt in (x for x in routes if x["budget"] == budget and x["algo"] == "KMeansTree-Frequency"):
            for q in (0, 7, nq - 1):
                md, nd, ns = tree.frequency_query(dataset["queries"][q], 50000, 500)
                assert rt["probes"][q] == list(orc.frequency_probes(md, ns, rt["votes"]))
    searches = _parse_searches(out + ".searches")
    assert len(searches) == 1 and searches[0]["ef"] == 0
    ivf = orc.ivf_build(dataset["pts"], dataset["clusters"])
    probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    bd, bi, _ = orc.ivf_query_bucket_batch(ivf, dataset["queries"], k, probes, S, 0, nthreads=4)
    for b in range(S):
        for q in range(nq):
            keep = [int(bi[q, b, x]) for x in range(k) if bi[q, b, x] != 0xFFFFFFFF and bd[q, b, x] <= dataset["kth"][q]]
            assert searches[0]["neighbors"][b][q] == keep[::-1]
        assert len(searches[0]["times"][b]) == nq and searches[0]["times"][b][0] > 0
    pareto = open(out + ".pareto").read().split("\n")
    assert pareto[0].startswith("partitioning,shard query,routing query,routing index,ef-search-shard") and len(pareto) > 2
    # the best recall on the front is 1.0: all shards probed = exact
    assert max(float(line.split(",")[8]) for line in pareto[1:] if line) > 0.999


CONVERT = os.path.join(ROOT, "gp-ann_amd", "bin", "Convert")


def test_convert_driver_replays_routes_over_searches(orc, dataset):
    """this build's Convert (the live branch of the reference's convert.cpp:62-89): .routes + .searches written through the host
    mirror -> the Pareto file PrintCombinationsOfRoutesAndSearches produces; host-only replay, no GPU work of its own"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
    f, k, S = dataset["files"], dataset["k"], dataset["S"]
    out = str(dataset["tmp"] / "cv")
    # routes and searches from the SmallScaleQueries-independent path: run the reference driver's flow through QueryAttribution if
    # it exists, else skip (the .routes/.searches writers are exercised there)
    if not os.path.exists(QA):
        pytest.skip("needs oracle/_ref/QueryAttribution to produce .routes / .searches")
    env = dict(os.environ, GPANN_MAX_ROUTER_BUDGET="20000")
    r = subprocess.run([QA, f["points"], f["queries"], f["gt"] + ".cv", str(k), f["clusters"], out, "KMeans", str(S)], capture_output=True, text=True,
                       timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([CONVERT, out + ".routes", out + ".searches", out + ".converted", "KMeans", f["queries"]], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    a = open(out + ".pareto").read().strip().split("\n")
    b = open(out + ".converted.pareto").read().strip().split("\n")
    assert a[0] == b[0] and len(b) > 2
    # row for row against the oracle's restatement of route_search_combination.cpp (oracle/attribution.py) on the same files
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import attribution as att
    routes, searches = att.read_routes(out + ".routes"), att.read_searches(out + ".searches")
    nq = len(dataset["queries"])
    want = att.pareto_rows(routes, searches, 10, nq, S, 40, "KMeans")             # Convert: k = 10, 40 requested shards (convert.cpp:88)
    assert a[0] == att.HEADER and sorted(b[1:]) == sorted(want)
    # the attribution driver computed its front from the unrounded times it measured; the files hold 6 significant digits.  Every row
    # of its front must be one of the restatement's (route, search, probes, hosts) combinations: same text columns, same recall and
    # probe count, QPS columns within the rounding of the times
    combos = {}
    for row, _, _ in att.combinations(routes, searches, k, nq, S, S, "KMeans"):   # QueryAttribution: requested shards = S
        c = row.split(",")
        combos[tuple(c[:6] + c[7:9] + c[13:])] = c
    assert len(a) - 1 <= len(combos)
    for row in a[1:]:
        c = row.split(",")
        ref = combos[tuple(c[:6] + c[7:9] + c[13:])]
        for x, y in zip((c[6], *c[9:13]), (ref[6], *ref[9:13])):
            assert abs(float(x) - float(y)) <= 2e-4 * abs(float(y)), (row, ref)
    # same routes and searches; Convert asks for 40 hosts like the reference (convert.cpp:88), the attribution driver for S, so the
    # fronts differ in length — both must reach the exact answer (all shards probed) and be sorted Pareto fronts of recall
    for front in (a, b):
        rec = [float(ln.split(",")[8]) for ln in front[1:]]
        assert max(rec) > 0.999 and all(0.0 <= x <= 1.0 for x in rec)
