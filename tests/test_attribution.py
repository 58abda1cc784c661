"""CPU: the oracle's restatement of the reference's route / search replay (oracle/attribution.py; reference
src/route_search_combination.cpp:9-63, :121-226) — hand-computed cases, and the host mirror's Convert driver (gp-ann_amd/bin/Convert,
host-only) against it, .pareto row for row, on random .routes / .searches files."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import attribution as att

CONVERT = os.path.join(ROOT, "gp-ann_amd", "bin", "Convert")


def _route(probes, t=0.5, inc=True, algo="KMeansTree", votes=0, budget=20000):
    return dict(routing_algorithm=algo, index_trainer="KMeansTree", votes=votes, ef_search=250, routing_time=t, try_increasing_num_shards=inc, budget=budget,
                num_centroids=64, min_cluster_size=350, buckets_to_probe=probes)


def test_hand_computed_increasing_probes():
    # 2 queries, 2 shards, k = 2.  q0 routes [0, 1], q1 routes [1, 0].
    search = dict(ef_search=0, neighbors=[[[5, 6], [7]], [[6], [8, 9]]], time_query_in_shard=[[1.0, 1.0], [3.0, 3.0]])
    route = _route([[0, 1], [1, 0]], t=2.0)
    steps = list(att.attribute_increasing(route, search, 2, 2, 2))
    # probe 1: q0 -> shard 0 (hits {5,6}), q1 -> shard 1 ({8,9}); work = [1, 3]
    assert steps[0] == ([1.0, 3.0], 4, 1.0)
    # probe 2: q0 adds shard 1 ({6}), q1 adds shard 0 ({7}): unique sets 2 and 3 -> min(., k) = 2 + 2; work = [2, 6]
    assert steps[1] == ([2.0, 6.0], 4, 2.0)
    rows = att.combinations([route], [search], 2, 2, 2, 2, "KMeans")
    # hosts run from num_shards = 2 to max(requested + 20, num_shards) = 22, for both probe counts
    assert len(rows) == 2 * 21
    first = rows[0][0].split(",")
    # 2 hosts: slowest shard 3.0 s, routing 2.0 / 2 hosts -> QPS = 2 / 4 = 0.5, per host 0.25; without routing 2 / 3
    assert first[:6] == ["KMeans", "HNSW", "KMeansTree", "KMeansTree", "0", "0"]
    assert first[6:] == ["1", "1", "1", "0.5", "0.25", "0.666667", "0.333333", "2", "2", "2", "20000", "350", "64"]
    # 3 hosts: the replica goes to shard 1 (3.0 -> 1.5), routing 2/3: QPS = 2 / (1.5 + 0.666..) = 0.923077
    assert rows[1][0].split(",")[9] == "0.923077" and rows[1][0].split(",")[13] == "3"
    # 4 hosts: shard 1 is still the slowest (1.5 > 1.0): 3.0 / 3 = 1.0; then a tie 1.0 / 1.0: max_element takes the FIRST (shard 0)
    assert rows[2][0].split(",")[11] == g(2 / 1.0) and rows[3][0].split(",")[11] == g(2 / 1.0)
    front = att.pareto_front(rows)
    assert all(front[i][2] >= front[i + 1][2] for i in range(len(front) - 1))
    assert not any(a is not b and a[1] <= b[1] and a[2] <= b[2] for a in front for b in front)


def g(x):
    return "%g" % x


def test_hand_computed_variable_probes():
    search = dict(ef_search=0, neighbors=[[[5, 6], [7]], [[6], [8, 9]]], time_query_in_shard=[[1.0, 1.0], [3.0, 3.0]])
    route = _route([[0, 1], [1]], inc=False)
    (work, hits, n_probes), = att.attribute_variable(route, search, 2, 2, 2)
    assert work == [1.0, 6.0] and hits == 2 + 2 and n_probes == 1.5


def _write_routes(path, routes):
    with open(path, "w") as f:
        f.write(f"{len(routes)}\n")
        for r in routes:
            f.write("R\n")
            f.write(f"{r['routing_algorithm']} {r['index_trainer']} {r['votes']} {r['ef_search']} {g(r['routing_time'])} "
                    f"{'true' if r['try_increasing_num_shards'] else 'false'} {len(r['buckets_to_probe'])} {r['budget']} {r['num_centroids']} {r['min_cluster_size']}\n")
            for order in r["buckets_to_probe"]:
                f.write("".join(f"{b} " for b in order) + "\n")


def _write_searches(path, searches):
    with open(path, "w") as f:
        f.write(f"{len(searches)}\n")
        for s in searches:
            S, Q = len(s["neighbors"]), len(s["neighbors"][0])
            f.write("S\n")
            f.write(f"{s['ef_search']} {S} {Q}\n")
            for b in range(S):
                for q in range(Q):
                    f.write("".join(f"{x} " for x in s["neighbors"][b][q]) + "\n")
            for b in range(S):
                f.write("".join(f"{g(t)} " for t in s["time_query_in_shard"][b]) + "\n")


@pytest.mark.skipif(not os.path.exists(CONVERT), reason="gp-ann_amd/bin/Convert missing: run __graft_entry__.build()")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_convert_driver_pareto_equals_the_restatement(tmp_path, seed):
    rng = np.random.default_rng(seed)
    Q, S, k = 60, 6, 10
    searches = []
    for ef in (0, 100):
        neigh = [[sorted(set(int(x) for x in rng.integers(0, 40, rng.integers(0, k + 1)))) for _ in range(Q)] for _ in range(S)]
        times = [[float(g(rng.uniform(1e-5, 1e-3) * (1 + 3 * (b == 2))))] * Q for b in range(S)]   # (one value per shard, as shard_searches.cpp:75)
        searches.append(dict(ef_search=ef, neighbors=neigh, time_query_in_shard=times))
    routes = []
    for i in range(4):
        probes = [list(map(int, rng.permutation(S))) for _ in range(Q)]
        routes.append(_route(probes, t=float(g(rng.uniform(1e-3, 5e-2))), algo="KMeansTree" if i % 2 == 0 else "KMeansTree-Frequency", votes=20 * i, budget=20000 * (i + 1)))
    # a route with a variable number of probes (try_increasing_num_shards false: route_search_combination.cpp:37-63)
    routes.append(_route([list(map(int, rng.permutation(S)[:rng.integers(1, S + 1)])) for _ in range(Q)], t=0.01, inc=False, algo="Pyramid"))
    rf, sf, qf, out = (str(tmp_path / n) for n in ("x.routes", "x.searches", "q.fbin", "x.out"))
    _write_routes(rf, routes)
    _write_searches(sf, searches)
    with open(qf, "wb") as f:
        np.array([Q, 4], np.uint32).tofile(f)
        np.zeros((Q, 4), np.float32).tofile(f)
    # the files round-trip through the restatement's readers
    assert att.read_routes(rf) == routes and att.read_searches(sf) == searches
    r = subprocess.run([CONVERT, rf, sf, out, "KMeans", qf], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = open(out + ".pareto").read().split("\n")
    assert got[0] == att.HEADER and got[-1] == ""
    got = got[1:-1]
    want = att.pareto_rows(routes, searches, 10, Q, S, 40, "KMeans")   # Convert: k = 10, 40 requested shards (convert.cpp:88)
    assert len(want) > 3
    # same rows; the order is by QPS descending, rows of equal QPS may be permuted (unstable std::sort)
    assert sorted(got) == sorted(want)
    qps = [float(x.split(",")[9]) for x in got]
    assert all(qps[i] >= qps[i + 1] for i in range(len(qps) - 1))
