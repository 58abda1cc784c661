"""GPU: the HIP path replayed on the committed golden vectors (tests/golden/hotpath_*.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["l2_d128", "l2_d100", "ip_d200", "l2_d7"])
def test_golden_vectors(orc, gp, name):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"hotpath_{name}.npz"))
    n, d, k, metric = int(z["n"]), int(z["d"]), int(z["k"]), int(z["metric"])
    pts = orc.synth_rows(555, 7, 16, 0.35, 0, n, d)          # the generator is part of the fixture definition
    a, b = z["tie_rows"]
    pts[a:b] = pts[a]
    queries = orc.synth_rows(556, 7, 16, 0.35, 0, 24, d)
    queries[0] = pts[a]
    offs = z["cluster_offsets"]
    clusters = [z["cluster_ids"][int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    shards = gp.Shards(pts, clusters, metric, 0)
    ids, dist, _ = shards.scan_buckets(queries, k, z["probes"], 3)
    assert np.array_equal(ids, z["scan_ids"]) and np.array_equal(dist.view(np.uint32), z["scan_dist"].view(np.uint32))
    gi, gd, _ = gp.ground_truth(pts, queries, k, metric)
    assert np.array_equal(gi, z["gt_ids"]) and np.array_equal(gd.view(np.uint32), z["gt_dist"].view(np.uint32))
    flat = {key[5:]: z[key] for key in z.files if key.startswith("tree_")}
    flat["d"], flat["S"] = int(flat["d"]), int(flat["S"])
    router = gp.Router(flat, metric, 0)
    for budget, pk, mk in ((int(z["router_budget"][0]), "router_probes", "router_min_dist"),
                           (int(z["router_budget"][1]), "router_probes_small", "router_min_dist_small")):
        probes, md, _ = router.query_batch(queries, budget)
        assert np.array_equal(probes, z[pk]) and np.array_equal(md.view(np.uint32), z[mk].view(np.uint32))
    # FrequencyQuery on the GPU against the oracle on the same tree
    tree = orc.kmtr_from_flat(flat, metric)
    md, nd, ns, nc, _ = router.frequency_query_batch(queries, 150, 40)
    for q in range(len(queries)):
        wmd, wnd, wns = tree.frequency_query(queries[q], 150, 40)
        assert np.array_equal(md[q], wmd) and nc[q] == len(wnd)
        assert np.array_equal(nd[q, :nc[q]], wnd) and np.array_equal(ns[q, :nc[q]], wns)
