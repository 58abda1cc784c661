"""CPU: pins of the oracle (and of the product's host mirror) against the REFERENCE's own code.

Two sources: (1) committed fixtures in tests/golden/ref_*.npz|txt, produced by tests/golden/make_golden.py from
oracle/_ref/ref_core = the reference's src/topn.h + src/defs.cpp + src/metis_io.cpp compiled from their own sources;
(2) when that binary is present (build container), live random cases."""
import os

import numpy as np
import pytest

import ref_core
from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


def _topn_cases():
    z = np.load(os.path.join(GOLD, "ref_topn.npz"))
    for i in range(int(z["n_cases"])):
        yield z[f"d{i}"], z[f"i{i}"], int(z[f"n{i}"]), z[f"ad{i}"], z[f"ai{i}"], z[f"dd{i}"], z[f"di{i}"]


def test_oracle_topn_matches_reference_fixture(orc):
    for d, ids, n, ad, ai, dd, di in _topn_cases():
        gd, gi = orc.topn(d, ids, n)
        assert np.array_equal(gd, ad) and np.array_equal(gi, ai)          # TopN::Take(): ascending (dist, id)
        assert np.array_equal(gd[::-1], dd) and np.array_equal(gi[::-1], di)   # ConvertTopKToNNVec: descending


def test_oracle_partition_conversions_match_reference_fixture(orc):
    z = np.load(os.path.join(GOLD, "ref_partition.npz"))
    clusters = orc.partition_to_clusters(z["partition"])
    assert len(clusters) == int(z["num_parts"])
    offs = z["cluster_offsets"]
    for b, c in enumerate(clusters):
        assert np.array_equal(c, z["cluster_ids"][int(offs[b]):int(offs[b + 1])])
    # RemapPartitionIDs compacts the ids because shard 2 is empty
    remapped = z["remapped"]
    assert set(np.unique(remapped)) == set(range(len(np.unique(z["partition"]))))
    order = {old: new for new, old in enumerate(sorted(np.unique(z["partition"])))}
    assert np.array_equal(remapped, np.vectorize(order.get)(z["partition"]))
    # cover of a partition: exactly one shard per point
    assert (z["cover_sizes"] == 1).all() and np.array_equal(z["cover_flat"], z["partition"])


def test_clusters_text_format_fixture():
    # ReadClusters keeps empty lines as empty shards and WriteClusters writes "id id ... \n" (metis_io.cpp:40-64)
    src = open(os.path.join(GOLD, "ref_clusters.txt")).read()
    out = open(os.path.join(GOLD, "ref_clusters.txt.out")).read()
    assert src == "3 1 4 1 5 \n\n9 2 6\n5 3 5 8 9 7 9 \n"
    assert out == "3 1 4 1 5 \n\n9 2 6 \n5 3 5 8 9 7 9 \n"


@pytest.mark.skipif(not ref_core.available(), reason="oracle/_ref/ref_core needs /root/reference (build container only)")
def test_oracle_topn_matches_reference_live(orc):
    ref = ref_core.RefCore()
    rng = np.random.default_rng(99)
    cases = []
    for _ in range(40):
        n_items = int(rng.integers(0, 400))
        n = int(rng.integers(1, 80))
        d = (rng.integers(0, 7, n_items) * np.float32(0.25)).astype(np.float32)
        if rng.random() < 0.3:
            d = rng.standard_normal(n_items).astype(np.float32)
        cases.append((d, rng.integers(0, 50, n_items).astype(np.uint32), n))
    for (d, ids, n), (rd, ri) in zip(cases, ref.topn(cases)):
        gd, gi = orc.topn(d, ids, n)
        assert np.array_equal(gd, rd) and np.array_equal(gi, ri)


@pytest.mark.skipif(not ref_core.available(), reason="oracle/_ref/ref_core needs /root/reference (build container only)")
def test_extract_points_in_bucket_live(orc):
    ref = ref_core.RefCore()
    rng = np.random.default_rng(5)
    pts = rng.standard_normal((50, 6)).astype(np.float32)
    bucket = rng.integers(0, 50, 20).astype(np.uint32)
    got = ref.extract(pts, bucket)
    clustered, offsets, perm = orc.ivf_build(pts, [bucket])     # InvertedIndex gathers rows the same way
    assert np.array_equal(got, pts[bucket]) and np.array_equal(clustered, got) and np.array_equal(perm, bucket)
