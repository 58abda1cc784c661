"""CPU: multirank_ref.merge_trees (tests/multirank_ref.py) — per-rank router parts (own shards trained, other shards empty roots) merge into the tree
KMeansTreeRouter::Train builds over all shards."""
import numpy as np

from conftest import make_mixture, random_clusters


def _part(full, owner, rank):
    """what a rank's Router.train(...).export() looks like: subtrees of its shards, a root without centroids elsewhere"""
    S, d = int(full["S"]), int(full["d"])
    ns, cf, chf, chc, sr, cen = [], [0], [], [], [], []
    for b in range(S):
        lo = int(full["shard_root"][b])
        hi = int(full["shard_root"][b + 1]) if b + 1 < S else len(full["node_shard"])
        base = len(ns)
        sr.append(base)
        if owner[b] != rank:
            ns.append(b); chc.append(0); chf.append(0); cf.append(cf[-1])
            continue
        for u in range(lo, hi):
            ns.append(b)
            cc = int(full["node_child_count"][u])
            chc.append(cc)
            chf.append(int(full["node_child_first"][u]) - lo + base if cc else 0)
            cf.append(cf[-1] + int(full["node_cent_first"][u + 1]) - int(full["node_cent_first"][u]))
        cen.append(full["centroids"][int(full["node_cent_first"][lo]):int(full["node_cent_first"][hi])])
    cen = np.concatenate(cen, axis=0) if cen else np.zeros((0, d), np.float32)
    return dict(S=S, d=d, node_shard=np.asarray(ns, np.int32), node_cent_first=np.asarray(cf, np.uint32), node_child_first=np.asarray(chf, np.uint32),
                node_child_count=np.asarray(chc, np.uint32), shard_root=np.asarray(sr, np.uint32), centroids=cen)


def test_merge_trees_roundtrip(orc):
    import multirank_ref
    from pygpann import comm as gcomm
    rng = np.random.default_rng(2)
    n, d, S = 8000, 10, 6
    pts = make_mixture(orc, n, d, n_centers=12)
    clusters = random_clusters(rng, n, S)
    full = orc.kmtr_train(pts, clusters, 8, 30, 900, 0, nthreads=4).flat()
    assert (full["node_child_count"] > 0).any()
    for world in (1, 2, 3):
        owner, _ = gcomm.lpt_owner(np.asarray([len(c) for c in clusters]), world)
        parts = [_part(full, owner, r) for r in range(world)]
        merged = multirank_ref.merge_trees(parts, owner)
        for key in ("node_shard", "node_cent_first", "node_child_first", "node_child_count", "shard_root"):
            assert np.array_equal(merged[key], full[key]), (world, key)
        assert np.array_equal(merged["centroids"].view(np.uint32), np.asarray(full["centroids"], np.float32).view(np.uint32))
