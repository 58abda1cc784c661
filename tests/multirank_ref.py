"""Test helpers: Python restatements of host-side pieces of the multi-rank path, used as expected values — the product's versions are
gpann_lpt_owner / gpann_mask_probes / gpann_comm_merge_router (csrc/comm.hip, include/gpann_comm.h)."""
import numpy as np

NO_ID = np.uint32(0xFFFFFFFF)


def lpt_owner(shard_sizes, world):
    """Longest-processing-time bin packing: biggest shard first onto the least loaded rank.  Returns (owner[S], load[world])."""
    sizes = np.asarray(shard_sizes, np.int64)
    owner = np.zeros(len(sizes), np.int32)
    load = np.zeros(world, np.int64)
    for b in np.argsort(-sizes, kind="stable"):
        r = int(np.argmin(load))
        owner[b] = r
        load[r] += sizes[b]
    return owner, load


def mask_probes(probes, owner, rank):
    """Probe lists with the shards of other ranks replaced by -1 (= skip in gpann_scan_buckets)."""
    p = np.array(probes, np.int32, copy=True)
    valid = p >= 0
    theirs = np.zeros_like(valid)
    theirs[valid] = owner[p[valid]] != rank
    p[theirs] = -1
    return p



def merge_trees(flats, owner):
    """Router trees trained per rank (Router.train over a store that holds only the rank's shards; the other shards get
    empty roots) -> the one tree KMeansTreeRouter::Train builds over all shards.  Shards are independent in training
    (kmeans_tree_router.cpp:20-31), and the flat layout keeps the nodes of a shard consecutive, so the merge is a
    concatenation of per-shard node ranges with re-based child / centroid indices."""
    S = int(flats[0]["S"])
    d = int(flats[0]["d"])
    node_shard, cent_first, child_first, child_count, shard_root, cents = [], [0], [], [], [], []
    for b in range(S):
        f = flats[int(owner[b])]
        lo = int(f["shard_root"][b])
        hi = int(f["shard_root"][b + 1]) if b + 1 < S else len(f["node_shard"])
        base = len(node_shard)
        shard_root.append(base)
        c_lo, c_hi = int(f["node_cent_first"][lo]), int(f["node_cent_first"][hi])
        for u in range(lo, hi):
            node_shard.append(b)
            cc = int(f["node_child_count"][u])
            child_count.append(cc)
            child_first.append(int(f["node_child_first"][u]) - lo + base if cc else 0)
            cent_first.append(cent_first[-1] + int(f["node_cent_first"][u + 1]) - int(f["node_cent_first"][u]))
        cents.append(np.asarray(f["centroids"], np.float32).reshape(-1, d)[c_lo:c_hi])
    return dict(S=S, d=d, node_shard=np.asarray(node_shard, np.int32), node_cent_first=np.asarray(cent_first, np.uint32),
                node_child_first=np.asarray(child_first, np.uint32), node_child_count=np.asarray(child_count, np.uint32),
                shard_root=np.asarray(shard_root, np.uint32), centroids=np.concatenate(cents, axis=0))
