"""Shard-parallel execution over several GPUs (one process per GPU, torch.distributed; backend nccl = RCCL on ROCm).

The path shards naturally: a shard lives on exactly one rank, every rank scans the (query, probe) tasks whose
shard it owns, and the only exchange is an all-gather of the fixed-shape per-slot results followed by a host-side selection.
These helpers are backend-agnostic (the CPU tests drive them over gloo)."""
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


def combine_slots(gathered_dist, gathered_ids, probes, owner):
    """gathered_*: [world][nq][nprobe][k] (every rank's slot arrays, +inf / NO_ID where it does not own the shard).
    Picks each slot from the rank that owns its shard -> ([nq][nprobe][k] dist, ids)."""
    gd = np.asarray(gathered_dist)
    gi = np.asarray(gathered_ids)
    p = np.asarray(probes)[:, :gd.shape[2]]
    sel = np.where(p >= 0, owner[np.clip(p, 0, len(owner) - 1)], 0).astype(np.int64)
    dist = np.take_along_axis(gd, sel[None, :, :, None], axis=0)[0]
    ids = np.take_along_axis(gi, sel[None, :, :, None], axis=0)[0]
    skipped = p < 0
    dist[skipped] = np.inf
    ids[skipped] = NO_ID
    return dist, ids


def all_gather_slots(dist_t, ids_t, torch_dist, world):
    """all_gather of this rank's slot tensors (torch tensors on the backend's device); returns numpy [world, ...] arrays."""
    import torch
    gd = [torch.empty_like(dist_t) for _ in range(world)]
    gi = [torch.empty_like(ids_t) for _ in range(world)]
    torch_dist.all_gather(gd, dist_t)
    torch_dist.all_gather(gi, ids_t)
    return torch.stack(gd).cpu().numpy(), torch.stack(gi).cpu().numpy().view(np.uint32)


def reduce_slots(dist_t, gids_t, torch_dist):
    """The same exchange as all_gather_slots + combine_slots in one collective per array: a slot is scanned by exactly one rank
    (the owner of its shard) and reads (+inf, -1) everywhere else, so the element-wise MIN of the distances and MAX of the
    global ids (int64, -1 = no id) over the ranks is the owner's result.  In place, on whatever device the backend uses."""
    torch_dist.all_reduce(dist_t, op=torch_dist.ReduceOp.MIN)
    torch_dist.all_reduce(gids_t, op=torch_dist.ReduceOp.MAX)
    return dist_t, gids_t


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
