"""CPU: the property GPANN_SCAN_FOR_MERGE rests on (include/gpann.h), checked with the oracle's DedupNeighbors
(small_scale_queries.cpp:15-23) on random per-slot lists: cutting every slot of a query after its last entry with distance <= tau_q
(tau_q = the smallest k-th distance among the query's slots) never changes the merged result — with ids repeated across slots, ties at
the bound, short and empty slots."""
import numpy as np
import pytest

NO_ID = 0xFFFFFFFF


def _random_slots(rng, nq, nprobe, k, n_ids, levels):
    """per-slot lists as an exact scan would return them: ascending distance, ids unique within a slot, the same id may appear in
    several slots of a query with the SAME distance (it is the same point); few distinct distance levels = many ties"""
    dist = np.full((nq, nprobe, k), np.inf, np.float32)
    ids = np.full((nq, nprobe, k), NO_ID, np.uint32)
    for q in range(nq):
        point_dist = rng.integers(0, levels, size=n_ids).astype(np.float32) * np.float32(0.25) - np.float32(1.0)   # (negative values: inner product)
        for j in range(nprobe):
            size = int(rng.choice([0, 3, k, 4 * k, 10 * k]))                     # rows of the "shard" this slot scanned
            members = rng.choice(n_ids, size=min(size, n_ids), replace=False)
            order = np.lexsort((members, point_dist[members]))[:k]
            dist[q, j, :len(order)] = point_dist[members][order]
            ids[q, j, :len(order)] = members[order]
    return dist, ids


def _merge(orc, dist, ids, k):
    nq = dist.shape[0]
    fd, fi = dist.reshape(nq, -1), ids.reshape(nq, -1)
    keep = fi != NO_ID
    offs = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.uint64)
    return orc.dedup_neighbors(fd[keep], fi[keep], offs, k)


@pytest.mark.parametrize("k,nprobe,levels", [(10, 4, 1000), (10, 6, 6), (1, 3, 4), (5, 2, 3), (10, 1, 50)])
def test_cut_lists_merge_to_the_same_neighbours(orc, k, nprobe, levels):
    rng = np.random.default_rng(k * 100 + nprobe)
    dist, ids = _random_slots(rng, 300, nprobe, k, 200, levels)
    tau = dist[:, :, -1].min(axis=1)
    cut = ~(dist <= tau[:, None, None])
    cd, ci = dist.copy(), ids.copy()
    cd[cut] = np.inf
    ci[cut] = NO_ID
    full, short = _merge(orc, dist, ids, k), _merge(orc, cd, ci, k)
    assert np.array_equal(full[2], short[2])
    assert np.array_equal(full[1], short[1])
    assert np.array_equal(full[0].view(np.uint32), short[0].view(np.uint32))
    if nprobe > 1 and levels > 10:
        assert (cut & (ids != NO_ID)).any()          # the bound does cut real entries in this configuration
