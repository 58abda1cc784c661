"""CPU: replica placement for query-load balance (gpann_plan_replicas / gpann_mask_probes_replicated, include/gpann_comm.h) — the greedy
"one more replica for the slowest shard" rule of the reference's QPS model (src/route_search_combination.cpp:149-183) as a real
shard -> rank placement.  Simulated on NS-shaped routes: 40 k-means shards whose sizes spread 40x (as bench.py's do), 10 000 queries
routed to shards in proportion to their size; makespan = max over ranks of the task rows it scans."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
from pygpann import comm as gc


def _ns_like(seed, S=40, n=100_000_000, nq=10_000):
    rng = np.random.default_rng(seed)
    w = rng.lognormal(0.0, 0.9, S)
    rows = np.maximum(50_000, (w / w.sum() * n)).astype(np.int64)
    first = rng.choice(S, size=nq, p=rows / rows.sum())            # nprobe = 1: the shard a query is routed to
    return rows, first


def _rank_task_rows(rows, probes, masks):
    return np.array([int(rows[m[m >= 0]].sum()) for m in masks], np.float64)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_replicas_balance_the_query_load(seed, world):
    rows, first = _ns_like(seed)
    S, nq = len(rows), len(first)
    probes = first.reshape(nq, 1).astype(np.int32)
    tasks = np.bincount(first, minlength=S)
    work = tasks.astype(np.float64) * rows
    # without replicas: LPT by rows (memory balance) leaves the QUERY load far from even
    owner, _ = gc.lpt_owner(rows, world)
    base = _rank_task_rows(rows, probes, [gc.mask_probes(probes, 1, owner, r) for r in range(world)])
    # with replicas: every rank may hold twice its fair share of the rows
    cap = int(max(2 * rows.sum() / world, 1.25 * rows.max()))
    rep_begin, rep_rank, rank_work, rank_rows = gc.plan_replicas(rows, work, world, cap, 1.05)
    masks = [gc.mask_probes_replicated(probes, 1, rep_begin, rep_rank, r) for r in range(world)]
    # every slot is scanned by exactly one rank, and that rank holds the shard
    stacked = np.stack(masks)
    assert ((stacked >= 0).sum(axis=0) == 1).all()
    for r in range(world):
        held = {b for b in range(S) if r in rep_rank[rep_begin[b]:rep_begin[b + 1]]}
        assert set(np.unique(masks[r][masks[r] >= 0])) <= held
    for b in range(S):                                               # a rank holds a shard at most once; capacity respected
        rr = rep_rank[rep_begin[b]:rep_begin[b + 1]]
        assert len(rr) >= 1 and len(set(rr)) == len(rr)
    assert (rank_rows <= cap).all() and rank_rows.sum() == sum(int(rows[b]) * (rep_begin[b + 1] - rep_begin[b]) for b in range(S))
    got = _rank_task_rows(rows, probes, masks)
    assert got.sum() == base.sum()                                   # the same work, spread differently
    ratio, base_ratio = got.max() / got.mean(), base.max() / base.mean()
    assert ratio <= 1.10, (ratio, base_ratio)                        # makespan within 10 % of the mean
    assert ratio <= base_ratio + 1e-9
    assert abs(rank_work.sum() - work.sum()) < 1e-6 * work.sum()


def test_no_capacity_for_replicas_falls_back_to_one_replica_each():
    rows, first = _ns_like(7)
    work = np.bincount(first, minlength=len(rows)).astype(np.float64) * rows
    cap = int(rows.sum() / 4 * 1.3)                                  # barely room for the primaries
    rep_begin, rep_rank, _, rank_rows = gc.plan_replicas(rows, work, 4, cap, 1.0)
    assert (rank_rows <= cap).all() and (np.diff(rep_begin) >= 1).all()
    with pytest.raises(Exception):
        gc.plan_replicas(rows, work, 4, int(rows.max()) - 1, 1.05)   # the largest shard fits nowhere


def test_single_rank_and_skipped_slots():
    rep_begin, rep_rank, rank_work, _ = gc.plan_replicas([10, 20, 30], [1.0, 2.0, 3.0], 1, 100, 1.05)
    assert rep_begin.tolist() == [0, 1, 2, 3] and rep_rank.tolist() == [0, 0, 0] and rank_work.tolist() == [6.0]
    p = np.array([[0, -1, 2], [1, 7, 0]], np.int32)                  # -1 and an out-of-range shard id stay skipped
    m = gc.mask_probes_replicated(p, 3, rep_begin, rep_rank, 0)
    assert m.tolist() == [[0, -1, 2], [1, -1, 0]]
