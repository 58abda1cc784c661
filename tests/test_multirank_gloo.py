"""CPU, world_size 2 over gloo: the shard-parallel exchange used by bench.py --gpus N (pygpann.multigpu).
Each rank scans only the shards it owns (the CPU oracle stands in for the GPU scan here), the per-slot results are
all-gathered and combined by owner; the outcome must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, make_mixture, random_clusters


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "gp-ann_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import pyoracle
    from pygpann import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = pyoracle.Oracle("strict")
    rng = np.random.default_rng(21)
    n, d, S, nq, k, nprobe = 4000, 16, 7, 60, 5, 3
    pts = orc.synth_rows(555, 7, 16, 0.35, 0, n, d)
    queries = orc.synth_rows(556, 7, 16, 0.35, 0, nq, d)
    clusters = random_clusters(rng, n, S, overlap=0.2)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    probes[5, 1] = -1                                     # a skipped slot
    owner, load = multigpu.lpt_owner([len(c) for c in clusters], world)
    assert load.sum() == sum(len(c) for c in clusters) and set(owner) == set(range(world))
    # this rank's store holds only its shards (others empty), like bench.py
    mine = [c if owner[b] == rank else np.zeros(0, np.uint32) for b, c in enumerate(clusters)]
    ivf = orc.ivf_build(pts, mine)
    lp = multigpu.mask_probes(probes[:, :nprobe], owner, rank)
    assert ((lp == -1) | (owner[np.clip(lp, 0, S - 1)] == rank)).all()
    od, oi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, lp, nprobe, 0)
    gd, gi = multigpu.all_gather_slots(torch.from_numpy(od), torch.from_numpy(oi.view(np.int32)), dist, world)
    cd, ci = multigpu.combine_slots(gd, gi, probes, owner)
    # the one-collective form of the same exchange (what bench.py runs): MIN over distances, MAX over int64 ids (-1 = none)
    rd = torch.from_numpy(od.copy())
    ri = torch.from_numpy(np.where(oi == 0xFFFFFFFF, -1, oi.astype(np.int64)))
    multigpu.reduce_slots(rd, ri, dist)
    assert np.array_equal(rd.numpy().view(np.uint32), cd.view(np.uint32)) and np.array_equal(ri.numpy().astype(np.uint32), ci)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    if rank == 0:
        full = orc.ivf_build(pts, clusters)
        wd, wi, _ = orc.ivf_query_bucket_batch(full, queries, k, probes, nprobe, 0)
        np.savez(os.path.join(out_dir, "result.npz"), ok=np.array([np.array_equal(ci, wi) and np.array_equal(cd.view(np.uint32), wd.view(np.uint32))]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_parallel_exchange(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert bool(np.load(tmp_path / "result.npz")["ok"][0])


def test_lpt_and_masks():
    from pygpann import multigpu
    owner, load = multigpu.lpt_owner([10, 9, 8, 1, 1, 1], 3)
    assert sorted(load.tolist()) == [10, 10, 10]
    p = np.array([[0, 3, -1], [5, 1, 2]], np.int32)
    m = multigpu.mask_probes(p, owner, int(owner[0]))
    assert m[0, 0] == 0 and m[0, 2] == -1
    assert all(m[i, j] == -1 or owner[m[i, j]] == owner[0] for i in range(2) for j in range(3))
