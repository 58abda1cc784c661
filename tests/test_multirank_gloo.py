"""CPU, world_size 2 over gloo: the N > 1 path of bench.py / DistributedBench with the PRODUCT's exchange library (libgpann_comm.so,
include/gpann_comm.h) — placement (gpann_lpt_owner, gpann_plan_replicas), probe masks (gpann_mask_probes[_replicated]), the result exchange
(gpann_comm_gather_slots, TCP backend on host buffers) and the host merge (gpann_merge_dedup) — cross-checked against the same exchange done
with torch.distributed (gloo all_gather + a numpy owner selection) and against the single-process oracle.  Each rank scans only the slots it
owns (the CPU oracle stands in for the GPU scan), with complete per-slot lists and with lists cut at the rank's merge bound
(GPANN_SCAN_FOR_MERGE: a rank only sees its own slots of a query, so its bound is looser than the single-process one — the merged
neighbours must still be the reference's)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, random_clusters

NO_ID = 0xFFFFFFFF


def _free_port():
    from conftest import free_port   # (below the ephemeral range: see there)
    return free_port(2)


def _cut_at_merge_bound(d, i):
    """GPANN_SCAN_FOR_MERGE on [nq][nprobe][k] lists (include/gpann.h)"""
    tau = d[:, :, -1].min(axis=1)
    cut = ~(d <= tau[:, None, None])
    d, i = d.copy(), i.copy()
    d[cut] = np.inf
    i[cut] = NO_ID
    return d, i


def _worker(rank, world, port, comm_port, out_dir, replicas):
    for p in (ROOT, os.path.join(ROOT, "gp-ann_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import pygpann
    import pyoracle
    from pygpann import comm as gcomm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = gcomm.Comm(gcomm.TCP, rank=rank, world=world, addr="127.0.0.1", port=comm_port)
    orc = pyoracle.Oracle("strict")
    rng = np.random.default_rng(21)
    n, d, S, nq, k, nprobe = 4000, 16, 7, 60, 5, 3
    pts = orc.synth_rows(555, 7, 16, 0.35, 0, n, d)
    queries = orc.synth_rows(556, 7, 16, 0.35, 0, nq, d)
    clusters = random_clusters(rng, n, S, overlap=0.2)            # ids repeat across shards: the merge drops copies
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    probes[5, 1] = -1                                             # a skipped slot
    sizes = np.asarray([len(c) for c in clusters], np.uint64)
    if replicas:
        work = sizes.astype(np.float64) * (1.0 + np.arange(S))
        rep_begin, rep_rank, _, _ = gcomm.plan_replicas(sizes, work, world, int(1.6 * sizes.sum() / world), 1.0)
        held = [b for b in range(S) if rank in rep_rank[rep_begin[b]:rep_begin[b + 1]]]
        lp = gcomm.mask_probes_replicated(probes, nprobe, rep_begin, rep_rank, rank)
    else:
        owner, load = gcomm.lpt_owner(sizes, world)
        assert load.sum() == sizes.sum() and set(owner.tolist()) == set(range(world))
        held = [b for b in range(S) if owner[b] == rank]
        lp = gcomm.mask_probes(probes, nprobe, owner, rank)
    assert ((lp == -1) | np.isin(lp, held)).all()
    # this rank's store holds only its shards (others empty), like bench.py
    ivf = orc.ivf_build(pts, [c if b in held else np.zeros(0, np.uint32) for b, c in enumerate(clusters)])
    od, oi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, lp, nprobe, 0)
    ok = True
    for cut in (False, True):
        sd, si = _cut_at_merge_bound(od, oi) if cut else (od, oi)
        # the product's exchange ...
        n_slots = sd.size
        out_d, out_i = np.empty_like(sd), np.empty_like(si)
        comm.gather_slots(np.ascontiguousarray(sd), np.ascontiguousarray(si), n_slots, out_d, out_i, np.empty(comm.gather_slots_scratch(n_slots), np.uint8))
        # ... against torch.distributed: all_gather of every rank's lists, each slot from the one rank where it is not empty
        gd = [torch.empty((nq, nprobe, k), dtype=torch.float32) for _ in range(world)]
        gi = [torch.empty((nq, nprobe, k), dtype=torch.int32) for _ in range(world)]
        dist.all_gather(gd, torch.from_numpy(np.ascontiguousarray(sd)))
        dist.all_gather(gi, torch.from_numpy(np.ascontiguousarray(si).view(np.int32)))
        gd = torch.stack(gd).numpy()
        gi = torch.stack(gi).numpy().view(np.uint32)
        has = gi != NO_ID
        assert (has.sum(axis=0) <= 1).all()                       # a slot entry is non-empty on at most one rank
        want_i = np.where(has.any(axis=0), gi.min(axis=0), NO_ID).astype(np.uint32)
        want_d = gd.min(axis=0)
        ok = ok and np.array_equal(out_i, want_i) and np.array_equal(out_d.view(np.uint32), want_d.view(np.uint32))
        # the host merge over the exchanged lists = the reference's merge over the single-process lists
        md, mi, mc = pygpann.merge_dedup(out_d, out_i, nprobe, k)
        full = orc.ivf_build(pts, clusters)
        wd, wi, _ = orc.ivf_query_bucket_batch(full, queries, k, probes, nprobe, 0)
        if not cut:
            ok = ok and np.array_equal(out_i, wi) and np.array_equal(out_d.view(np.uint32), wd.view(np.uint32))
        fd, fi, fc = pygpann.merge_dedup(wd, wi, nprobe, k)
        ok = ok and np.array_equal(mi, fi) and np.array_equal(md.view(np.uint32), fd.view(np.uint32)) and np.array_equal(mc, fc)
    # max-over-ranks timing reduction: the library's and torch.distributed's
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == world and comm.max_double(float(rank + 1)) == world
    flags = [torch.zeros(1, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(flags, torch.tensor([1 if ok else 0], dtype=torch.int32))
    if rank == 0:
        np.savez(os.path.join(out_dir, "result.npz"), ok=np.array([all(int(f.item()) == 1 for f in flags)]))
    comm.barrier()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("replicas", [False, True])
def test_two_rank_shard_parallel_exchange(tmp_path, replicas):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), _free_port(), str(tmp_path), replicas), nprocs=2, join=True)
    assert bool(np.load(tmp_path / "result.npz")["ok"][0])


def test_lpt_and_masks():
    sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
    from pygpann import comm as gcomm
    owner, load = gcomm.lpt_owner([10, 9, 8, 1, 1, 1], 3)
    assert sorted(load.tolist()) == [10, 10, 10]
    p = np.array([[0, 3, -1], [5, 1, 2]], np.int32)
    m = gcomm.mask_probes(p, 3, owner, int(owner[0]))
    assert m[0, 0] == 0 and m[0, 2] == -1
    assert all(m[i, j] == -1 or owner[m[i, j]] == owner[0] for i in range(2) for j in range(3))


def _abort_worker(rank, world, comm_port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "gp-ann_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import time
    from pygpann import GpannError
    from pygpann import comm as gcomm
    comm = gcomm.Comm(gcomm.TCP, rank=rank, world=world, addr="127.0.0.1", port=comm_port)
    comm.barrier()
    comm.wait(None, 5)                      # every rank alive, nothing enqueued: returns at once
    comm.check_peers()
    if rank == 1:
        comm.abort()                        # "this rank failed": no farewell collective, the connections are closed
        return
    t0 = time.time()
    lost, msg = False, ""
    while time.time() - t0 < 20.0:          # rank 0 notices without taking part in a collective
        comm.wait(None, 60)                 # (nothing pending: a success whatever rank 1 has done meanwhile — completed work is never an error)
        try:
            comm.check_peers()
            time.sleep(0.05)
        except GpannError as e:
            lost, msg = True, str(e)
            break
    failed_collective = False
    try:
        comm.barrier()
    except GpannError:
        failed_collective = True
    comm.abort()
    np.savez(os.path.join(out_dir, "abort.npz"), lost=np.array([lost]), after=np.array([failed_collective]), seconds=np.array([time.time() - t0]),
             msg=np.array([msg]))


def test_a_rank_that_leaves_is_noticed(tmp_path):
    """gpann_comm_abort / gpann_comm_wait (include/gpann_comm.h): the reference relies on mpirun to tear a broken job down; here a rank that
    fails closes its bootstrap connections and the others' waits return an error instead of blocking for ever"""
    import torch.multiprocessing as mp
    mp.spawn(_abort_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = np.load(tmp_path / "abort.npz")
    assert bool(r["lost"][0]) and bool(r["after"][0]) and float(r["seconds"][0]) < 20.0, r
    assert "has left" in str(r["msg"][0])
