"""CPU, world size 2 and 3 over the TCP backend of libgpann_comm.so (include/gpann_comm.h): the exchange logic of the multi-GPU path —
rendezvous, barrier, max over ranks, broadcast, all-gather, the owner selection of the per-slot result lists — and the placement
helpers against the Python restatements in tests/multirank_ref.py.  The CPU oracle stands in for the GPU scan; the outcome must equal the
single-process result.  (The RCCL backend runs the same calls on device buffers; it needs one GPU per rank.)"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, random_clusters


def _free_port():
    from conftest import free_port   # (below the ephemeral range: see there)
    return free_port(2)


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "gp-ann_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import pyoracle
    from pygpann import comm as gc
    import multirank_ref as multigpu
    c = gc.Comm(gc.TCP, rank=rank, world=world, addr="127.0.0.1", port=port, timeout_s=60)
    assert (c.rank, c.world, c.name) == (rank, world, "tcp")
    c.barrier()
    assert c.max_double(float(rank + 1)) == float(world)
    g = c.allgather_host(np.array([rank, 10 * rank], np.int32))
    assert g.tolist() == [[r, 10 * r] for r in range(world)]
    b = np.arange(7, dtype=np.float32) * (rank + 1)
    c.broadcast_host(b, root=world - 1)                       # a root other than rank 0 goes through rank 0
    assert np.array_equal(b, np.arange(7, dtype=np.float32) * world)

    orc = pyoracle.Oracle("strict")
    rng = np.random.default_rng(21)
    n, d, S, nq, k, nprobe = 4000, 16, 7, 60, 5, 3
    pts = orc.synth_rows(555, 7, 16, 0.35, 0, n, d)
    # the query block exists on rank 0 only and is broadcast (distributed_bench.cpp:39 has every rank read the file)
    queries = orc.synth_rows(556, 7, 16, 0.35, 0, nq, d) if rank == 0 else np.zeros((nq, d), np.float32)
    c.broadcast(queries, queries.nbytes, 0)
    assert np.array_equal(queries, orc.synth_rows(556, 7, 16, 0.35, 0, nq, d))
    clusters = random_clusters(rng, n, S, overlap=0.2)
    probes = np.stack([rng.permutation(S) for _ in range(nq)]).astype(np.int32)
    probes[5, 1] = -1                                          # a skipped slot
    sizes = [len(x) for x in clusters]
    owner, load = gc.lpt_owner(sizes, world)
    o2, l2 = multigpu.lpt_owner(sizes, world)
    assert np.array_equal(owner, o2) and np.array_equal(load, l2) and load.sum() == sum(sizes)
    lp = gc.mask_probes(probes, nprobe, owner, rank)
    assert np.array_equal(lp, multigpu.mask_probes(probes[:, :nprobe], owner, rank))
    # every rank routes a slice of the queries; the probe lists are all-gathered (here: the slices of `probes`)
    chunk = (nq + world - 1) // world
    mine = np.full((chunk, S), -1, np.int32)
    lo, hi = min(nq, rank * chunk), min(nq, (rank + 1) * chunk)
    mine[:hi - lo] = probes[lo:hi]
    allp = np.empty((world, chunk, S), np.int32)
    c.allgather(mine, allp, mine.nbytes)
    assert np.array_equal(allp.reshape(-1, S)[:nq], probes)
    # local scan of the owned shards, then the result exchange
    store = [x if owner[b] == rank else np.zeros(0, np.uint32) for b, x in enumerate(clusters)]
    ivf = orc.ivf_build(pts, store)
    od, oi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, lp, nprobe, 0)
    nslots = od.size
    out_d, out_i = np.empty_like(od), np.empty_like(oi)
    scratch = np.empty(c.gather_slots_scratch(nslots), np.uint8)
    c.gather_slots(od, oi, nslots, out_d, out_i, scratch)
    if rank == 0:
        full = orc.ivf_build(pts, clusters)
        wd, wi, _ = orc.ivf_query_bucket_batch(full, queries, k, probes, nprobe, 0)
        ok = np.array_equal(out_i, wi) and np.array_equal(out_d.view(np.uint32), wd.view(np.uint32))
        np.savez(os.path.join(out_dir, "result.npz"), ok=np.array([ok]))
    c.barrier()
    c.close()


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_backend_exchange(tmp_path, world):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert bool(np.load(tmp_path / "result.npz")["ok"][0])


def test_comm_library_exports_and_single_rank():
    sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))
    from pygpann import comm as gc
    L = gc.lib()
    assert not [s for s in gc.EXPORTS if not hasattr(L, s)]
    c = gc.Comm(gc.TCP, rank=0, world=1)
    c.barrier()
    assert c.max_double(3.5) == 3.5
    d = np.array([1.0, np.inf], np.float32)
    i = np.array([7, 0xFFFFFFFF], np.uint32)
    od, oi = np.empty_like(d), np.empty_like(i)
    c.gather_slots(d, i, 2, od, oi, np.empty(c.gather_slots_scratch(2), np.uint8))
    assert od.tolist() == [1.0, np.inf] and oi.tolist() == [7, 0xFFFFFFFF]
    # without a GPU the RCCL backend must refuse loudly, not fall back
    import pygpann
    if pygpann.device_count() == 0:
        with pytest.raises(pygpann.GpannError):
            gc.Comm(gc.RCCL, rank=0, world=1)
    owner, load = gc.lpt_owner([10, 9, 8, 1, 1, 1], 3)
    assert sorted(load.tolist()) == [10, 10, 10]
