"""GPU: this build's DistributedBench (gp-ann_amd/drivers/distributed_bench.cpp = the reference's distributed_bench.cpp:13-62 over the
multi-GPU DistributedQueryBenchmark) end to end through the C-ABI exchange layer (include/gpann_comm.h):
  * one rank, RCCL backend (ncclBroadcast / ncclAllGather with world size 1, the owner-selection kernel);
  * two and three ranks sharing GPU 0 over the TCP backend (RCCL refuses two ranks on one device; the exchange LOGIC is the same) — every
    rank holds only its shards, trains only their subtrees, routes only its slice of the queries.
The merged neighbours of all queries must equal the CPU oracle's single-process route + scan + DedupNeighbors, bit for bit."""
import os
import socket
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_mixture

pytestmark = pytest.mark.gpu

DB = os.path.join(ROOT, "gp-ann_amd", "bin", "DistributedBench")


def _free_port():
    """a free MASTER_PORT whose successors — the exchange library's default GPANN_COMM_PORT and the second communicator's — are free as well"""
    from conftest import free_port
    return free_port(3)


@pytest.fixture(scope="module")
def data(orc, tmp_path_factory):
    tmp = tmp_path_factory.mktemp("dist")
    n, d, nq, S, k, nprobe = 20000, 32, 301, 6, 10, 2
    pts = make_mixture(orc, n, d, n_centers=32)
    queries = make_mixture(orc, nq, d, seed=556, n_centers=32)
    part, _ = orc.kmeans(pts, S, 555, 0, nthreads=4)
    clusters = orc.partition_to_clusters(part)
    f = dict(points=str(tmp / "base.fbin"), queries=str(tmp / "query.fbin"), gt=str(tmp / "gt.bin"), part=str(tmp / "base.partition"))
    for path, a in ((f["points"], pts), (f["queries"], queries)):
        with open(path, "wb") as fh:
            np.array(a.shape, np.uint32).tofile(fh)
            np.ascontiguousarray(a, np.float32).tofile(fh)
    with open(f["part"], "w") as fh:
        fh.write("".join(f"{int(p)}\n" for p in part))
    gd, gi = orc.ground_truth(pts, queries, k, 0, nthreads=4)
    with open(f["gt"], "wb") as fh:
        np.array([nq, k], np.uint32).tofile(fh)
        gi.astype(np.uint32).tofile(fh)
        gd.astype(np.float32).tofile(fh)
    # expected: Train{64, 350, n / S} -> Query(50000) -> first 2 probes -> QueryBucket -> DedupNeighbors
    tree = orc.kmtr_train(pts, clusters, 64, 350, n // S, 0, nthreads=4)
    probes, _ = tree.query_batch(queries, 50000, nthreads=4)
    ivf = orc.ivf_build(pts, clusters)
    wd, wi, _ = orc.ivf_query_bucket_batch(ivf, queries, k, probes, nprobe, 0, nthreads=4)
    offs = np.arange(nq + 1, dtype=np.uint64) * (nprobe * k)
    md, mi, _ = orc.dedup_neighbors(wd[:, :nprobe].reshape(-1), wi[:, :nprobe].reshape(-1), offs, k)
    recall = orc.recall(md.reshape(-1), np.arange(nq + 1, dtype=np.uint64) * k, gd[:, k - 1].copy(), k)
    return dict(tmp=tmp, files=f, nq=nq, k=k, want_d=md.reshape(nq, k), want_i=mi.reshape(nq, k), recall=recall)


def _gpu_count():
    """GPUs of this box without initialising HIP in the pytest process (device_count() does not, on this image)"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def _rank_env(data, rank, world, backend, port, out, one_gpu_per_rank=False, **extra):
    dev = str(rank) if one_gpu_per_rank else "0"
    return dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=dev, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPANN_COMM_BACKEND=backend,
                GPANN_DIST_DEVICE=dev, GPANN_DIST_OUT=out, GPANN_DIST_REPS="2", GPANN_ROUTER_VERIFY="1", GPANN_ROUTER_STRICT="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)


def _join_ranks(procs, tag, limit_s=240.0):
    """waits for the rank processes; a rank that ends with an error (or the time limit) ends the others too — they would otherwise sit in
    the rendezvous / a collective until THEIR limits — and the test fails with every rank's output"""
    import tempfile
    import threading
    import time
    outs = [None] * len(procs)

    def drain(i):
        outs[i] = procs[i].communicate()
    threads = [threading.Thread(target=drain, args=(i,), daemon=True) for i in range(len(procs))]
    for t in threads:
        t.start()
    t0 = time.time()
    failed = None
    while any(t.is_alive() for t in threads):
        time.sleep(0.05)
        bad = [i for i, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() - t0 > limit_s:
            failed = f"rank(s) {bad} ended with an error" if bad else f"no completion within {limit_s:.0f} s"
            time.sleep(2.0)   # (the others may be about to notice by themselves: their messages are worth having)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
    for t in threads:
        t.join(30)
    if failed is None and all(p.returncode == 0 for p in procs):
        return outs
    report = f"[{tag}] {failed or 'a rank failed'}; return codes {[p.returncode for p in procs]}\n"
    for i, o in enumerate(outs):
        so, se = o if o else ("", "")
        report += f"--- rank {i} stdout tail:\n{so[-1500:]}\n--- rank {i} stderr tail:\n{se[-3000:]}\n"
    try:   # kept for the record of rare events (tools/stress_suite_context.sh collects gpurun_out)
        with open(os.path.join(ROOT, "gpurun_out", f"distributed_failure_{tag}_{os.getpid()}.txt"), "w") as fh:
            fh.write(report)
    except OSError:
        pass
    pytest.fail(report)


def _run(data, world, backend, tag, router_name, one_gpu_per_rank=False, **extra):
    f = data["files"]
    out = str(data["tmp"] / f"out_{tag}.bin")
    router = str(data["tmp"] / router_name)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = _rank_env(data, rank, world, backend, port, out, one_gpu_per_rank, **extra)
        procs.append(subprocess.Popen([DB, f["points"], f["queries"], f["gt"], str(data["k"]), f["part"], router], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = _join_ranks(procs, tag)
    raw = np.fromfile(out, np.uint32)
    nq, k = int(raw[0]), int(raw[1])
    assert (nq, k) == (data["nq"], data["k"])
    ids = raw[2:2 + nq * k].reshape(nq, k)
    dist = raw[2 + nq * k:2 + 2 * nq * k].reshape(nq, k)
    assert np.array_equal(ids, data["want_i"]) and np.array_equal(dist, data["want_d"].view(np.uint32)), tag
    line = [ln for ln in outs[0][0].split("\n") if ln.startswith("backend ")][0]
    assert f"ranks {world}" in line and abs(float(line.split("recall ")[1].split(",")[0]) - data["recall"]) < 1e-6
    assert "End-to-end time" in outs[0][0]
    if "GPANN_DIST_PIPELINE" in extra:
        stream = [ln for ln in outs[0][0].split("\n") if ln.startswith("Pipelined stream: ")]
        assert stream and "identical to the single batch's 1" in stream[0], outs[0][0][-800:]
    return line


def test_one_rank_rccl(data):
    assert os.path.exists(DB), "gp-ann_amd/bin/DistributedBench missing: run __graft_entry__.build()"
    line = _run(data, 1, "rccl", "r1", "router_r1.tree")
    assert line.startswith("backend rccl ")
    # second run: the router is loaded from the tree file the first run wrote
    assert os.path.exists(str(data["tmp"] / "router_r1.tree"))
    _run(data, 1, "rccl", "r1b", "router_r1.tree")


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu_over_tcp(data, world):
    line = _run(data, world, "tcp", f"t{world}", f"router_t{world}.tree")
    assert line.startswith("backend tcp")
    # ... and the tree the ranks trained part by part is the tree one rank trains
    a = open(str(data["tmp"] / f"router_t{world}.tree"), "rb").read()
    single = str(data["tmp"] / "router_r1.tree")
    if os.path.exists(single):
        assert a == open(single, "rb").read()


@pytest.mark.parametrize("world,backend", [(1, "rccl"), (2, "tcp"), (3, "tcp")])
def test_pipelined_stream_of_batches(data, world, backend):
    """GPANN_DIST_PIPELINE: after the reference's one-batch protocol the driver runs the block as a stream of batches through the three-stage
    pipeline (route-ahead thread on a second communicator, deferred scans, exchange on a third stream): the last batch's neighbours must be
    the single batch's, bit for bit (the driver checks and says so), and what rank 0 writes out must be the oracle's"""
    line = _run(data, world, backend, f"p{world}{backend}", f"router_p{world}.tree", GPANN_DIST_PIPELINE="5")
    assert line.startswith(f"backend {backend}")


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the 1-GPU boxes of rounds 1-4 skip this)")
def test_two_rank_rccl(data):
    """The N > 1 RCCL path itself — ncclCommInitRank with world 2, ncclBroadcast of the query block, the probe all-gather, the grouped
    all-gather of the slot lists and the owner selection — one rank per GPU over xGMI: neighbours of all queries bit-identical to the CPU
    oracle's single-process result, the tree trained part by part identical to the single-rank tree."""
    line = _run(data, 2, "rccl", "rccl2", "router_rccl2.tree", one_gpu_per_rank=True)
    assert line.startswith("backend rccl ") and "ranks 2" in line
    single = str(data["tmp"] / "router_r1.tree")
    if os.path.exists(single):
        assert open(str(data["tmp"] / "router_rccl2.tree"), "rb").read() == open(single, "rb").read()


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs (see test_two_rank_rccl)")
def test_two_rank_rccl_a_rank_dies_mid_run(data):
    """Failure handling of the RCCL path (gpann_comm_wait's ncclCommGetAsyncError / lost-peer branch, ncclCommAbort): rank 1 is killed while
    the ranks loop over batches; rank 0 must come back from its collective through gpann_comm_wait within GPANN_COMM_WAIT_S instead of hanging
    in RCCL, and end with an error that names the reason."""
    import signal
    import time
    f = data["files"]
    out = str(data["tmp"] / "out_rccl2_dead.bin")
    router = str(data["tmp"] / "router_rccl2.tree")
    port = _free_port()
    procs = []
    for rank in range(2):
        env = _rank_env(data, rank, 2, "rccl", port, out, True, GPANN_COMM_WAIT_S="10")
        env["GPANN_DIST_REPS"] = "200000"   # (keeps the ranks in the batch loop until rank 1 is killed)
        procs.append(subprocess.Popen([DB, f["points"], f["queries"], f["gt"], str(data["k"]), f["part"], router], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    time.sleep(20.0)   # set-up (store build, training or tree load, RCCL init) is over, the batch loop is running
    assert procs[1].poll() is None and procs[0].poll() is None, "a rank ended before it could be killed: " + str([p.poll() for p in procs])
    procs[1].send_signal(signal.SIGKILL)
    t0 = time.time()
    try:
        so, se = procs[0].communicate(timeout=90)
    except subprocess.TimeoutExpired:
        procs[0].kill()
        pytest.fail("rank 0 hung after rank 1 was killed (gpann_comm_wait did not abandon the collective)")
    procs[1].communicate()
    assert procs[0].returncode != 0
    assert time.time() - t0 < 60.0
    assert "collective abandoned" in se or "has left" in se or "connection" in se, se[-2000:]


@pytest.mark.parametrize("world", [2, 8])
def test_owner_selection_kernels_on_a_fabricated_world(gp, world):
    """SelectSlotsKernel / MapIdsKernel (csrc/comm.hip) with world > 1 on device buffers — the part of the RCCL path a 1-GPU box can
    run: a fabricated all-gather result for `world` ranks incl. replicated shards (a replicated shard's tasks are split by query id, so
    a slot is still non-empty on exactly one rank), every rank reporting LOCAL row ids with its own id map."""
    from pygpann import comm as gcomm
    rng = np.random.default_rng(world)
    nq, nprobe, k, S = 700, 3, 10, 24
    shard_rows = rng.integers(1000, 50000, S).astype(np.uint64)
    work = shard_rows.astype(np.float64) * rng.integers(1, 200, S)
    rep_begin, rep_rank, _, _ = gcomm.plan_replicas(shard_rows, work, world, int(2.5 * shard_rows.sum() / world), 1.0)
    assert int(rep_begin[S]) > S or world == 2            # replicas exist (8 ranks always have room)
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    probes[::11, 1] = -1                                  # skipped slots: empty on every rank
    n = nq * nprobe * k
    # every rank: its local rows are a random subset of the global ids, in ascending order (bench.py: my_ids)
    n_global = 1_000_000
    id_maps = [np.sort(rng.choice(n_global, 30000, replace=False)).astype(np.uint32) for _ in range(world)]
    rank_dist = np.full((world, nq, nprobe, k), np.inf, np.float32)
    rank_ids = np.full((world, nq, nprobe, k), 0xFFFFFFFF, np.uint32)
    want_dist = np.full((nq, nprobe, k), np.inf, np.float32)
    want_ids = np.full((nq, nprobe, k), 0xFFFFFFFF, np.uint32)
    owned = np.zeros((nq, nprobe), bool)
    for r in range(world):
        mine = gcomm.mask_probes_replicated(probes, nprobe, rep_begin, rep_rank, r)
        for q, j in zip(*np.nonzero(mine >= 0)):
            cnt = int(rng.integers(0, k + 1))             # short lists (cut at the merge bound / small shards) are part of the contract
            local = rng.choice(len(id_maps[r]), cnt, replace=False).astype(np.uint32)
            d = np.sort(rng.random(cnt).astype(np.float32))
            assert not owned[q, j]                        # one owner per slot, replicas included
            owned[q, j] = True
            rank_dist[r, q, j, :cnt], rank_ids[r, q, j, :cnt] = d, local
            want_dist[q, j, :cnt], want_ids[q, j, :cnt] = d, id_maps[r][local]
    assert np.array_equal(owned, probes >= 0)             # every valid slot has its owner
    got_d, got_i = gcomm.debug_select_slots(rank_dist.reshape(world, n), rank_ids.reshape(world, n), id_maps)
    assert np.array_equal(got_i, want_ids.reshape(-1))
    assert np.array_equal(got_d.view(np.uint32), want_dist.reshape(-1).view(np.uint32))
    # without id maps: ids pass through
    got_d2, got_i2 = gcomm.debug_select_slots(rank_dist.reshape(world, n), rank_ids.reshape(world, n))
    sel = rank_ids.reshape(world, n)
    assert np.array_equal(got_i2, np.where((sel != 0xFFFFFFFF).any(axis=0), sel.min(axis=0), 0xFFFFFFFF).astype(np.uint32))


def test_one_rank_rccl_wait_and_abort(gp):
    """gpann_comm_wait / gpann_comm_abort on the RCCL backend (one rank, in-process): a wait on enqueued collectives returns once they
    have completed and leaves their results in place; a wait with nothing pending is a success; abort tears the communicator down without
    a farewell collective (ncclCommAbort) and the handle is gone afterwards."""
    import torch
    from pygpann import comm as gcomm
    port = _free_port()
    comm = gcomm.Comm(gcomm.RCCL, rank=0, world=1, addr="127.0.0.1", port=port + 1, device=0)
    assert comm.name.startswith("rccl") and comm.world == 1
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    send = torch.arange(1 << 20, dtype=torch.int32, device=dev)
    recv = torch.zeros_like(send)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(8):
            comm.allgather(send.data_ptr(), recv.data_ptr(), send.numel() * 4, stream.cuda_stream)
    comm.wait(stream.cuda_stream, 30)          # pending -> completed
    assert stream.query()
    assert torch.equal(send, recv)
    comm.wait(stream.cuda_stream, 1)           # nothing pending: at once
    comm.check_peers()
    comm.abort()
    assert not comm.h.value
    comm.abort()                               # idempotent on a released handle
