#!/usr/bin/env python3
"""bench.py — QPS at recall@10 >= 0.9 of the route + shard-scan hot path on synthetic 100M x 128 float vectors.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: one rank per GPU — started by torch.distributed.run, or, without a
                                                                launcher, by bench.py itself: spawn_ranks)

One "step" = one pass of the hot path over one batch of 10 000 queries that are already resident in HBM:
    KMeansTreeRouter::Query (GPU walk + host std::sort)  ->  per-(query, probe) exact shard scan + top-k (GPU)
    ->  [N > 1: all-gather of the per-slot partial results over RCCL]  ->  host merge (DedupNeighbors).
The workload is BASELINE.json's north-star configuration (SURVEY.md §8 "NS"): 100M x 128 f32 Gaussian-mixture points
(1024 centres, sigma 0.35, seed 555), 10k queries (seed 556), 40 k-means shards, k = 10, nprobe = the smallest value whose
recall@10 reaches 0.9 (found by a sweep before the timed region; the ground truth is computed with the exact scan mode).
With N GPUs the 40 shards are spread over the ranks (LPT by size); total work is fixed, so scaling is "strong".

Prints ONE JSON line (rank 0).  Extra keys: `roofline` (the scan kernel: algorithmic bytes / HIP-event kernel time against
the 8 TB/s HBM peak), `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample of the same
workload) and `detail`.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
# The timed loop keeps three scan calls, the router and the lists' way out in flight on streams of their own; the HIP runtime multiplexes a
# process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and a stream that shares a queue with the rest pass waits for it.
# Read when the runtime is initialised: set before anything touches HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# N > 1: how long a rank waits for a collective before it gives the run up (gpann_comm_wait also returns as soon as another rank has left)
COMM_WAIT_S = int(os.environ.get("GPANN_COMM_WAIT_S", "600"))
_live_comm = []   # the communicator of this process, for the exit path of a failed run
_live_turns = []  # the timed loop's collective schedule (Turnstile), for the same exit path


def best_single_step_seconds(step_fn, barrier_fn, n=5, clock=time.perf_counter):
    """The shortest of n single steps, each closed by barrier_fn: what the choice among router configurations goes by.  (A mean lets one host hiccup
    decide: a 19 ms outlier in a 4.6 ms step once handed the timed loop to a configuration 2.6x slower.)"""
    best = float("inf")
    for _ in range(n):
        t0 = clock()
        step_fn()
        barrier_fn()
        best = min(best, clock() - t0)
    return best


class Turnstile:
    """ONE launch order, the same on every rank, for the collectives of the timed loop's TWO communicators (the route-ahead thread's probe
    all-gathers, this thread's slot exchanges).  Each communicator alone is driven by one thread in one order — but two communicators on one GPU
    whose collectives are launched in a different relative order on different ranks can wait for each other for ever when the GPU cannot
    co-schedule both kernels (RCCL / NCCL: concurrent communicators need a consistent launch order).  `order` lists the collectives of one
    run_steps call; slot(key) blocks until every earlier entry has been LAUNCHED (not completed), so the schedule costs no overlap."""

    def __init__(self, order):
        import threading
        self.pos = {key: i for i, key in enumerate(order)}
        self.turn = 0
        self.dead = False
        self.cv = threading.Condition()

    @contextlib.contextmanager
    def slot(self, key):
        with self.cv:
            self.cv.wait_for(lambda: self.dead or self.turn == self.pos[key])
            if self.dead:
                raise RuntimeError("the timed loop was given up (another thread of this rank failed)")
        try:
            yield
        finally:
            with self.cv:
                self.turn += 1
                self.cv.notify_all()

    def abort(self):
        with self.cv:
            self.dead = True
            self.cv.notify_all()
for p in (ROOT, os.path.join(ROOT, "gp-ann_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_I8_PEAK_TOPS = 5000.0       # dense int8 MFMA peak: 32x32x32 in the cycles of bf16 32x32x16, twice the K (MI355X_MICROARCH.md, matrix cores table)
VALU_F32_PEAK_TLANEOPS = 78.65   # 157.3 TFLOP/s fp32 vector counts an FMA as 2: unfused lane-ops (sub, mul, add) issue at half of it
POINT_SEED, QUERY_SEED, CENTER_SEED = 555, 556, 7
N_CENTERS, SIGMA = 1024, 0.35


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


@contextlib.contextmanager
def stdout_to_stderr():
    """the host mirror prints progress to std::cout like the reference does; keep bench's stdout to the one JSON line"""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        os.dup2(saved, 1)
        os.close(saved)


def lloyd_gpu(torch, pygpann, sample, k, rounds, seed, stream):
    """Flat k-means on the device (setup only): nearest-centroid kernel + torch scatter-add updates."""
    n, d = sample.shape
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    idx = torch.randperm(n, generator=g)[:k].to(sample.device)
    cent = sample[idx].clone().contiguous()
    assign = torch.empty(n, dtype=torch.int32, device=sample.device)
    for _ in range(rounds):
        pygpann.nearest_centroid_device(sample.data_ptr(), n, d, 0, cent.data_ptr(), cent.shape[0], assign.data_ptr(), stream)
        a64 = assign.to(torch.int64)
        # float64 accumulation: the scatter-add's atomic order then changes the float32 centroids (hence the shards and every
        # number of the run) only with negligible probability — runs are repeatable
        sums = torch.zeros(cent.shape, dtype=torch.float64, device=cent.device).index_add_(0, a64, sample.double()).to(sample.dtype)
        cnt = torch.bincount(a64, minlength=cent.shape[0]).clamp_(min=1).to(sample.dtype)
        keep = torch.bincount(a64, minlength=cent.shape[0]) > 0
        cent = torch.where(keep[:, None], sums / cnt[:, None], cent).contiguous()
    pygpann.nearest_centroid_device(sample.data_ptr(), n, d, 0, cent.data_ptr(), cent.shape[0], assign.data_ptr(), stream)
    return cent, assign


def _free_ports(span):
    """`span` consecutive free TCP ports on 127.0.0.1 below the kernel's ephemeral range (a port from bind(0) can be handed to a connecting
    rank as its SOURCE port: the socket then connects to itself — tests/conftest.py free_port has the story)"""
    import random
    import socket
    lo_eph = 32768
    try:
        lo_eph = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError):
        pass
    hi = max(12000, min(lo_eph, 32768) - span - 1)
    rng = random.Random(os.getpid() * 1000003 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(400):
        p0 = rng.randrange(11000, hi)
        socks = []
        try:
            for i in range(span):
                sk = socket.socket()
                socks.append(sk)
                sk.bind(("127.0.0.1", p0 + i))
            return p0
        except OSError:
            continue
        finally:
            for sk in socks:
                sk.close()
    raise SystemExit("bench.py: no free TCP ports for the ranks' bootstrap")


def spawn_ranks(n_gpus):
    """`python3 bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the launcher — the shape of
    the reference's distributed_bench.cpp:26-54, where every rank runs the same main() and rank 0 prints.  It starts N copies of itself as CHILD
    processes, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays rank 0's one JSON line and
    returns the worst exit code.  Nothing here touches the GPU (no HIP call, no torch.cuda.is_available(): torch.cuda.device_count() does not
    initialise the runtime on this image) and nothing is exec'ed: a process that has initialised the GPU must not be replaced by another."""
    import signal
    import subprocess
    if "GPANN_BENCH_FORCE_DEVICE" not in os.environ:   # (debugging aid: every rank on the one named GPU, --backend tcp)
        import torch
        have = torch.cuda.device_count()
        if have < n_gpus:
            print(f"bench.py: --gpus {n_gpus} but this node has {have} GPU(s) visible; nothing was started "
                  f"(GPANN_BENCH_FORCE_DEVICE=<id> with --backend tcp runs every rank on one GPU)", file=sys.stderr, flush=True)
            return 2
    port = _free_ports(4)   # MASTER_PORT, the exchange's bootstrap (+1) and the routing communicator's (+2), one spare
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(n_gpus), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "LOCAL_WORLD_SIZE": str(n_gpus),
                 "GPANN_BENCH_SPAWNED": "1"})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env0.pop("GPANN_COMM_PORT", None)
    procs = []
    for r in range(n_gpus):
        env = dict(env0)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        # rank 0's stdout is this process's stdout (its one JSON line); the other ranks print nothing there, and if they do it goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else sys.stderr,
                                      start_new_session=True))
    log(f"launcher: started {n_gpus} ranks (pids {[p.pid for p in procs]}), MASTER_PORT {port}")
    worst, failed_first = 0, None
    try:
        live = set(range(n_gpus))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and failed_first is None:
                    failed_first = (r, rc)
            if failed_first is not None and live:
                # a rank has failed: the others come back through gpann_comm_wait on their own; give them a moment, then end exactly the processes started here
                deadline = time.time() + float(os.environ.get("GPANN_BENCH_SPAWN_GRACE_S", "30"))
                while live and time.time() < deadline:
                    for r in sorted(live):
                        if procs[r].poll() is not None:
                            live.discard(r)
                    time.sleep(0.2)
                for r in sorted(live):
                    try:
                        os.killpg(procs[r].pid, signal.SIGTERM)
                    except OSError:
                        pass
                for r in sorted(live):
                    try:
                        procs[r].wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        try:
                            os.killpg(procs[r].pid, signal.SIGKILL)
                        except OSError:
                            pass
                live.clear()
            time.sleep(0.1)
    except KeyboardInterrupt:
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGTERM)
                except OSError:
                    pass
        worst = 130
    if failed_first is not None:
        print(f"bench.py: rank {failed_first[0]} ended with code {failed_first[1]}; the run has no result line", file=sys.stderr, flush=True)
        worst = failed_first[1] if failed_first[1] > 0 else 1
    return worst


def main():
    # `python3 bench.py --gpus N` without a launcher: start the N ranks as children (before anything touches the GPU) and relay rank 0's line
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        known, _ = pre.parse_known_args()
        if known.gpus > 1:
            sys.exit(spawn_ranks(known.gpus))
    # stdout carries exactly ONE line, the result: everything any library prints there meanwhile (gloo's connection notes, the
    # host trainer's progress lines, ...) goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", dest="n", type=int, default=int(os.environ.get("GPANN_BENCH_N", 100_000_000)))
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--shards", type=int, default=40)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--target-recall", type=float, default=0.9)
    ap.add_argument("--config", choices=["ns", "c2", "c3", "c4", "c5"], default="ns",
                    help="BASELINE.json configuration shapes: ns = the metric's own (100M x 128 L2, 40 shards; the default), c2 = 1M x 128, 8 shards, "
                         "c3 = 10M x 96, 16 shards, nprobe 4, c4 = 100M x 200 inner product, 40 shards, c5 = 100-dim, 60 OVERLAPPING shards (20 %% of the "
                         "points in a second shard), nprobe swept, 200M points = the largest size one GPU holds "
                         "(sets --points/--d/--shards/--metric/--nprobe/--overlap; GPANN_BENCH_N still overrides the size)")
    ap.add_argument("--overlap", type=float, default=0.0,
                    help="SPANN-style overlapping shards (reference src/overlapping_partitioning.cpp:143-235): this fraction of the points — those "
                         "closest to a representative of another shard — is ALSO stored in that shard; the merge drops the copies "
                         "(small_scale_queries.cpp:15-23)")
    ap.add_argument("--metric", choices=["l2", "ip"], default="l2", help="ip = the reference's MIPS_DISTANCE build: 1 - <p, q>")
    ap.add_argument("--nprobe", type=int, default=0, help="probe exactly this many shards (0: the smallest count that reaches --target-recall)")
    ap.add_argument("--dist", choices=["mixture", "uniform", "offset"], default="mixture",
                    help="synthetic cloud: mixture = 1024 Gaussian clusters (the metric's workload); uniform = ONE Gaussian blob (no cluster structure: "
                         "cells cannot prune); offset = the mixture moved by +100 in every coordinate (non-centred, SIFT-like: the bf16 bound's error "
                         "term grows with (R + |q|)^2)")
    ap.add_argument("--sigma", type=float, default=SIGMA,
                    help="standard deviation of the mixture's components (0.35: the metric's workload; larger = the clusters overlap, routing gets "
                         "harder and recall 0.9 needs more probes)")
    ap.add_argument("--search-budget", type=int, default=50000)
    ap.add_argument("--router-centroids", type=int, default=64)
    ap.add_argument("--router-min-cluster", type=int, default=350)
    ap.add_argument("--router-budget", type=str, default="20000,100000",
                    help="index sizes (centroids) of the k-means tree to try, the reference's grid (routes.cpp:130-157); the configuration "
                         "with the best QPS at the target recall is the one timed, as in the reference's Pareto evaluation")
    ap.add_argument("--scan-mode", choices=["exact", "filter"], default=os.environ.get("GPANN_BENCH_SCAN_MODE", "filter"),
                    help="exact: VALU scan of every row; filter: same exact results with the MFMA pre-filter (scan_filter.h)")
    ap.add_argument("--backend", default=os.environ.get("GPANN_BENCH_BACKEND", "rccl"), choices=["rccl", "tcp"],
                    help="exchange backend of libgpann_comm.so for N > 1: rccl (RCCL over xGMI, device buffers) or tcp (debugging: the same "
                         "payloads through host sockets, e.g. several ranks on one GPU)")
    ap.add_argument("--replica-capacity", type=float, default=float(os.environ.get("GPANN_BENCH_REPLICA_CAPACITY", 2.0)),
                    help="N > 1: rows a rank may hold, in units of its fair share (rows / N); the room beyond 1.0 is filled with replicas of the "
                         "shards that attract the most query work (0: no replicas, LPT by rows)")
    ap.add_argument("--no-merge-bound", action="store_true",
                    help="timed steps: complete per-slot lists instead of GPANN_SCAN_FOR_MERGE (lists cut at the query's merge bound — the merged "
                         "neighbours are bit-identical either way; only matters for nprobe > 1)")
    ap.add_argument("--emulate-world", type=str, default="",
                    help="one GPU only, e.g. 2,4,8: after the timed run, for every listed world size W and every rank r < W build ONLY rank r's "
                         "shards of the W-way placement (LPT + replica plan, exactly what --gpus W would place) from the same points, run its "
                         "steps alone on this GPU — routing of its query slice, scan of the slots it owns for all queries, D2H, host merge — and "
                         "report per-rank ms/step and the implied W-GPU QPS = nq / (max over ranks + exchange); the exchange is timed on this GPU "
                         "through the real RCCL calls with one rank, its wire time is an estimate.  A hardware-free strong-scaling prediction, "
                         "not a measurement of W GPUs (`scale_emulated` in the JSON line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hard-workload", action="store_true",
                    help="skip detail.hard_workload: after the default run (NS, one rank) the same benchmark is run once more, for a few steps and without its "
                         "CPU leg, on a cloud whose clusters OVERLAP (component sigma 0.8 instead of 0.35: recall 0.9 then needs more than one probe and the "
                         "filter can prune far less) — the driver-timed record shows how the number depends on the data")
    ap.add_argument("--no-exact-compare", action="store_true", help="skip the extra steps in exact (VALU) scan mode")
    args = ap.parse_args()
    preset = {"ns": {}, "c2": dict(n=1_000_000, d=128, shards=8), "c3": dict(n=10_000_000, d=96, shards=16, nprobe=4),
              "c4": dict(n=100_000_000, d=200, shards=40, metric="ip"),
              "c5": dict(n=200_000_000, d=100, shards=60, overlap=0.2)}[args.config]
    for key, val in preset.items():
        setattr(args, key, val)
    if args.config != "ns" and "GPANN_BENCH_N" in os.environ:
        args.n = int(os.environ["GPANN_BENCH_N"])
    metric = 1 if args.metric == "ip" else 0
    n_centers, sigma = (1, 1.0) if args.dist == "uniform" else (N_CENTERS, args.sigma)
    offset = 100.0 if args.dist == "offset" else 0.0
    args.metric_id = metric
    ops_per_dim = 2 if metric else 3   # unfused VALU lane-ops per (pair, dimension): mul + add (inner product) / sub + mul + add (L2)

    import torch
    import pygpann
    from pygpann import comm as gcomm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # multi: the exchange layer is in use.  GPANN_BENCH_FORCE_COMM=1 (debugging aid) runs the N > 1 code path — communicator, RCCL
    # collectives, owner selection, id maps — with a single rank, which is as far as a 1-GPU box can exercise it on device buffers.
    multi = world > 1 or os.environ.get("GPANN_BENCH_FORCE_COMM") == "1"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # GPANN_BENCH_FORCE_DEVICE: debugging aid — run every rank on one GPU (only sensible with --backend gloo and a small --n)
    if "GPANN_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["GPANN_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        # the exchange layer is the C-ABI library libgpann_comm.so (include/gpann_comm.h): RCCL collectives on this process's HIP stream.
        # RCCL over xGMI is the measured configuration: if it cannot be initialised the run FAILS — a number taken over host sockets must
        # be asked for explicitly (--backend tcp) and says so in the JSON line ("backend")
        comm = gcomm.Comm(gcomm.RCCL if args.backend == "rccl" else gcomm.TCP, device=local_rank)   # RANK / WORLD_SIZE / MASTER_* from the environment; the device the buffers are on
        _live_comm.append(comm)
        if comm.world != args.gpus or comm.rank != rank:
            raise SystemExit(f"communicator has {comm.world} ranks (this is rank {comm.rank}), --gpus {args.gpus}")
        # a second communicator for the ROUTING exchange of the timed loop (round 4): batch i + 1 is routed, its probe lists all-gathered and
        # waited for on a second stream by a second thread while batch i is scanned — collectives of one communicator must be issued in one
        # order by one thread at a time, so the route-ahead thread gets its own (bootstrap port: the first communicator's + 1)
        comm_r = None
        if os.environ.get("GPANN_BENCH_ROUTE_AHEAD", "1") != "0":
            base_port = int(os.environ.get("GPANN_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 1))
            comm_r = gcomm.Comm(gcomm.RCCL if args.backend == "rccl" else gcomm.TCP, rank=rank, world=world, addr=os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                port=base_port + 1, device=local_rank, timeout_s=int(os.environ.get("GPANN_COMM_TIMEOUT", "600")))
            _live_comm.append(comm_r)
    else:
        comm = comm_r = None
    on_dev = args.backend == "rccl"
    stream = torch.cuda.current_stream().cuda_stream
    n, d, nq, S, k = args.n, args.d, args.nq, args.shards, args.k

    t_setup = time.time()
    # ---------------------------------------------------------------- synthetic inputs, generated straight into HBM
    queries = torch.empty((nq, d), dtype=torch.float32, device=dev)
    def synth(t, seed, row0, rows):
        pygpann.synth_rows_device(t.data_ptr(), seed, CENTER_SEED, n_centers, sigma, row0, rows, d, stream)
        if offset:
            t[:rows] += offset
    synth(queries, QUERY_SEED, 0, nq)
    n_sample = min(n, 2_000_000)
    sample = torch.empty((n_sample, d), dtype=torch.float32, device=dev)
    synth(sample, POINT_SEED, 0, n_sample)
    shard_cent, sample_assign = lloyd_gpu(torch, pygpann, sample, S, 20, 555, stream)
    if multi:
        # the device k-means sums with float atomics, so its centroids differ in the last bits from run to run: every rank
        # must cut the cloud with rank 0's centroids or rows would be owned by two ranks / by none
        if on_dev:
            comm.broadcast(shard_cent.data_ptr(), shard_cent.numel() * 4, 0, stream)
        else:
            c = shard_cent.cpu().numpy()
            comm.broadcast(c, c.nbytes, 0)
            shard_cent.copy_(torch.from_numpy(c))
        pygpann.nearest_centroid_device(sample.data_ptr(), n_sample, d, 0, shard_cent.data_ptr(), S, sample_assign.data_ptr(), stream)
    del sample, sample_assign
    log(f"shard centroids ready in {time.time() - t_setup:.1f}s")

    # shard assignment of all rows, chunked; every rank keeps only the rows of the shards it owns
    chunk = 4_000_000
    counts = torch.zeros(S, dtype=torch.int64, device=dev)
    assign_all = torch.empty(n, dtype=torch.int32, device=dev)
    # --overlap: the reference's MakeOverlappingWithCentroids rates every (point, foreign shard) pair by the distance to that shard's
    # nearest representative and realises the `overlap * n` best ratings.  Here, at 10^8 points: one representative per shard (its
    # centroid), the best foreign shard per point, and the rating threshold = the `overlap` quantile of the first chunk's ratings
    # (points are i.i.d. over the chunks).  second_all[i] = that shard, or -1.
    second_all = torch.full((n,), -1, dtype=torch.int32, device=dev) if args.overlap > 0 else None
    overlap_thr = None
    cent_sq = (shard_cent.double() ** 2).sum(1).float()
    buf = torch.empty((min(chunk, n), d), dtype=torch.float32, device=dev)
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        synth(buf, POINT_SEED, r0, m)
        pygpann.nearest_centroid_device(buf.data_ptr(), m, d, 0, shard_cent.data_ptr(), S, assign_all[r0:r0 + m].data_ptr(), stream)
        if second_all is not None:
            x = buf[:m]
            d2 = (x * x).sum(1, keepdim=True) + cent_sq[None, :] - 2.0 * (x @ shard_cent.T)
            d2.scatter_(1, assign_all[r0:r0 + m].to(torch.int64)[:, None], float("inf"))   # not the shard the point already lives in
            rating, second = d2.min(dim=1)
            if overlap_thr is None:
                overlap_thr = torch.kthvalue(rating, max(1, int(args.overlap * m))).values
            second_all[r0:r0 + m] = torch.where(rating <= overlap_thr, second.to(torch.int32), torch.full_like(second, -1, dtype=torch.int32))
            del d2, rating, second
    counts_t = torch.bincount(assign_all.to(torch.int64), minlength=S)
    if second_all is not None:
        counts_t += torch.bincount(second_all[second_all >= 0].to(torch.int64), minlength=S)
    counts = counts_t.cpu().numpy()
    store_rows = int(counts.sum())   # rows of the shard store (> n with overlapping shards)
    # Placement.  Rows alone (LPT) balance memory, not work: a shard's work per batch is (queries routed to it) x rows, and the popular
    # shards are also the big ones.  The query load is estimated before anything is built — queries per shard = nearest shard centroid —
    # and the hottest shards get replicas while a rank has room (gpann_plan_replicas; the reference models exactly this when it
    # predicts QPS, route_search_combination.cpp:149-183).  A replicated shard's tasks are split over its replicas by query id.
    placement = None
    if multi and args.replica_capacity > 0:
        qa = torch.empty(nq, dtype=torch.int32, device=dev)
        pygpann.nearest_centroid_device(queries.data_ptr(), nq, d, 0, shard_cent.data_ptr(), S, qa.data_ptr(), stream)
        est_tasks = np.bincount(qa.cpu().numpy().astype(np.int64), minlength=S).astype(np.float64) + 0.5
        cap_rows = int(max(args.replica_capacity * counts.sum() / world, 1.25 * counts.max()))
        # (GPANN_BENCH_REPLICA_IMBALANCE: the heaviest rank's estimated work over the mean at which no more replicas are added; 1.0 fills the room)
        rep_begin, rep_rank, rank_work, load = gcomm.plan_replicas(counts, est_tasks * counts, world, cap_rows,
                                                                   float(os.environ.get("GPANN_BENCH_REPLICA_IMBALANCE", 1.05)))
        owner = rep_rank[rep_begin[:-1]].astype(np.int32)          # the primary of every shard (trains its subtree)
        mine = np.array([b for b in range(S) if rank in rep_rank[rep_begin[b]:rep_begin[b + 1]]], np.int64)
        placement = {"replicas": int(rep_begin[S]) - S, "capacity_rows": cap_rows, "rank_rows": [int(x) for x in load],
                     "estimated_work_max_over_mean": float(rank_work.max() / rank_work.mean())}
    else:
        owner, load = gcomm.lpt_owner(counts, world)
        mine = np.flatnonzero(owner == rank)
    log(f"shard sizes min/max {counts.min()}/{counts.max()}, rank rows {[int(x) for x in load]}" + (f", placement {placement}" if placement else ""))
    # local point set = rows of my shards, in global-id order; Clusters hold GLOBAL ids, local_rows maps them to local rows
    mine_mask = torch.zeros(S, dtype=torch.bool, device=dev)
    mine_mask[torch.as_tensor(mine, device=dev)] = True
    keep = mine_mask[assign_all.to(torch.int64)]
    if second_all is not None:
        keep |= (second_all >= 0) & mine_mask[second_all.clamp(min=0).to(torch.int64)]
    my_ids = torch.nonzero(keep).squeeze(1)                      # global ids I own, ascending
    n_local = int(my_ids.numel())
    local_points = torch.empty((n_local, d), dtype=torch.float32, device=dev)
    pos = 0
    for r0 in range(0, n, chunk):
        m = min(chunk, n - r0)
        kc = keep[r0:r0 + m]
        c = int(kc.sum())
        if c == 0:
            continue
        synth(buf, POINT_SEED, r0, m)
        local_points[pos:pos + c] = buf[:m][kc]
        pos += c
    del buf, keep
    # (local row, shard) pairs: every point with its own shard, the overlapping ones once more with their second shard
    my_assign = assign_all[my_ids].to(torch.int64)
    del assign_all
    pair_row = torch.arange(n_local, dtype=torch.int64, device=dev)
    pair_shard = my_assign
    if second_all is not None:
        my_second = second_all[my_ids].to(torch.int64)
        del second_all
        prim_ok = mine_mask[my_assign]
        sec_ok = (my_second >= 0) & mine_mask[my_second.clamp(min=0)]
        pair_row = torch.cat([pair_row[prim_ok], pair_row[sec_ok]])
        pair_shard = torch.cat([my_assign[prim_ok], my_second[sec_ok]])
        del my_second, prim_ok, sec_ok
    order = torch.sort(pair_shard, stable=True).indices       # grouped by shard: a shard's own points (ascending id), then its extra ones
    sorted_assign = pair_shard[order].cpu().numpy()
    local_rows_by_shard = pair_row[order].to(torch.int32).cpu().numpy().astype(np.uint32)
    del order, my_assign, pair_row, pair_shard
    # Clusters over the LOCAL point array (all S shards; shards of other ranks are empty here)
    bounds = np.searchsorted(sorted_assign, np.arange(S + 1))
    clusters_local = [local_rows_by_shard[bounds[b]:bounds[b + 1]] for b in range(S)]
    torch.cuda.synchronize()
    t_index = time.time()
    # two steps: the rows (InvertedIndex gather) while the generated point array still exists, the filter's planes / cells after it is gone
    # (100M x 200: 80 GB + 80 GB + 102 GB would not fit)
    shards = pygpann.Shards(local_points.data_ptr(), clusters_local, metric, local_rank, n=n_local, d=d, plain=True)
    emulate_worlds = [int(x) for x in args.emulate_world.split(",") if x] if world == 1 else []
    points_kept = local_points if emulate_worlds else None   # (--emulate-world builds the emulated ranks' stores from the same point array)
    del local_points
    torch.cuda.empty_cache()
    if args.scan_mode == "filter":
        shards.build_filter()
    index_build_s = time.time() - t_index   # InvertedIndex gather + (filter mode support data) bf16 planes, cells, radii
    shards.set_scan_mode(1 if args.scan_mode == "filter" else 0)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_resident_gb = (total_b - free_b) / 1e9   # everything this process holds in HBM once the store is built (rows, planes, queries, workspaces)
    def train_router(index_budget):
        """KMeansTreeRouter::Train (reference src/kmeans_tree_router.cpp:9-78; options as routes.cpp:130-157) on the GPU over ALL rows
        of the shards this rank holds; with several ranks every rank trains the subtrees of its own shards and the parts are merged"""
        t_train = time.time()
        rt = pygpann.Router.train(shards, n, args.router_centroids, args.router_min_cluster, index_budget)
        if multi:
            part = rt
            rt = comm.merge_router(part, owner, local_rank)   # every rank's subtrees -> the router over all shards, replicated
            part.close()
        train_s = time.time() - t_train
        tr = rt.export()
        log(f"router (index budget {index_budget}) trained on the GPU in {train_s:.2f}s: {len(tr['node_shard'])} nodes, {tr['centroids'].shape[0]} centroids")
        return rt, tr, train_s

    log(f"rank {rank}: {n_local} points = {len(local_rows_by_shard)} store rows in {len(mine)} shards resident; setup {time.time() - t_setup:.1f}s")

    def to_global(ids_local):
        """scan results carry ids of the local point array; map them to the global ids of the synthetic cloud"""
        out = np.full(ids_local.shape, 0xFFFFFFFF, np.uint32)
        v = ids_local != 0xFFFFFFFF
        # permutation: local row id -> global id (my_ids is ascending, local row i = i-th owned global id)
        out[v] = my_ids_host[ids_local[v]]
        return out

    my_ids_host = my_ids.cpu().numpy().astype(np.uint32)
    my_ids_dev = my_ids.to(torch.int32)   # local row -> global point id (the id map of gpann_comm_gather_slots; ids < 2^31 here)
    del my_ids

    out_ids = torch.empty((nq, S, k), dtype=torch.int32, device=dev)
    out_dist = torch.empty((nq, S, k), dtype=torch.float32, device=dev)
    host_out = {}   # pinned result buffers per (nq, nprobe)
    dev_out = {}    # device result buffers per nprobe

    def scan_slots(probes, nprobe, timed=False, aux=False, buf=0, for_merge=False, in_turn=None):
        """per-(query, slot) exact top-k on this rank's shards, gathered over ranks; returns (dist, global ids) on host.
        buf: which of the two pinned host buffer sets receives the results (the timed loop merges batch i on a worker thread while
        batch i + 1 is on the GPU)"""
        if not multi:
            lp = np.ascontiguousarray(probes[:, :nprobe])
        elif placement:
            lp = gcomm.mask_probes_replicated(probes, nprobe, rep_begin, rep_rank, rank)
        else:
            lp = gcomm.mask_probes(probes, nprobe, owner, rank)
        if nprobe not in dev_out:   # (a slice + .contiguous() would allocate and launch a copy kernel in every step)
            dev_out[nprobe] = ((out_ids, out_dist) if nprobe == S else
                               (torch.empty((nq, nprobe, k), dtype=torch.int32, device=dev), torch.empty((nq, nprobe, k), dtype=torch.float32, device=dev)))
        oi, od = dev_out[nprobe]
        ms = shards.scan_buckets_device(queries.data_ptr(), nq, k, lp, nprobe, oi.data_ptr(), od.data_ptr(), stream, timed=timed, aux=aux, for_merge=for_merge)
        key = (nq, nprobe, buf)
        if multi:
            # the result exchange (gpann_comm_gather_slots): local row ids -> global ids, ONE all-gather of the per-slot lists over RCCL
            # on this stream, owner selection on the device; the selected lists come back through pinned host buffers
            n_slots = nq * nprobe * k
            if key not in host_out:
                host_out[key] = (torch.empty((nq, nprobe, k), dtype=torch.float32).pin_memory(), torch.empty((nq, nprobe, k), dtype=torch.int32).pin_memory(),
                                 torch.empty((nq, nprobe, k), dtype=torch.float32, device=dev), torch.empty((nq, nprobe, k), dtype=torch.int32, device=dev),
                                 torch.empty(comm.gather_slots_scratch(n_slots), dtype=torch.uint8, device=dev if on_dev else "cpu"))
            hd, hi, gd, gi, scratch = host_out[key]
            in_turn = contextlib.nullcontext() if in_turn is None else in_turn
            if on_dev:
                with in_turn:
                    comm.gather_slots(od.data_ptr(), oi.data_ptr(), n_slots, gd.data_ptr(), gi.data_ptr(), scratch.data_ptr(), id_map=my_ids_dev.data_ptr(), stream=stream)
                hd.copy_(gd, non_blocking=True)
                hi.copy_(gi, non_blocking=True)
                comm.wait(stream, COMM_WAIT_S)   # (a synchronise that gives up when a rank has left: gpann_comm_wait)
            else:
                hd.copy_(od, non_blocking=True)
                hi.copy_(oi, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                sd_, si_ = hd.numpy().copy(), hi.numpy().view(np.uint32).copy()
                with in_turn:
                    comm.gather_slots(sd_, si_, n_slots, hd.numpy(), hi.numpy().view(np.uint32), scratch.numpy(), id_map=my_ids_host)
            dist_h = hd.numpy()
            ids_h = hi.numpy().view(np.uint32)
        else:
            # one rank holds every row in global-id order: local row id = global id.  Results come back through pinned host buffers
            if key not in host_out:
                host_out[key] = (torch.empty((nq, nprobe, k), dtype=torch.float32).pin_memory(), torch.empty((nq, nprobe, k), dtype=torch.int32).pin_memory())
            hd, hi = host_out[key]
            hd.copy_(od, non_blocking=True)
            hi.copy_(oi, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            dist_h = hd.numpy()
            ids_h = hi.numpy().view(np.uint32)
        return dist_h, ids_h, ms

    chunk_q = (nq + world - 1) // world
    probes_buf = torch.empty((nq, S), dtype=torch.int32).pin_memory().numpy()   # pinned: the router's D2H copy lands in it directly

    probes_top = {}   # pinned [nq][n_top] arrays

    def route(n_top=None, j=0, st=None, cm=None, turn=None, turn_key=None):
        """KMeansTreeRouter::Query for the whole batch.  n_top: only the first n_top shards of every order come back (the timed steps
        probe nprobe shards; the sweep before them needs the whole order).  N > 1: the router is replicated, every rank routes its slice
        of the queries and the probe lists ([nq][S] int32, 1.6 MB) are all-gathered — on stream `st` through communicator `cm` into buffer
        set `j` (the timed loop's route-ahead thread passes its own of each; `turn`: the loop's collective schedule, Turnstile)."""
        in_turn = turn.slot(turn_key) if turn is not None else contextlib.nullcontext()
        st = stream if st is None else st
        cm = comm if cm is None else cm
        if not multi:   # (one reusable host array for the probe lists; min_dist itself is not needed here)
            buf = probes_buf
            if n_top is not None and n_top < S:
                if n_top not in probes_top:
                    probes_top[n_top] = torch.empty((nq, n_top), dtype=torch.int32).pin_memory().numpy()
                buf = probes_top[n_top]
            pr, _, r_ms = router.query_batch(queries.data_ptr(), args.search_budget, st, on_device=True, nq=nq, out_probes=buf,
                                             want_min_dist=False, n_top=None if buf is probes_buf else n_top)
            return pr, r_ms
        lo = min(nq, rank * chunk_q)
        hi = min(nq, lo + chunk_q)
        r_ms = 0.0
        mine_t, all_t, dev_mine, dev_all = route_sets[j]
        mine, allp = mine_t.numpy(), all_t.numpy()
        mine[:] = -1
        if hi > lo:
            _, _, r_ms = router.query_batch(queries.data_ptr() + lo * d * 4, args.search_budget, st, on_device=True, nq=hi - lo,
                                            out_probes=mine[:hi - lo], want_min_dist=False)
        if on_dev:   # all-gather of the probe lists over RCCL on this stream
            with torch.cuda.stream(torch.cuda.ExternalStream(st, device=dev)) if st != stream else contextlib.nullcontext():
                dev_mine.copy_(mine_t, non_blocking=True)
                with in_turn:   # (the launch has its place in the loop's schedule; the wait below does not hold it)
                    cm.allgather(dev_mine.data_ptr(), dev_all.data_ptr(), mine.nbytes, st)
                all_t.copy_(dev_all, non_blocking=True)
            cm.wait(st, COMM_WAIT_S)
        else:
            with in_turn:
                cm.allgather(mine, allp, mine.nbytes)
        return allp.reshape(-1, S)[:nq], r_ms

    if multi:
        def _route_set():
            mine_t = torch.empty((chunk_q, S), dtype=torch.int32).pin_memory()
            all_t = torch.empty((world, chunk_q, S), dtype=torch.int32).pin_memory()
            dm = torch.empty((chunk_q, S), dtype=torch.int32, device=dev) if on_dev else None
            da = torch.empty((world, chunk_q, S), dtype=torch.int32, device=dev) if on_dev else None
            return mine_t, all_t, dm, da
        route_sets = [_route_set() for _ in range(4)]   # [0]: the calls outside the timed loop; [1] .. [3]: the route-ahead thread, in turn

    # ---------------------------------------------------------------- ground truth (same kernel, every shard) + nprobe sweep
    t0 = time.time()
    all_probes = np.tile(np.arange(S, dtype=np.int32), (nq, 1))
    # the ground truth comes from the plain exact (VALU) scan whatever mode is timed: a defect of the filtered pipeline cannot
    # hide in its own ground truth
    shards.set_scan_mode(0)
    gt_slot_d, gt_slot_i, _ = scan_slots(all_probes, S, aux=True)
    shards.set_scan_mode(1 if args.scan_mode == "filter" else 0)
    gt_slot_d, gt_slot_i = gt_slot_d.copy(), gt_slot_i.copy()   # (scan_slots hands out views of its reusable pinned buffers)
    gt_d, gt_i, _ = pygpann.merge_dedup(gt_slot_d, gt_slot_i, S, k)
    kth = gt_d[:, k - 1].copy()
    log(f"ground truth over {n} points in {time.time() - t0:.1f}s")

    def find_nprobe(probes):
        """nprobe sweep on the host: the ground-truth pass already produced the exact top-k of EVERY (query, shard), so the
        result of probing the first j shards of a route is a gather + DedupNeighbors (small_scale_queries.cpp:149-157)"""
        curve = []
        routed_d = np.take_along_axis(gt_slot_d, probes[:, :, None].astype(np.int64), axis=1)
        routed_i = np.take_along_axis(gt_slot_i, probes[:, :, None].astype(np.int64), axis=1)
        for np_try in range(1, S + 1):
            md, mi, _ = pygpann.merge_dedup(routed_d, routed_i, np_try, k)
            rec = float(np.count_nonzero(md <= kth[:, None])) / (nq * k)
            curve.append(round(rec, 6))
            if (args.nprobe and np_try == args.nprobe) or (not args.nprobe and rec >= args.target_recall):
                return np_try, curve
        return S, curve

    # ---------------------------------------------------------------- the timed hot path
    trace = os.environ.get("GPANN_BENCH_TRACE") == "1"
    merge_bound = not args.no_merge_bound   # the step's product is the MERGED list: slots may be cut at the query's merge bound (include/gpann.h)
    def step(buf=0, merge_later=None):
        """one batch: route -> scan -> merge.  merge_later (the timed loop): the merge is handed to a worker thread and runs while the
        next batch is on the GPU; its future is returned instead of the merged lists."""
        t_a = time.perf_counter()
        pr, r_ms = route(n_top=nprobe)
        t_b = time.perf_counter()
        sd, si, s_ms = scan_slots(pr, nprobe, timed=True, buf=buf, for_merge=merge_bound)
        if merge_later is not None:
            return merge_later.submit(pygpann.merge_dedup, sd, si, nprobe, k), None, r_ms, s_ms
        t_c = time.perf_counter()
        md, mi, _ = pygpann.merge_dedup(sd, si, nprobe, k)
        t_d = time.perf_counter()
        if trace:
            log(f"step: route {1e3 * (t_b - t_a):.2f} ms (kernels {r_ms:.2f}), scan+gather {1e3 * (t_c - t_b):.2f} ms (kernels {s_ms:.2f}), "
                f"merge {1e3 * (t_d - t_c):.2f} ms")
        return md, mi, r_ms, s_ms

    def barrier():
        if multi:
            comm.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- router configurations (the reference's grid), best one is timed
    candidates = []
    for index_budget in [int(x) for x in str(args.router_budget).split(",") if x]:
        router, tree, router_train_s = train_router(index_budget)
        probes, _ = route()
        nprobe, recall_curve = find_nprobe(probes)
        log(f"index budget {index_budget}: recall curve {recall_curve} -> nprobe {nprobe}")
        step()
        barrier()
        # the BEST of five single steps: the choice of the timed configuration must not hang on one host hiccup (a box was seen to spend 19 ms in a
        # step that takes 4.6 — averaged over two steps, as this trial used to be, that handed the timed loop to the 100k-centroid router: 1.1M QPS for 2.9M)
        trial = best_single_step_seconds(step, barrier, 5)
        if multi:
            trial = comm.max_double(trial)
        candidates.append(dict(index_budget=index_budget, router=router, tree=tree, train_s=router_train_s, nprobe=nprobe, recall_curve=recall_curve,
                               trial_ms=1000.0 * trial, probes=probes.copy()))
        if os.environ.get("GPANN_BENCH_FREQUENCY") == "1" and not multi:
            # KMeansTreeRouter::FrequencyQuery (500 votes, the reference's default) on the same batch: the bounded replay against the canonical table
            hq = queries.cpu().numpy()
            fq = {}
            outs = {}
            for name, env in (("bounded", None), ("canonical_table", "0")):
                router.set_option("GPANN_ROUTER_BOUND_TABLE", env)
                outs[name] = router.frequency_query_batch(hq, args.search_budget, 500)
                t0 = time.perf_counter()
                for _ in range(2):
                    router.frequency_query_batch(hq, args.search_budget, 500)
                fq[name + "_ms"] = 1000.0 * (time.perf_counter() - t0) / 2
            router.set_option("GPANN_ROUTER_BOUND_TABLE", None)
            a_, b_ = outs["bounded"], outs["canonical_table"]
            fq["identical"] = bool(all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a_[:4], b_[:4])))
            candidates[-1]["frequency_query"] = fq
            log(f"index budget {index_budget}: FrequencyQuery(500 votes) {fq}")
        log(f"index budget {index_budget}: {candidates[-1]['trial_ms']:.2f} ms per step")
    del gt_slot_d, gt_slot_i
    best = min(candidates, key=lambda c: c["trial_ms"])
    router, tree, router_train_s, nprobe, recall_curve = best["router"], best["tree"], best["train_s"], best["nprobe"], best["recall_curve"]
    probes = best["probes"]   # the timed configuration's routes (the CPU baseline's sample is chosen from them)
    log(f"timed configuration: index budget {best['index_budget']}, nprobe {nprobe}")

    scan_ms, router_ms, algo_bytes, filter_ms, rest_ms, step_wall_ms, stats = [], [], [], [], [], [], None
    host_trace = []   # GPANN_BENCH_TRACE=1: (event, batch, ms since the timed loop began) of the main thread
    router_bound_ms, router_bound_flops = [], 0.0
    import concurrent.futures
    # The timed loop is a software pipeline over the batches, one step = route + scan + merge of one batch, K of each inside the clock:
    #  * the host merge of batch i (DedupNeighbors, ~0.1 ms) runs on a worker thread while batch i + 1 is on the GPU; the results land
    #    in two alternating pinned buffer sets; every merge has finished before the clock stops;
    #  * (one rank) batch i + 1 is ROUTED on a second HIP stream by a second worker thread while batch i is scanned: the router's
    #    kernels fill what the scan leaves idle (its host-side planning, the tail of its launches) and the router -> scan hand-over
    #    (probe lists through pinned host arrays) leaves the critical path.  GPANN_BENCH_ROUTE_AHEAD=0: strictly one batch at a time.
    merger = concurrent.futures.ThreadPoolExecutor(1)
    # (several ranks, round 4: the same with the slice routing + the probe all-gather + its wait on the second stream, through the second communicator)
    route_ahead = os.environ.get("GPANN_BENCH_ROUTE_AHEAD", "1") != "0" and (not multi or comm_r is not None)
    # (one rank, round 4) ... and the SCAN of batch i + 1 is planned, staged and enqueued behind batch i before batch i is waited for
    # (GPANN_SCAN_DEFER / gpann_scan_finish, include/gpann.h): the host side of a scan call — task lists, group tables, the uploads, ~35 launches —
    # runs under the previous batch's kernels instead of between two batches.  GPANN_BENCH_SCAN_AHEAD=0: the scan call completes before it returns.
    # (several ranks: the scan is deferred the same way — into device arrays, two sets — and the exchange of batch i's slot lists (gather_slots,
    # the copies to the host) runs on a THIRD stream once its scan has been completed, next to the scan of batch i + 1)
    scan_ahead = route_ahead and os.environ.get("GPANN_BENCH_SCAN_AHEAD", "1") != "0"
    # (round 5) ... on the handle's own streams (GPANN_SCAN_OVERLAP: a front stream per call slot, one rest stream, one tail stream): the passes before the rest pass of batch i + 1 — short launches that leave
    # most of the chip idle — run BESIDE the rest pass of batch i, which is bound by HBM and gives up compute units every few work items.
    # GPANN_BENCH_OVERLAP=0: both calls on one stream, batch i + 1 strictly behind batch i.
    scan_overlap = scan_ahead and os.environ.get("GPANN_BENCH_OVERLAP", "1") != "0"
    if route_ahead:
        r_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("GPANN_BENCH_ROUTER_PRIORITY", "0")))
        r_bufs = [torch.empty((nq, nprobe), dtype=torch.int32).pin_memory().numpy() for _ in range(3)]
        router_pool = concurrent.futures.ThreadPoolExecutor(1)
        def route_job(j, turn=None):
            if multi:   # my slice, the all-gather of the probe lists and its wait: all on the second stream, buffer sets 1 / 2 alternating
                torch.cuda.set_device(local_rank)
                pr, r_ms = route(j=1 + j % 3, st=r_stream.cuda_stream, cm=comm_r, turn=turn, turn_key=("R", j))
                return pr, r_ms, router.stats()
            pr, _, r_ms = router.query_batch(queries.data_ptr(), args.search_budget, r_stream.cuda_stream, on_device=True, nq=nq, out_probes=r_bufs[j % 3],
                                             want_min_dist=False, n_top=None if nprobe == S else nprobe)
            return pr, r_ms, router.stats()

    x_stream = torch.cuda.Stream(device=dev) if multi else None
    # the deferred scan calls are issued on a stream of their own, not on the legacy default stream: under GPANN_SCAN_OVERLAP the caller's stream
    # only orders a call's START (an event recorded on it), and an event on the default stream waits for every blocking stream of the process —
    # the router's among them: the front of batch i + 2 would wait for the routing of batch i + 3
    s_stream = torch.cuda.Stream(device=dev)
    dev_sets = {}

    def scan_begin(pr, buf):
        """enqueue the scan of one batch (deferred call): the per-slot lists travel to the pinned host buffers `buf` behind the kernels"""
        key = (nq, nprobe, buf)
        if multi:   # this rank's slots into device arrays of set `buf`; the exchange follows in scan_finish
            lp = gcomm.mask_probes_replicated(pr, nprobe, rep_begin, rep_rank, rank) if placement else gcomm.mask_probes(pr, nprobe, owner, rank)
            if buf not in dev_sets:
                dev_sets[buf] = (torch.empty((nq, nprobe, k), dtype=torch.int32, device=dev), torch.empty((nq, nprobe, k), dtype=torch.float32, device=dev))
            oi, od = dev_sets[buf]
            shards.scan_buckets_device(queries.data_ptr(), nq, k, lp, nprobe, oi.data_ptr(), od.data_ptr(), s_stream.cuda_stream if scan_overlap else stream,
                                       for_merge=merge_bound, defer=True, overlap=scan_overlap)
            return
        if key not in host_out:
            host_out[key] = (torch.empty((nq, nprobe, k), dtype=torch.float32).pin_memory(), torch.empty((nq, nprobe, k), dtype=torch.int32).pin_memory())
        hd, hi = host_out[key]
        shards.scan_buckets_device(queries.data_ptr(), nq, k, np.ascontiguousarray(pr[:, :nprobe]), nprobe, hi.data_ptr(), hd.data_ptr(),
                                   s_stream.cuda_stream if scan_overlap else stream, for_merge=merge_bound, defer=True, host_out=True, overlap=scan_overlap)

    def scan_finish(buf, in_turn=None):
        ms = shards.finish()
        in_turn = contextlib.nullcontext() if in_turn is None else in_turn
        if multi:   # the batch's scan is complete (an overflowed filter call redone): its slot lists are exchanged on the third stream
            key = (nq, nprobe, buf)
            n_slots = nq * nprobe * k
            if key not in host_out:
                host_out[key] = (torch.empty((nq, nprobe, k), dtype=torch.float32).pin_memory(), torch.empty((nq, nprobe, k), dtype=torch.int32).pin_memory(),
                                 torch.empty((nq, nprobe, k), dtype=torch.float32, device=dev), torch.empty((nq, nprobe, k), dtype=torch.int32, device=dev),
                                 torch.empty(comm.gather_slots_scratch(n_slots), dtype=torch.uint8, device=dev if on_dev else "cpu"))
            hd, hi, gd, gi, scratch = host_out[key]
            oi, od = dev_sets[buf]
            with torch.cuda.stream(x_stream):
                if on_dev:
                    with in_turn:
                        comm.gather_slots(od.data_ptr(), oi.data_ptr(), n_slots, gd.data_ptr(), gi.data_ptr(), scratch.data_ptr(), id_map=my_ids_dev.data_ptr(),
                                          stream=x_stream.cuda_stream)
                    hd.copy_(gd, non_blocking=True)
                    hi.copy_(gi, non_blocking=True)
                else:
                    hd.copy_(od, non_blocking=True)
                    hi.copy_(oi, non_blocking=True)
            if on_dev:
                comm.wait(x_stream.cuda_stream, COMM_WAIT_S)
            else:
                x_stream.synchronize()
                sd_, si_ = hd.numpy().copy(), hi.numpy().view(np.uint32).copy()
                with in_turn:
                    comm.gather_slots(sd_, si_, n_slots, hd.numpy(), hi.numpy().view(np.uint32), scratch.numpy(), id_map=my_ids_host)
            return hd.numpy(), hi.numpy().view(np.uint32), ms
        hd, hi = host_out[(nq, nprobe, buf)]
        return hd.numpy(), hi.numpy().view(np.uint32), ms

    def run_steps(count, record):
        """count steps back to back; returns the last merged lists"""
        pending = None
        md = mi = None
        def note(t_step, r_ms, s_ms, rst):
            if record is None:
                return
            nonlocal_stats = shards.stats()
            step_wall_ms.append(round(1e3 * (time.perf_counter() - t_step), 3))
            scan_ms.append(nonlocal_stats["scan_ms"])
            filter_ms.append(nonlocal_stats["filter_ms"])
            rest_ms.append(nonlocal_stats["filter_rest_ms"])
            router_ms.append(r_ms)
            router_bound_ms.append(rst["last_bound_ms"])
            algo_bytes.append(nonlocal_stats["algorithmic_bytes"])
            record["stats"] = nonlocal_stats
            record["router_bound_flops"] = rst["last_bound_flops"]
        # several ranks: the loop's collectives — R j: the probe all-gather of batch j (route-ahead thread, second communicator), X i: the slot
        # exchange of batch i (this thread) — are LAUNCHED in one order on every rank (Turnstile)
        turn = None
        # scans in flight while one is being completed: 1 (the deferred call of round 4), 2 with GPANN_SCAN_OVERLAP — the front stream then always
        # holds the next front, so the rest passes follow each other without a gap; the routes run one batch further ahead than the scans
        scans_ahead = (max(1, min(2, int(os.environ.get("GPANN_BENCH_SCANS_AHEAD", "2")))) if scan_overlap else 1) if scan_ahead else 0
        # the routes run routes_ahead batches ahead of the scan being begun: beside three scans the router's launches take up to ten times their
        # time alone, and a route that is not there when its scan is to begin stalls the whole pipeline
        routes_ahead = max(1, min(2, int(os.environ.get("GPANN_BENCH_ROUTES_AHEAD", "2")))) if scan_overlap else 1
        if multi and route_ahead and count:
            r_ahead = scans_ahead + routes_ahead
            order = [("R", j) for j in range(min(r_ahead, count))]
            for it in range(count):
                if it + r_ahead < count:
                    order.append(("R", it + r_ahead))
                order.append(("X", it))
            turn = Turnstile(order)
            _live_turns.append(turn)
        x_turn = (lambda it: turn.slot(("X", it))) if turn is not None else (lambda it: None)
        if scan_ahead and count:
            depth = scans_ahead + 1     # host / device buffer sets: one per scan in flight
            routes = {j: router_pool.submit(route_job, j, turn) for j in range(min(routes_ahead, count))}
            info = {}

            t_origin = time.perf_counter()
            stamp = (lambda what, j: host_trace.append((what, j, round(1e3 * (time.perf_counter() - t_origin), 3)))) if (trace and record is not None) else (lambda what, j: None)

            def begin(j):
                """the routing of batch j has arrived: the next one goes to the router thread, the scan of batch j is planned and enqueued"""
                stamp("route_wait", j)
                pr, r_ms, rst = routes.pop(j).result()
                info[j] = (r_ms, rst)
                if j + routes_ahead < count:
                    routes[j + routes_ahead] = router_pool.submit(route_job, j + routes_ahead, turn)
                stamp("begin", j)
                scan_begin(pr, j % depth)
                stamp("begun", j)
            for j in range(min(scans_ahead, count)):
                begin(j)
            t_step = time.perf_counter()
            for it in range(count):
                if pending is not None:   # the merge of batch it - 1 (submitted a whole batch ago): its host buffers are handed to the batch begun next
                    md, mi, _ = pending.result()
                    pending = None
                if it + scans_ahead < count:
                    begin(it + scans_ahead)     # (queued behind the batches the GPU is still working on)
                stamp("finish", it)
                sd, si, s_ms = scan_finish(it % depth, x_turn(it))
                stamp("finished", it)
                pending = merger.submit(pygpann.merge_dedup, sd, si, nprobe, k)
                note(t_step, info[it][0], s_ms, info.pop(it)[1])
                t_step = time.perf_counter()
            if pending is not None:
                md, mi, _ = pending.result()
            if record is not None:
                record["last_buf"] = (count - 1) % depth     # the buffer set the last step's lists went to
            return md, mi
        nxt = router_pool.submit(route_job, 0, turn) if route_ahead and count else None
        for it in range(count):
            t_step = time.perf_counter()
            if route_ahead:
                pr, r_ms, rst = nxt.result()
                nxt = router_pool.submit(route_job, it + 1, turn) if it + 1 < count else None
                sd, si, s_ms = scan_slots(pr, nprobe, timed=True, buf=it & 1, for_merge=merge_bound, in_turn=x_turn(it))
                fut = merger.submit(pygpann.merge_dedup, sd, si, nprobe, k)
            else:
                fut, _, r_ms, s_ms = step(buf=it & 1, merge_later=merger)
                rst = router.stats()
            if pending is not None:
                md, mi, _ = pending.result()
            pending = fut
            note(t_step, r_ms, s_ms, rst)
        if pending is not None:
            md, mi, _ = pending.result()
        if record is not None and count:
            record["last_buf"] = (count - 1) & 1
        return md, mi

    run_steps(args.warmup, None)
    import gc
    gc.collect()
    gc.disable()   # (a generation-2 collection in the middle of a 9 ms step is a visible outlier)
    rec = {}
    barrier()
    t0 = time.perf_counter()
    md, mi = run_steps(args.steps, rec)
    stats = rec.get("stats")
    router_bound_flops = rec.get("router_bound_flops", 0.0)
    barrier()
    elapsed = time.perf_counter() - t0
    # the same step strictly one batch at a time (route -> scan -> D2H -> merge, nothing of another batch in flight): the per-batch wall
    # time the metric's "10k batched queries" literally describes; `value` above is the pipelined throughput
    lat, rest_alone_ms, scan_alone_ms = [], [], []
    for _ in range(2):
        step()
    barrier()
    for _ in range(max(5, min(20, args.steps))):
        t1 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t1))
        st_alone = shards.stats()     # (the launches of a step with the chip to themselves: what the rest pass takes when nothing runs beside it)
        rest_alone_ms.append(st_alone["filter_rest_ms"])
        scan_alone_ms.append(st_alone["scan_ms"])
    latency_ms_one_batch = float(np.mean(lat))
    cu_sweep = None
    if os.environ.get("GPANN_BENCH_CU_SWEEP") and not multi:
        # measurement aid (not part of any result): the scan of one batch on a stream that may use only the first N compute units
        # (hipExtStreamCreateWithCUMask; the mask's bits interleave the XCDs) — how much of the chip the rest pass needs, what the passes before it lose
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        cu_sweep = []
        lp = np.ascontiguousarray(probes[:, :nprobe])
        oi = torch.empty((nq, nprobe, k), dtype=torch.int32, device=dev)
        od = torch.empty((nq, nprobe, k), dtype=torch.float32, device=dev)
        for n_cu in [int(x) for x in os.environ["GPANN_BENCH_CU_SWEEP"].split(",") if x]:
            words = (ctypes.c_uint32 * 8)(*[(((1 << max(0, min(32, n_cu - 32 * w))) - 1) & 0xFFFFFFFF) for w in range(8)])
            st_m = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st_m), 8, words)
            if rc != 0:
                cu_sweep.append({"cus": n_cu, "error": f"hipExtStreamCreateWithCUMask -> {rc}"})
                continue
            rows = []
            for it in range(6):
                t1 = time.perf_counter()
                shards.scan_buckets_device(queries.data_ptr(), nq, k, lp, nprobe, oi.data_ptr(), od.data_ptr(), st_m.value, timed=True, for_merge=merge_bound)
                hip.hipStreamSynchronize(st_m)
                wall = 1e3 * (time.perf_counter() - t1)
                stt = shards.stats()
                _, _, r_ms = router.query_batch(queries.data_ptr(), args.search_budget, st_m.value, on_device=True, nq=nq, out_probes=r_bufs[0] if route_ahead else None,
                                                want_min_dist=False, n_top=None if nprobe == S else nprobe)
                if it:
                    rows.append((wall, stt["scan_ms"], stt["filter_ms"], stt["filter_rest_ms"], r_ms))
            a = np.array(rows).mean(axis=0)
            cu_sweep.append({"cus": n_cu, "scan_wall_ms": round(float(a[0]), 3), "scan_kernels_ms": round(float(a[1]), 3), "filter_ms": round(float(a[2]), 3),
                             "rest_pass_ms": round(float(a[3]), 3), "router_kernels_ms": round(float(a[4]), 3)})
            log(f"CU sweep: {cu_sweep[-1]}")
            hip.hipStreamDestroy(st_m)
    merger.shutdown()
    if route_ahead:
        router_pool.shutdown()
    gc.enable()
    if multi:
        elapsed = comm.max_double(elapsed)
    recall = float(np.count_nonzero(md <= kth[:, None])) / (nq * k)
    qps = nq * args.steps / elapsed
    # what the merge of the last step had to drop: ids that came back from more than one probed shard (overlapping shards only)
    last_ids = np.sort(host_out[(nq, nprobe, rec.get("last_buf", 0))][1].numpy().view(np.uint32).reshape(nq, -1), axis=1)
    dup_ids = int(np.count_nonzero((last_ids[:, 1:] == last_ids[:, :-1]) & (last_ids[:, 1:] != 0xFFFFFFFF)))
    slot_entries = int(np.count_nonzero(last_ids != 0xFFFFFFFF))

    result = {
        "metric": "QPS @ recall@10=0.9, 10k batched queries" if not args.nprobe else f"QPS @ nprobe={args.nprobe}, 10k batched queries",
        "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps, "latency_ms_one_batch": latency_ms_one_batch, "qps_one_batch_at_a_time": 1000.0 * nq / latency_ms_one_batch,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: {n}x{d} f32 {'inner product' if metric else 'L2'}, "
                               + (f"Gaussian mixture ({n_centers} centres, sigma {sigma})" if args.dist != "uniform" else "one Gaussian blob (no clusters)")
                               + (f" + {offset:g} in every coordinate" if offset else "") + f", {nq} queries, {S} k-means shards"
                               + (f" overlapping SPANN-style ({store_rows - n} points = {100.0 * (store_rows - n) / n:.1f} % also in a second shard: {store_rows} store rows)" if args.overlap > 0 else "") + ", "
                               f"k={k}, KMeansTreeRouter ({tree['centroids'].shape[0]} centroids in {len(tree['node_shard'])} nodes, search budget {args.search_budget}) + exact "
                               f"shard scan, nprobe={nprobe}",
                   "n": n, "d": d, "metric": args.metric, "dist": args.dist, "queries": nq, "shards": S, "k": k, "nprobe": nprobe, "recall_at_k": recall,
                   "overlap": args.overlap, "store_rows": store_rows,
                   "parallelism": f"shards over {world} GPU(s), LPT"},
    }
    avg_scan_s = float(np.mean(scan_ms)) / 1000.0
    hbm_alg = float(np.mean(algo_bytes)) / avg_scan_s / 1e9 if avg_scan_s > 0 else 0.0
    filtered = bool(stats["filtered"])
    # per step: the cells filter kernel is launched three times (head, sample, rest), the exact kernel once; the rest pass — the
    # launch that streams (nearly) the whole store once — is the dominant kernel
    if filtered:
        f_s = float(np.mean(filter_ms)) / 1000.0
        r_s = float(np.mean(rest_ms)) / 1000.0
        # matrix-core work: the wave-instructions the filter kernels ISSUED, counted by the kernels themselves (gpann_scan_stats.filter_mfma_*: KS per
        # plane product of a (row tile, query tile) pair; the 8-bit passes' second half only where the H.P' test did not rule the product out), each
        # dtype against its own dense peak.  The two dtypes share the filter launches' time (f_s: the head pass and the sample..rest interval, HIP
        # events), so each fraction is a lower bound of that dtype's own utilisation and their sum is at most 1; the rest pass alone — int8 only when
        # it runs on the 8-bit plane — is divided by its own launch time
        bf16_tflops = stats["filter_mfma_bf16"] * 32768.0 / f_s / 1e12 if f_s > 0 else 0.0     # v_mfma_f32_32x32x16_bf16: 32 x 32 x 16 x 2 flop
        i8_tops = stats["filter_mfma_i8"] * 65536.0 / f_s / 1e12 if f_s > 0 else 0.0           # v_mfma_i32_32x32x32_i8: 32 x 32 x 32 x 2 integer ops
        rest_i8_tops = stats["filter_rest_mfma_i8"] * 65536.0 / r_s / 1e12 if r_s > 0 else 0.0
        # bytes the rest pass has to read: its row tiles x (bf16 hi plane [+ lo plane] + the row's norm word)
        ks_read = stats["filter_plane_ksteps"]   # 16-dim k-steps a plane row is stored with and the kernel loads (32 B each; ceil(d / 16) rounded up to a kernel instance)
        # (bf16 hi plane: 32 B per k-step of 16 dimensions; 8-bit plane: 32 B per k-step of 32 dimensions — the library reports which)
        rest_bytes = stats["filter_rest_rows"] * float(stats["filter_rest_row_bytes"])
        rest_gbps = rest_bytes / r_s / 1e9 if r_s > 0 else 0.0
        traffic, traffic_src = lookup_traffic(n, d, S, nprobe, "FilterKernelCells<", {"": 1}, largest_launch=True, kernel_bytes=rest_bytes)
        # step level: the bytes every launch of the step has to read — plane rows of all filter launches (two-plane launches twice the plane
        # bytes), the rows re-scored exactly (candidates + seed rows, d x 4 each), the router's centroid planes (both passes, both planes)
        # (the sample pass — the hi-plane rows outside the rest pass — streams the int8 plane too unless the store has none or GPANN_FILTER_Q8_FROM_PASS=2:
        # until round 5 it was charged the bf16 hi plane's 260 B per row, 1.3 GB too many at NS)
        sample_on_q8 = stats["filter_rest_row_bytes"] < 32 * ks_read + 4 and shards.get_option("GPANN_FILTER_Q8_FROM_PASS") != "2"
        row_hi = float(stats["filter_rest_row_bytes"]) if sample_on_q8 else 32.0 * ks_read + 4.0
        hi_rows = stats["filter_rows"] - stats["filter_two_plane_rows"] - stats["filter_rest_rows"]
        step_bytes = (rest_bytes + hi_rows * row_hi + stats["filter_two_plane_rows"] * (64.0 * ks_read + 4.0)
                      + (stats["filter_candidates"] + stats["filter_seed_rows"]) * d * 4.0
                      + 2.0 * tree["centroids"].shape[0] * 64.0 * ks_read)
        step_s = elapsed / args.steps
        ra_s = float(np.mean(rest_alone_ms)) / 1000.0 if rest_alone_ms else 0.0
        result["roofline"] = {"bound": "hbm", "achieved": rest_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": rest_gbps / HBM_PEAK_GBPS,
                              "traffic": traffic, "traffic_source": traffic_src,
                              "kernel_ms_alone": 1000.0 * ra_s, "achieved_alone": rest_bytes / ra_s / 1e9 if ra_s > 0 else None,
                              "frac_alone": rest_bytes / ra_s / 1e9 / HBM_PEAK_GBPS if ra_s > 0 else None,
                              "overlap_note": ("kernel_ms / achieved / frac are the rest pass's launches INSIDE the timed loop, where (GPANN_SCAN_OVERLAP) the fronts of "
                                               "the next two batches and the router run beside it on the same chip and the same HBM — by design its time there "
                                               "is longer than alone; kernel_ms_alone / frac_alone = the same launch of the same batch in the one-batch loop of this "
                                               "run (nothing beside it); frac_step = what all launches of a step together make of the HBM peak"
                                               if scan_overlap else "nothing runs beside the rest pass in this run's timed loop"),
                              "frac_step": step_bytes / step_s / 1e9 / HBM_PEAK_GBPS, "step_bytes": step_bytes,
                              "frac_step_note": "bytes ALL launches of a step have to read (plane rows of every filter pass, exactly re-scored rows, the router's "
                                                "centroid planes) / ms_per_step / 8 TB/s: what the step as a whole makes of the memory system",
                              "kernel": "gpann::FilterKernelCells (rest pass)", "kernel_ms": 1000.0 * r_s, "kernel_bytes": rest_bytes,
                              "kernel_row_bytes": stats["filter_rest_row_bytes"],
                              "kernel_plane": "int8 (per-cell scaled residuals)" if stats["filter_rest_row_bytes"] < 32 * ks_read + 4 else "bf16 hi",
                              "note": "achieved = the bytes the dominant launch (the rest pass of FilterKernelCells) HAS TO read — its row tiles x "
                                      "kernel_row_bytes (the int8 or bf16 hi plane it streams + the norm word) — / its own HIP-event time on the scan's stream; frac = achieved / 8 TB/s.  traffic = the PMC "
                                      "figure (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE) of that launch from the committed pass named in "
                                      "traffic_source (null if there is none for this workload).  algorithmic_GBps = SURVEY 8d's figure — rows x d x 4 "
                                      "per (query, probed shard), the bytes the REFERENCE streams — / HIP-event time of all scan kernels of the step; it "
                                      "exceeds the HBM peak because the exact answer is produced without that work (cell pruning, bf16 matrix-core bound, "
                                      "canonical re-scoring of the candidates only) and is not a roofline fraction",
                              "algorithmic_GBps": hbm_alg, "algorithmic_bytes_per_step": float(np.mean(algo_bytes)),
                              "filter_kernels_ms": 1000.0 * f_s,
                              "filter_mfma_bf16_TFLOPs": bf16_tflops, "filter_mfma_frac_bf16": bf16_tflops / MFMA_BF16_PEAK_TFLOPS,
                              "filter_mfma_i8_TOPs": i8_tops, "filter_mfma_frac_i8": i8_tops / MFMA_I8_PEAK_TOPS,
                              "rest_pass_mfma_i8_TOPs": rest_i8_tops, "rest_pass_mfma_frac_i8": rest_i8_tops / MFMA_I8_PEAK_TOPS,
                              "filter_mfma_issued": {"bf16_32x32x16": stats["filter_mfma_bf16"], "i8_32x32x32": stats["filter_mfma_i8"],
                                                     "i8_32x32x32_rest_pass": stats["filter_rest_mfma_i8"]},
                              "filter_mfma_note": "issued matrix instructions counted by the kernels (not pairs x d): bf16 against 2.5 PFLOP/s, int8 against 5 POP/s "
                                                  "(MI355X_MICROARCH.md: 2x the bf16 rate per clock); filter_* over the filter launches' HIP-event time, rest_pass_* over "
                                                  "the rest pass's own",
                              "filter_pairs": stats["filter_pairs"], "filter_candidates": stats["filter_candidates"], "filter_dirty_tasks": stats["filter_dirty_tasks"],
                              "filter_rows": stats["filter_rows"], "filter_rest_rows": stats["filter_rest_rows"]}
    else:
        traffic, traffic_src = lookup_traffic(n, d, S, nprobe, "ScanTileKernel<0, 0>", {"": 1})
        lane_ops = stats["padded_pairs"] * d * ops_per_dim / avg_scan_s if avg_scan_s > 0 else 0.0
        result["roofline"] = {"bound": "valu", "achieved": lane_ops / 1e12, "peak": VALU_F32_PEAK_TLANEOPS, "unit": "T lane-ops/s",
                              "frac": lane_ops / 1e12 / VALU_F32_PEAK_TLANEOPS, "traffic": traffic, "traffic_source": traffic_src,
                              "kernel": "gpann::ScanTileKernel", "kernel_ms": 1000.0 * avg_scan_s,
                              "note": "exact mode: every HBM row is reused by up to 64 queries per workgroup, so the kernel is bound by FP32 VALU issue (3 unfused "
                                      "lane-ops per pair and dimension; peak = 157.3 TFLOP/s / 2 since nothing may be fused); algorithmic_GBps = SURVEY 8d's bytes "
                                      "/ kernel time",
                              "algorithmic_GBps": hbm_alg, "algorithmic_bytes_per_step": float(np.mean(algo_bytes)),
                              "valu_lane_ops_per_s": lane_ops}
    rb_s = float(np.mean(router_bound_ms)) / 1000.0 if router_bound_ms else 0.0
    if rb_s > 0:
        result["roofline"].update({"router_mfma_ms": 1000.0 * rb_s, "router_mfma_TFLOPs": router_bound_flops / rb_s / 1e12,
                                   "router_mfma_frac": router_bound_flops / rb_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                   "router_mfma_note": "RouterBoundKernel passes 1 + 2 (HIP events on the router's stream): 2 passes x 3 bf16 products x 2 nq C d flops"})
    result["backend"] = comm.name if multi else "none (1 rank)"      # what libgpann_comm.so reports: "rccl <version>" / "tcp"
    result["world_size_seen"] = comm.world if multi else 1
    if multi:
        # the work every rank actually did in the last timed step: task rows (pairs) of its scan
        per_rank = comm.allgather_host(np.array([float(stats["pairs"])], np.float64)).reshape(-1)
        result["placement"] = dict(placement or {"replicas": 0}, rank_task_rows=[float(x) for x in per_rank],
                                   task_rows_max_over_mean=float(per_rank.max() / max(per_rank.mean(), 1.0)))
    result["index_build_s"] = index_build_s
    result["hbm_resident_GB"] = hbm_resident_gb
    result["config"]["slot_lists"] = ("cut at the query's merge bound (GPANN_SCAN_FOR_MERGE): the merged neighbours are bit-identical to the merge of the "
                                      "complete QueryBucket lists" if merge_bound else "complete QueryBucket lists")
    result["config"]["pipeline"] = (("batches overlap: batch i+2 is routed (second stream), the scan of batch i+1 is planned and enqueued behind batch i (deferred "
                                     "scan call" + (" on the store's own streams: its passes before the rest pass run BESIDE the rest pass of batch i, GPANN_SCAN_OVERLAP"
                                                    if scan_overlap else "") + ") and batch i-1 is merged on the host while batch i is scanned" if scan_ahead else
                                     "batches overlap: batch i+1 is routed (second stream) and batch i-1 merged (host) while batch i is scanned")
                                    if route_ahead else "batch i-1 is merged on the host while batch i is routed and scanned")
    if multi and route_ahead:
        result["config"]["pipeline"] = ("batches overlap: every rank routes its slice of batch i+2 and the probe lists are all-gathered (second stream, second "
                                        "communicator), the scan of batch i+1 is planned and enqueued behind batch i (deferred scan call), the slot lists of batch i "
                                        "are exchanged on a third stream once its scan is complete; batch i-1 is merged on the host" if scan_ahead else
                                        "batches overlap: every rank routes its slice of batch i+1 and the probe lists are all-gathered (second stream, second "
                                        "communicator) while batch i is scanned and its slot lists are exchanged; batch i-1 is merged on the host")
    result["detail"] = {"merge": {"slot_entries": slot_entries, "duplicate_ids_dropped": dup_ids,
                                  "note": "DedupNeighbors over the per-slot lists of the last timed step: entries whose id also came back from another probed shard"},
                        "scan_mode": args.scan_mode, "filtered": filtered, "recall_curve": recall_curve, "router_ms": float(np.mean(router_ms)), "step_wall_ms": step_wall_ms,
                        "scan_kernels_ms": float(np.mean(scan_ms)), "filter_kernels_ms": float(np.mean(filter_ms)), "merge_kernel_ms": stats["merge_ms"],
                        "units": stats["units"], "tasks": stats["tasks"], "pairs": stats["pairs"], "exact_pairs_padded": stats["padded_pairs"],
                        "router_train_s": router_train_s, "router_index_budget": best["index_budget"],
                        "router_configs": [{kk: c[kk] for kk in ("index_budget", "nprobe", "trial_ms", "train_s", "recall_curve", "frequency_query") if kk in c} | {"centroids": int(c["tree"]["centroids"].shape[0]), "nodes": int(len(c["tree"]["node_shard"]))} for c in candidates],
                        "router_nodes": int(len(tree["node_shard"])),
                        "router_centroids": int(tree["centroids"].shape[0]), "setup_s": time.time() - t_setup}
    # the router's always-on self-checks over the whole run (gpann_router_get_stats): events where the matrix-core bound path left a minimum unset and the
    # block was re-evaluated canonically (bound_selfcheck), blocks GPANN_ROUTER_VERIFY compared and found different (verify_failures) — both must be 0
    rs_all = router.stats()
    result["detail"]["router_selfcheck_events"] = int(rs_all["bound_selfcheck"])
    result["detail"]["router_verify_failures"] = int(rs_all["verify_failures"])
    result["detail"]["router_bound_calls"] = int(rs_all["bound_calls"])
    if cu_sweep is not None:
        result["detail"]["cu_sweep"] = cu_sweep
    if host_trace:
        result["detail"]["host_trace"] = host_trace[:160]
        result["detail"]["per_step"] = {"router_ms": [round(x, 3) for x in router_ms], "scan_call_ms": [round(x, 3) for x in scan_ms], "rest_ms": [round(x, 3) for x in rest_ms]}

    # tuning aid (off by default): GPANN_BENCH_SWEEP="GPANN_FILTER_SEED_ROWS=64,128;GPANN_FILTER_HEAD_ROWS=512,0" times a few steps for every
    # combination of the listed options of the store (gpann_shards_set_option: read by every call) and checks that the results do not change
    if os.environ.get("GPANN_BENCH_SWEEP") and world == 1:
        import itertools
        knobs = [(kv.split("=")[0], kv.split("=")[1].split(",")) for kv in os.environ["GPANN_BENCH_SWEEP"].split(";") if kv]
        sweep = []
        for combo in itertools.product(*[v for _, v in knobs]):
            for (name, _), val in zip(knobs, combo):
                shards.set_option(name, val)
            step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(6):
                md_s, mi_s, _, _ = step()
            barrier()
            dt = (time.perf_counter() - t0) / 6
            st_s = shards.stats()
            same = bool(np.array_equal(mi_s, mi) and np.array_equal(md_s.view(np.uint32), md.view(np.uint32)))
            sweep.append({"knobs": dict(zip([n_ for n_, _ in knobs], combo)), "ms_per_step": 1000.0 * dt, "scan_kernels_ms": st_s["scan_ms"], "rest_ms": st_s["filter_rest_ms"],
                          "candidates": st_s["filter_candidates"], "rest_rows": st_s["filter_rest_rows"], "dirty_tasks": st_s["filter_dirty_tasks"], "filtered": st_s["filtered"],
                          "identical": same})
            log(f"sweep {sweep[-1]}")
        for name, _ in knobs:
            shards.set_option(name, None)
        result["detail"]["sweep"] = sweep

    # the same steps with the plain exact (VALU) scan, for reference: identical results, no matrix cores
    # (always: a filtered result that differs from the exact one fails the run — --no-exact-compare is accepted and ignored)
    if args.scan_mode == "filter":
        shards.set_scan_mode(0)
        step()
        barrier()
        t0 = time.perf_counter()
        md0, mi0, _, _ = step()
        barrier()
        dt = time.perf_counter() - t0
        st0 = shards.stats()
        same = bool(np.array_equal(mi0, mi) and np.array_equal(md0.view(np.uint32), md.view(np.uint32)))
        result["detail"]["exact_mode"] = {"qps": nq / dt, "ms_per_step": 1000.0 * dt, "scan_kernel_ms": st0["scan_ms"],
                                          "valu_lane_ops_per_s": st0["padded_pairs"] * d * ops_per_dim / (st0["scan_ms"] / 1000.0),
                                          "hbm_algorithmic_GBps": st0["algorithmic_bytes"] / (st0["scan_ms"] / 1000.0) / 1e9,
                                          "results_identical_to_filter_mode": same}
        shards.set_scan_mode(1)
        # the kernel north_star literally describes (ScanTileKernel: no matrix cores in the scan) against the peak of UNFUSED fp32 lane-ops
        result["roofline"]["exact_mode_frac"] = result["detail"]["exact_mode"]["valu_lane_ops_per_s"] / 1e12 / VALU_F32_PEAK_TLANEOPS
        result["roofline"]["exact_mode_note"] = ("ScanTileKernel (scan mode exact, same step, identical results): 3 unfused fp32 lane-ops per (pair, dimension) / kernel "
                                                 f"time / {VALU_F32_PEAK_TLANEOPS} T lane-ops/s")
        if not same:
            raise SystemExit("bench: the filtered scan's ids / distances differ from the exact scan's on the same step — no result")

    # ---------------------------------------------------------------- --emulate-world: every rank of a W-GPU run, alone on this GPU
    if emulate_worlds:
        result["scale_emulated"] = emulate_worlds_on_one_gpu(
            args, emulate_worlds, torch, pygpann, gcomm, dev, local_rank, stream, points_kept, clusters_local, counts, queries, probes, nprobe, router, merge_bound,
            one_gpu_ms=1000.0 * elapsed / args.steps)
        del points_kept

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only): the oracle on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(args, shards, tree, queries, probes, nprobe, kth, counts, md, mi)
        except Exception as e:   # the baseline must never take the headline number down with it
            result["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        if result["cpu_baseline"].get("results_match_gpu") is False:
            raise SystemExit("bench: the GPU's neighbours differ from the CPU oracle's on the sampled queries — no result\n" + json.dumps(result["cpu_baseline"]))
    # ---------------------------------------------------------------- the same benchmark on data the filter can prune far less (one rank, default workload only)
    if (rank == 0 and world == 1 and args.config == "ns" and args.dist == "mixture" and args.sigma == SIGMA and args.n == 100_000_000 and not args.no_hard_workload
            and not args.nprobe and args.scan_mode == "filter" and os.environ.get("GPANN_BENCH_HARD_WORKLOAD", "1") != "0"):
        import subprocess
        t_h = time.time()
        try:
            # a child process (this one keeps its 74 GB resident; both fit): same script, sigma 0.8, 5 steps, no CPU leg, no grandchildren
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--sigma", "0.8", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-hard-workload",
                                 "--router-budget", "20000"], capture_output=True, text=True, timeout=900)
            line = [ln for ln in cp.stdout.split("\n") if ln.startswith("{")]
            if cp.returncode != 0 or not line:
                raise RuntimeError(f"rc {cp.returncode}: {cp.stderr[-400:]}")
            h = json.loads(line[-1])
            result["detail"]["hard_workload"] = {
                "workload": h["config"]["workload"], "value": h["value"], "unit": h["unit"], "ms_per_step": h["ms_per_step"], "steps": h["steps"],
                "nprobe": h["config"]["nprobe"], "recall_at_k": h["config"]["recall_at_k"], "recall_curve": h["detail"]["recall_curve"],
                "rest_pass_ms": h["roofline"].get("kernel_ms"), "rest_pass_frac_of_hbm_peak": h["roofline"].get("frac"),
                "rest_rows": h["roofline"].get("filter_rest_rows"), "store_rows": h["config"]["store_rows"], "candidates": h["roofline"].get("filter_candidates"),
                "duplicate_ids_dropped": h["detail"]["merge"]["duplicate_ids_dropped"],
                "results_identical_to_exact_mode": h["detail"].get("exact_mode", {}).get("results_identical_to_filter_mode"),
                "seconds": time.time() - t_h,
                "note": "bench.py --sigma 0.8 --steps 5 --warmup 2 --no-cpu-baseline in a child process after the timed run: the clusters overlap, one probe no "
                        "longer reaches recall 0.9 and the cells prune far less — the headline number is a property of clusterable data"}
        except Exception as e:   # never takes the headline number down with it
            result["detail"]["hard_workload"] = {"error": str(e)[:600], "seconds": time.time() - t_h}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if multi:
        comm.barrier()
        if comm_r is not None:
            comm_r.close()
        comm.close()


def emulate_worlds_on_one_gpu(args, worlds, torch, pygpann, gcomm, dev, device_index, stream, points, clusters, counts, queries, probes, nprobe, router,
                              merge_bound, one_gpu_ms):
    """--emulate-world (see its help).  Per world size W and rank r: the shards the W-GPU run would give rank r (same calls: plan_replicas /
    lpt_owner, mask_probes[_replicated]) as a store of their own, then the rank's share of a step, timed like the real loop (merge of batch
    i on a worker thread under batch i + 1).  Returns the `scale_emulated` object."""
    import concurrent.futures
    nq, d, S, k = args.nq, args.d, args.shards, args.k
    metric = args.metric_id
    out = {"note": "per-rank steps of a W-GPU run executed one rank at a time on ONE GPU (same placement, same kernels, same host code); "
                   "predicted_qps = nq / predicted_ms_per_step (see `model` per world; `..._exchange_in_series` = the rounds 2-3 model: slowest rank + exchange + wire); "
                   "exchange_ms = the step's collectives timed through "
                   "RCCL with ONE rank on this GPU (launch + kernel cost, no wire) + wire_ms_estimate (bytes / 50 GB/s per xGMI link direction, "
                   "ring of W - 1 steps); NOT a measurement on W GPUs",
           "one_gpu_ms_per_step": one_gpu_ms, "worlds": []}
    # the exchange, through the real calls with one rank (RCCL needs one process per device: a single-rank communicator is what one GPU can run)
    exch_ms = None
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        c1 = gcomm.Comm(gcomm.RCCL, rank=0, world=1, addr="127.0.0.1", port=int(os.environ.get("GPANN_COMM_PORT", int(os.environ["MASTER_PORT"]) + 1)), device=device_index)
        n_slots = nq * nprobe * k
        sd = torch.zeros(n_slots, dtype=torch.float32, device=dev)
        si = torch.zeros(n_slots, dtype=torch.int32, device=dev)
        gd, gi = torch.empty_like(sd), torch.empty_like(si)
        scratch = torch.empty(c1.gather_slots_scratch(n_slots), dtype=torch.uint8, device=dev)
        idm = torch.arange(max(args.n, 1), dtype=torch.int32, device=dev)
        pr_mine = torch.zeros((nq, S), dtype=torch.int32, device=dev)
        pr_all = torch.empty_like(pr_mine)
        def exchange():
            c1.broadcast(queries.data_ptr(), queries.numel() * 4, 0, stream)
            c1.allgather(pr_mine.data_ptr(), pr_all.data_ptr(), pr_mine.numel() * 4, stream)
            c1.gather_slots(sd.data_ptr(), si.data_ptr(), n_slots, gd.data_ptr(), gi.data_ptr(), scratch.data_ptr(), id_map=idm.data_ptr(), stream=stream)
        for _ in range(3):
            exchange()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            exchange()
        torch.cuda.synchronize()
        exch_ms = 1000.0 * (time.perf_counter() - t0) / 20
        out["exchange_backend"] = c1.name
        c1.close()
        del sd, si, gd, gi, scratch, idm, pr_mine, pr_all
    except Exception as e:   # the prediction then carries no exchange term and says so
        out["exchange_error"] = str(e)
    out["exchange_ms_one_rank_rccl"] = exch_ms
    est_tasks = np.bincount(probes[:, 0].astype(np.int64), minlength=S).astype(np.float64) + 0.5   # (the real run estimates before routing: nearest shard centroid)
    for W in worlds:
        if args.replica_capacity > 0:
            cap_rows = int(max(args.replica_capacity * counts.sum() / W, 1.25 * counts.max()))
            rep_begin, rep_rank, rank_work, load = gcomm.plan_replicas(counts, est_tasks * counts, W, cap_rows, float(os.environ.get("GPANN_BENCH_REPLICA_IMBALANCE", 1.05)))
            holders = [set(int(x) for x in rep_rank[rep_begin[b]:rep_begin[b + 1]]) for b in range(S)]
        else:
            owner, load = gcomm.lpt_owner(counts, W)
            holders = [{int(owner[b])} for b in range(S)]
        chunk_q = (nq + W - 1) // W
        ranks = []
        only = os.environ.get("GPANN_BENCH_EMULATE_RANKS")   # (profiling aid: "0" or "0,3": only these ranks of every world)
        for r in range(W):
            if only and str(r) not in only.split(","):
                continue
            mine = [b for b in range(S) if r in holders[b]]
            cl = [clusters[b] if b in mine else np.zeros(0, np.uint32) for b in range(S)]
            t_b = time.time()
            sh = pygpann.Shards(points.data_ptr(), cl, metric, device_index, n=points.shape[0], d=d, plain=True)
            if args.scan_mode == "filter":
                sh.build_filter()
            sh.set_scan_mode(1 if args.scan_mode == "filter" else 0)
            build_s = time.time() - t_b
            oi = torch.empty((nq, nprobe, k), dtype=torch.int32, device=dev)
            od = torch.empty((nq, nprobe, k), dtype=torch.float32, device=dev)
            hbuf = [(torch.empty((nq, nprobe, k), dtype=torch.float32).pin_memory(), torch.empty((nq, nprobe, k), dtype=torch.int32).pin_memory()) for _ in range(3)]
            lo, hi = min(nq, r * chunk_q), min(nq, (r + 1) * chunk_q)
            route_bufs = [torch.empty((max(hi - lo, 1), S), dtype=torch.int32).pin_memory().numpy() for _ in range(2)]
            merger = concurrent.futures.ThreadPoolExecutor(1)
            router_pool = concurrent.futures.ThreadPoolExecutor(1)
            r_stream = torch.cuda.Stream(device=dev)
            tasks_rows = [0.0]
            route_ahead = os.environ.get("GPANN_BENCH_ROUTE_AHEAD", "1") != "0"

            def route_share(j, st):   # this rank's share of the routing of batch j
                if hi > lo:
                    router.query_batch(queries.data_ptr() + lo * d * 4, args.search_budget, st, on_device=True, nq=hi - lo, out_probes=route_bufs[j & 1][:hi - lo], want_min_dist=False)
                return True

            scan_ahead = route_ahead and os.environ.get("GPANN_BENCH_SCAN_AHEAD", "1") != "0"
            x_stream = torch.cuda.Stream(device=dev)
            dsets = [(oi, od), (torch.empty_like(oi), torch.empty_like(od)), (torch.empty_like(oi), torch.empty_like(od))]
            emu_overlap = scan_ahead and os.environ.get("GPANN_BENCH_OVERLAP", "1") != "0"
            emu_depth = 3 if emu_overlap else 2     # scans in flight (the real loop: run_steps)
            s_stream = torch.cuda.Stream(device=dev)

            def one(i, routed):
                """like the real loop of a rank with GPANN_BENCH_SCAN_AHEAD=0: batch i + 1 is routed on a second stream by a second thread while batch i is scanned"""
                if routed is not None:
                    routed.result()
                else:
                    route_share(i, stream)
                nxt = router_pool.submit(route_share, i + 1, r_stream.cuda_stream) if route_ahead else None
                lp = (gcomm.mask_probes_replicated(probes, nprobe, rep_begin, rep_rank, r) if args.replica_capacity > 0 else gcomm.mask_probes(probes, nprobe, owner, r))
                sh.scan_buckets_device(queries.data_ptr(), nq, k, lp, nprobe, oi.data_ptr(), od.data_ptr(), stream, timed=True, for_merge=merge_bound)
                hd, hi_ = hbuf[i & 1]
                hd.copy_(od, non_blocking=True)
                hi_.copy_(oi, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                tasks_rows[0] = float(sh.stats()["pairs"])
                return merger.submit(pygpann.merge_dedup, hd.numpy(), hi_.numpy().view(np.uint32), nprobe, k), nxt

            host_t = np.zeros(6)   # GPANN_BENCH_TRACE: where the main thread's time goes (wait for the routing, mask, enqueue, finish, copies, merge wait)

            def begin(i, routed):
                """the real loop's scan_begin: the routing of batch i has arrived (batch i + 1 goes to the router thread), the scan is planned and ENQUEUED (deferred call)"""
                t_a = time.perf_counter()
                routed.result()
                nxt = router_pool.submit(route_share, i + 1, r_stream.cuda_stream)
                t_b = time.perf_counter()
                lp = (gcomm.mask_probes_replicated(probes, nprobe, rep_begin, rep_rank, r) if args.replica_capacity > 0 else gcomm.mask_probes(probes, nprobe, owner, r))
                t_c = time.perf_counter()
                di, dd = dsets[i % emu_depth]
                sh.scan_buckets_device(queries.data_ptr(), nq, k, lp, nprobe, di.data_ptr(), dd.data_ptr(), s_stream.cuda_stream if emu_overlap else stream,
                                       for_merge=merge_bound, defer=True, overlap=emu_overlap)
                host_t[:3] += (t_b - t_a, t_c - t_b, time.perf_counter() - t_c)
                return nxt

            def finish(i):
                """the real loop's scan_finish: the scan of batch i is completed; its lists leave on the third stream (where the real loop runs gather_slots first)"""
                t_a = time.perf_counter()
                sh.finish()
                t_b = time.perf_counter()
                di, dd = dsets[i % emu_depth]
                hd, hi_ = hbuf[i % emu_depth]
                with torch.cuda.stream(x_stream):
                    hd.copy_(dd, non_blocking=True)
                    hi_.copy_(di, non_blocking=True)
                x_stream.synchronize()
                tasks_rows[0] = float(sh.stats()["pairs"])
                host_t[3:5] += (t_b - t_a, time.perf_counter() - t_b)
                return merger.submit(pygpann.merge_dedup, hd.numpy(), hi_.numpy().view(np.uint32), nprobe, k)

            def loop(count):
                if scan_ahead:
                    pending = None
                    ahead = emu_depth - 1
                    nxt = router_pool.submit(route_share, 0, r_stream.cuda_stream)
                    for j in range(min(ahead, count)):
                        nxt = begin(j, nxt)
                    for i in range(count):
                        if pending is not None:
                            t_a = time.perf_counter()
                            pending.result()
                            host_t[5] += time.perf_counter() - t_a
                        if i + ahead < count:
                            nxt = begin(i + ahead, nxt)
                        pending = finish(i)
                    pending.result()
                    nxt.result()
                    torch.cuda.synchronize()
                    return
                pending, routed = None, (router_pool.submit(route_share, 0, r_stream.cuda_stream) if route_ahead else None)
                for i in range(count):
                    f, routed = one(i, routed)
                    if pending is not None:
                        pending.result()
                    pending = f
                pending.result()
                if routed is not None:
                    routed.result()
                torch.cuda.synchronize()

            loop(max(2, args.warmup))
            steps = max(10, args.steps)
            t0 = time.perf_counter()
            loop(steps)
            ms = 1000.0 * (time.perf_counter() - t0) / steps
            st = sh.stats()
            ranks.append({"rank": r, "shards": len(mine), "rows": int(sum(int(counts[b]) for b in mine)), "ms_per_step": ms, "scan_kernels_ms": st["scan_ms"],
                          "rest_pass_ms": st["filter_rest_ms"], "task_rows": tasks_rows[0], "store_build_s": build_s})
            log(f"emulated world {W} rank {r}: {len(mine)} shards, {ranks[-1]['rows']} rows, {ms:.3f} ms/step (scan kernels {st['scan_ms']:.3f})")
            if os.environ.get("GPANN_BENCH_TRACE") == "1" and scan_ahead:
                n_it = max(1, steps + max(2, args.warmup))
                log("   main thread, ms per step: wait for the routing %.3f, mask %.3f, plan + enqueue %.3f, finish %.3f, copies + sync %.3f, wait for the merge %.3f"
                    % tuple(1000.0 * host_t / n_it))
            merger.shutdown()
            router_pool.shutdown()
            sh.close()
            del oi, od, hbuf, dsets
            torch.cuda.empty_cache()
        worst = max(x["ms_per_step"] for x in ranks)
        # wire time of a STEP's collectives on W GPUs: ring all-gather, (W - 1) steps of one rank's block over one xGMI link direction (~50 GB/s achievable):
        # the probe lists and the per-slot (dist, id) lists.  (The query block is broadcast once per block of batches, not per step: the timed loop's
        # queries are resident — rounds 2-3 charged its 5 MB to every step.)
        wire_bytes = (W - 1) * (chunk_q * S * 4 + 2 * nq * nprobe * k * 4)
        wire_ms = 1000.0 * wire_bytes / 50e9
        serial = worst + (exch_ms or 0.0) + wire_ms
        # The pipelined loop (GPANN_BENCH_SCAN_AHEAD, the default) exchanges batch i's lists on a third stream while batch i + 1 is scanned and routes batch
        # i + 2 on a second: the exchange sits beside the scan, not behind it — a step then takes what the slower of the two takes: the rank's own step (GPU
        # and host work, measured above WITH the deferred scans) or its host-side chain plan + enqueue (measured: the one-rank scan call's host time) + exchange.
        host_chain = 0.35 + (exch_ms or 0.0) + wire_ms   # (0.35 ms: mask + plan + staging + ~30 launches of a scan call, bench.py's own trace)
        pipelined = max(worst, host_chain) if os.environ.get("GPANN_BENCH_SCAN_AHEAD", "1") != "0" else serial
        total = pipelined
        out["worlds"].append({"world": W, "ranks": ranks, "max_ms_per_step": worst, "mean_ms_per_step": float(np.mean([x["ms_per_step"] for x in ranks])),
                              "wire_ms_estimate": wire_ms, "predicted_ms_per_step": total, "predicted_qps": 1000.0 * nq / total,
                              "predicted_speedup_over_one_gpu": one_gpu_ms / total,
                              "predicted_ms_per_step_exchange_in_series": serial, "predicted_speedup_exchange_in_series": one_gpu_ms / serial,
                              "model": "max(slowest rank's step, host chain + exchange): the exchange of batch i runs beside the scan of batch i + 1 (third stream)"
                                       if total is pipelined and pipelined != serial else "slowest rank's step + exchange + wire",
                              "replicas": int(sum(len(h) for h in holders) - S)})
        log(f"emulated world {W}: max {worst:.3f} ms/step -> predicted {1000.0 * nq / total:.0f} QPS ({one_gpu_ms / total:.2f}x one GPU)")
    return out


def lookup_traffic(n, d, S, nprobe, kernel, launches_per_step, largest_launch=False, kernel_bytes=None):
    """HBM bytes per STEP of the dominant kernel from the committed PMC pass (profiles/*/…_pmc_traffic.json, FETCH_SIZE doubled per
    the gfx950 correction + WRITE_SIZE), if one exists for exactly this workload; else None.  `launches_per_step` maps a kernel
    name fragment to its launches per step (the cells filter runs as one full-precision launch and two hi-plane launches);
    largest_launch: the biggest single launch of the matching kernels instead (the rest pass among the hi-plane launches).
    kernel_bytes: the bytes this run's dominant launch had to read — a committed pass whose own run (its `dominant_kernel_bytes`) differs
    from that by more than 2 % was taken from another kernel / layout and is NOT reported (None + a note naming the stale file)."""
    import glob
    best, stale = None, None
    import re
    natural = lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", p)]   # ... v9 < v10: the newest pass wins
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*_pmc_traffic.json")), key=natural):
        try:
            j = json.load(open(path))
            w = j["workload"]
            if (w["n"], w["d"], w["shards"], w["nprobe"]) != (n, d, S, nprobe):
                continue
            if kernel_bytes is not None:
                have = j.get("dominant_kernel_bytes")
                if not have or abs(float(have) / kernel_bytes - 1.0) > 0.02:
                    stale = os.path.relpath(path, ROOT)
                    continue
            total, found = 0.0, False
            for name, v in j["kernels"].items():
                for frag, per_step in launches_per_step.items():
                    if kernel in name and frag in name:
                        if largest_launch:
                            total = max(total, max(2048.0 * f + 1024.0 * w for f, w in zip(v["fetch_kib"], v["write_kib"])))
                        else:
                            total += float(v["hbm_bytes_per_launch"]) * per_step
                        found = True
            if found:
                best = (total, os.path.relpath(path, ROOT))
        except Exception:
            continue
    if best:
        return best
    return (None, f"stale: {stale} was measured on a launch whose bytes differ by more than 2 % from this run's" if stale else None)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, shards, tree, queries, probes, nprobe, kth, counts, gpu_dist, gpu_ids):
    """Time the strict CPU oracle (oracle/liboracle.so, `kind: port`) on the host cores over a bounded sample of the same workload —
    the queries whose first `nprobe` probes fall into a few shards, scanned against those shards' real rows — and compare its
    neighbours with the GPU's for those queries (`results_match_gpu`).  Threads: all hardware threads, and min(32, hw) beside it
    (the reference caps its query loops at 32 workers, routes.cpp:194 / shard_searches.cpp:48)."""
    import concurrent.futures
    import pyoracle
    orc = pyoracle.Oracle("strict")
    cores = os.cpu_count() or 1
    S, k, d = args.shards, args.k, args.d
    want_queries = min(len(probes), int(os.environ.get("GPANN_BENCH_CPU_QUERIES", 2560)))
    # shards in a fixed pseudo-random order until the sample holds enough queries (bounded host memory: <= 20 shards / 24 GB of rows).
    # (Until round 3 the shards were taken in ASCENDING SIZE: the sampled queries then scan smaller shards than the batch's average — 1.96M
    # rows per query against 2.98M at NS — which flattered the CPU by ~1.5x.  A random choice is size-representative; both averages are
    # reported — sample_rows_per_query, batch_rows_per_query — and value_scaled_to_batch rescales by their ratio.)
    first = probes[:, :nprobe]
    chosen, sel_q, pairs = [], np.zeros(len(first), bool), 0
    for b in np.random.default_rng(12345).permutation(len(counts)):
        if (sum(int(counts[x]) for x in chosen) + int(counts[b])) * d * 4 > 24e9:
            continue
        chosen.append(int(b))
        sel_q = np.all(np.isin(first, chosen), axis=1)
        if int(sel_q.sum()) >= want_queries or len(chosen) >= 20:
            break
    qidx = np.flatnonzero(sel_q)[:want_queries]
    if len(qidx) == 0:
        return {"value": None, "unit": "queries/s", "cores": cores, "kind": "port", "sample": "no query falls entirely into the sampled shards"}
    pairs = int(sum(int(counts[first[q]].sum()) for q in qidx))
    valid = first >= 0
    batch_rows_per_query = float(np.where(valid, counts[np.where(valid, first, 0)], 0).sum()) / max(1, len(first))
    sample_rows_per_query = pairs / len(qidx)
    # the sampled shards' rows on the host: the pages are first touched by many threads, each taking every cores-th 2 MiB block, so
    # they are spread over the host's memory nodes (a single-threaded concatenate puts the whole sample on one node)
    offs = shards.offsets()
    new_id = {b: i for i, b in enumerate(chosen)}
    sizes = np.array([int(offs[b + 1] - offs[b]) for b in chosen], np.int64)
    total_rows = int(sizes.sum())
    clustered = np.empty((total_rows, d), np.float32)
    flat = clustered.reshape(-1).view(np.uint8)
    block = 2 << 20
    n_blocks = (flat.size + block - 1) // block
    touchers = min(cores, 64)

    def touch(i):
        for blk in range(i, n_blocks, touchers):
            flat[blk * block:min(flat.size, (blk + 1) * block)] = 0
    with concurrent.futures.ThreadPoolExecutor(touchers) as ex:
        list(ex.map(touch, range(touchers)))
    ivf_perm = np.empty(total_rows, np.uint32)
    pos = 0
    for b, m in zip(chosen, sizes):
        perm_b, rows_b = shards.export_range(int(offs[b]), int(m))
        clustered[pos:pos + m] = rows_b
        ivf_perm[pos:pos + m] = perm_b
        pos += int(m)
        del rows_b
    ivf_offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    q_host = np.ascontiguousarray(queries.cpu().numpy()[qidx])
    ivf = (clustered, ivf_offsets, ivf_perm)
    pr = np.vectorize(new_id.get)(first[qidx]).astype(np.int32)
    offsets_all = np.arange(len(qidx) + 1, dtype=np.uint64) * (nprobe * k)

    def run(o, qsel, threads):
        """route + scan + merge of the sampled queries `qsel` on `threads` host threads; returns (seconds, dist, ids)"""
        otree = o.kmtr_from_flat(tree, args.metric_id)
        t0 = time.perf_counter()
        otree.query_batch(q_host[qsel], args.search_budget, nthreads=threads)
        od, oi, _ = o.ivf_query_bucket_batch(ivf, q_host[qsel], k, pr[qsel], nprobe, args.metric_id, nthreads=threads)
        md, mi, _ = o.dedup_neighbors(od.reshape(-1), oi.reshape(-1), offsets_all[:len(qsel) + 1], k)
        return time.perf_counter() - t0, md, mi

    everything = np.arange(len(qidx))
    # the reference's own thread cap (routes.cpp:194: 32 workers) on the WHOLE sample — on a 256-thread host it is the faster setting (the per-query
    # streams of 256 threads thrash the memory system) and it is the leg whose neighbours are compared with the GPU's; all hardware threads, and the
    # flags upstream ships, on an eighth of the sample each (until round 3 they ran on all of it: three minutes of a default run)
    t32 = min(32, cores)
    dt32, cd, ci = run(orc, everything, t32)
    # parity of the timed GPU path with the CPU path on the sample: ids and distance bits
    cd, ci = np.asarray(cd).reshape(len(qidx), k), np.asarray(ci).reshape(len(qidx), k)
    match = bool(np.array_equal(ci, gpu_ids[qidx]) and np.array_equal(cd.view(np.uint32), np.ascontiguousarray(gpu_dist[qidx]).view(np.uint32)))
    sub = everything[:max(1, len(qidx) // 8)]
    dt, _, _ = run(orc, sub, cores)
    # the same sub-sample with the flags upstream ships (-O3 -march=native -ffast-math, CMakeLists.txt:15): not the parity oracle
    # (results may differ in the last bits), reported so the strict flags do not flatter the GPU
    fast = fast32 = None
    try:
        o_fast = pyoracle.Oracle("refflags")
        dtf, _, _ = run(o_fast, sub, cores)
        fast = len(sub) / dtf
        # ... and on the reference's 32 workers over the WHOLE sample: the cell next to value_32_threads (same queries, same threads, upstream's flags)
        dtf32, _, _ = run(o_fast, everything, t32)
        fast32 = len(qidx) / dtf32
    except Exception:
        pass
    # `value` = the better of the two thread counts (on a 256-thread host the reference's own cap of 32 workers is the faster setting: the
    # per-query streams of 256 threads thrash the memory system); both are reported
    v_all, v_32 = len(sub) / dt, len(qidx) / dt32
    return {"value": max(v_all, v_32), "unit": "queries/s", "cores": cores if v_all >= v_32 else t32, "kind": "port", "cpu_model": cpu_model(),
            "value_all_threads": v_all, "threads_all": cores,
            "value_reference_flags": fast, "value_32_threads": v_32, "threads_32": t32, "value_reference_flags_32_threads": fast32, "results_match_gpu": match,
            "sample_queries": int(len(qidx)), "sample_pairs": pairs, "sample_GBps": pairs * d * 4 / dt32 / 1e9,
            "sample_rows_per_query": sample_rows_per_query, "batch_rows_per_query": batch_rows_per_query,
            "value_scaled_to_batch": max(v_all, v_32) * sample_rows_per_query / max(batch_rows_per_query, 1.0),
            "sample_choice": "shards in a fixed pseudo-random order (seed 12345) until enough queries fall entirely into them: size-representative; "
                             "value_scaled_to_batch = value x sample_rows_per_query / batch_rows_per_query (the CPU path's time is proportional to the rows it streams)",
            "sample": f"{len(qidx)} of {len(first)} queries (those routed entirely into shards {chosen}), same nprobe={nprobe}; "
                      f"{pairs:.3g} (query,row) pairs = {pairs * d * 4 / 1e9:.0f} GB streamed in {dt32:.1f}s on {t32} threads — the reference's own cap — (rows "
                      f"first-touched by {touchers} threads in interleaved 2 MiB blocks); strict build (-O2, no fast-math/FMA/AVX); value_all_threads: {len(sub)} of "
                      f"them on {cores} threads in {dt:.1f}s; value_reference_flags: those {len(sub)} with -O3 -march=native -ffast-math on {cores} threads; "
                      f"value_reference_flags_32_threads: all {len(qidx)} with those flags on {t32} threads (cell for cell beside value_32_threads)",
            "seconds": dt32}


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        # a rank that fails leaves without waiting for collectives the others may be in (ncclCommDestroy could block on them); closing the
        # bootstrap connections is what makes their gpann_comm_wait return
        for t in _live_turns:   # (a route-ahead job waiting for its place in the schedule leaves instead of waiting for ever)
            t.abort()
        for c in _live_comm:
            c.abort()
        raise
