#!/usr/bin/env python3
"""The C++ driver (gp-ann_amd/bin/DistributedBench) on a synthetic file set: the reference's one-batch protocol ("End-to-end time") next to the
pipelined stream of batches (GPANN_DIST_PIPELINE) — ranks share GPU 0 over the TCP backend unless --rccl (one rank).
    python tools/dist_stream_demo.py [--n 2000000] [--d 64] [--nq 10000] [--shards 12] [--ranks 2] [--batches 10] [--rccl]"""
import argparse
import os
import subprocess
import sys
import tempfile
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--d", type=int, default=64)
ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--shards", type=int, default=12)
ap.add_argument("--ranks", type=int, default=2)
ap.add_argument("--batches", type=int, default=10)
ap.add_argument("--rccl", action="store_true")
a = ap.parse_args()
rng = np.random.default_rng(3)
centres = rng.standard_normal((256, a.d)).astype(np.float32)
pts = centres[rng.integers(0, 256, a.n)] + 0.35 * rng.standard_normal((a.n, a.d)).astype(np.float32)
queries = centres[rng.integers(0, 256, a.nq)] + 0.35 * rng.standard_normal((a.nq, a.d)).astype(np.float32)
heads = centres[rng.choice(256, a.shards, replace=False)]
part = np.empty(a.n, np.int32)
for i in range(0, a.n, 1 << 18):
    x = pts[i:i + (1 << 18)]
    part[i:i + (1 << 18)] = np.argmin((x * x).sum(1, keepdims=True) - 2.0 * x @ heads.T + (heads * heads).sum(1), axis=1)
tmp = tempfile.mkdtemp(prefix="gpann_dist_demo_")
f = {k: os.path.join(tmp, v) for k, v in dict(points="base.fbin", queries="query.fbin", gt="none.bin", part="base.partition", router="router.tree").items()}
for path, arr in ((f["points"], pts), (f["queries"], queries)):
    with open(path, "wb") as fh:
        np.array(arr.shape, np.uint32).tofile(fh)
        arr.tofile(fh)
np.savetxt(f["part"], part, fmt="%d")
world = 1 if a.rccl else a.ranks
port = 29700 + os.getpid() % 200
procs = []
for rank in range(world):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPANN_COMM_BACKEND="rccl" if a.rccl else "tcp",
               GPANN_DIST_DEVICE="0", GPANN_DIST_REPS="5", GPANN_DIST_PIPELINE=str(a.batches), GPANN_DIST_PROBES="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen([os.path.join(ROOT, "gp-ann_amd", "bin", "DistributedBench"), f["points"], f["queries"], f["gt"], "10", f["part"], f["router"]], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
rc = 0
for rank, p in enumerate(procs):
    so, se = p.communicate(timeout=900)
    rc |= p.returncode
    if rank == 0 or p.returncode:
        print(so[-1200:])
        if p.returncode:
            print(se[-2000:], file=sys.stderr)
print(f"{a.n} x {a.d} points, {a.nq} queries, {a.shards} shards, {world} rank(s) on GPU 0, backend {'rccl' if a.rccl else 'tcp'}")
sys.exit(rc)
