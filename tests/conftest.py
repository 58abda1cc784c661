import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gp-ann_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Every router call of the test session (and of the driver binaries the tests start: they inherit the environment) runs with the
# library's checks armed: GPANN_ROUTER_VERIFY compares every minimum of the matrix-core bound path with the canonical evaluation of all
# (query, centroid) pairs, GPANN_ROUTER_STRICT turns a fired self-check into GPANN_ERR_INTERNAL instead of a silent re-evaluation.
os.environ.setdefault("GPANN_ROUTER_VERIFY", "1")
os.environ.setdefault("GPANN_ROUTER_STRICT", "1")
os.environ.setdefault("GPANN_VERIFY_REPORT", os.path.join(ROOT, "gpurun_out", "router_verify_report.txt"))
# The test stores are small: by the library's default rule (GPANN_FILTER_SMALL_SHARD_ROWS: shards that average fewer than 262144 rows run without the
# shard-sample and head passes) they would never run the five-pass pipeline of the large stores the benchmark measures.  The session therefore switches
# the rule off and the tests that are ABOUT it (tests/test_gpu_small_shards.py, the odd seeds of the fuzz test, the C1 / C2 shapes, the bench-line tests)
# take the library's default back.
os.environ.setdefault("GPANN_FILTER_SMALL_SHARD_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)


@pytest.fixture(scope="session")
def orc():
    """The strict CPU oracle (oracle/liboracle.so) — the checker, never the thing under test in -m gpu tests."""
    import pyoracle
    pyoracle.build()
    return pyoracle.Oracle("strict")


@pytest.fixture(scope="session")
def gp():
    """The product binding; GPU tests fail loudly (no fallback) if the HIP library is missing."""
    import pygpann
    pygpann.lib()
    return pygpann


def free_port(span=2):
    """`span` consecutive free TCP ports on 127.0.0.1 BELOW the kernel's ephemeral range; returns the first.  (A port from bind(0) lies inside
    the ephemeral range: a rank that connects to it before rank 0 listens can be handed that very port as its SOURCE port and connect to itself —
    rank 0 then fails with "Address already in use" and the others wait for their timeouts: 2 of 110 three-rank runs on one box in round 4.)"""
    import random
    import socket
    lo_eph = 32768
    try:
        lo_eph = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError):
        pass
    hi = max(12000, min(lo_eph, 32768) - span - 1)
    rng = random.Random(os.getpid() * 1000003 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        p = rng.randrange(11000, hi)
        socks = []
        try:
            for i in range(span):
                s = socket.socket()
                socks.append(s)
                s.bind(("127.0.0.1", p + i))
            return p
        except OSError:
            continue
        finally:
            for s in socks:
                s.close()
    raise RuntimeError("no free ports")


def make_mixture(orc, n, d, seed=555, n_centers=16, sigma=0.35, center_seed=7):
    return orc.synth_rows(seed, center_seed, n_centers, sigma, 0, n, d, nthreads=4)


def random_clusters(rng, n, S, overlap=0.0, empty_shard=False):
    """Random shard membership lists in random order (ids may repeat across shards when overlap > 0)."""
    part = rng.integers(0, S - (1 if empty_shard else 0), size=n)
    clusters = [list(np.flatnonzero(part == b)) for b in range(S)]
    if overlap > 0:
        extra = rng.choice(n, size=int(n * overlap), replace=False)
        for i in extra:
            b = int(rng.integers(0, S - (1 if empty_shard else 0)))
            if part[i] != b:
                clusters[b].append(int(i))
    out = []
    for c in clusters:
        c = np.asarray(c, dtype=np.uint32)
        rng.shuffle(c)
        out.append(c)
    return out
