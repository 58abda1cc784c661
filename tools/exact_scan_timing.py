#!/usr/bin/env python3
"""Kernel time of the exact (VALU) scan — ScanTileKernel — on a synthetic store, for comparing builds of the library (GPANN_HIP_LIBRARY=...):
    python3 tools/exact_scan_timing.py [rows] [d]
prints ms per call (HIP events inside the library) and the fp32 lane-operations per second it corresponds to (3 per pair and dimension)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gp-ann_amd")]
import torch  # noqa: E402
import pygpann  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
S, nq, k = 4, 2048, 10
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
pts = torch.empty((n, d), dtype=torch.float32, device=dev)
pygpann.synth_rows_device(pts.data_ptr(), 555, 7, 1024, 0.35, 0, n, d, st)
q = torch.empty((nq, d), dtype=torch.float32, device=dev)
pygpann.synth_rows_device(q.data_ptr(), 556, 7, 1024, 0.35, 0, nq, d, st)
torch.cuda.synchronize()
clusters = [np.arange(b * (n // S), (b + 1) * (n // S), dtype=np.uint32) for b in range(S)]
sh = pygpann.Shards(pts.data_ptr(), clusters, 0, 0, n=n, d=d, plain=True)
oi = torch.empty((nq, 1, k), dtype=torch.int32, device=dev)
od = torch.empty((nq, 1, k), dtype=torch.float32, device=dev)
probes = (np.arange(nq, dtype=np.int32) % S).reshape(nq, 1)
ms = []
for rep in range(5):
    ms.append(sh.scan_buckets_device(q.data_ptr(), nq, k, probes, 1, oi.data_ptr(), od.data_ptr(), st, timed=True))
torch.cuda.synchronize()
s = sh.stats()
best = min(ms[1:])
print(f"[exact scan] {pygpann.LIB_PATH}: {n} x {d}, {nq} queries: {best:.3f} ms per call (runs {[round(x, 2) for x in ms]}), "
      f"{3.0 * s['pairs'] * d / (best * 1e-3) / 1e12:.2f} T lane-ops/s, checksum {int(oi.to(torch.int64).sum().item())}")
