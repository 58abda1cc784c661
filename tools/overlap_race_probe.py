#!/usr/bin/env python3
"""Stress probe for GPANN_SCAN_OVERLAP (include/gpann.h): the pipeline of tests/test_gpu_filter.py::test_overlapped_stream_of_batches_many_cells
repeated many times on one store, every batch's lists compared bit for bit with the same call made alone (no flag, one stream, nothing beside it).
A mismatch is printed entry by entry (query, probe slot, position: got / want) and classified — a list that misses rows the plain call found, a list
cut at a different merge bound, foreign rows — so that a rare event leaves something to reason about.

    python3 tools/overlap_race_probe.py --rest-items 2 --d 64 --repeats 200

Needs a GPU; data is made with numpy (no oracle: the reference is the product's own plain call)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gp-ann_amd"))


def mixture(rng, centers, n, sigma=0.35):
    pick = rng.integers(0, centers.shape[0], size=n)
    return (centers[pick] + sigma * rng.standard_normal((n, centers.shape[1]))).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rest-items", default="2")
    ap.add_argument("--front-cus", default="64")
    ap.add_argument("--rest-cus", default="128")
    ap.add_argument("--cell-rows", default="1500")
    ap.add_argument("--seg-tiles", default="8")
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--n", type=int, default=150000)
    ap.add_argument("--nq", type=int, default=700)
    ap.add_argument("--shards", type=int, default=3)
    ap.add_argument("--nprobe", type=int, default=2)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--repeats", type=int, default=100)
    ap.add_argument("--seconds", type=float, default=0.0, help="run pipelines until this much time has passed (overrides --repeats)")
    ap.add_argument("--depth", type=int, default=2, help="calls begun ahead of the one being finished (the handle has three call slots)")
    ap.add_argument("--no-overlap", action="store_true", help="deferred calls without GPANN_SCAN_OVERLAP (the caller's stream only)")
    ap.add_argument("--no-for-merge", action="store_true", help="complete per-slot lists instead of GPANN_SCAN_FOR_MERGE")
    ap.add_argument("--set", action="append", default=[], help="NAME=VALUE: a further GPANN_* option for the store")
    args = ap.parse_args()

    os.environ["GPANN_CELL_ROWS"] = args.cell_rows
    os.environ["GPANN_FILTER_SEG_TILES"] = args.seg_tiles
    os.environ["GPANN_OVERLAP_REST_ITEMS"] = args.rest_items
    os.environ["GPANN_OVERLAP_FRONT_CUS"] = args.front_cus
    os.environ["GPANN_OVERLAP_REST_CUS"] = args.rest_cus
    for kv in args.set:
        name, value = kv.split("=", 1)
        os.environ[name] = value
    import pygpann as gp
    gp.lib()
    hip = C.CDLL("libamdhip64.so")

    def ok(rc):
        assert rc == 0, f"hip error {rc}"

    rng = np.random.default_rng(101 + args.d)
    d, n, S, nq, k, nprobe = args.d, args.n, args.shards, args.nq, 10, args.nprobe
    centers = rng.standard_normal((48, d)).astype(np.float32)
    pts = mixture(rng, centers, n)
    if args.metric == 1:
        pts /= np.maximum(1.0, np.linalg.norm(pts, axis=1, keepdims=True)).astype(np.float32)
    part = rng.integers(0, S, size=n)
    clusters = []
    for b in range(S):
        c = np.flatnonzero(part == b).astype(np.uint32)
        extra = rng.choice(n, size=n // (10 * S), replace=False).astype(np.uint32)
        c = np.unique(np.concatenate([c, extra]))
        rng.shuffle(c)
        clusters.append(c)
    batches = [mixture(rng, centers, nq) for _ in range(args.batches)]
    probes = np.stack([rng.permutation(S)[:nprobe] for _ in range(nq)]).astype(np.int32)
    shards = gp.Shards(pts, clusters, args.metric, 0)
    shards.set_scan_mode(1)
    st = C.c_void_p()
    ok(hip.hipStreamCreate(C.byref(st)))
    nb = nq * nprobe * k * 4
    q_dev, h_ids, h_dist = [], [], []
    for q in batches:
        p = C.c_void_p()
        ok(hip.hipMalloc(C.byref(p), C.c_size_t(q.nbytes)))
        ok(hip.hipMemcpy(p, q.ctypes.data_as(C.c_void_p), C.c_size_t(q.nbytes), 1))
        q_dev.append(p)
        for lst in (h_ids, h_dist):
            h = C.c_void_p()
            ok(hip.hipHostMalloc(C.byref(h), C.c_size_t(nb), 0))
            lst.append(h)
    ok(hip.hipDeviceSynchronize())

    def host_view(h):
        return np.ctypeslib.as_array(C.cast(h, C.POINTER(C.c_uint32)), shape=(nq * nprobe * k,)).reshape(nq, nprobe, k).copy()

    want = []
    for q in batches:
        ids, dist, _ = shards.scan_buckets(q, k, probes, nprobe, for_merge=not args.no_for_merge)
        ids2, dist2, _ = shards.scan_buckets(q, k, probes, nprobe, for_merge=not args.no_for_merge)
        assert np.array_equal(ids, ids2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32)), "the plain call is not repeatable"
        want.append((ids.copy(), dist.view(np.uint32).copy()))
    print(f"[probe] store {n} x {d}, metric {args.metric}, {S} shards, {args.batches} batches of {nq} queries, nprobe {nprobe}; rest_items {args.rest_items} "
          f"front/rest CUs {args.front_cus}/{args.rest_cus} overlap {not args.no_overlap} depth {args.depth}", flush=True)

    def begin(i):
        shards.scan_buckets_device(q_dev[i].value, nq, k, probes, nprobe, h_ids[i].value, h_dist[i].value, st.value, defer=True, host_out=True, for_merge=not args.no_for_merge,
                                   overlap=not args.no_overlap)

    events = 0
    t0 = time.time()
    rep = -1
    overlapped_calls = 0
    while True:
        rep += 1
        if (time.time() - t0 >= args.seconds) if args.seconds > 0 else (rep >= args.repeats):
            break
        for h in h_ids + h_dist:
            C.memset(h, 0xA5, nb)
        nbat = len(batches)
        for j in range(min(args.depth, nbat)):
            begin(j)
        stats = []
        for i in range(nbat):
            if i + args.depth < nbat:
                begin(i + args.depth)
            shards.finish()
            stats.append(shards.stats())
            overlapped_calls += int(stats[-1].get("overlapped", 0))
        for i in range(nbat):
            got_i, got_d = host_view(h_ids[i]), host_view(h_dist[i])
            w_i, w_d = want[i]
            if np.array_equal(got_i, w_i) and np.array_equal(got_d, w_d):
                continue
            events += 1
            bad = np.argwhere((got_i != w_i) | (got_d != w_d))
            lists = sorted({(int(a), int(b)) for a, b, _ in bad})
            print(f"[probe] EVENT repeat {rep} batch {i} (slot {i % 3}): {len(bad)} entries in {len(lists)} lists differ; stats filtered={stats[i]['filtered']} "
                  f"candidates={stats[i].get('filter_candidates')} dirty={stats[i].get('filter_dirty_tasks')}", flush=True)
            for (qq, jj) in lists[:6]:
                gd, wd = got_d[qq, jj].view(np.float32), w_d[qq, jj].view(np.float32)
                gi, wi = got_i[qq, jj], w_i[qq, jj]
                missing = [int(x) for x in wi if x != 0xFFFFFFFF and x not in set(gi.tolist())]
                foreign = [int(x) for x in gi if x != 0xFFFFFFFF and x not in set(wi.tolist())]
                print(f"    query {qq} slot {jj} (shard {probes[qq, jj]}): missing {missing} foreign {foreign}")
                print(f"      got  ids {gi.tolist()}\n      got  d   {[float(x) for x in gd]}\n      want ids {wi.tolist()}\n      want d   {[float(x) for x in wd]}")
            # is it repeatable?  the same batch alone, with the flag
            C.memset(h_ids[i], 0xA5, nb)
            C.memset(h_dist[i], 0xA5, nb)
            begin(i)
            shards.finish()
            again = np.array_equal(host_view(h_ids[i]), w_i) and np.array_equal(host_view(h_dist[i]), w_d)
            print(f"    the same batch once more, alone with the flag: {'equal to the plain call' if again else 'DIFFERS again'}", flush=True)
    print(f"[probe] {rep} repeats x {len(batches)} batches in {time.time() - t0:.1f} s ({overlapped_calls} calls ran overlapped): {events} events", flush=True)
    shards.close()
    return 1 if events else 0


if __name__ == "__main__":
    sys.exit(main())
