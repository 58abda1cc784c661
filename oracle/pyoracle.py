"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never
from the product package (gp-ann_amd/).  See oracle/gpann_oracle.h for what each call restates (reference file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
INF = np.float32(np.inf)
NO_ID = np.uint32(0xFFFFFFFF)

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile liboracle.so / liboracle_refflags.so (and _ref/ref_core when /root/reference is present)."""
    targets = ["all"]
    if os.path.isdir(os.environ.get("GPANN_REFERENCE", "/root/reference")):
        targets.append("ref")
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-C", _HERE, "-j4"] + targets, stdout=subprocess.DEVNULL)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def clusters_to_csr(clusters):
    """list of id arrays -> (ids_concat u32, offsets u64)"""
    offs = np.zeros(len(clusters) + 1, dtype=np.uint64)
    for b, c in enumerate(clusters):
        offs[b + 1] = offs[b] + len(c)
    ids = np.concatenate([np.asarray(c, dtype=np.uint32) for c in clusters]) if len(clusters) else np.zeros(0, np.uint32)
    return _u32(ids), offs


class Oracle:
    def __init__(self, variant="strict"):
        name = "liboracle.so" if variant == "strict" else "liboracle_refflags.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [_f32p, _f32p, C.c_uint32, C.c_int]
        L.orc_distance_many.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_int, _f32p, _f32p]
        L.orc_topn.restype = C.c_uint32
        L.orc_topn.argtypes = [_f32p, _u32p, C.c_uint64, C.c_uint32, _f32p, _u32p]
        L.orc_ivf_build.argtypes = [_f32p, C.c_uint32, _u32p, _u64p, C.c_uint32, _f32p, _i32p, _u32p]
        L.orc_ivf_query_bucket.restype = C.c_uint32
        L.orc_ivf_query_bucket.argtypes = [_f32p, C.c_uint32, C.c_int, _i32p, _u32p, _f32p, C.c_uint32, C.c_int32, _f32p, _u32p, _u32p]
        L.orc_ivf_query.restype = C.c_uint32
        L.orc_ivf_query.argtypes = [_f32p, C.c_uint32, C.c_int, _i32p, _u32p, _f32p, C.c_uint32, _i32p, C.c_uint32, _f32p, _u32p]
        L.orc_ivf_query_bucket_batch.argtypes = [_f32p, C.c_uint32, C.c_int, _i32p, _u32p, _f32p, C.c_uint32, C.c_uint32, _i32p,
                                                 C.c_uint32, C.c_uint32, _f32p, _u32p, _u32p, C.c_int]
        L.orc_ground_truth.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_int, _f32p, C.c_uint32, C.c_uint32, _f32p, _u32p, C.c_int]
        L.orc_distance_to_kth.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_int, _f32p, C.c_uint32, C.c_uint32, _f32p, C.c_int]
        L.orc_convert_ground_truth.restype = C.c_uint64
        L.orc_convert_ground_truth.argtypes = [_f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _f32p, C.c_uint32, C.c_int, _f32p, _f32p]
        L.orc_recall.restype = C.c_double
        L.orc_recall.argtypes = [_f32p, _u64p, C.c_uint32, _f32p, C.c_uint32]
        L.orc_dedup_neighbors.argtypes = [_f32p, _u32p, _u64p, C.c_uint32, C.c_uint32, _f32p, _u32p, _u32p]
        L.orc_kmeans.restype = C.c_uint32
        L.orc_kmeans.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_int, _i32p, _f32p, C.c_int]
        L.orc_kmtr_train.restype = C.c_void_p
        L.orc_kmtr_train.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_int, _u32p, _u64p, C.c_uint32, C.c_uint64, C.c_uint64,
                                     C.c_int64, C.c_int]
        L.orc_kmtr_from_flat.restype = C.c_void_p
        L.orc_kmtr_from_flat.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _i32p, _u32p, _u32p, _u32p, _u32p, _f32p]
        L.orc_kmtr_sizes.argtypes = [C.c_void_p, _u32p, _u32p, _u32p, _u64p]
        L.orc_kmtr_export.argtypes = [C.c_void_p, _i32p, _u32p, _u32p, _u32p, _u32p, _f32p]
        L.orc_kmtr_extract_partition.argtypes = [C.c_void_p, _i32p]
        L.orc_kmtr_query.argtypes = [C.c_void_p, _f32p, C.c_int, _i32p, _f32p]
        L.orc_kmtr_query_batch.argtypes = [C.c_void_p, _f32p, C.c_uint32, C.c_int, _i32p, _f32p, C.c_int]
        L.orc_kmtr_frequency_query.restype = C.c_uint32
        L.orc_kmtr_frequency_query.argtypes = [C.c_void_p, _f32p, C.c_int, C.c_int, _f32p, _f32p, _u32p]
        L.orc_frequency_probes.argtypes = [_f32p, C.c_uint32, _u32p, C.c_uint32, C.c_int, _i32p]
        L.orc_kmtr_free.argtypes = [C.c_void_p]
        L.orc_partition_to_clusters.restype = C.c_uint32
        L.orc_partition_to_clusters.argtypes = [_i32p, C.c_uint64, _u32p, _u64p]
        L.orc_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, _f32p, C.c_int]

    # ---- distance / topn -------------------------------------------------------------------------------------
    def distance(self, p, q, metric=0):
        p, q = _f32(p), _f32(q)
        return np.float32(self.lib.orc_distance(_ptr(p, _f32p), _ptr(q, _f32p), p.shape[0], metric))

    def distance_many(self, points, q, metric=0):
        points, q = _f32(points), _f32(q)
        out = np.empty(points.shape[0], np.float32)
        self.lib.orc_distance_many(_ptr(points, _f32p), points.shape[0], points.shape[1], metric, _ptr(q, _f32p), _ptr(out, _f32p))
        return out

    def topn(self, dists, ids, n):
        dists, ids = _f32(dists), _u32(ids)
        od, oi = np.empty(n, np.float32), np.empty(n, np.uint32)
        c = self.lib.orc_topn(_ptr(dists, _f32p), _ptr(ids, _u32p), len(dists), n, _ptr(od, _f32p), _ptr(oi, _u32p))
        return od[:c], oi[:c]

    # ---- inverted index ----------------------------------------------------------------------------------------
    def ivf_build(self, points, clusters):
        points = _f32(points)
        ids, offs = clusters_to_csr(clusters)
        S, d = len(clusters), points.shape[1]
        total = int(offs[-1])
        clustered = np.empty((total, d), np.float32)
        offsets = np.empty(S + 1, np.int32)
        perm = np.empty(total, np.uint32)
        self.lib.orc_ivf_build(_ptr(points, _f32p), d, _ptr(ids, _u32p), _ptr(offs, _u64p), S, _ptr(clustered, _f32p),
                               _ptr(offsets, _i32p), _ptr(perm, _u32p))
        return clustered, offsets, perm

    def ivf_query_bucket(self, ivf, Q, k, bucket, metric=0):
        clustered, offsets, perm = ivf
        Q = _f32(Q)
        od, oi, orow = np.empty(k, np.float32), np.empty(k, np.uint32), np.empty(k, np.uint32)
        c = self.lib.orc_ivf_query_bucket(_ptr(clustered, _f32p), clustered.shape[1], metric, _ptr(offsets, _i32p), _ptr(perm, _u32p),
                                          _ptr(Q, _f32p), k, bucket, _ptr(od, _f32p), _ptr(oi, _u32p), _ptr(orow, _u32p))
        return od[:c], oi[:c], orow[:c]

    def ivf_query(self, ivf, Q, k, probes, nprobe, metric=0):
        clustered, offsets, perm = ivf
        Q, probes = _f32(Q), _i32(probes)
        od, oi = np.empty(k, np.float32), np.empty(k, np.uint32)
        c = self.lib.orc_ivf_query(_ptr(clustered, _f32p), clustered.shape[1], metric, _ptr(offsets, _i32p), _ptr(perm, _u32p),
                                   _ptr(Q, _f32p), k, _ptr(probes, _i32p), nprobe, _ptr(od, _f32p), _ptr(oi, _u32p))
        return od[:c], oi[:c]

    def ivf_query_bucket_batch(self, ivf, queries, k, probes, nprobe, metric=0, nthreads=1):
        clustered, offsets, perm = ivf
        queries, probes = _f32(queries), _i32(probes)
        nq = queries.shape[0]
        stride = probes.shape[1]
        od = np.empty((nq, nprobe, k), np.float32)
        oi = np.empty((nq, nprobe, k), np.uint32)
        oc = np.empty((nq, nprobe), np.uint32)
        self.lib.orc_ivf_query_bucket_batch(_ptr(clustered, _f32p), clustered.shape[1], metric, _ptr(offsets, _i32p), _ptr(perm, _u32p),
                                            _ptr(queries, _f32p), nq, k, _ptr(probes, _i32p), stride, nprobe, _ptr(od, _f32p),
                                            _ptr(oi, _u32p), _ptr(oc, _u32p), nthreads)
        return od, oi, oc

    # ---- ground truth / recall ---------------------------------------------------------------------------------
    def ground_truth(self, points, queries, k, metric=0, nthreads=1):
        points, queries = _f32(points), _f32(queries)
        nq = queries.shape[0]
        od, oi = np.empty((nq, k), np.float32), np.empty((nq, k), np.uint32)
        self.lib.orc_ground_truth(_ptr(points, _f32p), points.shape[0], points.shape[1], metric, _ptr(queries, _f32p), nq, k,
                                  _ptr(od, _f32p), _ptr(oi, _u32p), nthreads)
        return od, oi

    def distance_to_kth(self, points, queries, k, metric=0, nthreads=1):
        points, queries = _f32(points), _f32(queries)
        out = np.empty(queries.shape[0], np.float32)
        self.lib.orc_distance_to_kth(_ptr(points, _f32p), points.shape[0], points.shape[1], metric, _ptr(queries, _f32p),
                                     queries.shape[0], k, _ptr(out, _f32p), nthreads)
        return out

    def convert_ground_truth(self, gt_dist, gt_ids, k, points, queries, metric=0):
        gt_dist, gt_ids = _f32(gt_dist).copy(), _u32(gt_ids).copy()
        points, queries = _f32(points), _f32(queries)
        nq, ks = gt_dist.shape
        kth = np.empty(nq, np.float32)
        wrong = self.lib.orc_convert_ground_truth(_ptr(gt_dist, _f32p), _ptr(gt_ids, _u32p), nq, ks, k, _ptr(points, _f32p),
                                                  points.shape[1], metric, _ptr(queries, _f32p), _ptr(kth, _f32p))
        return kth, gt_dist, gt_ids, int(wrong)

    def recall(self, dists, offsets, kth, k):
        dists, offsets, kth = _f32(dists), _u64(offsets), _f32(kth)
        return float(self.lib.orc_recall(_ptr(dists, _f32p), _ptr(offsets, _u64p), len(kth), _ptr(kth, _f32p), k))

    def dedup_neighbors(self, dists, ids, offsets, k):
        dists, ids, offsets = _f32(dists), _u32(ids), _u64(offsets)
        nq = len(offsets) - 1
        od, oi, oc = np.empty((nq, k), np.float32), np.empty((nq, k), np.uint32), np.empty(nq, np.uint32)
        self.lib.orc_dedup_neighbors(_ptr(dists, _f32p), _ptr(ids, _u32p), _ptr(offsets, _u64p), nq, k, _ptr(od, _f32p),
                                     _ptr(oi, _u32p), _ptr(oc, _u32p))
        return od, oi, oc

    # ---- k-means / router --------------------------------------------------------------------------------------
    def kmeans(self, points, k, seed, metric=0, nthreads=1):
        points = _f32(points)
        part = np.empty(points.shape[0], np.int32)
        cent = np.empty((k, points.shape[1]), np.float32)
        kk = self.lib.orc_kmeans(_ptr(points, _f32p), points.shape[0], points.shape[1], metric, k, seed, _ptr(part, _i32p),
                                 _ptr(cent, _f32p), nthreads)
        return part, cent[:kk]

    def partition_to_clusters(self, partition):
        partition = _i32(partition)
        n = len(partition)
        S = int(partition.max()) + 1 if n else 0
        ids = np.empty(n, np.uint32)
        offs = np.empty(S + 1, np.uint64)
        s = self.lib.orc_partition_to_clusters(_ptr(partition, _i32p), n, _ptr(ids, _u32p), _ptr(offs, _u64p))
        assert s == S
        return [ids[int(offs[b]):int(offs[b + 1])] for b in range(S)]

    def kmtr_train(self, points, clusters, num_centroids=64, min_cluster_size=250, budget=50000, metric=0, nthreads=1):
        points = _f32(points)
        ids, offs = clusters_to_csr(clusters)
        h = self.lib.orc_kmtr_train(_ptr(points, _f32p), points.shape[0], points.shape[1], metric, _ptr(ids, _u32p), _ptr(offs, _u64p),
                                    len(clusters), num_centroids, min_cluster_size, budget, nthreads)
        return Tree(self, h)

    def kmtr_from_flat(self, flat, metric=0):
        h = self.lib.orc_kmtr_from_flat(flat["d"], metric, flat["S"], len(flat["node_shard"]), _ptr(_i32(flat["node_shard"]), _i32p),
                                        _ptr(_u32(flat["node_cent_first"]), _u32p), _ptr(_u32(flat["node_child_first"]), _u32p),
                                        _ptr(_u32(flat["node_child_count"]), _u32p), _ptr(_u32(flat["shard_root"]), _u32p),
                                        _ptr(_f32(flat["centroids"]), _f32p))
        return Tree(self, h)

    def frequency_probes(self, min_dist, nn_shard, num_voting):
        min_dist, nn_shard = _f32(min_dist), _u32(nn_shard)
        out = np.empty(len(min_dist), np.int32)
        self.lib.orc_frequency_probes(_ptr(min_dist, _f32p), len(min_dist), _ptr(nn_shard, _u32p), len(nn_shard), num_voting,
                                      _ptr(out, _i32p))
        return out

    # ---- synthetic data ----------------------------------------------------------------------------------------
    def synth_rows(self, seed, center_seed, n_centers, sigma, row0, nrows, d, nthreads=1):
        out = np.empty((nrows, d), np.float32)
        self.lib.orc_synth_rows(seed, center_seed, n_centers, sigma, row0, nrows, d, _ptr(out, _f32p), nthreads)
        return out


class Tree:
    """An oracle-side k-means tree router (flat layout, see gpann_oracle.h)."""

    def __init__(self, orc, handle):
        self.orc, self.h = orc, handle
        d, S, nn, nc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        orc.lib.orc_kmtr_sizes(self.h, C.byref(d), C.byref(S), C.byref(nn), C.byref(nc))
        self.d, self.S, self.n_nodes, self.n_centroids = d.value, S.value, nn.value, nc.value

    def __del__(self):
        if getattr(self, "h", None):
            self.orc.lib.orc_kmtr_free(self.h)
            self.h = None

    def flat(self):
        f = dict(d=self.d, S=self.S,
                 node_shard=np.empty(self.n_nodes, np.int32), node_cent_first=np.empty(self.n_nodes + 1, np.uint32),
                 node_child_first=np.empty(self.n_nodes, np.uint32), node_child_count=np.empty(self.n_nodes, np.uint32),
                 shard_root=np.empty(self.S, np.uint32), centroids=np.empty((self.n_centroids, self.d), np.float32))
        self.orc.lib.orc_kmtr_export(self.h, _ptr(f["node_shard"], _i32p), _ptr(f["node_cent_first"], _u32p),
                                     _ptr(f["node_child_first"], _u32p), _ptr(f["node_child_count"], _u32p),
                                     _ptr(f["shard_root"], _u32p), _ptr(f["centroids"], _f32p))
        return f

    def extract_partition(self):
        out = np.empty(self.n_centroids, np.int32)
        self.orc.lib.orc_kmtr_extract_partition(self.h, _ptr(out, _i32p))
        return out

    def query(self, Q, budget):
        Q = _f32(Q)
        probes, md = np.empty(self.S, np.int32), np.empty(self.S, np.float32)
        self.orc.lib.orc_kmtr_query(self.h, _ptr(Q, _f32p), budget, _ptr(probes, _i32p), _ptr(md, _f32p))
        return probes, md

    def query_batch(self, queries, budget, nthreads=1):
        queries = _f32(queries)
        nq = queries.shape[0]
        probes, md = np.empty((nq, self.S), np.int32), np.empty((nq, self.S), np.float32)
        self.orc.lib.orc_kmtr_query_batch(self.h, _ptr(queries, _f32p), nq, budget, _ptr(probes, _i32p), _ptr(md, _f32p), nthreads)
        return probes, md

    def frequency_query(self, Q, budget, num_voting):
        Q = _f32(Q)
        md = np.empty(self.S, np.float32)
        nd, ns = np.empty(num_voting, np.float32), np.empty(num_voting, np.uint32)
        c = self.orc.lib.orc_kmtr_frequency_query(self.h, _ptr(Q, _f32p), budget, num_voting, _ptr(md, _f32p), _ptr(nd, _f32p),
                                                  _ptr(ns, _u32p))
        return md, nd[:c], ns[:c]
