"""ctypes access to the extern "C" surface of libgpann_host.so (gp-ann_amd/host/host_capi.cpp): the host-side mirror of
the reference's router training and k-means.  Loading it needs libgpann_hip.so next to it but no GPU."""
import ctypes as C
import os

import numpy as np

from . import LIB_PATH, clusters_to_csr

HOST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libgpann_host.so")
_f32p, _u32p, _i32p, _u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(HOST_LIB_PATH)
        L.gpann_host_last_error.restype = C.c_char_p
        L.gpann_host_configure.argtypes = [C.c_int, C.c_int, C.c_int]
        L.gpann_host_router_train.restype = C.c_void_p
        L.gpann_host_router_train.argtypes = [_f32p, C.c_uint64, C.c_uint32, _u32p, _u64p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int64]
        L.gpann_host_router_free.argtypes = [C.c_void_p]
        L.gpann_host_router_sizes.argtypes = [C.c_void_p, _u32p, _u64p]
        L.gpann_host_router_export.argtypes = [C.c_void_p, _i32p, _u32p, _u32p, _u32p, _u32p, _f32p]
        L.gpann_host_kmeans.restype = C.c_int64
        L.gpann_host_kmeans.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _i32p, _f32p]
        L.gpann_host_partition.restype = C.c_int64
        L.gpann_host_partition.argtypes = [C.c_int, _f32p, C.c_uint64, C.c_uint32, C.c_int, C.c_double, C.c_uint64, C.c_int, _i32p]
        L.gpann_host_overlap_spann.restype = C.c_int64
        L.gpann_host_overlap_spann.argtypes = [_f32p, C.c_uint64, C.c_uint32, _i32p, C.c_int, C.c_double, C.c_double, C.c_int, _u32p, C.c_uint64, _u64p]
        L.gpann_host_read_points.argtypes = [C.c_char_p, C.c_int64, _u64p, _u32p, _f32p, C.c_uint64]
        L.gpann_host_write_points.argtypes = [C.c_char_p, _f32p, C.c_uint64, C.c_uint32]
        L.gpann_host_ground_truth_roundtrip.argtypes = [C.c_char_p, _f32p, _u32p, C.c_uint32, C.c_uint32, _f32p, _u32p]
        L.gpann_host_distance.restype = C.c_float
        L.gpann_host_distance.argtypes = [_f32p, _f32p, C.c_uint32]
        L.gpann_host_router_query_threads.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _i32p, _u32p, _u32p, _u32p, _u32p, _f32p, _f32p, C.c_uint32, C.c_int,
                                                      C.c_int, C.c_int, _i32p, C.POINTER(C.c_double), _u64p]
        L.gpann_host_index_query_bucket_threads.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_int, _u32p, _u64p, C.c_uint32, _f32p, C.c_uint32, C.c_uint32, _i32p,
                                                            C.c_int, C.c_int, _u32p, _f32p, C.POINTER(C.c_double), _u64p, C.c_void_p]
        _lib = L
    return _lib


def configure(mips=False, device=0, threads=0):
    lib().gpann_host_configure(int(mips), device, threads)


def train_router(points, clusters, num_centroids=64, min_cluster_size=250, budget=50000, mips=False, threads=0):
    """KMeansTreeRouter::Train on the host (reference src/kmeans_tree_router.cpp:9-78) -> flat tree dict."""
    points = np.ascontiguousarray(points, np.float32)
    ids, offs = clusters_to_csr(clusters)
    configure(mips, 0, threads)
    n, d = points.shape
    S = len(clusters)
    h = lib().gpann_host_router_train(points.ctypes.data_as(_f32p), n, d, ids.ctypes.data_as(_u32p), offs.ctypes.data_as(_u64p), S, num_centroids,
                                      min_cluster_size, budget)
    if not h:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    try:
        nn, nc = C.c_uint32(), C.c_uint64()
        lib().gpann_host_router_sizes(h, C.byref(nn), C.byref(nc))
        f = dict(d=d, S=S, node_shard=np.empty(nn.value, np.int32), node_cent_first=np.empty(nn.value + 1, np.uint32),
                 node_child_first=np.empty(nn.value, np.uint32), node_child_count=np.empty(nn.value, np.uint32),
                 shard_root=np.empty(S, np.uint32), centroids=np.empty((nc.value, d), np.float32))
        lib().gpann_host_router_export(h, f["node_shard"].ctypes.data_as(_i32p), f["node_cent_first"].ctypes.data_as(_u32p),
                                       f["node_child_first"].ctypes.data_as(_u32p), f["node_child_count"].ctypes.data_as(_u32p),
                                       f["shard_root"].ctypes.data_as(_u32p), f["centroids"].ctypes.data_as(_f32p))
        return f
    finally:
        lib().gpann_host_router_free(h)


def router_query_threads(tree, queries, budget, threads, coalesce, mips=False):
    """The reference's calling pattern on the host mirror (needs a GPU): `threads` host threads, each calling KMeansTreeRouter::Query(Q, budget)
    for its share of the queries, one query per call (routes.cpp:194-200) — with the mirror's coalescing queue on or off.
    Returns (probes [nq][S], seconds of the query loop, device batches it took)."""
    queries = np.ascontiguousarray(queries, np.float32)
    nq, d = queries.shape
    S = int(tree["S"])
    cents = np.ascontiguousarray(tree["centroids"], np.float32)
    probes = np.empty((nq, S), np.int32)
    sec, batches = C.c_double(0), C.c_uint64(0)
    arr = {k_: np.ascontiguousarray(tree[k_]) for k_ in ("node_shard", "node_cent_first", "node_child_first", "node_child_count", "shard_root")}
    rc = lib().gpann_host_router_query_threads(d, int(mips), S, len(arr["node_shard"]), arr["node_shard"].astype(np.int32).ctypes.data_as(_i32p),
                                               arr["node_cent_first"].astype(np.uint32).ctypes.data_as(_u32p), arr["node_child_first"].astype(np.uint32).ctypes.data_as(_u32p),
                                               arr["node_child_count"].astype(np.uint32).ctypes.data_as(_u32p), arr["shard_root"].astype(np.uint32).ctypes.data_as(_u32p),
                                               cents.ctypes.data_as(_f32p), queries.ctypes.data_as(_f32p), nq, int(budget), int(threads), int(bool(coalesce)),
                                               probes.ctypes.data_as(_i32p), C.byref(sec), C.byref(batches))
    if rc != 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    return probes, sec.value, batches.value


def index_query_bucket_threads(points, clusters, queries, k, buckets, threads, coalesce, mips=False, tolerate_failures=False):
    """InvertedIndex::QueryBucket(Q, k, buckets[q]) from `threads` host threads, one call per query (shard_searches.cpp:48) — coalescing queue on or
    off.  Returns (ids [nq][k], dist [nq][k], seconds, device batches) — tolerate_failures: + failed [nq] (1 where that query's call threw; the others'
    results stand)."""
    points = np.ascontiguousarray(points, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    buckets = np.ascontiguousarray(buckets, np.int32)
    ids_csr, offs = clusters_to_csr(clusters)
    nq, d = queries.shape
    ids = np.empty((nq, k), np.uint32)
    dist = np.empty((nq, k), np.float32)
    sec, batches = C.c_double(0), C.c_uint64(0)
    failed = np.zeros(nq, np.uint8) if tolerate_failures else None
    rc = lib().gpann_host_index_query_bucket_threads(points.ctypes.data_as(_f32p), points.shape[0], d, int(mips), ids_csr.ctypes.data_as(_u32p), offs.ctypes.data_as(_u64p),
                                                     len(clusters), queries.ctypes.data_as(_f32p), nq, int(k), buckets.ctypes.data_as(_i32p), int(threads),
                                                     int(bool(coalesce)), ids.ctypes.data_as(_u32p), dist.ctypes.data_as(_f32p), C.byref(sec), C.byref(batches),
                                                     failed.ctypes.data_as(C.c_void_p) if tolerate_failures else None)
    if rc != 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    if tolerate_failures:
        return ids, dist, sec.value, batches.value, failed
    return ids, dist, sec.value, batches.value


def kmeans(points, k, seed, mips=False, threads=0):
    points = np.ascontiguousarray(points, np.float32)
    configure(mips, 0, threads)
    part = np.empty(points.shape[0], np.int32)
    cent = np.empty((k, points.shape[1]), np.float32)
    kk = lib().gpann_host_kmeans(points.ctypes.data_as(_f32p), points.shape[0], points.shape[1], k, seed, part.ctypes.data_as(_i32p),
                                 cent.ctypes.data_as(_f32p))
    if kk < 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    return part, cent[:kk]


def distance(p, q, mips=False):
    p, q = np.ascontiguousarray(p, np.float32), np.ascontiguousarray(q, np.float32)
    configure(mips, 0, 0)
    return np.float32(lib().gpann_host_distance(p.ctypes.data_as(_f32p), q.ctypes.data_as(_f32p), p.shape[0]))


PART_KMEANS, PART_RECURSIVE, PART_REBALANCING = 0, 1, 2


def partition(which, points, k=-1, eps=0.05, max_cluster_size=0, mips=False, device=False, threads=0):
    """KMeansPartitioning / RecursiveKMeansPartitioning / RebalancingKMeansPartitioning (reference src/partitioning.cpp:12-146);
    device=True runs k-means on the GPU (gpann_kmeans)."""
    points = np.ascontiguousarray(points, np.float32)
    configure(mips, 0, threads)
    part = np.empty(points.shape[0], np.int32)
    r = lib().gpann_host_partition(which, points.ctypes.data_as(_f32p), points.shape[0], points.shape[1], int(k), float(eps), int(max_cluster_size),
                                   int(device), part.ctypes.data_as(_i32p))
    if r < 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    return part


def overlap_spann(points, partition, requested_num_clusters, eps, overlap, mips=False, device=False, threads=0):
    """OverlappingKMeansPartitioningSPANN (reference src/overlapping_partitioning.cpp:143-235) -> list of id arrays."""
    points = np.ascontiguousarray(points, np.float32)
    partition = np.ascontiguousarray(partition, np.int32)
    configure(mips, 0, threads)
    n = points.shape[0]
    cap = 6 * n + 16
    ids = np.empty(cap, np.uint32)
    offs = np.empty(int(partition.max()) + 2 if n else 1, np.uint64)
    S = lib().gpann_host_overlap_spann(points.ctypes.data_as(_f32p), n, points.shape[1], partition.ctypes.data_as(_i32p), int(requested_num_clusters),
                                       float(eps), float(overlap), int(device), ids.ctypes.data_as(_u32p), cap, offs.ctypes.data_as(_u64p))
    if S < 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    return [ids[int(offs[b]):int(offs[b + 1])].copy() for b in range(S)]


def read_points(path, size=-1):
    """ReadPoints (reference src/points_io.cpp:114-124): .fbin / .u8bin / .i8bin -> float32 [n, d]; `size` overrides n for .fbin only."""
    n, d = C.c_uint64(), C.c_uint32()
    if lib().gpann_host_read_points(path.encode(), size, C.byref(n), C.byref(d), None, 0) != 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    out = np.empty((n.value, d.value), np.float32)
    if lib().gpann_host_read_points(path.encode(), size, C.byref(n), C.byref(d), out.ctypes.data_as(_f32p), out.size) != 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    return out


def write_points(path, points):
    points = np.ascontiguousarray(points, np.float32)
    if lib().gpann_host_write_points(path.encode(), points.ctypes.data_as(_f32p), points.shape[0], points.shape[1]) != 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())


def ground_truth_roundtrip(path, dist, ids):
    """WriteGroundTruth then ReadGroundTruth (reference src/points_io.cpp:136-179)."""
    dist, ids = np.ascontiguousarray(dist, np.float32), np.ascontiguousarray(ids, np.uint32)
    od, oi = np.empty_like(dist), np.empty_like(ids)
    if lib().gpann_host_ground_truth_roundtrip(path.encode(), dist.ctypes.data_as(_f32p), ids.ctypes.data_as(_u32p), dist.shape[0], dist.shape[1],
                                               od.ctypes.data_as(_f32p), oi.ctypes.data_as(_u32p)) != 0:
        raise RuntimeError(lib().gpann_host_last_error().decode())
    return od, oi
