"""pygpann — thin ctypes binding of libgpann_hip.so (include/gpann.h) used by the tests and bench.py.

The product is the C-ABI shared library plus the C++ host mirror under gp-ann_amd/host/; this module only marshals
numpy arrays / raw device pointers into that ABI.  It never falls back to a CPU implementation: if the library or a GPU
is missing, calls raise GpannError.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libgpann_hip.so")
if os.environ.get("GPANN_HIP_LIBRARY"):   # another build of the same library: the co-tenancy-safe flavour (make cotenant-safe -> lib_cotenant/), or the measurement build of tools/router_event_experiment.sh
    LIB_PATH = os.path.abspath(os.environ["GPANN_HIP_LIBRARY"])

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_UNSUPPORTED = range(6)
METRIC_L2, METRIC_MIPS = 0, 1
QUERIES_ON_DEVICE, OUTPUT_ON_DEVICE, POINTS_ON_DEVICE, AUX_PASS, STORE_PLAIN, SCAN_FOR_MERGE, SCAN_DEFER, SCAN_OVERLAP = 1, 2, 4, 8, 16, 32, 64, 128
MAX_K = 64
NO_ID = np.uint32(0xFFFFFFFF)

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)

EXPORTS = [
    "gpann_last_error", "gpann_version", "gpann_abi_version", "gpann_scan_last_stats_sized", "gpann_device_count", "gpann_shards_create", "gpann_shards_destroy", "gpann_shards_build_filter", "gpann_shards_info",
    "gpann_shards_export", "gpann_shards_export_range", "gpann_scan_buckets", "gpann_scan_merged", "gpann_shard_search_hits", "gpann_ground_truth",
    "gpann_scan_last_stats", "gpann_scan_finish", "gpann_shards_set_scan_mode", "gpann_router_create", "gpann_router_destroy", "gpann_router_train", "gpann_kmeans", "gpann_router_info", "gpann_router_get_stats", "gpann_router_export", "gpann_router_query_batch", "gpann_router_query_batch_top", "gpann_router_frequency_query_batch", "gpann_merge_dedup",
    "gpann_synth_rows_device", "gpann_nearest_centroid_device",
    "gpann_option_count", "gpann_option_name", "gpann_option_help", "gpann_option_read_at_build",
    "gpann_shards_set_option", "gpann_shards_get_option", "gpann_router_set_option", "gpann_router_get_option",
]


class GpannError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"gpann status {status}: {msg}")
        self.status = status


class ScanStats(C.Structure):
    _fields_ = [("pairs", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("tasks", C.c_uint64), ("units", C.c_uint64),
                ("padded_pairs", C.c_uint64), ("filter_pairs", C.c_uint64), ("filter_candidates", C.c_uint64), ("filtered", C.c_uint32), ("filter_plane_ksteps", C.c_uint32),
                ("scan_ms", C.c_float), ("merge_ms", C.c_float), ("filter_ms", C.c_float), ("filter_dirty_tasks", C.c_uint32),
                ("filter_rows", C.c_uint64), ("filter_rest_rows", C.c_uint64), ("filter_rest_ms", C.c_float), ("filter_rest_hi_only", C.c_uint32),
                ("filter_two_plane_rows", C.c_uint64), ("filter_seed_rows", C.c_uint64),
                ("filter_rest_row_bytes", C.c_uint32), ("overlapped", C.c_uint32),
                ("filter_mfma_bf16", C.c_uint64), ("filter_mfma_i8", C.c_uint64), ("filter_rest_mfma_i8", C.c_uint64)]


class RouterStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("bound_calls", "bound_selfcheck", "bound_overflow", "verified", "verify_failures", "host_sorted", "heap_redo")] + \
               [("last_bound_ms", C.c_double), ("last_bound_flops", C.c_double), ("probed", C.c_uint64), ("probe_failures", C.c_uint64)] + \
               [(n, C.c_uint64) for n in ("table_blocks", "table_queries", "table_band_nodes", "table_exact", "table_band_overflow", "table_rescans")]


def build(force=False):
    """Compile libgpann_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _PKG, "clean"])
    subprocess.check_call(["make", "-C", _PKG, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library (no implicit build: a missing library is an error, not a fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpannError(-1, f"{LIB_PATH} not found — build it with `make -C gp-ann_amd` (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(LIB_PATH)
    L.gpann_last_error.restype = C.c_char_p
    L.gpann_version.restype = C.c_char_p
    L.gpann_device_count.argtypes = [C.POINTER(C.c_int)]
    L.gpann_option_count.restype = C.c_uint32
    L.gpann_option_name.restype = C.c_char_p
    L.gpann_option_name.argtypes = [C.c_uint32]
    L.gpann_option_help.restype = C.c_char_p
    L.gpann_option_help.argtypes = [C.c_uint32]
    L.gpann_option_read_at_build.argtypes = [C.c_uint32]
    for fn in (L.gpann_shards_set_option, L.gpann_router_set_option):
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    for fn in (L.gpann_shards_get_option, L.gpann_router_get_option):
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.gpann_shards_create.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, _u32p, _u64p, C.c_uint32, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]
    L.gpann_shards_destroy.argtypes = [C.c_void_p]
    L.gpann_shards_build_filter.argtypes = [C.c_void_p]
    L.gpann_shards_info.argtypes = [C.c_void_p, _u64p, _u32p, _u32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gpann_shards_export.argtypes = [C.c_void_p, _i64p, _u32p, _f32p]
    L.gpann_shards_export_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, _u32p, _f32p]
    L.gpann_scan_buckets.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _i32p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, _f32p]
    L.gpann_scan_merged.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _i32p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, _f32p]
    L.gpann_shard_search_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, _u32p, _u32p, C.c_int, C.c_void_p, _f32p]
    L.gpann_ground_truth.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p, _f32p]
    L.gpann_scan_last_stats.argtypes = [C.c_void_p, C.POINTER(ScanStats)]
    L.gpann_scan_last_stats_sized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.gpann_abi_version.restype = C.c_uint32
    L.gpann_scan_finish.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.gpann_shards_set_scan_mode.argtypes = [C.c_void_p, C.c_int]
    L.gpann_router_create.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _i32p, _u32p, _u32p, _u32p, _u32p, _f32p, C.c_int,
                                      C.POINTER(C.c_void_p)]
    L.gpann_router_destroy.argtypes = [C.c_void_p]
    L.gpann_router_train.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, C.POINTER(C.c_void_p)]
    L.gpann_kmeans.argtypes = [C.c_void_p, C.c_uint32, C.c_int, _i32p, _f32p, _u32p]
    L.gpann_router_info.argtypes = [C.c_void_p, _u32p, C.POINTER(C.c_int), _u32p, _u32p, _u64p]
    L.gpann_router_get_stats.argtypes = [C.c_void_p, C.POINTER(RouterStats)]
    L.gpann_router_export.argtypes = [C.c_void_p, _i32p, _u32p, _u32p, _u32p, _u32p, _f32p]
    L.gpann_router_query_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, _i32p, _f32p, C.c_int, C.c_void_p, _f32p]
    L.gpann_router_query_batch_top.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, _i32p, _f32p, C.c_int, C.c_void_p, _f32p]
    L.gpann_router_frequency_query_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, _f32p, _f32p, _u32p, _u32p,
                                                     C.c_int, C.c_void_p, _f32p]
    L.gpann_merge_dedup.argtypes = [_f32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _f32p, _u32p, _u32p]
    L.gpann_synth_rows_device.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p,
                                          C.c_void_p]
    L.gpann_nearest_centroid_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    _lib = L
    return L


def _check(status):
    if status != OK:
        raise GpannError(status, lib().gpann_last_error().decode())


def options():
    """[(name, read_at_build, help)]: the table of handle options (gpann_option_*)"""
    L = lib()
    return [(L.gpann_option_name(i).decode(), bool(L.gpann_option_read_at_build(i)), L.gpann_option_help(i).decode()) for i in range(L.gpann_option_count())]


def device_count():
    c = C.c_int(0)
    _check(lib().gpann_device_count(C.byref(c)))
    return c.value


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _vp(a):
    """numpy array -> void*; int -> raw (device) pointer; None -> NULL"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return a.ctypes.data_as(C.c_void_p)


def clusters_to_csr(clusters):
    offs = np.zeros(len(clusters) + 1, dtype=np.uint64)
    for b, c in enumerate(clusters):
        offs[b + 1] = offs[b] + len(c)
    ids = np.concatenate([np.asarray(c, dtype=np.uint32) for c in clusters]) if len(clusters) else np.zeros(0, np.uint32)
    return _np(ids, np.uint32), offs


class Shards:
    """Device-resident shard store = InvertedIndex(points, clusters) (reference src/inverted_index.h:12-41)."""

    def __init__(self, points, clusters, metric=METRIC_L2, device=0, n=None, d=None, plain=False):
        """points: numpy [n,d] float32 (host) or an int device pointer (then n, d must be given).
        plain: rows only, no filter support data (GPANN_STORE_PLAIN) — for stores that only feed training / partitioning."""
        ids, offs = clusters_to_csr(clusters)
        flags = STORE_PLAIN if plain else 0
        if isinstance(points, (int, np.integer)):
            flags |= POINTS_ON_DEVICE
            pts = C.c_void_p(int(points))
        else:
            points = _np(points, np.float32)
            n, d = points.shape
            self._keep = points
            pts = _vp(points)
        self.h = C.c_void_p()
        self.k_max = MAX_K
        _check(lib().gpann_shards_create(pts, n, d, metric, ids.ctypes.data_as(_u32p), offs.ctypes.data_as(_u64p), len(clusters), device,
                                         flags, C.byref(self.h)))
        self.S, self.d, self.metric, self.device = len(clusters), d, metric, device
        self.rows = int(offs[-1])

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().gpann_shards_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_filter(self):
        """filter support data of a store created with plain=True, built after the caller freed its own copy of the points"""
        _check(lib().gpann_shards_build_filter(self.h))

    def kmeans(self, k, seed):
        """RandomSample(k, seed) + KMeans over all rows as one point set (gpann_kmeans): (partition[rows] int32, centroids[k', d])."""
        part = np.empty(self.rows, np.int32)
        cent = np.empty((int(k), self.d), np.float32)
        kk = C.c_uint32(0)
        _check(lib().gpann_kmeans(self.h, C.c_uint32(int(k)), C.c_int(int(seed)), part.ctypes.data_as(_i32p), cent.ctypes.data_as(_f32p), C.byref(kk)))
        return part, cent[:kk.value].copy()

    def set_option(self, name, value):
        """a knob of this handle (gpann_shards_set_option; include/gpann.h "Options"): value None = back to the default.  The handle took the
        environment's GPANN_* variables when it was created and does not look at the environment again."""
        _check(lib().gpann_shards_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def get_option(self, name):
        buf = C.create_string_buffer(256)
        _check(lib().gpann_shards_get_option(self.h, name.encode(), buf, 256))
        return buf.value.decode()

    def set_scan_mode(self, mode):
        """0 = exact VALU scan, 1 = exact scan with the MFMA pre-filter (same results)"""
        _check(lib().gpann_shards_set_scan_mode(self.h, int(mode)))

    def export(self):
        offsets = np.empty(self.S + 1, np.int64)
        perm = np.empty(self.rows, np.uint32)
        pts = np.empty((self.rows, self.d), np.float32)
        _check(lib().gpann_shards_export(self.h, offsets.ctypes.data_as(_i64p), perm.ctypes.data_as(_u32p), pts.ctypes.data_as(_f32p)))
        return offsets, perm, pts

    def offsets(self):
        offsets = np.empty(self.S + 1, np.int64)
        _check(lib().gpann_shards_export(self.h, offsets.ctypes.data_as(_i64p), None, None))
        return offsets

    def export_range(self, row_begin, row_count):
        perm = np.empty(row_count, np.uint32)
        pts = np.empty((row_count, self.d), np.float32)
        _check(lib().gpann_shards_export_range(self.h, row_begin, row_count, perm.ctypes.data_as(_u32p), pts.ctypes.data_as(_f32p)))
        return perm, pts

    def scan_buckets(self, queries, k, probes, nprobe=None, stream=None, want_rows=False, for_merge=False):
        """Batched QueryBucket.  queries [nq,d]; probes [nq,stride] int32.  Returns ids, dist [nq,nprobe,k] (+rows), ms.
        for_merge: GPANN_SCAN_FOR_MERGE — per-slot lists cut at the query's merge bound (include/gpann.h)."""
        queries = _np(queries, np.float32)
        probes = _np(probes, np.int32)
        if probes.ndim == 1:
            probes = probes.reshape(-1, 1)
        nq = queries.shape[0]
        stride = probes.shape[1] if probes.size else (nprobe or 0)
        nprobe = stride if nprobe is None else nprobe
        ids = np.empty((nq, nprobe, k), np.uint32)
        dist = np.empty((nq, nprobe, k), np.float32)
        rows = np.empty((nq, nprobe, k), np.uint32) if want_rows else None
        ms = C.c_float(0)
        _check(lib().gpann_scan_buckets(self.h, _vp(queries), nq, k, probes.ctypes.data_as(_i32p), stride, nprobe, _vp(ids), _vp(dist),
                                        _vp(rows), SCAN_FOR_MERGE if for_merge else 0, C.c_void_p(stream or 0), C.byref(ms)))
        if want_rows:
            return ids, dist, rows, ms.value
        return ids, dist, ms.value

    def scan_buckets_device(self, queries_ptr, nq, k, probes, nprobe, out_ids_ptr, out_dist_ptr, stream=None, timed=False, aux=False, for_merge=False,
                            defer=False, host_out=False, overlap=False):
        """Same with queries resident on the device (raw pointer) and the outputs on the device (raw pointers) or — host_out — in host
        arrays given by address (pinned ones for an asynchronous copy); asynchronous unless timed.  defer: GPANN_SCAN_DEFER — the call only
        enqueues its work and finish() completes it; two more calls may be issued before that (the handle has three call slots).  overlap (with
        defer): GPANN_SCAN_OVERLAP — the call runs on the handle's own streams (a front stream per slot, one rest stream, one tail stream), its front
        beside the rest pass of the calls before it; results are in place after finish(); stats()["overlapped"] says whether the flag took effect."""
        probes = _np(probes, np.int32)
        stride = probes.shape[1]
        ms = C.c_float(0)
        flags = QUERIES_ON_DEVICE | (0 if host_out else OUTPUT_ON_DEVICE) | (AUX_PASS if aux else 0) | (SCAN_FOR_MERGE if for_merge else 0) | (SCAN_DEFER if defer else 0) | \
            (SCAN_OVERLAP if overlap and defer else 0)
        _check(lib().gpann_scan_buckets(self.h, C.c_void_p(int(queries_ptr)), nq, k, probes.ctypes.data_as(_i32p), stride, nprobe,
                                        C.c_void_p(int(out_ids_ptr)), C.c_void_p(int(out_dist_ptr)) if out_dist_ptr else None, None,
                                        flags, C.c_void_p(stream or 0), C.byref(ms) if (timed and not defer) else None))
        return ms.value

    def finish(self):
        """completes the oldest deferred scan call (gpann_scan_finish); returns its kernel time in ms"""
        ms = C.c_float(0)
        _check(lib().gpann_scan_finish(self.h, C.byref(ms)))
        return ms.value

    def scan_merged(self, queries, k, probes, nprobe, stream=None):
        queries = _np(queries, np.float32)
        probes = _np(probes, np.int32)
        nq = queries.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dist = np.empty((nq, k), np.float32)
        ms = C.c_float(0)
        _check(lib().gpann_scan_merged(self.h, _vp(queries), nq, k, probes.ctypes.data_as(_i32p), probes.shape[1], nprobe, _vp(ids), _vp(dist), 0,
                                       C.c_void_p(stream or 0), C.byref(ms)))
        return ids, dist, ms.value

    def shard_search_hits(self, queries, k, distance_to_kth, stream=None):
        queries = _np(queries, np.float32)
        kth = _np(distance_to_kth, np.float32)
        nq = queries.shape[0]
        ids = np.empty((self.S, nq, k), np.uint32)
        cnt = np.empty((self.S, nq), np.uint32)
        ms = C.c_float(0)
        _check(lib().gpann_shard_search_hits(self.h, _vp(queries), nq, k, kth.ctypes.data_as(_f32p), ids.ctypes.data_as(_u32p),
                                             cnt.ctypes.data_as(_u32p), 0, C.c_void_p(stream or 0), C.byref(ms)))
        return ids, cnt, ms.value

    def stats(self):
        st = ScanStats()
        _check(lib().gpann_scan_last_stats_sized(self.h, C.byref(st), C.sizeof(st)))   # (sized: this binding and the library may be of different revisions)
        return {f: getattr(st, f) for f, _ in ScanStats._fields_}


def ground_truth(points, queries, k, metric=METRIC_L2, device=0, store=None):
    """ComputeGroundTruth (reference src/recall.h:25-39): ids, dist [nq,k] ascending (dist,id)."""
    queries = _np(queries, np.float32)
    nq = queries.shape[0]
    ids = np.empty((nq, k), np.uint32)
    dist = np.empty((nq, k), np.float32)
    ms = C.c_float(0)
    if store is not None:
        _check(lib().gpann_ground_truth(store.h, None, 0, store.d, metric, _vp(queries), nq, k, _vp(ids), _vp(dist), device, 0, None,
                                        C.byref(ms)))
    else:
        points = _np(points, np.float32)
        _check(lib().gpann_ground_truth(None, _vp(points), points.shape[0], points.shape[1], metric, _vp(queries), nq, k, _vp(ids), _vp(dist),
                                        device, 0, None, C.byref(ms)))
    return ids, dist, ms.value


class Router:
    """Device-resident flattened KMeansTreeRouter (reference src/kmeans_tree_router.{h,cpp})."""

    def __init__(self, flat, metric=METRIC_L2, device=0):
        self.S, self.d = int(flat["S"]), int(flat["d"])
        ns = _np(flat["node_shard"], np.int32)
        cf = _np(flat["node_cent_first"], np.uint32)
        chf = _np(flat["node_child_first"], np.uint32)
        chc = _np(flat["node_child_count"], np.uint32)
        sr = _np(flat["shard_root"], np.uint32)
        cen = _np(flat["centroids"], np.float32)
        self.h = C.c_void_p()
        _check(lib().gpann_router_create(self.d, metric, self.S, len(ns), ns.ctypes.data_as(_i32p), cf.ctypes.data_as(_u32p),
                                         chf.ctypes.data_as(_u32p), chc.ctypes.data_as(_u32p), sr.ctypes.data_as(_u32p),
                                         cen.ctypes.data_as(_f32p), device, C.byref(self.h)))

    @classmethod
    def train(cls, shards, n_points, num_centroids=64, min_cluster_size=250, budget=50000):
        """KMeansTreeRouter::Train on the GPU over the rows of a shard store (gpann_router_train)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p()
        _check(lib().gpann_router_train(shards.h, C.c_uint64(int(n_points)), C.c_uint32(int(num_centroids)), C.c_uint32(int(min_cluster_size)),
                                        C.c_int64(int(budget)), C.byref(self.h)))
        d, S = C.c_uint32(0), C.c_uint32(0)
        _check(lib().gpann_router_info(self.h, C.byref(d), None, C.byref(S), None, None))
        self.S, self.d = S.value, d.value
        return self

    def export(self):
        """The flattened tree as the dict Router(flat) takes."""
        d, S, n_nodes, metric, nc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0), C.c_uint64(0)
        _check(lib().gpann_router_info(self.h, C.byref(d), C.byref(metric), C.byref(S), C.byref(n_nodes), C.byref(nc)))
        n = n_nodes.value
        flat = dict(S=S.value, d=d.value, metric=metric.value, node_shard=np.empty(n, np.int32), node_cent_first=np.empty(n + 1, np.uint32),
                    node_child_first=np.empty(n, np.uint32), node_child_count=np.empty(n, np.uint32), shard_root=np.empty(S.value, np.uint32),
                    centroids=np.empty((nc.value, d.value), np.float32))
        _check(lib().gpann_router_export(self.h, flat["node_shard"].ctypes.data_as(_i32p), flat["node_cent_first"].ctypes.data_as(_u32p),
                                         flat["node_child_first"].ctypes.data_as(_u32p), flat["node_child_count"].ctypes.data_as(_u32p),
                                         flat["shard_root"].ctypes.data_as(_u32p), flat["centroids"].ctypes.data_as(_f32p)))
        return flat

    def set_option(self, name, value):
        """a knob of this handle (gpann_router_set_option): value None = back to the default"""
        _check(lib().gpann_router_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def get_option(self, name):
        buf = C.create_string_buffer(256)
        _check(lib().gpann_router_get_option(self.h, name.encode(), buf, 256))
        return buf.value.decode()

    def stats(self):
        """Counters of this handle (gpann_router_get_stats); bound_selfcheck and verify_failures must stay 0."""
        st = RouterStats()
        _check(lib().gpann_router_get_stats(self.h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in RouterStats._fields_}

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().gpann_router_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query_batch(self, queries, budget, stream=None, on_device=False, nq=None, out_probes=None, want_min_dist=True, n_top=None):
        """Batched KMeansTreeRouter::Query.  Returns probes [nq,S] int32, min_dist [nq,S] float32 (None unless want_min_dist), ms.
        out_probes: a reusable [nq,S] int32 array to fill instead of a fresh one.  n_top: only the first n_top shards of every order
        (probes [nq,n_top]; gpann_router_query_batch_top)."""
        if on_device:
            qp = C.c_void_p(int(queries))
            flags = QUERIES_ON_DEVICE
        else:
            queries = _np(queries, np.float32)
            nq = queries.shape[0]
            qp = _vp(queries)
            flags = 0
        cols = self.S if n_top is None else int(n_top)
        probes = out_probes if out_probes is not None else np.empty((nq, cols), np.int32)
        assert probes.shape == (nq, cols) and probes.dtype == np.int32 and probes.flags.c_contiguous
        md = np.empty((nq, self.S), np.float32) if want_min_dist else None
        ms = C.c_float(0)
        _check(lib().gpann_router_query_batch_top(self.h, qp, nq, budget, cols, probes.ctypes.data_as(_i32p), md.ctypes.data_as(_f32p) if want_min_dist else None,
                                                  flags, C.c_void_p(stream or 0), C.byref(ms)))
        return probes, md, ms.value


def _router_frequency_query_batch(self, queries, budget, num_voting, stream=None):
    """Batched FrequencyQuery: min_dist [nq,S], nn_dist / nn_shard [nq,V], nn_count [nq], ms."""
    queries = _np(queries, np.float32)
    nq = queries.shape[0]
    md = np.empty((nq, self.S), np.float32)
    nd = np.full((nq, num_voting), np.inf, np.float32)
    ns = np.full((nq, num_voting), 0xFFFFFFFF, np.uint32)
    nc = np.empty(nq, np.uint32)
    ms = C.c_float(0)
    _check(lib().gpann_router_frequency_query_batch(self.h, _vp(queries), nq, budget, num_voting, md.ctypes.data_as(_f32p),
                                                    nd.ctypes.data_as(_f32p), ns.ctypes.data_as(_u32p), nc.ctypes.data_as(_u32p), 0,
                                                    C.c_void_p(stream or 0), C.byref(ms)))
    return md, nd, ns, nc, ms.value


Router.frequency_query_batch = _router_frequency_query_batch


def merge_dedup(dist, ids, nprobe_used, k):
    """DedupNeighbors (reference small_scale_queries.cpp:15-23) over [nq,nprobe,k] per-slot results."""
    dist, ids = _np(dist, np.float32), _np(ids, np.uint32)
    nq, nprobe, kk = ids.shape
    assert kk == k
    od, oi, oc = np.empty((nq, k), np.float32), np.empty((nq, k), np.uint32), np.empty(nq, np.uint32)
    _check(lib().gpann_merge_dedup(dist.ctypes.data_as(_f32p), ids.ctypes.data_as(_u32p), nq, nprobe, nprobe_used, k, od.ctypes.data_as(_f32p),
                                   oi.ctypes.data_as(_u32p), oc.ctypes.data_as(_u32p)))
    return od, oi, oc


def synth_rows_device(out_ptr, seed, center_seed, n_centers, sigma, row0, nrows, d, stream=None):
    _check(lib().gpann_synth_rows_device(seed, center_seed, n_centers, sigma, row0, nrows, d, C.c_void_p(int(out_ptr)), C.c_void_p(stream or 0)))


def nearest_centroid_device(points_ptr, n, d, metric, centroids_ptr, c, out_ptr, stream=None):
    _check(lib().gpann_nearest_centroid_device(C.c_void_p(int(points_ptr)), n, d, metric, C.c_void_p(int(centroids_ptr)), c,
                                               C.c_void_p(int(out_ptr)), C.c_void_p(stream or 0)))
