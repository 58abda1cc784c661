"""ctypes binding of libgpann_comm.so (include/gpann_comm.h): the exchange steps of the multi-GPU path — RCCL collectives on device
buffers (backend RCCL) or the same payloads through rank 0 over sockets on host buffers (backend TCP, CPU tests).  Loading the library
needs RCCL and libgpann_hip.so next to it, but no GPU."""
import ctypes as C
import os

import numpy as np

from . import LIB_PATH, GpannError, OK

COMM_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libgpann_comm.so")
RCCL, TCP = 0, 1
EXPORTS = ["gpann_comm_last_error", "gpann_comm_init", "gpann_comm_init_from_env", "gpann_comm_init_from_env_on", "gpann_comm_destroy", "gpann_comm_abort", "gpann_comm_wait", "gpann_comm_check_peers", "gpann_comm_info", "gpann_comm_backend_name",
           "gpann_comm_barrier", "gpann_comm_max_double", "gpann_comm_allgather_host", "gpann_comm_broadcast_host", "gpann_comm_broadcast",
           "gpann_comm_allgather", "gpann_comm_gather_slots_scratch", "gpann_comm_gather_slots", "gpann_comm_debug_select_slots", "gpann_lpt_owner", "gpann_mask_probes", "gpann_plan_replicas", "gpann_mask_probes_replicated",
           "gpann_comm_merge_router"]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(COMM_LIB_PATH):
            raise GpannError(-1, f"{COMM_LIB_PATH} not found — build it with `make -C gp-ann_amd`")
        L = C.CDLL(COMM_LIB_PATH)
        L.gpann_comm_last_error.restype = C.c_char_p
        L.gpann_comm_backend_name.restype = C.c_char_p
        L.gpann_comm_backend_name.argtypes = [C.c_void_p]
        L.gpann_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gpann_comm_init_from_env.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.gpann_comm_init_from_env_on.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gpann_comm_destroy.argtypes = [C.c_void_p]
        L.gpann_comm_abort.argtypes = [C.c_void_p]
        L.gpann_comm_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.gpann_comm_check_peers.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.gpann_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gpann_comm_barrier.argtypes = [C.c_void_p]
        L.gpann_comm_max_double.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.gpann_comm_allgather_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.gpann_comm_broadcast_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.gpann_comm_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.gpann_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gpann_comm_gather_slots_scratch.restype = C.c_size_t
        L.gpann_comm_gather_slots_scratch.argtypes = [C.c_int, C.c_size_t]
        L.gpann_comm_gather_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gpann_lpt_owner.argtypes = [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.gpann_mask_probes.argtypes = [C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.c_int,
                                        C.POINTER(C.c_int32)]
        L.gpann_plan_replicas.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.gpann_mask_probes_replicated.argtypes = [C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_uint32,
                                                   C.c_int, C.POINTER(C.c_int32)]
        L.gpann_comm_merge_router.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_void_p)]
        _lib = L
    return _lib


def _check(status):
    if status != OK:
        raise GpannError(status, lib().gpann_comm_last_error().decode())


def _ptr(a):
    """numpy array -> host pointer; int -> raw (device) pointer"""
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return a.ctypes.data_as(C.c_void_p)


def debug_select_slots(rank_dist, rank_ids, id_maps=None, device=0):
    """the device half of gather_slots on a fabricated world (gpann_comm_debug_select_slots): rank_dist / rank_ids [world][n] host arrays,
    id_maps: list of the ranks' local-row -> global-id maps (or None).  Returns (dist[n], ids[n])."""
    rank_dist = np.ascontiguousarray(rank_dist, np.float32)
    rank_ids = np.ascontiguousarray(rank_ids, np.uint32)
    world, n = rank_ids.shape
    out_d, out_i = np.empty(n, np.float32), np.empty(n, np.uint32)
    maps = begin = None
    if id_maps is not None:
        begin = np.concatenate([[0], np.cumsum([len(m) for m in id_maps])]).astype(np.uint64)
        maps = np.ascontiguousarray(np.concatenate([np.asarray(m, np.uint32) for m in id_maps]), np.uint32)
    _check(lib().gpann_comm_debug_select_slots(C.c_int(world), C.c_size_t(n), _ptr(rank_dist), _ptr(rank_ids), _ptr(maps) if maps is not None else None,
                                               _ptr(begin) if begin is not None else None, _ptr(out_d), _ptr(out_i), C.c_int(device)))
    return out_d, out_i


def lpt_owner(shard_rows, world):
    """LPT bin packing of the shards onto `world` ranks (gpann_lpt_owner) -> (owner[S] int32, load[world] int64)"""
    rows = np.ascontiguousarray(shard_rows, np.uint64)
    owner = np.empty(len(rows), np.int32)
    load = np.empty(world, np.uint64)
    _check(lib().gpann_lpt_owner(rows.ctypes.data_as(C.POINTER(C.c_uint64)), len(rows), world, owner.ctypes.data_as(C.POINTER(C.c_int32)),
                                 load.ctypes.data_as(C.POINTER(C.c_uint64))))
    return owner, load.astype(np.int64)


def mask_probes(probes, nprobe, owner, rank, out=None):
    """the probe slots this rank scans: probes[:, :nprobe] with the shards of other ranks replaced by -1 (gpann_mask_probes)"""
    probes = np.ascontiguousarray(probes, np.int32)
    owner = np.ascontiguousarray(owner, np.int32)
    nq, stride = probes.shape
    if out is None:
        out = np.empty((nq, nprobe), np.int32)
    _check(lib().gpann_mask_probes(probes.ctypes.data_as(C.POINTER(C.c_int32)), nq, stride, nprobe, owner.ctypes.data_as(C.POINTER(C.c_int32)), len(owner),
                                   rank, out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def plan_replicas(shard_rows, shard_work, world, capacity_rows, max_imbalance=1.05):
    """Replica placement for query-load balance (gpann_plan_replicas): returns (rep_begin[S+1] uint32, rep_rank int32 (CSR of the ranks
    holding each shard, primary first), rank_work[world], rank_rows[world])."""
    rows = np.ascontiguousarray(shard_rows, np.uint64)
    work = np.ascontiguousarray(shard_work, np.float64)
    S = len(rows)
    rep_begin = np.empty(S + 1, np.uint32)
    rep_rank = np.empty(S * world, np.int32)
    rank_work = np.empty(world, np.float64)
    rank_rows = np.empty(world, np.uint64)
    _check(lib().gpann_plan_replicas(rows.ctypes.data_as(C.POINTER(C.c_uint64)), work.ctypes.data_as(C.POINTER(C.c_double)), S, world, int(capacity_rows),
                                     float(max_imbalance), rep_begin.ctypes.data_as(C.POINTER(C.c_uint32)), rep_rank.ctypes.data_as(C.POINTER(C.c_int32)),
                                     rank_work.ctypes.data_as(C.POINTER(C.c_double)), rank_rows.ctypes.data_as(C.POINTER(C.c_uint64))))
    return rep_begin, rep_rank[:int(rep_begin[S])].copy(), rank_work, rank_rows.astype(np.int64)


def mask_probes_replicated(probes, nprobe, rep_begin, rep_rank, rank, out=None):
    """the probe slots this rank scans under a replica placement: slot (q, b) belongs to replica q mod r_b of shard b"""
    probes = np.ascontiguousarray(probes, np.int32)
    rep_begin = np.ascontiguousarray(rep_begin, np.uint32)
    rep_rank = np.ascontiguousarray(rep_rank, np.int32)
    nq, stride = probes.shape
    if out is None:
        out = np.empty((nq, nprobe), np.int32)
    _check(lib().gpann_mask_probes_replicated(probes.ctypes.data_as(C.POINTER(C.c_int32)), nq, stride, nprobe, rep_begin.ctypes.data_as(C.POINTER(C.c_uint32)),
                                              rep_rank.ctypes.data_as(C.POINTER(C.c_int32)), len(rep_begin) - 1, rank, out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


class Comm:
    def __init__(self, backend, rank=None, world=None, addr="127.0.0.1", port=None, device=None, timeout_s=120):
        """rank=None: RANK / WORLD_SIZE / MASTER_ADDR / GPANN_COMM_PORT (default MASTER_PORT + 1) from the environment; device=None: LOCAL_RANK
        (pass the device the caller's buffers live on if that is another one)"""
        self.h = C.c_void_p()
        if rank is None:
            _check(lib().gpann_comm_init_from_env_on(backend, -1 if device is None else int(device), C.byref(self.h)))
        else:
            _check(lib().gpann_comm_init(rank, world, addr.encode(), int(port or 0), backend, int(device or 0), timeout_s, C.byref(self.h)))
        r, w, b = C.c_int(), C.c_int(), C.c_int()
        _check(lib().gpann_comm_info(self.h, C.byref(r), C.byref(w), C.byref(b)))
        self.rank, self.world, self.backend = r.value, w.value, b.value
        self.name = lib().gpann_comm_backend_name(self.h).decode()

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            lib().gpann_comm_destroy(self.h)
            self.h = C.c_void_p()

    def abort(self):
        """leave after an error: no waiting for outstanding collectives, the bootstrap connections are closed (the other ranks' wait() notices)"""
        if getattr(self, "h", None) is not None and self.h.value:
            lib().gpann_comm_abort(self.h)
            self.h = C.c_void_p()

    def wait(self, stream=None, timeout_s=600):
        """wait for the collectives enqueued on `stream` (RCCL).  Completed work is a success whatever the peers did afterwards; while the
        work is pending the other ranks are watched: raises GpannError when a rank has left, RCCL reports an asynchronous error or nothing
        completes within timeout_s (this rank's RCCL side is then aborted) — instead of a stream synchronise that would never return"""
        _check(lib().gpann_comm_wait(self.h, C.c_void_p(stream or 0), int(timeout_s)))

    def check_peers(self):
        """raises GpannError if a rank has hung up (gpann_comm_check_peers); never blocks"""
        lost = C.c_int(-1)
        _check(lib().gpann_comm_check_peers(self.h, C.byref(lost)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def barrier(self):
        _check(lib().gpann_comm_barrier(self.h))

    def max_double(self, x):
        v = C.c_double(float(x))
        _check(lib().gpann_comm_max_double(self.h, C.byref(v)))
        return v.value

    def allgather_host(self, a):
        a = np.ascontiguousarray(a)
        out = np.empty((self.world,) + a.shape, a.dtype)
        _check(lib().gpann_comm_allgather_host(self.h, _ptr(a), _ptr(out), a.nbytes))
        return out

    def broadcast_host(self, a, root=0):
        assert a.flags.c_contiguous
        _check(lib().gpann_comm_broadcast_host(self.h, _ptr(a), a.nbytes, root))
        return a

    def broadcast(self, buf, nbytes, root=0, stream=None):
        """buf: device pointer (RCCL) or numpy array (TCP)"""
        _check(lib().gpann_comm_broadcast(self.h, _ptr(buf), nbytes, root, C.c_void_p(stream or 0)))

    def allgather(self, send, recv, nbytes_per_rank, stream=None):
        _check(lib().gpann_comm_allgather(self.h, _ptr(send), _ptr(recv), nbytes_per_rank, C.c_void_p(stream or 0)))

    def gather_slots_scratch(self, n):
        return lib().gpann_comm_gather_slots_scratch(self.world, n)

    def gather_slots(self, dist, ids, n, out_dist, out_ids, scratch, id_map=None, stream=None):
        """per-slot lists of this rank -> every slot from its owner.  Arguments: device pointers (RCCL) or numpy arrays (TCP).
        id_map: local row id -> global point id, applied to this rank's ids before the exchange."""
        _check(lib().gpann_comm_gather_slots(self.h, _ptr(dist), _ptr(ids), n, _ptr(out_dist), _ptr(out_ids), _ptr(scratch),
                                             None if id_map is None else _ptr(id_map), C.c_void_p(stream or 0)))

    def merge_router(self, router, owner, device=0):
        """all ranks: Router trained over this rank's shards -> the Router over all shards (gpann_comm_merge_router)"""
        from . import Router
        owner = np.ascontiguousarray(owner, np.int32)
        merged = Router.__new__(Router)
        merged.h = C.c_void_p()
        _check(lib().gpann_comm_merge_router(self.h, router.h, owner.ctypes.data_as(C.POINTER(C.c_int32)), device, C.byref(merged.h)))
        merged.S, merged.d = router.S, router.d
        return merged
