/* gpann_comm.h — C-ABI of libgpann_comm.so: the exchange steps of the multi-GPU route + shard-scan path.
 *
 * One process per GPU, shards are the unit of distribution (rank = owner of a set of shards).  What this library replaces in the
 * reference is the MPI layer of its (unfinished) distributed driver:
 *
 *   gpann_comm_init / _init_from_env   <- MPI_Init, MPI_Comm_rank, MPI_Comm_size          distributed_bench.cpp:27-31
 *   gpann_comm_barrier                 <- MPI_Barrier                                     distributed_bench.cpp:45,52
 *   gpann_comm_max_double              <- MPI_Reduce(MPI_MAX) of the elapsed time         distributed_bench.cpp:53-55
 *   gpann_comm_broadcast               <- every rank reading the query file itself        distributed_bench.cpp:39 (here: rank 0 reads, RCCL broadcast)
 *   gpann_comm_allgather               <- (probe lists of the routed query slices)        src/distributed_query_benchmark.h:104-108
 *   gpann_comm_gather_slots            <- the request / response queues that carry a query to the owners of its probed shards and the
 *                                         neighbours back                                src/distributed_query_benchmark.h:110-208
 *   gpann_lpt_owner, gpann_mask_probes <- "partition[point] == rank" (one shard per rank) src/distributed_query_benchmark.h:49-53
 *   gpann_comm_merge_router            <- LoadRouter on every rank                        src/distributed_query_benchmark.h:99-101
 *
 * Backends: GPANN_COMM_RCCL — device buffers, RCCL collectives over xGMI on the caller's HIP stream (ncclBroadcast / ncclAllGather);
 * GPANN_COMM_TCP — host buffers over plain sockets through rank 0 (CPU-only tests of the same exchange logic; never chosen silently).
 * Both bootstrap over one TCP connection per rank to rank 0 (address / port: arguments, or MASTER_ADDR and GPANN_COMM_PORT, default
 * MASTER_PORT + 1, as exported by torch.distributed.run), which also carries the host-side scalars (barrier, max of a double) and,
 * for RCCL, the ncclUniqueId.
 *
 * Conventions as in gpann.h: status codes, no exceptions, caller-allocated outputs; gpann_comm_last_error() describes the last failure
 * in the calling thread.
 */
#ifndef GPANN_COMM_H
#define GPANN_COMM_H

#include <stddef.h>
#include <stdint.h>

#include "gpann.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GPANN_COMM_RCCL = 0,
    GPANN_COMM_TCP = 1
};

typedef struct gpann_comm gpann_comm;

const char* gpann_comm_last_error(void);

/* device: HIP device of this rank (RCCL backend; ignored by TCP) — the communicator lives there, and on return it IS the calling thread's
 * current HIP device (a rank process that has not touched HIP yet selects its GPU through this call; a caller that keeps its buffers on
 * another device must pass THAT device, or call hipSetDevice itself afterwards).  timeout_s: how long to wait for the other ranks. */
gpann_status gpann_comm_init(int rank, int world, const char* master_addr, int port, int backend, int device, int timeout_s, gpann_comm** out);
/* RANK, WORLD_SIZE, MASTER_ADDR, GPANN_COMM_PORT (default MASTER_PORT + 1), LOCAL_RANK (device), GPANN_COMM_TIMEOUT (seconds to wait for
 * the other ranks, default 600: the first `import torch` of a rank on a fresh machine can take minutes) from the environment */
gpann_status gpann_comm_init_from_env(int backend, gpann_comm** out);
/* the same with the HIP device given by the caller (device < 0: LOCAL_RANK) — for callers that place their buffers on another device than
 * LOCAL_RANK names (GPANN_DIST_DEVICE, GPANN_BENCH_FORCE_DEVICE): the communicator has to live where the buffers do */
gpann_status gpann_comm_init_from_env_on(int backend, int device, gpann_comm** out);
gpann_status gpann_comm_destroy(gpann_comm* c);
/* Failure handling (the reference relies on mpirun tearing the job down; distributed_bench.cpp has none).
 * gpann_comm_abort: for a rank that leaves after an error — the communicator is torn down WITHOUT waiting for outstanding collectives
 * (ncclCommAbort; ncclCommDestroy may block on them), the bootstrap connections are closed, which is how the other ranks notice; frees c.
 * gpann_comm_wait: wait for the work enqueued on `stream` (RCCL backend; TCP: nothing to wait for) for at most timeout_s seconds.  Work that
 * HAS COMPLETED is a success whatever the other ranks did afterwards (a peer that finished the same collective and left a moment earlier is
 * not an error of this call).  While the work is pending the bootstrap connections and RCCL's asynchronous error state are watched: when a
 * rank has hung up, RCCL reports an error or the time is over, this rank's RCCL side is aborted — the collective it waits for would never
 * complete — and an error is returned; the handle then stays valid for gpann_comm_abort / gpann_comm_destroy only.  Use it instead of a bare
 * stream synchronise after gpann_comm_broadcast / _allgather / _gather_slots.
 * gpann_comm_check_peers: has a rank hung up?  GPANN_OK and *lost_rank = -1 if all bootstrap connections are alive; otherwise
 * GPANN_ERR_INTERNAL and the rank (rank 0 sees every rank, the others only rank 0).  Never blocks, touches neither the stream nor RCCL. */
gpann_status gpann_comm_abort(gpann_comm* c);
gpann_status gpann_comm_wait(gpann_comm* c, void* stream, int timeout_s);
gpann_status gpann_comm_check_peers(gpann_comm* c, int* lost_rank /* may be NULL */);
gpann_status gpann_comm_info(const gpann_comm* c, int* rank, int* world, int* backend);
/* "rccl <version>" / "tcp": what the JSON line of bench.py reports as the backend that actually ran */
const char* gpann_comm_backend_name(const gpann_comm* c);

gpann_status gpann_comm_barrier(gpann_comm* c);
gpann_status gpann_comm_max_double(gpann_comm* c, double* value /* in: this rank's; out: max over ranks */);
/* small host buffers through the bootstrap connections (setup: shapes, tree parts): recv holds world * bytes */
gpann_status gpann_comm_allgather_host(gpann_comm* c, const void* send, void* recv, size_t bytes);
gpann_status gpann_comm_broadcast_host(gpann_comm* c, void* buf, size_t bytes, int root);

/* Data path.  RCCL: device pointers, enqueued on `stream` (hipStream_t; the call returns without synchronising).  TCP: host pointers. */
gpann_status gpann_comm_broadcast(gpann_comm* c, void* buf, size_t bytes, int root, void* stream);
gpann_status gpann_comm_allgather(gpann_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream);

/* The result exchange.  dist / ids: this rank's per-slot lists [n] (n = queries * probes * k) with (+inf, GPANN_NO_ID) wherever the
 * probed shard belongs to another rank — a slot is non-empty on exactly one rank.  out_dist / out_ids [n]: every slot from its owner.
 * One all-gather of the (dist, id) lists + an owner-selection pass (device kernel / host loop).  `scratch` must hold
 * gpann_comm_gather_slots_scratch(world, n) bytes in the backend's memory space. */
size_t gpann_comm_gather_slots_scratch(int world, size_t n);
/* id_map (may be NULL; same memory space as ids): ids[i] != GPANN_NO_ID is replaced by id_map[ids[i]] before the exchange — a rank's store
 * is built over its own rows, so its scan reports LOCAL row ids; the map holds their global point ids. */
gpann_status gpann_comm_gather_slots(gpann_comm* c, const float* dist, const uint32_t* ids, size_t n, float* out_dist, uint32_t* out_ids, void* scratch,
                                     const uint32_t* id_map, void* stream);

/* Testing hook: the DEVICE half of gpann_comm_gather_slots for a fabricated world, without a communicator — what a 1-GPU box can check of the
 * RCCL path with more than one rank.  rank_dist / rank_ids [world][n] (host): what each rank would contribute, ids still LOCAL when id_maps is
 * given; id_maps (host, may be NULL): the ranks' id maps back to back, rank r's at id_maps[id_map_begin[r] .. id_map_begin[r + 1]).  Runs
 * MapIdsKernel on every rank's block and SelectSlotsKernel over the gathered buffer on `device`; out_dist / out_ids [n] (host). */
gpann_status gpann_comm_debug_select_slots(int world, size_t n, const float* rank_dist, const uint32_t* rank_ids, const uint32_t* id_maps,
                                           const uint64_t* id_map_begin, float* out_dist, uint32_t* out_ids, int device);

/* ---- placement helpers (host) ---- */
/* Longest-processing-time bin packing of S shards onto `world` ranks by descending size (ties: lower shard id first; the lighter
 * rank, then the lower rank, takes the shard).  owner[S], load[world] (rows per rank; may be NULL). */
gpann_status gpann_lpt_owner(const uint64_t* shard_rows, uint32_t S, uint32_t world, int32_t* owner, uint64_t* load);
/* out[q][j] = probes[q][j] if owner[probes[q][j]] == rank else -1 (the probe slots this rank scans) */
gpann_status gpann_mask_probes(const int32_t* probes, uint32_t nq, uint32_t probe_stride, uint32_t nprobe, const int32_t* owner, uint32_t S, int rank,
                               int32_t* out /* [nq][nprobe] */);

/* ---- replicas: query-load balance (the reference MODELS replicas of the slowest shard when it predicts QPS, src/route_search_combination.cpp:149-183;
 * here the same greedy rule produces a real placement) ----
 * shard_work[b]: the work shard b attracts per batch, e.g. (queries routed to b) x rows_b from a routed sample batch.  Every shard starts
 * with one replica; while the heaviest rank exceeds max_imbalance x the mean, the shard with the largest work PER REPLICA gets one more
 * replica (`assigned_hosts[max_shard]++`, :181-182), as long as some rank that does not hold it yet has room (capacity_rows per rank);
 * replicas are placed LPT by work per replica.  The tasks of a shard are split evenly over its replicas (gpann_mask_probes_replicated:
 * query q goes to replica q mod r_b).
 *   rep_begin[S+1], rep_rank[<= S * world]: CSR of the ranks holding each shard; rep_rank[rep_begin[b]] is the shard's primary (trains its subtree)
 *   rank_work[world], rank_rows[world] (may be NULL): per-rank totals of the returned placement */
gpann_status gpann_plan_replicas(const uint64_t* shard_rows, const double* shard_work, uint32_t S, uint32_t world, uint64_t capacity_rows,
                                 double max_imbalance, uint32_t* rep_begin, int32_t* rep_rank, double* rank_work, uint64_t* rank_rows);
/* out[q][j] = probes[q][j] if this rank is replica (q mod r_b) of shard b = probes[q][j], else -1 */
gpann_status gpann_mask_probes_replicated(const int32_t* probes, uint32_t nq, uint32_t probe_stride, uint32_t nprobe, const uint32_t* rep_begin,
                                          const int32_t* rep_rank, uint32_t S, int rank, int32_t* out /* [nq][nprobe] */);

/* ---- router parts ---- */
/* Every rank trains the subtrees of ITS shards (gpann_router_train on a store whose other shards are empty gives those shards an
 * empty root).  This call all-gathers the parts through the bootstrap connections and builds, on every rank, the router over all S
 * shards: shard b's subtree is taken from rank owner[b].  The result is the tree a single process would have trained. */
gpann_status gpann_comm_merge_router(gpann_comm* c, gpann_router* mine, const int32_t* owner, int device, gpann_router** out);

#ifdef __cplusplus
}
#endif
#endif /* GPANN_COMM_H */
