/* gpann.h — C-ABI of libgpann_hip.so: the MI355X (gfx950) implementation of gp-ann's route + shard-scan hot path.
 *
 * This is the drop-in boundary.  The reference is a single-process C++ program with no FFI; the
 * entry points below are what a binding for this path would have to cover, one per reference interface:
 *
 *   gpann_shards_create            <- InvertedIndex::InvertedIndex(PointSet&, const Clusters&)      src/inverted_index.h:12-41
 *   gpann_scan_buckets             <- InvertedIndex::QueryBucket(float*, int k, int bucket)         src/inverted_index.h:62-75
 *                                     batched over (query, probe slot), as driven by small_scale_queries.cpp:149-155
 *   gpann_scan_merged              <- InvertedIndex::Query(float*, int k, probes, nprobe)           src/inverted_index.h:43-60
 *   gpann_ground_truth             <- ComputeGroundTruth(points, queries, k)                        src/recall.h:25-39
 *   gpann_shard_search_hits        <- RunInShardSearches(...) hit lists (exact scan instead of HNSW) src/shard_searches.cpp:11-104
 *   gpann_router_create            <- KMeansTreeRouter (trained tree, flattened as ExtractPoints)    src/kmeans_tree_router.cpp:174-201
 *   gpann_router_train             <- KMeansTreeRouter::Train / TrainRecursive + KMeans             src/kmeans_tree_router.cpp:9-91, src/kmeans.cpp:12-244
 *   gpann_router_query_batch       <- KMeansTreeRouter::Query(float* Q, int budget)                 src/kmeans_tree_router.cpp:93-130
 *   gpann_router_frequency_query_batch <- KMeansTreeRouter::FrequencyQuery                      src/kmeans_tree_router.cpp:132-172
 *   gpann_merge_dedup              <- DedupNeighbors(neighbors, k)                                  small_scale_queries.cpp:15-23
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns a gpann_status (0 = OK) and never
 * throws; outputs are caller-allocated.  `metric`: 0 = squared L2, 1 = MIPS distance 1 - <p,q> (the reference's
 * compile-time MIPS_DISTANCE switch, src/dist.cpp:75-81, is a runtime argument here).  Distances are evaluated in
 * strict IEEE binary32 in the reference's scalar operation order (src/dist.cpp:17-37, :42-52), so returned ids and
 * distances are bit-identical to the reference's CPU path built without -ffast-math/AVX.
 *
 * Memory spaces: pointer arguments documented as "host|device" are interpreted according to the GPANN_*_ON_DEVICE bits
 * in `flags`; all other pointers are host pointers.  `stream` is a hipStream_t (NULL = default stream).  Calls that
 * return results to host memory synchronise the stream before returning.
 *
 * There is no CPU fallback: if no gfx950 device is usable every entry point returns GPANN_ERR_NO_DEVICE.
 */
#ifndef GPANN_H
#define GPANN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gpann_status {
    GPANN_OK = 0,
    GPANN_ERR_INVALID = 1,     /* bad argument (null pointer, k == 0 or k > GPANN_MAX_K, shard id out of range ...) */
    GPANN_ERR_NO_DEVICE = 2,   /* no usable HIP device */
    GPANN_ERR_HIP = 3,         /* a HIP runtime call failed; see gpann_last_error() */
    GPANN_ERR_OOM = 4,         /* device or host allocation failed */
    GPANN_ERR_UNSUPPORTED = 5,
    GPANN_ERR_INTERNAL = 6     /* an internal consistency check failed; see gpann_last_error() */
} gpann_status;

enum {
    GPANN_METRIC_L2 = 0,
    GPANN_METRIC_MIPS = 1
};

enum {
    GPANN_QUERIES_ON_DEVICE = 1 << 0,  /* `queries` is a device pointer */
    GPANN_OUTPUT_ON_DEVICE = 1 << 1,   /* out_* are device pointers (no synchronisation on return) */
    GPANN_POINTS_ON_DEVICE = 1 << 2,   /* `points` is a device pointer (gpann_shards_create) */
    GPANN_STORE_PLAIN = 1 << 4,        /* gpann_shards_create: rows, offsets and permutation only — no filter support data (cells,
                                          bf16 planes); for stores that only feed training / partitioning.  Scans on it are exact-mode */
    GPANN_AUX_PASS = 1 << 3,           /* profiling label: run the identical scan kernel under a second symbol name so that
                                          auxiliary passes (ground truth) stay out of the hot path's per-kernel statistics */
    GPANN_SCAN_FOR_MERGE = 1 << 5,     /* gpann_scan_buckets: the caller only merges the slots of a query (DedupNeighbors,
                                          small_scale_queries.cpp:15-23, or one TopN over them, inverted_index.h:43-60).  Slot j of query q then
                                          holds the reference's per-slot list TRUNCATED after its last entry with distance <= tau_q, where
                                          tau_q = min over the query's slots of the slot's k-th distance (+inf for a slot with < k rows);
                                          the cut entries read (+inf, GPANN_NO_ID).  Every list of q has k distinct ids (ids are unique within a
                                          shard) at distance <= its own k-th, so the merged k-th distance is <= tau_q and no cut entry can be
                                          among the first k of the merge: the merged result is bit-identical to the one over the full lists.
                                          In filter mode tau_q bounds every shard of the query, not only the one that produced it — rows and
                                          cells beyond it are never evaluated. */
    GPANN_SCAN_OVERLAP = 1 << 7,       /* gpann_scan_buckets, filter mode, with GPANN_SCAN_DEFER (ignored without it, and in exact mode): the call's device
                                          work runs on three streams of the HANDLE instead of `stream` — the passes before the rest pass (tables,
                                          seed, shard-sample, head, sample and their re-scoring: short launches that leave most of the chip idle)
                                          on one, the rest pass's launch (it streams the store once: HBM-bound) on the second, what follows it on
                                          the third — so that the fronts of calls i + 1, i + 2 run BESIDE the rest pass of call i and the rest
                                          passes follow each other without a gap (every call in flight owns its device workspaces).  `stream` only orders the call's START (the call waits for what `stream` holds when it
                                          is issued); its results are in place when gpann_scan_finish returns, NOT when `stream` has drained.
                                          What the flag may NOT change is the result: every list entry is still the canonical distance of a real row.
                                          Kernels of different calls then share compute units, and on this part that needs two properties of the BUILT
                                          library, both checked (tests/test_capi_exports.py on the code objects; the first also asked of the runtime
                                          for the kernels a store launches — a build without it IGNORES this flag and says so on stderr):
                                          every matrix-core kernel is allocated all 256 vector registers and leaves its SIMDs together (nothing of
                                          another launch runs beside int8 matrix wavefronts), and the library contains no packed fp32 instruction (the
                                          results of v_pk_*_f32 were measured to go wrong in lanes 48 - 63 beside another kernel's int8 matrix
                                          instructions; scalar fp32 is not disturbed).  Round 5, before either: about one overlapped call in 500 lost a
                                          neighbour at d <= 64; since: 0 events in 1.8 million overlapped calls (profiles/r06/NOTES.md).
                                          gpann_scan_stats.overlapped says whether the flag took effect for a call. */
    GPANN_SCAN_DEFER = 1 << 6          /* gpann_scan_buckets: plan the call and ENQUEUE its work — uploads, kernels, the copies into host output
                                          arrays — then return; gpann_scan_finish completes it (waits, reads the pipeline's counters, redoes the call
                                          with the plain exact scan should the data have defeated the filter, statistics, timing).  TWO further
                                          calls may be issued on the handle before the finish: their host-side planning and uploads then run while
                                          the device works on the first (a call issued with three unfinished calls outstanding completes the oldest
                                          one itself).  Contract until the finish: the same `stream` for the calls, device queries in the layout the
                                          kernels read in place (GPANN_QUERIES_ON_DEVICE, d % 4 == 0, 16-byte aligned — otherwise the call simply
                                          completes before it returns), queries and output arrays left alone (device outputs: not reused by the
                                          next call). */
};

#define GPANN_MAX_K 64u                 /* top-k lists are held one entry per wavefront lane */
#define GPANN_NO_ID 0xFFFFFFFFu         /* id / row written to unused result slots; their distance is +inf */

typedef struct gpann_shards gpann_shards;   /* device-resident shard store (the InvertedIndex layout) */
typedef struct gpann_router gpann_router;   /* device-resident flattened k-means tree router */

/* Threads.  The reference's callers run Query / QueryBucket / FrequencyQuery from many threads at once, one query per call
 * (routes.cpp:194-200, shard_searches.cpp:48).  Every entry point here may be called from any thread; a handle (gpann_shards, gpann_router)
 * owns ONE set of device workspaces, so concurrent calls on the SAME handle take turns (a per-handle lock) — correct, and serial: the batch
 * entry points are how the GPU is kept busy.  Calls on different handles run concurrently (bench.py routes batch i + 1 on one handle and
 * stream while batch i is scanned on another). */
/* thread-local, human-readable description of the last failure in the calling thread */
const char* gpann_last_error(void);
const char* gpann_version(void);
/* ABI revision of this header; gpann_abi_version() is the library's.  It is raised whenever an exported signature or the layout of a struct a caller
 * allocates changes (6: gpann_scan_stats ends with the matrix-core issue counters, reserved3 became `overlapped`, gpann_scan_last_stats_sized added).
 * A caller that may meet a library of another revision compares the two and reads gpann_scan_stats through gpann_scan_last_stats_sized. */
#define GPANN_ABI_VERSION 6u
uint32_t gpann_abi_version(void);

/* Options.  Every knob that changes what a call does is a named option of the HANDLE (string values; names with or without the GPANN_
 * prefix).  A handle takes a snapshot of the process environment's GPANN_* variables when it is created — so a deployment's defaults can
 * still be given in the environment — and never reads the environment again: gpann_*_set_option is how a caller changes a knob afterwards
 * (value NULL: back to the built-in default), gpann_*_get_option reads it back ("" = not set).  Unknown names are GPANN_ERR_INVALID.  Options
 * marked "read at build" (gpann_option_read_at_build) shape the store / router while it is built: set them in the environment before the
 * handle is created (or, for a store built in two steps, before gpann_shards_build_filter).  Set options while no call is in flight on the
 * handle.  gpann_option_count / _name / _help enumerate the table (INTEGRATION.md lists it).  Only debug OUTPUT is still switched by the
 * environment where it is produced: GPANN_CELLS_DEBUG, GPANN_ROUTER_DEBUG, and the path GPANN_VERIFY_REPORT. */
uint32_t gpann_option_count(void);
const char* gpann_option_name(uint32_t i);
const char* gpann_option_help(uint32_t i);
int gpann_option_read_at_build(uint32_t i);
gpann_status gpann_device_count(int* count);

/* ------------------------------------------------------------------------------------------------ shard store */
/* Gathers the points of every shard into one shard-contiguous row-major array in HBM (row stride = d rounded up to
 * 4 floats), plus offsets[S+1] and permutation[row] -> global id, exactly the InvertedIndex layout.
 *   points          host|device, n x d float32 row-major
 *   ids_concat      host, concatenation of Clusters[0..S) (global ids, may repeat across shards)
 *   shard_offsets   host, S+1 prefix offsets into ids_concat
 */
gpann_status gpann_shards_create(const float* points, uint64_t n, uint32_t d, int metric, const uint32_t* ids_concat,
                                 const uint64_t* shard_offsets, uint32_t S, int device, int flags, gpann_shards** out);
gpann_status gpann_shards_destroy(gpann_shards* s);
gpann_status gpann_shards_set_option(gpann_shards* s, const char* name, const char* value);
gpann_status gpann_shards_get_option(const gpann_shards* s, const char* name, char* out, size_t cap);
/* Builds the filter support data (bf16 planes grouped by cell, radii) of a store created with GPANN_STORE_PLAIN — i.e. in a second step,
 * after the caller has released its own copy of the points: at 100M x 200 the rows are 80 GB and the planes 102 GB, and three copies
 * do not fit one GPU.  No-op if the data is already there; GPANN_ERR_UNSUPPORTED if the store's data is unfit (non-finite, d > 256).
 * If the data still cannot be built the status names the kind of failure — GPANN_ERR_OOM: an allocation (retry with less resident),
 * GPANN_ERR_HIP: a device / kernel error, GPANN_ERR_INTERNAL: the cell clustering produced no cells — and gpann_last_error() the step. */
gpann_status gpann_shards_build_filter(gpann_shards* s);
/* rows = sum of shard sizes; any out pointer may be NULL */
gpann_status gpann_shards_info(const gpann_shards* s, uint64_t* rows, uint32_t* d, uint32_t* S, int* metric, int* device);
/* copies offsets (S+1 x int64) / permutation (rows x u32) / clustered points (rows x d f32, unpadded) to host */
gpann_status gpann_shards_export(const gpann_shards* s, int64_t* offsets, uint32_t* permutation, float* clustered);
/* same for the row range [row_begin, row_begin + row_count) only (e.g. one shard): permutation / clustered may be NULL */
gpann_status gpann_shards_export_range(const gpann_shards* s, uint64_t row_begin, uint64_t row_count, uint32_t* permutation,
                                       float* clustered);

/* Batched QueryBucket.  For q < nq, j < nprobe: exact top-k of shard probes[q*probe_stride + j] (negative = skip).
 *   queries   host|device nq x d
 *   probes    host, int32
 *   out_ids   host|device [nq][nprobe][k] global ids, ascending by (distance, row); unused slots GPANN_NO_ID
 *   out_dist  host|device [nq][nprobe][k] (may be NULL); unused slots +inf
 *   out_rows  host|device [nq][nprobe][k] row index in the clustered layout = the reference's TopN key (may be NULL)
 *   elapsed_ms  (may be NULL) device time of the scan + merge kernels, measured with HIP events on `stream`
 */
gpann_status gpann_scan_buckets(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const int32_t* probes,
                                uint32_t probe_stride, uint32_t nprobe, uint32_t* out_ids, float* out_dist, uint32_t* out_rows,
                                int flags, void* stream, float* elapsed_ms);
/* Completes the oldest gpann_scan_buckets call issued with GPANN_SCAN_DEFER (no-op if there is none): on return its results are in place —
 * in the host arrays of a call without GPANN_OUTPUT_ON_DEVICE — gpann_scan_last_stats describes it and *elapsed_ms (may be NULL) is its kernel time. */
gpann_status gpann_scan_finish(gpann_shards* s, float* elapsed_ms);

/* Batched InvertedIndex::Query: ONE top-k over the union of the first nprobe probed shards (no dedupe), [nq][k]. */
gpann_status gpann_scan_merged(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const int32_t* probes,
                               uint32_t probe_stride, uint32_t nprobe, uint32_t* out_ids, float* out_dist, int flags, void* stream,
                               float* elapsed_ms);

/* RunInShardSearches with the exact scan: every query against every shard; for shard b and query q the global ids whose
 * distance <= distance_to_kth[q], in DESCENDING distance order (the reference pops a max-heap, shard_searches.cpp:78-89).
 *   out_ids    host [S][nq][k], unused slots GPANN_NO_ID;  out_count host [S][nq]
 */
gpann_status gpann_shard_search_hits(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const float* distance_to_kth,
                                     uint32_t* out_ids, uint32_t* out_count, int flags, void* stream, float* elapsed_ms);

/* ComputeGroundTruth: exact k-NN of every query over ALL n points, ids are global point ids, ascending (dist, id).
 * points: host|device.  Requires nothing to be resident; builds a temporary 1-shard store unless `s_all` (a store
 * created with S = 1 and ids = 0..n-1 IN ORDER: another row order would break ties by row, not by id, and is refused with GPANN_ERR_INVALID) is
 * passed, in which case `points` may be NULL. */
gpann_status gpann_ground_truth(gpann_shards* s_all, const float* points, uint64_t n, uint32_t d, int metric, const float* queries,
                                uint32_t nq, uint32_t k, uint32_t* out_ids, float* out_dist, int device, int flags, void* stream,
                                float* elapsed_ms);

/* Statistics of the most recent scan call on this store (for roofline accounting). */
typedef struct gpann_scan_stats {
    uint64_t pairs;              /* (query, point) distance evaluations requested by the caller: sum over tasks of shard rows */
    uint64_t algorithmic_bytes;  /* pairs * d * 4  +  tasks * k * 8    */
    uint64_t tasks;              /* (query, probe slot) pairs with a valid shard */
    uint64_t units;              /* workgroups launched by the scan kernel */
    uint64_t padded_pairs;       /* exact (VALU) distance evaluations actually executed (tile padding included) */
    uint64_t filter_pairs;       /* (query, row) pairs evaluated by the MFMA pre-filter (0 in exact mode) */
    uint64_t filter_candidates;  /* rows that passed the pre-filter and were re-scored exactly (all passes) */
    uint32_t filtered;           /* 1 if the MFMA-filtered pipeline produced the result, 0 if the plain exact scan did */
    uint32_t filter_plane_ksteps;/* 16-dimension k-steps a plane row of the filter is stored with (32 bytes each per row and plane); 0 in exact mode */
    float scan_ms;               /* scan kernels (exact mode: ScanTileKernel; filter mode: every kernel of the pipeline), HIP events */
    float merge_ms;              /* merge kernel(s), HIP events */
    float filter_ms;             /* of scan_ms: the MFMA filter kernels (threshold + filter launches of all passes) */
    uint32_t filter_dirty_tasks; /* tasks whose candidate bucket overflowed in some pass and whose shard was therefore evaluated exactly, row by row */
    uint64_t filter_rows;        /* rows streamed by the filter kernels: sum over their workgroups of the row segment (all passes) */
    uint64_t filter_rest_rows;   /* of filter_rows: the last ("rest") pass, the launch that streams nearly the whole store once */
    float filter_rest_ms;        /* that one FilterKernelCells launch alone, HIP events */
    uint32_t filter_rest_hi_only;/* 1 if it read only the hi plane (d * 2 bytes per row), 0 if hi + lo (d * 4) */
    uint64_t filter_two_plane_rows; /* of filter_rows: rows streamed by the launches that read both planes (shard-sample and head passes) */
    uint64_t filter_seed_rows;   /* rows evaluated exactly for the first lists (filter mode: tasks x seed rows) */
    uint32_t filter_rest_row_bytes; /* bytes the rest pass reads per row: 32 x k-steps (bf16 hi plane) or 32 x ceil(d / 32) (8-bit plane) + the norm word */
    uint32_t overlapped;         /* 1 if GPANN_SCAN_OVERLAP took effect: the call's device work ran on the store's own streams, beside other calls */
    uint64_t filter_mfma_bf16;   /* v_mfma_f32_32x32x16_bf16 wave-instructions the filter kernels ISSUED (32 x 32 x 16 x 2 flop each), counted by the kernels */
    uint64_t filter_mfma_i8;     /* v_mfma_i32_32x32x32_i8 wave-instructions issued (32 x 32 x 32 x 2 integer ops each): H.P' of every product + L.P' where it was not skipped */
    uint64_t filter_rest_mfma_i8;/* of filter_mfma_i8: the rest pass alone (0 when it ran on the bf16 plane) */
} gpann_scan_stats;
gpann_status gpann_scan_last_stats(const gpann_shards* s, gpann_scan_stats* out);   /* writes sizeof(gpann_scan_stats) of THIS header's revision */
/* The same for a caller whose header may be of another revision than the library: gpann_scan_stats only ever grows at its end, so the first
 * min(out_size, the library's size) bytes are the fields both sides know; the rest of `out` is zeroed.  Pass sizeof(gpann_scan_stats). */
gpann_status gpann_scan_last_stats_sized(const gpann_shards* s, void* out, size_t out_size);

/* Scan mode of a store: 0 = exact VALU scan of every row (default); 1 = the same exact results with a matrix-core
 * pre-filter (bf16x3 MFMA bound + exact re-scoring of the candidates, gp-ann_amd/csrc/scan_filter.h).  Both modes return
 * bit-identical ids and distances; mode 1 silently uses the plain scan where the filter does not apply (d > 256, non-finite
 * data, candidate overflow).  The environment variable GPANN_SCAN_MODE=exact|filter overrides the handle setting. */
gpann_status gpann_shards_set_scan_mode(gpann_shards* s, int mode);

/* ------------------------------------------------------------------------------------------------------ router */
/* Flattened KMeansTreeRouter.  Nodes are numbered per shard in BFS order (the order of ExtractPoints); the centroids of
 * node u are rows node_cent_first[u] .. node_cent_first[u+1]) of `centroids`; child i of node u (it exists iff
 * i < node_child_count[u]) is node node_child_first[u] + i and belongs to centroid i (kmeans_tree_router.cpp:119-121).
 * All arrays are host pointers and are copied. */
gpann_status gpann_router_create(uint32_t d, int metric, uint32_t S, uint32_t n_nodes, const int32_t* node_shard,
                                 const uint32_t* node_cent_first, const uint32_t* node_child_first, const uint32_t* node_child_count,
                                 const uint32_t* shard_root, const float* centroids, int device, gpann_router** out);
gpann_status gpann_router_destroy(gpann_router* r);
gpann_status gpann_router_set_option(gpann_router* r, const char* name, const char* value);
gpann_status gpann_router_get_option(const gpann_router* r, const char* name, char* out, size_t cap);

/* KMeansTreeRouter::Train (src/kmeans_tree_router.cpp:9-91) on the GPU, over the rows of a shard store: shard b of the
 * store is Clusters[b] in the reference's call, in the same order.  All k-means problems of one tree level run as one
 * batch of launches (gp-ann_amd/csrc/kmeans.hip); sampling (std::sample / std::mt19937, seeds 555*b and seed+i), the 20
 * Lloyd rounds, the sequential point-order centroid sums, empty-cluster removal, std::partition by min_cluster_size and
 * the budget rules are the reference's, so the tree is the one the reference builds (bit-identical while every node has
 * at most max(5e6, 200*centroids*d) points; beyond that the reference's own sums are unordered atomics and this build
 * takes its block partials in block order).
 * An empty shard gets a root without centroids (additive; lets a rank that stores only its own shards train their subtrees).
 *   n_points          PointSet::n of the reference's call (the denominator of the per-shard budget share, :28)
 *   num_centroids, min_cluster_size, budget     KMeansTreeRouterOptions (kmeans_tree_router.h:6-11) */
gpann_status gpann_router_train(gpann_shards* shards, uint64_t n_points, uint32_t num_centroids, uint32_t min_cluster_size, int64_t budget,
                                gpann_router** out);

/* RandomSample(points, k, seed) + KMeans(points, centroids) — reference src/kmeans.cpp:207-244, the first step of every k-means
 * partitioner (partition.cpp:23-26; src/partitioning.cpp:19-26, :94-98) — over all rows of `shards` as ONE point set (row r of the
 * store = point r of the call; build the store with a single identity shard).  Same engine and the same guarantees as
 * gpann_router_train.
 *   out_partition   host, rows x int32: cluster of every row after empty clusters were removed (may be NULL)
 *   out_centroids   host, k x d (the first *out_k rows are written; may be NULL)
 *   out_k           surviving centroids (PointSet::n of `centroids` after the call) */
gpann_status gpann_kmeans(gpann_shards* shards, uint32_t k, int seed, int32_t* out_partition, float* out_centroids, uint32_t* out_k);

/* Size and contents of a router's flattened tree (the arrays gpann_router_create takes; host outputs, any may be NULL). */
gpann_status gpann_router_info(const gpann_router* r, uint32_t* d, int* metric, uint32_t* S, uint32_t* n_nodes, uint64_t* n_centroids);
gpann_status gpann_router_export(const gpann_router* r, int32_t* node_shard, uint32_t* node_cent_first, uint32_t* node_child_first,
                                 uint32_t* node_child_count, uint32_t* shard_root, float* centroids);

/* Counters of a router handle since its creation (additive; no reference counterpart).  `bound_selfcheck` and `verify_failures` must stay 0:
 * the first counts query blocks whose matrix-core bound path (visit-everything case) left a minimum unset and were re-evaluated
 * canonically, the second the blocks that differed from the canonical evaluation under GPANN_ROUTER_VERIFY=1 (those calls return
 * GPANN_ERR_INTERNAL). */
typedef struct gpann_router_stats {
    uint64_t bound_calls;       /* query blocks answered through the matrix-core bound path */
    uint64_t bound_selfcheck;   /* ... whose self-check fired (canonical evaluation of every pair instead) */
    uint64_t bound_overflow;    /* ... whose candidate list overflowed (canonical evaluation instead) */
    uint64_t verified;          /* GPANN_ROUTER_VERIFY=1: blocks compared, minimum by minimum, with the canonical evaluation */
    uint64_t verify_failures;
    uint64_t host_sorted;       /* queries whose final order came from the host's std::sort (equal min_dist values) */
    uint64_t heap_redo;         /* queries replayed with the real priority queue after a sorted / threshold replay */
    double last_bound_ms;       /* most recent call: HIP-event time of its RouterBoundKernel launches (pass 1 + margins + pass 2) */
    double last_bound_flops;    /* most recent call: matrix-core flops of those launches, 2 passes x 3 bf16 products x 2 nq C d */
    uint64_t probed;            /* GPANN_ROUTER_PROBE=1 (debugging aid): blocks whose bound pipeline was run twice on the same inputs ... */
    uint64_t probe_failures;    /* ... and whose two runs differed in any bit (maxima, thresholds, candidate set, minima): GPANN_ERR_INTERNAL */
    /* the bounded replay (search budget inside the tree): distance table from the matrix cores, canonical evaluation of the cut-off band and
     * of the per-shard minimum candidates only; last_bound_ms / last_bound_flops then describe its table launches */
    uint64_t table_blocks;         /* query blocks answered through it */
    uint64_t table_queries;        /* ... queries in those blocks */
    uint64_t table_band_nodes;     /* ... tree nodes inside the cut-off band (they got canonical keys) */
    uint64_t table_exact;          /* ... canonical distance evaluations (band keys + minimum candidates) */
    uint64_t table_band_overflow;  /* ... queries whose band did not fit (replayed on a canonical row, counted in heap_redo too) */
    uint64_t table_rescans;        /* ... (query, shard) ranges read twice because one lane held two minimum candidates */
} gpann_router_stats;
gpann_status gpann_router_get_stats(const gpann_router* r, gpann_router_stats* out);

/* Batched KMeansTreeRouter::Query.  The best-first walk (all distance evaluations, the priority queue and the budget
 * rule) runs on the GPU.  The final order (std::sort over min_dist, kmeans_tree_router.cpp:125-127) is computed on the device for
 * the queries whose min_dist values are pairwise different (any correct sort gives the reference's permutation) and, for queries with
 * equal values, by that very std::sort call on the host, so tie order is inherited from libstdc++.
 *   queries       host|device nq x d
 *   out_probes    host [nq][S] shard ids by ascending min_dist            (may be NULL)
 *   out_min_dist  host [nq][S]; FLT_MAX for shards whose root was never opened (may be NULL)
 */
gpann_status gpann_router_query_batch(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, int32_t* out_probes,
                                      float* out_min_dist, int flags, void* stream, float* elapsed_ms);

/* The same, returning only the first n_top shards of every query's order: out_probes host [nq][n_top], 1 <= n_top <= S (additive: a caller
 * that probes nprobe shards does not need — and does not have to copy back — the other S - nprobe columns). */
gpann_status gpann_router_query_batch_top(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t n_top, int32_t* out_probes,
                                          float* out_min_dist, int flags, void* stream, float* elapsed_ms);

/* Batched KMeansTreeRouter::FrequencyQuery (kmeans_tree_router.cpp:132-172): the same walk, plus for every query the
 * num_voting lexicographically smallest (distance, shard) pairs over all evaluated centroids, ascending (what
 * FrequencyQueryData holds after Init()).  out_nn_* are host [nq][num_voting]; out_nn_count host [nq]. */
gpann_status gpann_router_frequency_query_batch(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t num_voting,
                                                float* out_min_dist, float* out_nn_dist, uint32_t* out_nn_shard, uint32_t* out_nn_count,
                                                int flags, void* stream, float* elapsed_ms);

/* ------------------------------------------------------------------------------------------------------- merge */
/* DedupNeighbors over per-slot results: for each query, the first `nprobe_used` slots of in_* [nq][nprobe][k] are
 * concatenated, sorted by id, de-duplicated on (dist,id), sorted by (dist,id) and cut to k.  Host arrays.
 * out_* [nq][k]; out_count[nq] (may be NULL). */
gpann_status gpann_merge_dedup(const float* in_dist, const uint32_t* in_ids, uint32_t nq, uint32_t nprobe, uint32_t nprobe_used,
                               uint32_t k, float* out_dist, uint32_t* out_ids, uint32_t* out_count);

/* ---------------------------------------------------------------------------------- synthetic inputs (bench support) */
/* Fills `out` (DEVICE pointer, nrows x d) with rows row0.. of the Gaussian-mixture generator of include/gpann_synth.h. */
gpann_status gpann_synth_rows_device(uint64_t seed, uint64_t center_seed, uint32_t n_centers, float sigma, uint64_t row0,
                                     uint64_t nrows, uint32_t d, float* out, void* stream);
/* For every row of `points` (DEVICE, n x d) the index of the nearest of `centroids` (DEVICE, c x d) under `metric`
 * (first minimum wins, kmeans.cpp:12-23 Top1Neighbor) -> out_assign (DEVICE int32[n]).  Used to shard synthetic data. */
gpann_status gpann_nearest_centroid_device(const float* points, uint64_t n, uint32_t d, int metric, const float* centroids, uint32_t c,
                                           int32_t* out_assign, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPANN_H */
