/* gpann_oracle.h — C-ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference's route + shard-scan path
 * (larsgottesbueren/gp-ann @ 2024-10-20).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.  The product (gp-ann_amd/) never links, imports or
 * calls anything in oracle/.
 *
 * PARITY PIN STATUS: partially pinned.  The reference has no tests, golden vectors or fixtures, and
 * its hot-path sources do not compile here without third-party headers that are absent (parlaylib, hnswlib:
 * src/dist.cpp:3, src/inverted_index.h:27, src/kmeans_tree_router.cpp:5, src/recall.h:8).  What DOES compile from
 * the reference's own files — src/topn.h, src/defs.{h,cpp}, src/metis_io.cpp — is built into oracle/_ref/ref_core
 * (see oracle/Makefile) and pins TopN selection/tie-break/Take order and the Partition/Clusters/Cover conversions.
 * The distance functions, the IVF scan, ground truth/recall and KMeansTreeRouter::{Train,Query,FrequencyQuery} are
 * restated from the cited lines and are "parity unpinned" beyond that.
 *
 * metric: 0 = squared L2 (reference default build), 1 = MIPS "distance" 1 - <p,q> (reference -DMIPS_DISTANCE).
 */
#ifndef GPANN_ORACLE_H
#define GPANN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- distance (src/dist.cpp:9-52,75-81; scalar branch, strict IEEE, no FMA) -------------------------------- */
float orc_distance(const float* p, const float* q, uint32_t d, int metric);
void orc_distance_many(const float* points, uint64_t n, uint32_t d, int metric, const float* q, float* out);

/* ---- TopN (src/topn.h:6-32): n lexicographically smallest (dist,id), returned ascending -------------------- */
/* returns the number of entries written (min(n_items, n)) */
uint32_t orc_topn(const float* dists, const uint32_t* ids, uint64_t n_items, uint32_t n, float* out_dist, uint32_t* out_ids);

/* ---- InvertedIndex (src/inverted_index.h) ------------------------------------------------------------------- */
/* ctor :12-41.  ids_concat/cluster_offsets = the Clusters as CSR.  out_offsets has S+1 int32 entries. */
void orc_ivf_build(const float* points, uint32_t d, const uint32_t* ids_concat, const uint64_t* cluster_offsets, uint32_t S,
                   float* out_clustered, int32_t* out_offsets, uint32_t* out_permutation);
/* QueryBucket :62-75.  Returns #results (<= k).  out_rows (optional) = pre-remap row indices. */
uint32_t orc_ivf_query_bucket(const float* clustered, uint32_t d, int metric, const int32_t* offsets, const uint32_t* permutation,
                              const float* Q, uint32_t k, int32_t bucket, float* out_dist, uint32_t* out_ids, uint32_t* out_rows);
/* Query :43-60 (one TopN over the first nprobe entries of probes). */
uint32_t orc_ivf_query(const float* clustered, uint32_t d, int metric, const int32_t* offsets, const uint32_t* permutation,
                       const float* Q, uint32_t k, const int32_t* probes, uint32_t nprobe, float* out_dist, uint32_t* out_ids);
/* batched QueryBucket: for q < nq, j < nprobe: bucket probes[q*probe_stride + j] (negative = skip, count 0).
 * Outputs [nq][nprobe][k]; unused slots: dist = +inf, id = 0xFFFFFFFF.  out_count [nq][nprobe]. */
void orc_ivf_query_bucket_batch(const float* clustered, uint32_t d, int metric, const int32_t* offsets, const uint32_t* permutation,
                                const float* queries, uint32_t nq, uint32_t k, const int32_t* probes, uint32_t probe_stride,
                                uint32_t nprobe, float* out_dist, uint32_t* out_ids, uint32_t* out_count, int nthreads);

/* ---- ground truth / recall (src/recall.h) -------------------------------------------------------------------- */
/* ComputeGroundTruth :25-39; out [nq][k] ascending (dist,id); requires n >= k */
void orc_ground_truth(const float* points, uint64_t n, uint32_t d, int metric, const float* queries, uint32_t nq, uint32_t k,
                      float* out_dist, uint32_t* out_ids, int nthreads);
/* ComputeDistanceToKthNeighbor :10-23 */
void orc_distance_to_kth(const float* points, uint64_t n, uint32_t d, int metric, const float* queries, uint32_t nq, uint32_t k,
                         float* out_kth, int nthreads);
/* ConvertGroundTruthToDistanceToKthNeighbor :74-137.  gt_* are [nq][k_stored] and are UPDATED in place (distances
 * recomputed, first k re-sorted by distance if needed).  Returns the number of lists that needed re-sorting. */
uint64_t orc_convert_ground_truth(float* gt_dist, uint32_t* gt_ids, uint32_t nq, uint32_t k_stored, uint32_t k,
                                  const float* points, uint32_t d, int metric, const float* queries, float* out_kth);
/* Recall :139-150.  neighbors of query q are dists[offsets[q] .. offsets[q+1]) */
double orc_recall(const float* dists, const uint64_t* offsets, uint32_t nq, const float* kth, uint32_t k);

/* ---- merge (small_scale_queries.cpp:15-23 DedupNeighbors) ---------------------------------------------------- */
/* per query: sort by id, unique (whole pair), sort by (dist,id), truncate to k.  in: CSR (offsets nq+1).
 * out: [nq][k] padded with (+inf, 0xFFFFFFFF); out_count[nq]. */
void orc_dedup_neighbors(const float* dists, const uint32_t* ids, const uint64_t* offsets, uint32_t nq, uint32_t k,
                         float* out_dist, uint32_t* out_ids, uint32_t* out_count);

/* ---- k-means (src/kmeans.cpp:12-50,72-149,207-244) ----------------------------------------------------------- */
/* RandomSample(seed) + KMeans: 20 Lloyd rounds; returns #centroids left (empty ones removed);
 * out_partition[n]; out_centroids (optional) [k*d] (first `return value` rows valid). */
uint32_t orc_kmeans(const float* points, uint64_t n, uint32_t d, int metric, uint32_t k, int seed, int32_t* out_partition,
                    float* out_centroids, int nthreads);

/* ---- KMeansTreeRouter (src/kmeans_tree_router.{h,cpp}) ------------------------------------------------------- */
typedef struct orc_kmtr orc_kmtr;
/* Train :9-78 */
orc_kmtr* orc_kmtr_train(const float* points, uint64_t n, uint32_t d, int metric, const uint32_t* ids_concat,
                         const uint64_t* cluster_offsets, uint32_t S, uint64_t num_centroids, uint64_t min_cluster_size,
                         int64_t budget, int nthreads);
/* flat layout: nodes numbered per shard in BFS order (= ExtractPoints order :174-201); children of a node are
 * consecutive; child i belongs to centroid i. */
orc_kmtr* orc_kmtr_from_flat(uint32_t d, int metric, uint32_t S, uint32_t n_nodes, const int32_t* node_shard,
                             const uint32_t* node_cent_first /*n_nodes+1*/, const uint32_t* node_child_first,
                             const uint32_t* node_child_count, const uint32_t* shard_root /*S*/, const float* centroids);
void orc_kmtr_sizes(const orc_kmtr* t, uint32_t* d, uint32_t* S, uint32_t* n_nodes, uint64_t* n_centroids);
void orc_kmtr_export(const orc_kmtr* t, int32_t* node_shard, uint32_t* node_cent_first, uint32_t* node_child_first,
                     uint32_t* node_child_count, uint32_t* shard_root, float* centroids);
/* ExtractPoints :174-201 : shard id per centroid (centroid rows are the flat centroid array itself) */
void orc_kmtr_extract_partition(const orc_kmtr* t, int32_t* out_shard_of_centroid);
/* Query :93-130.  out_probes[S]; out_min_dist[S] optional */
void orc_kmtr_query(const orc_kmtr* t, const float* Q, int budget, int32_t* out_probes, float* out_min_dist);
void orc_kmtr_query_batch(const orc_kmtr* t, const float* queries, uint32_t nq, int budget, int32_t* out_probes,
                          float* out_min_dist, int nthreads);
/* FrequencyQuery :132-172 (+ Init sort, kmeans_tree_router.h:22).  out_nn_* have room for num_voting entries;
 * returns the number of near neighbours written. */
uint32_t orc_kmtr_frequency_query(const orc_kmtr* t, const float* Q, int budget, int num_voting, float* out_min_dist,
                                  float* out_nn_dist, uint32_t* out_nn_shard);
/* FrequencyQueryData::Query kmeans_tree_router.h:23-44 */
void orc_frequency_probes(const float* min_dist, uint32_t S, const uint32_t* nn_shard, uint32_t nn_count, int num_voting,
                          int32_t* out_probes);
void orc_kmtr_free(orc_kmtr* t);

/* ---- defs.cpp helpers ---------------------------------------------------------------------------------------- */
/* ConvertPartitionToClusters defs.cpp:30-37 as CSR; returns S.  out_offsets needs max(partition)+2 entries. */
uint32_t orc_partition_to_clusters(const int32_t* partition, uint64_t n, uint32_t* out_ids_concat, uint64_t* out_offsets);

/* ---- synthetic inputs (include/gpann_synth.h) ---------------------------------------------------------------- */
void orc_synth_rows(uint64_t seed, uint64_t center_seed, uint32_t n_centers, float sigma, uint64_t row0, uint64_t nrows,
                    uint32_t d, float* out, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
