// gpann_host.h — the host-side C++ mirror of the reference's interface for the route + shard-scan path.
//
// Same type names, member names, function signatures and error behaviour as the reference headers (cited per
// declaration), so that the reference's own drivers (run_query_attribution.cpp, small_scale_queries.cpp) compile against
// this header set unchanged; the bodies are new and route every distance-heavy step through libgpann_hip.so (C-ABI in
// include/gpann.h) on an MI355X.  Host-only logic (training, text formats, attribution arithmetic) stays on the host as in
// the reference.  The thin headers defs.h / dist.h / topn.h / inverted_index.h / kmeans_tree_router.h / recall.h /
// shard_searches.h / routes.h / route_search_combination.h / points_io.h / metis_io.h / kmeans.h next to this file
// forward here.
//
// Differences from the reference, all additive or forced by the missing third-party code:
//   * the metric is a run-time switch (SetMetricMIPS) as well as the reference's compile-time -DMIPS_DISTANCE;
//   * batched entry points (QueryBatch, QueryBucketBatch) next to the one-query-per-call signatures, which remain and
//     run as batches of one;
//   * RunInShardSearches performs the exact scan instead of building hnswlib indices (hnswlib is not part of the
//     reference checkout); it returns one ShardSearch (ef_search = 0 meaning "exact") instead of nine identical ones;
//   * HNSW/Pyramid/SPANN routers (hnsw_router.h) are out of scope.
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <fstream>
#include <functional>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

struct gpann_shards;
struct gpann_router;

// ---------------------------------------------------------------------------------------------------- src/topn.h:6-32
struct TopN {
    using value_type = std::pair<float, uint32_t>;
    explicit TopN(size_t n) : n(n) {}
    size_t n;
    std::priority_queue<value_type> pq;
    void Add(value_type e);
    std::vector<value_type> Take();   // ascending
    const value_type& Top() const { return pq.top(); }
};

// --------------------------------------------------------------------------------------------------- src/defs.h:10-108
struct PointSet {
    std::vector<float> coordinates;
    size_t d = 0, n = 0;
    float* GetPoint(size_t i) { return &coordinates[i * d]; }
    const float* GetPoint(size_t i) const { return &coordinates[i * d]; }
    void Drop() { std::vector<float>().swap(coordinates); }
    void Alloc() { coordinates.resize(n * d, 0.f); }
    void Resize(size_t new_n) { n = new_n; coordinates.resize(new_n * d); }
    bool empty() const { return n == 0; }
};

using Partition = std::vector<int>;
using Clusters = std::vector<std::vector<uint32_t>>;
using Cover = std::vector<std::vector<int>>;
using AdjGraph = std::vector<std::vector<int>>;
using NNVec = std::vector<std::pair<float, uint32_t>>;

PointSet ExtractPointsInBucket(const std::vector<uint32_t>& bucket, PointSet& points);   // defs.cpp:11-23
int NumPartsInPartition(const Partition& partition);                                     // defs.cpp:25-28
Clusters ConvertPartitionToClusters(const Partition& partition);                         // defs.cpp:30-37
Cover ConvertPartitionToCover(const Partition& partition);                               // defs.cpp:39-45
Cover ConvertClustersToCover(const Clusters& clusters);                                  // defs.cpp:47-62
void RemapPartitionIDs(std::vector<int>& partition);                                     // defs.cpp:64-79
NNVec ConvertTopKToNNVec(TopN& top_k);                                                   // defs.cpp:81-85 (descending)
TopN ClosestLeaders(PointSet& points, PointSet& leader_points, uint32_t my_id, int k);   // defs.cpp:87-96

struct HNSWParameters {   // defs.h:52-56 (kept for signature compatibility; unused by the exact scan)
    size_t M = 32;
    size_t ef_construction = 200;
    size_t ef_search = 250;
};

using Duration = std::chrono::duration<double>;
using Timepoint = decltype(std::chrono::high_resolution_clock::now());

struct Timer {   // defs.h:61-98
    bool running = false;
    Timepoint start;
    Duration total_duration = Duration(0.0);
    Timepoint Start();
    double Stop();
    double Restart();
    double ElapsedRunning();
};

template <typename T1, typename T2>
inline auto idiv_ceil(T1 a, T2 b) {   // defs.h:101-104
    return static_cast<T1>((static_cast<unsigned long long>(a) + b - 1) / b);
}
inline std::pair<size_t, size_t> bounds(size_t i, size_t n, size_t chunk_size) {   // defs.h:106-108
    return {std::min(n, i * chunk_size), std::min(n, (i + 1) * chunk_size)};
}
inline bool DoubleEquals(double x, double y, double eps = 1e-12) { return std::abs(x - y) < eps; }

// ----------------------------------------------------------------------------------------------------- src/dist.h:3-15
// Host evaluation of the canonical (strict binary32, unfused) sequences; bit-identical to the device kernels.
float sqr_l2_dist(const float* a, const float* b, unsigned size);
float inner_product(float* p, float* q, unsigned d);
float mips_distance(float* p, float* q, unsigned d);
float vec_norm(float* p, unsigned d);
bool L2Normalize(float* p, unsigned d);
float distance(float* p, float* q, unsigned d);       // metric switch, dist.cpp:75-81
float pos_distance(float* p, float* q, unsigned d);   // dist.cpp:83-88

namespace gpann_host {
// Run-time form of the reference's compile-time MIPS_DISTANCE option.  Default: L2, or MIPS when this library itself
// was built with -DMIPS_DISTANCE.  Objects capture the metric when they are constructed / trained.
void SetMetricMIPS(bool mips);
bool MetricIsMIPS();
void SetDevice(int device);   // HIP device used by subsequently constructed objects (default 0)
int Device();
void SetNumThreads(int n);    // host worker threads for training / IO (default: hardware concurrency)
int NumThreads();
// Scan mode of subsequently constructed InvertedIndex / RunInShardSearches stores: true (default) = exact results through the
// matrix-core pre-filter (csrc/scan_filter.h), false = plain exact VALU scan.  Both return identical neighbours.
void SetFilteredScan(bool on);
bool FilteredScan();
// KMeansTreeRouter::Train runs on the GPU (gpann_router_train, csrc/kmeans.hip) by default; false selects the host trainer
// (router_host.cpp).  Both build the same tree, bit for bit.
void SetDeviceTraining(bool on);
bool DeviceTraining();
// Concurrent one-query calls — KMeansTreeRouter::Query, InvertedIndex::Query / QueryBucket, the way the reference's callers issue them from
// up to 32 threads (routes.cpp:194-200, shard_searches.cpp:48) — are coalesced into device batches: whoever finds no batch in flight runs
// everything that has queued up as ONE launch sequence.  Default on; off = every call a batch of one behind the handle's lock.
void SetCoalescing(bool on);
bool Coalescing();
}   // namespace gpann_host

// --------------------------------------------------------------------------------------------- src/points_io.h:5-11
PointSet ReadPoints(const std::string& path, int64_t size = -1);   // .fbin / .u8bin / .i8bin
void WritePoints(PointSet& points, const std::string& path);
std::vector<NNVec> ReadGroundTruth(const std::string& path);
void WriteGroundTruth(const std::string& path, const std::vector<NNVec>& ground_truth);

// ---------------------------------------------------------------------------------------------- src/metis_io.h:5-13
std::vector<int> ReadMetisPartition(const std::string& path);
void WriteMetisPartition(const std::vector<int>& partition, const std::string& path);
Clusters ReadClusters(const std::string& path);
void WriteClusters(const Clusters& clusters, const std::string& path);
void WriteMetisGraph(const std::string& path, const AdjGraph& graph);

// ------------------------------------------------------------------------------------------------- src/kmeans.h:5-8
PointSet RandomSample(PointSet& points, size_t num_samples, int seed);
std::vector<int> KMeans(PointSet& points, PointSet& centroids);

// --------------------------------------------------------------------------------------- src/kmeans_tree_router.h
struct KMeansTreeRouterOptions {
    size_t num_centroids = 64;
    size_t min_cluster_size = 250;
    int64_t budget = 50000;
    int64_t search_budget = 50000;
};

class KMeansTreeRouter {
public:
    KMeansTreeRouter() = default;
    ~KMeansTreeRouter();
    KMeansTreeRouter(const KMeansTreeRouter&) = delete;
    KMeansTreeRouter& operator=(const KMeansTreeRouter&) = delete;
    KMeansTreeRouter(KMeansTreeRouter&& other) noexcept;
    KMeansTreeRouter& operator=(KMeansTreeRouter&& other) noexcept;

    void Train(PointSet& points, const Clusters& clusters, KMeansTreeRouterOptions options);   // kmeans_tree_router.cpp:9-78 (GPU)
    // additive: train over the rows an InvertedIndex already holds in HBM (built from the same clusters); num_points = PointSet::n
    void Train(struct InvertedIndex& index, size_t num_points, KMeansTreeRouterOptions options);
    void TrainOnHost(PointSet& points, const Clusters& clusters, KMeansTreeRouterOptions options);   // the same tree, built by host threads

    std::vector<int> Query(float* Q, int budget);   // :93-130, one query = a GPU batch of one

    // additive: the whole query set in one GPU launch; result[q] is what Query(queries.GetPoint(q), budget) returns
    std::vector<std::vector<int>> QueryBatch(PointSet& queries, int budget);

    struct FrequencyQueryData {   // kmeans_tree_router.h:19-45
        std::vector<float> min_dist;
        NNVec near_neighbors;
        void Init() { std::sort(near_neighbors.begin(), near_neighbors.end()); }
        std::vector<int> Query(int num_shards, int num_voting_neighbors);
    };
    FrequencyQueryData FrequencyQuery(float* Q, int budget, int num_voting_neighbors);   // :132-172, GPU walk
    // additive: all queries in one GPU batch
    std::vector<FrequencyQueryData> FrequencyQueryBatch(PointSet& queries, int budget, int num_voting_neighbors);

    std::pair<PointSet, std::vector<int>> ExtractPoints();   // :174-201
    // additive: only the min_dist rows of Query (out[q * S + b]; FLT_MAX = shard never opened) for nq queries stored row-major at Q
    void MinDistBatch(const float* Q, size_t nq, int budget, float* out);

    // Flat tree (the layout gpann_router_create takes, include/gpann.h): nodes per shard in BFS order.
    struct Flat {
        uint32_t d = 0, S = 0;
        bool mips = false;
        std::vector<int32_t> node_shard;
        std::vector<uint32_t> node_cent_first, node_child_first, node_child_count, shard_root;
        std::vector<float> centroids;
    };
    const Flat& flat() const { return tree_; }
    void LoadFlat(Flat tree);   // adopt an already trained tree

private:
    void EnsureOnDevice();
    void TrainOnDevice(gpann_shards* store, size_t num_points, KMeansTreeRouterOptions options);
    Flat tree_;
    gpann_router* device_router_ = nullptr;
};

// ------------------------------------------------------------------------------------------- src/partitioning.h
// The k-means shard builders (src/partitioning.cpp:12-146); k-means itself runs on the GPU (gpann_kmeans).
Partition RecursiveKMeansPartitioning(PointSet& points, size_t max_cluster_size, int depth = 0, int num_clusters = -1);
Partition RebalancingKMeansPartitioning(PointSet& points, size_t max_cluster_size, int num_clusters = -1);
Partition KMeansPartitioning(PointSet& points, int num_clusters, double epsilon);
// declared for source compatibility; they throw std::runtime_error (KaMinPar / hnswlib are not part of the reference checkout)
Partition GraphPartitioning(PointSet& points, int num_clusters, double epsilon, bool strong, const std::string& graph_output_path = "");
Partition PyramidPartitioning(PointSet& points, int num_clusters, double epsilon, const std::string& routing_index_path = "");
Partition OurPyramidPartitioning(PointSet& points, int num_clusters, double epsilon, const std::string& routing_index_path,
                                 double coarsening_rate = 0.002);

// ------------------------------------------------------------------------------- src/overlapping_partitioning.h
// SPANN-style overlap (src/overlapping_partitioning.cpp:143-235): representatives from the GPU trainer, point x representative
// minima from the GPU router, greedy extra assignments on the host.
void MakeOverlappingWithCentroids(PointSet& points, Clusters& clusters, size_t max_cluster_size, size_t num_extra_assignments);
Clusters OverlappingKMeansPartitioningSPANN(PointSet& points, const Partition& partition, int requested_num_clusters, double epsilon,
                                            double overlap);
Clusters OverlappingGraphPartitioning(PointSet& points, int requested_num_clusters, double epsilon, double overlap, bool strong);   // throws

// ------------------------------------------------------------------------------------------ src/inverted_index.h
struct InvertedIndex {
    // The reference's public members stay readable; the scanned copy lives in HBM.
    PointSet clustered_points;            // left empty: the shard-contiguous copy is device resident (ExportClusteredPoints())
    std::vector<int> offsets;             // S + 1
    std::vector<uint32_t> permutation;    // row -> point id

    InvertedIndex(PointSet& points, const Clusters& clusters);   // inverted_index.h:12-41
    ~InvertedIndex();
    InvertedIndex(const InvertedIndex&) = delete;
    InvertedIndex& operator=(const InvertedIndex&) = delete;

    NNVec Query(float* Q, int k, const std::vector<int>& buckets_to_probe, size_t num_buckets_to_probe);   // :43-60
    NNVec QueryBucket(float* Q, int k, int bucket);                                                         // :62-75

    // additive, batched: result[q][j] = QueryBucket(queries.GetPoint(q), k, probes[q][j]) for j < num_probes
    std::vector<std::vector<NNVec>> QueryBucketBatch(PointSet& queries, int k, const std::vector<std::vector<int>>& probes, size_t num_probes);
    // additive, batched Query(): one merged top-k per query over its first num_probes probes
    std::vector<NNVec> QueryBatch(PointSet& queries, int k, const std::vector<std::vector<int>>& probes, size_t num_probes);

    PointSet ExportClusteredPoints() const;
    gpann_shards* device_handle() { return shards_; }
    double last_scan_ms = 0.0;   // device time of the most recent scan (kernels only)

private:
    gpann_shards* shards_ = nullptr;
    bool mips_ = false;
    size_t d_ = 0;
};

// ------------------------------------------------------------------------------------------------- src/recall.h
std::vector<float> ComputeDistanceToKthNeighbor(PointSet& points, PointSet& queries, int k);   // :10-23
std::vector<NNVec> ComputeGroundTruth(PointSet& points, PointSet& queries, int k);             // :25-39 (GPU scan)
void OracleRecall(const std::vector<NNVec>& ground_truth, const std::vector<int>& partition, int num_neighbors);   // :41-68
std::vector<float> ConvertGroundTruthToDistanceToKthNeighbor(std::vector<NNVec>& ground_truth, int k, PointSet& points, PointSet& queries);   // :74-137
double Recall(const std::vector<NNVec>& neighbors_per_query, const std::vector<float>& distance_to_kth_neighbor, int k);   // :139-150

// DedupNeighbors, small_scale_queries.cpp:15-23 (a static function of that driver in the reference)
void DedupNeighbors(std::vector<NNVec>& neighbors, size_t num_neighbors);

// ------------------------------------------------------------------------------------------ src/shard_searches.h
struct ShardSearch {
    ShardSearch() {}
    void Init(size_t ef_search, int num_shards, size_t num_queries);
    size_t ef_search = 0;
    std::vector<std::vector<std::vector<uint32_t>>> neighbors;      // [shard][query][hit]
    std::vector<std::vector<double>> time_query_in_shard;           // [shard][query]
    std::string Serialize() const;
    static ShardSearch Deserialize(std::ifstream& in);
};
void SerializeShardSearches(const std::vector<ShardSearch>& shard_searches, const std::string& output_file);
std::vector<ShardSearch> DeserializeShardSearches(const std::string& input_file);
std::vector<ShardSearch> RunInShardSearches(PointSet& points, PointSet& queries, HNSWParameters hnsw_parameters, int num_neighbors,
                                            const Clusters& clusters, int num_shards, const std::vector<float>& distance_to_kth_neighbor);

// -------------------------------------------------------------------------------------------------- src/routes.h
struct RoutingConfig {
    std::string routing_algorithm = "None";
    std::string index_trainer = "KMeansTree";
    size_t hnsw_num_voting_neighbors = 0;
    size_t hnsw_ef_search = 250;
    double routing_time = 0.0;
    size_t routing_distance_calcs = 0;
    bool try_increasing_num_shards = false;
    KMeansTreeRouterOptions routing_index_options;
    std::vector<std::vector<int>> buckets_to_probe;
    std::string Serialize() const;
    static RoutingConfig Deserialize(std::ifstream& in);
};
double MaxFirstShardRoutingRecall(const std::vector<std::vector<int>>& buckets_to_probe, const std::vector<NNVec>& ground_truth,
                                  int num_neighbors, const Cover& cover);
void SerializeRoutes(const std::vector<RoutingConfig>& routes, const std::string& output_file);
std::vector<RoutingConfig> DeserializeRoutes(const std::string& input_file);
// KMeansTreeRouter part of routes.cpp:164-313 (the HNSW / Pyramid / SPANN routers need hnswlib and are not built)
std::vector<RoutingConfig> IterateRoutingConfigs(PointSet& points, PointSet& queries, const Clusters& clusters, int num_shards,
                                                 KMeansTreeRouterOptions routing_index_options_blueprint, const std::vector<NNVec>& ground_truth,
                                                 int num_neighbors, const std::string& routing_index_file, const std::string& pyramid_index_file,
                                                 const std::string& our_pyramid_index_file);

// ------------------------------------------------------------------------------ src/route_search_combination.h
struct EmitResult {
    std::vector<double> local_work;
    size_t total_hits;
    double n_probes;
};
void AttributeRecallAndQueryTimeIncreasingNumProbes(const RoutingConfig& route, const ShardSearch& search, size_t num_queries, size_t num_shards,
                                                    int num_neighbors, std::function<void(EmitResult)>& emit);
void AttributeRecallAndQueryTimeVariableNumProbes(const RoutingConfig& route, const ShardSearch& search, size_t num_queries, size_t num_shards,
                                                  int num_neighbors, std::function<void(EmitResult)>& emit);
void MaxShardSearchRecall(const std::vector<ShardSearch>& shard_searches, int num_neighbors, int num_queries, int num_shards,
                          int num_requested_shards);
void MaxRoutingRecall(const std::vector<RoutingConfig>& routes, const std::vector<NNVec>& ground_truth, int num_neighbors,
                      const std::vector<int>& partition, int num_shards);
void PrintCombinationsOfRoutesAndSearches(const std::vector<RoutingConfig>& routes, const std::vector<ShardSearch>& shard_searches,
                                          const std::string& output_file, int num_neighbors, int num_queries, int num_shards,
                                          int num_requested_shards, const std::string& part_method);
