// partition_host.cpp — host mirror of the reference's k-means shard builders: RecursiveKMeansPartitioning,
// RebalancingKMeansPartitioning, KMeansPartitioning (src/partitioning.cpp:12-146) and the SPANN-style overlap
// MakeOverlappingWithCentroids / OverlappingKMeansPartitioningSPANN (src/overlapping_partitioning.cpp:143-235).
//
// The distance-heavy steps run on the GPU: RandomSample + KMeans as one gpann_kmeans call over a plain single-shard store
// (csrc/kmeans.hip; same clusters and centroids as the host KMeans, bit for bit), and the overlap's point x centroid minima as
// visit-everything router batches (csrc/router.hip).  The control flow around them — recursion on overloaded clusters, id
// renumbering, the sequential re-migration and the greedy extra assignments — is inherently serial and stays on the host like
// the reference's.  gpann_host::SetDeviceTraining(false) selects the host k-means / brute-force minima instead (same output).
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <numeric>

#include "gpann_host.h"
#include "host_util.h"

namespace {

// RandomSample(points, k, seed) followed by KMeans(points, centroids): kmeans.cpp:207-244
Partition SampleThenKMeans(PointSet& points, int k, int seed, PointSet& centroids) {
    if (!gpann_host::DeviceTraining()) {
        centroids = RandomSample(points, (size_t)k, seed);
        return KMeans(points, centroids);
    }
    std::vector<uint32_t> ids(points.n);
    std::iota(ids.begin(), ids.end(), 0u);
    const uint64_t offsets[2] = {0, (uint64_t)points.n};
    gpann_shards* store = nullptr;
    gpann_host::Check(gpann_shards_create(points.coordinates.data(), points.n, (uint32_t)points.d,
                                          gpann_host::MetricIsMIPS() ? GPANN_METRIC_MIPS : GPANN_METRIC_L2, ids.data(), offsets, 1,
                                          gpann_host::Device(), GPANN_STORE_PLAIN, &store),
                      "gpann_shards_create");
    std::vector<int32_t> owner(points.n);
    centroids.d = points.d;
    centroids.n = (size_t)k;
    centroids.coordinates.assign((size_t)k * points.d, 0.f);
    uint32_t left = 0;
    const gpann_status st = gpann_kmeans(store, (uint32_t)k, seed, owner.data(), centroids.coordinates.data(), &left);
    gpann_shards_destroy(store);
    gpann_host::Check(st, "gpann_kmeans");
    centroids.Resize(left);
    return Partition(owner.begin(), owner.end());
}

std::vector<size_t> PartSizes(const Partition& partition, int parts) {
    std::vector<size_t> sizes(parts, 0);
    for (int p : partition) sizes[p]++;
    return sizes;
}

}   // namespace

// partitioning.cpp:12-88
Partition RecursiveKMeansPartitioning(PointSet& points, size_t max_cluster_size, int depth, int num_clusters) {
    if (num_clusters < 0) num_clusters = static_cast<int>(std::ceil(double(points.n) / max_cluster_size));
    if (num_clusters == 0) return Partition(points.n, 0);
    Timer timer;
    timer.Start();
    PointSet centroids;
    Partition partition = SampleThenKMeans(points, num_clusters, 555, centroids);
    std::cout << "k-means at depth " << depth << " took " << timer.Stop() << " s" << std::endl;

    num_clusters = NumPartsInPartition(partition);
    const std::vector<size_t> sizes = PartSizes(partition, num_clusters);
    const size_t heavy = std::count_if(sizes.begin(), sizes.end(), [&](size_t s) { return s > max_cluster_size; });
    if (heavy > 0)
        std::cout << "At depth " << depth << " there are " << heavy << " / " << num_clusters << " too heavy clusters. Refine them" << std::endl;

    // an overloaded cluster is split by a recursive call; its first sub-cluster keeps the id, the others get fresh ids (:49-85)
    int fresh = num_clusters;
    for (int part = 0; part < num_clusters; ++part) {
        if (sizes[part] <= max_cluster_size) continue;
        std::cout << "Cluster " << part << " / " << num_clusters << " at depth " << depth << " is overloaded " << sizes[part] << " / "
                  << max_cluster_size << std::endl;
        std::vector<uint32_t> members;
        members.reserve(sizes[part]);
        for (uint32_t v = 0; v < partition.size(); ++v)
            if (partition[v] == part) members.push_back(v);
        PointSet subset = ExtractPointsInBucket(members, points);
        const Partition sub = RecursiveKMeansPartitioning(subset, max_cluster_size, depth + 1);
        const int last_sub = *std::max_element(sub.begin(), sub.end());
        std::cout << "Cluster " << part << " / " << num_clusters << " at depth " << depth << " was overloaded " << sizes[part] << " / "
                  << max_cluster_size << " and got split into " << last_sub + 1 << " sub-clusters" << std::endl;
        for (size_t i = 0; i < members.size(); ++i)
            if (sub[i] != 0) partition[members[i]] = fresh + sub[i] - 1;
        fresh += last_sub;
    }
    return partition;
}

// partitioning.cpp:90-139
Partition RebalancingKMeansPartitioning(PointSet& points, size_t max_cluster_size, int num_clusters) {
    if (num_clusters < 0) num_clusters = static_cast<int>(std::ceil(double(points.n) / max_cluster_size));
    if (num_clusters == 0) return Partition(points.n, 0);
    Timer timer;
    timer.Start();
    PointSet centroids;
    Partition partition = SampleThenKMeans(points, num_clusters, 555, centroids);
    std::cout << "k-means took " << timer.Stop() << " s" << std::endl;

    num_clusters = NumPartsInPartition(partition);
    const std::vector<size_t> sizes = PartSizes(partition, num_clusters);
    const size_t heavy = std::count_if(sizes.begin(), sizes.end(), [&](size_t s) { return s > max_cluster_size; });
    if (heavy == 0) return partition;
    std::cout << "There are " << heavy << " / " << num_clusters << " too heavy clusters. Rebalance stuff" << std::endl;

    // strictly sequential: every move changes which clusters still have room (:117-137); centroids are not updated
    Clusters clusters = ConvertPartitionToClusters(partition);
    for (int c = 0; c < num_clusters; ++c) {
        while (clusters[c].size() > max_cluster_size) {
            const uint32_t v = clusters[c].back();
            float best = std::numeric_limits<float>::max();
            int target = -1;
            for (size_t j = 0; j < clusters.size(); ++j) {
                if (clusters[j].size() >= max_cluster_size) continue;
                const float dist = distance(points.GetPoint(v), centroids.GetPoint(j), (unsigned)points.d);
                if (dist < best) {
                    best = dist;
                    target = (int)j;
                }
            }
            if (target < 0) throw std::runtime_error("RebalancingKMeansPartitioning: no cluster has room left");   // the reference asserts
            clusters[target].push_back(v);
            partition[v] = target;
            clusters[c].pop_back();
        }
    }
    return partition;
}

// partitioning.cpp:141-144
Partition KMeansPartitioning(PointSet& points, int num_clusters, double epsilon) {
    const size_t max_cluster_size = points.n * (1 + epsilon) / num_clusters;
    return RecursiveKMeansPartitioning(points, max_cluster_size, 0, num_clusters);
}

// overlapping_partitioning.cpp:143-227
void MakeOverlappingWithCentroids(PointSet& points, Clusters& clusters, size_t max_cluster_size, size_t num_extra_assignments) {
    clusters.erase(std::remove_if(clusters.begin(), clusters.end(), [](const auto& c) { return c.empty(); }), clusters.end());
    const Cover cover = ConvertClustersToCover(clusters);
    const size_t S = clusters.size();

    Timer timer;
    timer.Start();
    // step 1: <= 64 representatives per cluster = the root level of a k-means tree router over the clusters (:151-158)
    const size_t num_centroids = 64;
    KMeansTreeRouterOptions options;
    options.num_centroids = num_centroids;
    options.min_cluster_size = 350;
    options.budget = static_cast<int64_t>(S * num_centroids);
    options.search_budget = 0;
    KMeansTreeRouter router;
    router.Train(points, clusters, options);
    auto [reps, rep_cluster] = router.ExtractPoints();
    std::cout << "Got reps. " << reps.n << " " << rep_cluster.size() << " Took " << timer.Restart() << " s" << std::endl;

    // step 2: per point the five nearest clusters (by nearest representative) it is not yet part of and that still have room
    struct Rating {
        float dist;
        int target_cluster;
        uint32_t point_id;
    };
    const size_t keep = 5;
    std::vector<Rating> ratings(points.n * keep);
    std::vector<uint8_t> used(points.n, 0);
    std::vector<uint8_t> has_room(S);
    for (size_t c = 0; c < S; ++c) has_room[c] = clusters[c].size() < max_cluster_size;
    const float kNone = std::numeric_limits<float>::max();

    // min over a cluster's representatives, for a block of points at a time: one visit-everything router batch on the GPU
    // (min_dist of KMeansTreeRouter::Query with an unlimited budget IS that minimum), or the plain loops on the host
    const size_t block = std::max<size_t>(1, std::min<size_t>(points.n, (size_t(1) << 26) / std::max<size_t>(S, 1)));
    std::vector<float> min_dist;
    for (size_t first = 0; first < points.n; first += block) {
        const size_t count = std::min(block, points.n - first);
        min_dist.assign(count * S, kNone);
        if (gpann_host::DeviceTraining()) {
            router.MinDistBatch(points.GetPoint(first), count, std::numeric_limits<int>::max(), min_dist.data());
        } else {
            gpann_host::ParallelFor(0, count, 0, [&](size_t i) {
                float* mine = &min_dist[i * S];
                for (size_t j = 0; j < reps.n; ++j) {
                    const float dist = distance(points.GetPoint(first + i), reps.GetPoint(j), (unsigned)points.d);
                    if (dist < mine[rep_cluster[j]]) mine[rep_cluster[j]] = dist;
                }
            });
        }
        gpann_host::ParallelFor(0, count, 0, [&](size_t i) {
            const size_t u = first + i;
            float* mine = &min_dist[i * S];
            for (int c : cover[u]) mine[c] = kNone;
            std::vector<int> targets;
            for (size_t c = 0; c < S; ++c)
                if (mine[c] != kNone && has_room[c]) targets.push_back((int)c);
            std::sort(targets.begin(), targets.end(), [&](int l, int r) { return mine[l] < mine[r]; });   // the reference's call (:186)
            const size_t kept = std::min(targets.size(), keep);
            for (size_t t = 0; t < kept; ++t) ratings[u * keep + t] = Rating{mine[targets[t]], targets[t], (uint32_t)u};
            used[u] = (uint8_t)kept;
        });
    }
    std::cout << "got cluster rankings. Took " << timer.Restart() << std::endl;

    // flatten in point order, then order by distance.  The reference's parlay::sort_inplace leaves the order of equal distances
    // open; here equal distances keep their (point, rank) order.
    size_t total = 0;
    for (size_t u = 0; u < points.n; ++u) {
        for (size_t t = 0; t < used[u]; ++t) ratings[total++] = ratings[u * keep + t];
    }
    ratings.resize(total);
    std::stable_sort(ratings.begin(), ratings.end(), [](const Rating& l, const Rating& r) { return l.dist < r.dist; });
    std::cout << "Flatten and sort took " << timer.Restart() << std::endl;

    // step 3: greedy, nearest first, while the target has room (:209-221; the counter is the reference's unsigned one)
    size_t left = num_extra_assignments, steps = 0;
    for (const Rating& r : ratings) {
        if (clusters[r.target_cluster].size() < max_cluster_size) {
            --left;
            clusters[r.target_cluster].push_back(r.point_id);
        }
        ++steps;
        if (left == 0) break;
    }
    std::cout << "Finished kmeans overlap partitioning. " << left << " possible assignments unused. Moves inspected: " << steps << " Time "
              << timer.Stop() << std::endl;
    std::cout << "Total time for overlap " << timer.total_duration.count() << " seconds" << std::endl;
}

// overlapping_partitioning.cpp:229-235
Clusters OverlappingKMeansPartitioningSPANN(PointSet& points, const Partition& partition, int requested_num_clusters, double epsilon, double overlap) {
    const size_t num_extra_assignments = overlap * points.n;
    const size_t max_cluster_size = (1.0 + epsilon) * points.n / requested_num_clusters;
    Clusters clusters = ConvertPartitionToClusters(partition);
    MakeOverlappingWithCentroids(points, clusters, max_cluster_size, num_extra_assignments);
    return clusters;
}

// The graph / HNSW based partitioners need KaMinPar and hnswlib, which are not part of the reference checkout (.gitmodules only).
namespace {
[[noreturn]] void NeedsThirdParty(const char* what) {
    throw std::runtime_error(std::string(what) + ": needs KaMinPar / hnswlib, which this build does not contain (k-means partitioners only)");
}
}   // namespace
Partition GraphPartitioning(PointSet&, int, double, bool, const std::string&) { NeedsThirdParty("GraphPartitioning"); }
Partition PyramidPartitioning(PointSet&, int, double, const std::string&) { NeedsThirdParty("PyramidPartitioning"); }
Partition OurPyramidPartitioning(PointSet&, int, double, const std::string&, double) { NeedsThirdParty("OurPyramidPartitioning"); }
Clusters OverlappingGraphPartitioning(PointSet&, int, double, double, bool) { NeedsThirdParty("OverlappingGraphPartitioning"); }
