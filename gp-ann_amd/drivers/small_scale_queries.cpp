// SmallScaleQueries — route + scan + merge + recall on one data set, as a CSV of (probes, latency, recall) rows.
// Contract (reference small_scale_queries.cpp:25-165; only what a consumer of the experiment can observe):
//   argv   input-points  queries  ground-truth-file  num-neighbors  partition  part-method  out-file
//   files  ground truth is read if the file exists, otherwise computed (on the GPU) and written in the big-ann layout; the partition is a clusters file
//   CSV    header `partitioning,routing,shard query,probes,latency,routing latency, query latency,recall`; one row per number of probes 1..S, cumulative
//          (probe p adds the p-th shard of every query's route to what the earlier probes found, then DedupNeighbors + Recall)
//   router KMeansTreeRouterOptions{num_centroids 32, min_cluster_size 200, budget 50000, search_budget 5000} (small_scale_queries.cpp:98-103)
// Rows produced: routing = KMTR, shard query = BruteForce — the hot path of this build.  The reference's HNSW-router / IVF-HNSW rows and its unpartitioned
// branch need hnswlib and are not built.  Every stage is ONE device batch (QueryBatch, QueryBucketBatch); latencies are batch time / #queries.
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>

#include "inverted_index.h"
#include "kmeans_tree_router.h"
#include "metis_io.h"
#include "points_io.h"
#include "recall.h"

namespace {

struct Args {
    std::string points, queries, ground_truth, partition, part_method, csv;
    int k = 10;
};

double Seconds(Timer& t) { return t.Stop(); }

// distance to the k-th true neighbour of every query: what Recall() compares against (recall.h:74-137)
std::vector<float> RecallThresholds(const Args& a, PointSet& points, PointSet& queries) {
    std::vector<NNVec> truth;
    if (std::filesystem::exists(a.ground_truth)) {
        truth = ReadGroundTruth(a.ground_truth);
    } else {
        std::cout << "no ground truth at " << a.ground_truth << ": computing it" << std::endl;
        truth = ComputeGroundTruth(points, queries, a.k);
        WriteGroundTruth(a.ground_truth, truth);
    }
    return ConvertGroundTruthToDistanceToKthNeighbor(truth, a.k, points, queries);
}

KMeansTreeRouterOptions SmallScaleRouter() {
    KMeansTreeRouterOptions o;
    o.num_centroids = 32;
    o.min_cluster_size = 200;
    o.budget = 50000;
    o.search_budget = 5000;
    return o;
}

}   // namespace

int main(int argc, const char* argv[]) {
    if (argc != 8) {
        std::cerr << "Usage ./SmallScaleQueries input-points queries ground-truth-file num-neighbors partition part-method out-file" << std::endl;
        std::abort();
    }
    Args a{argv[1], argv[2], argv[3], argv[5], argv[6], argv[7], std::stoi(argv[4])};
    if (const char* m = std::getenv("GPANN_METRIC")) gpann_host::SetMetricMIPS(std::string(m) == "mips");
    if (a.part_method == "None" || !std::filesystem::exists(a.partition)) {
        std::cerr << "SmallScaleQueries: no partition (" << a.partition << ", method " << a.part_method
                  << "): the unpartitioned experiment of the reference is HNSW on the whole input and is not part of this build" << std::endl;
        return 1;
    }
    PointSet points = ReadPoints(a.points), queries = ReadPoints(a.queries);
    const std::vector<float> kth_distance = RecallThresholds(a, points, queries);
    const Clusters shards = ReadClusters(a.partition);
    const size_t nq = queries.n, S = shards.size();

    Timer clock;
    const KMeansTreeRouterOptions opt = SmallScaleRouter();
    KMeansTreeRouter router;
    clock.Start();
    router.Train(points, shards, opt);
    std::cout << "router trained in " << Seconds(clock) << " s" << std::endl;
    clock.Start();
    const std::vector<std::vector<int>> route = router.QueryBatch(queries, (int)opt.search_budget);   // a full permutation of the shards per query
    const double t_route = Seconds(clock);
    std::cout << "routed " << nq << " queries in " << t_route << " s (" << nq / t_route << " QPS)" << std::endl;
    clock.Start();
    InvertedIndex index(points, shards);
    std::cout << "shard store built in " << Seconds(clock) << " s" << std::endl;

    std::ofstream csv(a.csv);
    csv << "partitioning,routing,shard query,probes,latency,routing latency, query latency,recall" << std::endl;
    std::vector<NNVec> found(nq);
    std::vector<std::vector<int>> one_probe(nq, std::vector<int>(1));
    double t_scan = 0.0;
    for (size_t p = 0; p < S; ++p) {
        for (size_t q = 0; q < nq; ++q) one_probe[q][0] = route[q][p];
        clock.Start();
        const auto lists = index.QueryBucketBatch(queries, a.k, one_probe, 1);   // QueryBucket(q, k, route[q][p]) for every q
        t_scan += Seconds(clock);
        for (size_t q = 0; q < nq; ++q) found[q].insert(found[q].end(), lists[q][0].begin(), lists[q][0].end());
        DedupNeighbors(found, a.k);
        const double recall = Recall(found, kth_distance, a.k);
        std::cout << "probes " << p + 1 << ": recall " << recall << ", scan time so far " << t_scan << " s" << std::endl;
        csv << a.part_method << ",KMTR,BruteForce," << p + 1 << "," << (t_route + t_scan) / nq << "," << t_route / nq << "," << t_scan / nq << "," << recall
            << std::endl;
    }
    return 0;
}
