// small_scale_queries.cpp — MI355X build of the reference's SmallScaleQueries experiment (small_scale_queries.cpp:25-165):
//   read points / queries / ground truth -> KMeansTreeRouter::Train{32, 200, 50000, 5000} -> route every query ->
//   InvertedIndex -> for nprobe = 1..S: QueryBucket of the nprobe-th probe, DedupNeighbors, Recall -> CSV.
// Same argv, same CSV columns and file formats.  Only the rows the hot path covers are produced: routing = KMTR,
// shard query = BruteForce.  (The HNSW router / IVF-HNSW rows of the reference need hnswlib.)  Routing and the per-probe
// scans are issued as one GPU batch each instead of one query at a time; the reported latencies are batch time / #queries.
#include <filesystem>
#include <iostream>

#include "inverted_index.h"
#include "kmeans_tree_router.h"
#include "metis_io.h"
#include "points_io.h"
#include "recall.h"

int main(int argc, const char* argv[]) {
    if (argc != 8) {
        std::cerr << "Usage ./SmallScaleQueries input-points queries ground-truth-file num-neighbors partition part-method out-file" << std::endl;
        std::abort();
    }
    const std::string point_file = argv[1], query_file = argv[2], ground_truth_file = argv[3];
    const int num_neighbors = std::stoi(argv[4]);
    const std::string partition_file = argv[5], part_method = argv[6], out_file = argv[7];
    if (const char* m = std::getenv("GPANN_METRIC")) gpann_host::SetMetricMIPS(std::string(m) == "mips");

    PointSet points = ReadPoints(point_file);
    PointSet queries = ReadPoints(query_file);

    std::vector<NNVec> ground_truth;
    if (std::filesystem::exists(ground_truth_file)) {
        ground_truth = ReadGroundTruth(ground_truth_file);
    } else {
        std::cout << "start computing ground truth" << std::endl;
        ground_truth = ComputeGroundTruth(points, queries, num_neighbors);
        std::cout << "computed ground truth" << std::endl;
        WriteGroundTruth(ground_truth_file, ground_truth);
        std::cout << "wrote ground truth to file " << ground_truth_file << std::endl;
    }
    std::vector<float> distance_to_kth_neighbor = ConvertGroundTruthToDistanceToKthNeighbor(ground_truth, num_neighbors, points, queries);
    std::cout << "finished converting ground truth to distances" << std::endl;

    if (!std::filesystem::exists(partition_file) || part_method == "None") {
        std::cerr << "Not partitioned: the unpartitioned branch of the reference runs HNSW and is not part of this build." << std::endl;
        return 1;
    }
    Clusters clusters = ReadClusters(partition_file);
    const int num_shards = (int)clusters.size();

    Timer timer;
    timer.Start();
    KMeansTreeRouterOptions options;
    options.num_centroids = 32;
    options.min_cluster_size = 200;
    options.budget = 50000;
    options.search_budget = 5000;
    KMeansTreeRouter router;
    router.Train(points, clusters, options);
    std::cout << "Training KMTR took " << timer.Stop() << " seconds." << std::endl;

    timer.Start();
    std::vector<std::vector<int>> probes = router.QueryBatch(queries, (int)options.search_budget);
    const double routing_time = timer.Stop();
    std::cout << "KMTR routing took " << routing_time << " seconds. That's " << 1000.0 * routing_time / queries.n << "ms per query, or "
              << queries.n / routing_time << " QPS" << std::endl;

    timer.Start();
    InvertedIndex ivf(points, clusters);
    std::cout << "Building IVF took " << timer.Stop() << " seconds." << std::endl;

    std::ofstream out(out_file);
    out << "partitioning,routing,shard query,probes,latency,routing latency, query latency,recall" << std::endl;
    std::cout << "Start queries" << std::endl;

    std::vector<NNVec> neighbors(queries.n);
    double time = 0;
    for (int num_probes = 1; num_probes <= num_shards; ++num_probes) {
        // QueryBucket(q, k, probes[q][num_probes - 1]) for every query, as one batch
        std::vector<std::vector<int>> this_probe(queries.n);
        for (size_t q = 0; q < queries.n; ++q) this_probe[q] = {probes[q][num_probes - 1]};
        timer.Start();
        auto found = ivf.QueryBucketBatch(queries, num_neighbors, this_probe, 1);
        time += timer.Stop();
        for (size_t q = 0; q < queries.n; ++q) neighbors[q].insert(neighbors[q].end(), found[q][0].begin(), found[q][0].end());
        DedupNeighbors(neighbors, num_neighbors);
        const double recall = Recall(neighbors, distance_to_kth_neighbor, num_neighbors);
        const double latency = (routing_time + time) / queries.n;
        std::cout << "router = KMTR query = IVF nprobes = " << num_probes << " recall = " << recall << " time = " << time
                  << " avg latency = " << 1000.0 * latency << " ms" << std::endl;
        out << part_method << ",KMTR,BruteForce," << num_probes << "," << latency << "," << routing_time / queries.n << "," << time / queries.n << ","
            << recall << std::endl;
    }
    return 0;
}
