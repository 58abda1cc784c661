// partition.cpp — this build's Partition driver (reference partition.cpp:36-144): same argv, same output file names
// (<output>.k=<k>.<method>[.o=<overlap>]) and formats (METIS partition for Random, clusters file otherwise).  The k-means family
// of methods is built (k-means on the GPU); the graph / HNSW based ones need KaMinPar / hnswlib, which the reference checkout does
// not contain, and end like an unknown method does in the reference.
#include <cmath>
#include <fstream>
#include <iostream>
#include <random>

#include "kmeans.h"
#include "metis_io.h"
#include "overlapping_partitioning.h"
#include "partitioning.h"
#include "points_io.h"

int main(int argc, const char* argv[]) {
    if (argc != 6 && argc != 7) {
        std::cerr << "Usage ./Partition input-points output-path num-clusters partitioning-method (default|strong) [overlap]" << std::endl;
        std::abort();
    }
    const std::string input_file = argv[1], output_file = argv[2], k_str = argv[3], config = argv[5];
    std::string part_method = argv[4];
    const int k = std::stoi(k_str);
    std::string part_file = output_file + ".k=" + k_str + "." + part_method;
    if (config != "strong" && config != "default") throw std::runtime_error("Unknown config: " + config);
    double overlap = 0.0;
    if (argc == 7) {
        overlap = std::stod(argv[6]);
        part_file += std::string(".o=") + argv[6];
    }

    if (part_method == "Random") {   // :64-80: n/k ids per block, shuffled with mt19937(555); only the header of the input is read
        uint32_t n = 0;
        {
            std::ifstream in(input_file, std::ios::binary);
            in.read(reinterpret_cast<char*>(&n), sizeof(uint32_t));
        }
        std::vector<int> partition;
        partition.reserve(n);
        for (int b = 0; b < k; ++b) partition.insert(partition.end(), n / k, b);
        std::mt19937 prng(555);
        std::shuffle(partition.begin(), partition.end(), prng);
        WriteMetisPartition(partition, part_file);
        return 0;
    }

    PointSet points = ReadPoints(input_file);
    std::cout << "Finished reading points" << std::endl;

    const double eps = 0.05;
    const size_t max_cluster_size = (1.0 + eps) * points.n / k;   // the bound RKM / ORKM pass on (:103, :106)
    std::vector<int> partition;
    Clusters clusters;
    if (part_method == "KMeans") {
        partition = KMeansPartitioning(points, k, eps);
    } else if (part_method == "FlatKMeans") {   // :23-26
        PointSet centroids = RandomSample(points, k, 555);
        partition = KMeans(points, centroids);
    } else if (part_method == "RKM") {
        partition = RebalancingKMeansPartitioning(points, max_cluster_size, k);
    } else if (part_method == "ORKM") {
        const int adjusted_num_clusters = (int)std::ceil(k * (1.0 + overlap));
        auto rkm = RebalancingKMeansPartitioning(points, max_cluster_size, adjusted_num_clusters);
        clusters = OverlappingKMeansPartitioningSPANN(points, rkm, k, eps, overlap);
    } else if (part_method == "OKM") {
        Timer timer;
        timer.Start();
        auto kmp = KMeansPartitioning(points, k, eps);
        std::cout << "KM took " << timer.Stop() << " seconds" << std::endl;
        clusters = OverlappingKMeansPartitioningSPANN(points, kmp, k, eps, overlap);
    } else {
        std::cout << "Unsupported partitioning method " << part_method << " . The supported options are [Random, KMeans, FlatKMeans, RKM, ORKM, OKM]"
                  << std::endl;
        std::abort();
    }
    std::cout << "Finished partitioning" << std::endl;
    if (clusters.empty()) clusters = ConvertPartitionToClusters(partition);
    WriteClusters(clusters, part_file);
    return 0;
}
