// Partition — shard synthesis driver of this build.  Contract taken from the reference's driver (partition.cpp:36-144), nothing else:
//   argv      input-points  output-path  num-clusters  method  default|strong  [overlap]
//   output    <output-path>.k=<num-clusters>.<method>[.o=<overlap>]   ("Random": a METIS partition, one shard id per line; every other method: a
//             clusters file, one line of point ids per shard)
// Methods are rows of a table (kMethods): what each needs from the command line and the function that produces the shards.  Only the k-means family is
// built (k-means runs on the GPU through libgpann_host); the graph / HNSW methods of the reference need KaMinPar / hnswlib and are not in the table.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "kmeans.h"
#include "metis_io.h"
#include "overlapping_partitioning.h"
#include "partitioning.h"
#include "points_io.h"

namespace {

constexpr double kImbalance = 0.05;   // eps of every balanced method (reference partition.cpp:87)
constexpr int kSeed = 555;

struct Request {
    std::string input, method, out_file;
    int shards = 0;
    double overlap = 0.0;
};

// what a method hands back: a partition (one shard per point) or shards that may overlap
struct Shards {
    std::vector<int> partition;
    Clusters clusters;
};

size_t ShardCapacity(const PointSet& points, int shards) { return (size_t)((1.0 + kImbalance) * points.n / shards); }

Shards FlatLloyd(PointSet& points, const Request& rq) {   // RandomSample + 20 Lloyd rounds, no balancing
    PointSet seeds = RandomSample(points, rq.shards, kSeed);
    return {KMeans(points, seeds), {}};
}
Shards BalancedByRecursion(PointSet& points, const Request& rq) { return {KMeansPartitioning(points, rq.shards, kImbalance), {}}; }
Shards BalancedByMigration(PointSet& points, const Request& rq) {
    return {RebalancingKMeansPartitioning(points, ShardCapacity(points, rq.shards), rq.shards), {}};
}
Shards OverlapOnMigration(PointSet& points, const Request& rq) {   // more, smaller base clusters leave room for the extra assignments
    const int base_clusters = (int)std::ceil(rq.shards * (1.0 + rq.overlap));
    std::vector<int> base = RebalancingKMeansPartitioning(points, ShardCapacity(points, rq.shards), base_clusters);
    return {{}, OverlappingKMeansPartitioningSPANN(points, base, rq.shards, kImbalance, rq.overlap)};
}
Shards OverlapOnRecursion(PointSet& points, const Request& rq) {
    Timer clock;
    clock.Start();
    std::vector<int> base = KMeansPartitioning(points, rq.shards, kImbalance);
    std::cout << "KM took " << clock.Stop() << " seconds" << std::endl;
    return {{}, OverlappingKMeansPartitioningSPANN(points, base, rq.shards, kImbalance, rq.overlap)};
}

struct Method {
    const char* name;
    std::function<Shards(PointSet&, const Request&)> run;
};
const Method kMethods[] = {
    {"KMeans", BalancedByRecursion}, {"FlatKMeans", FlatLloyd},        {"RKM", BalancedByMigration},
    {"ORKM", OverlapOnMigration},    {"OKM", OverlapOnRecursion},
};

// "Random" never looks at the coordinates: the header's point count, floor(n / shards) ids per shard, one shuffle
int WriteRandomPartition(const Request& rq) {
    uint32_t n_points = 0;
    std::ifstream header(rq.input, std::ios::binary);
    header.read(reinterpret_cast<char*>(&n_points), sizeof n_points);
    std::vector<int> shard_of;
    shard_of.reserve(n_points);
    const size_t per_shard = n_points / (uint32_t)rq.shards;
    for (int b = 0; b < rq.shards; ++b) shard_of.insert(shard_of.end(), per_shard, b);
    std::mt19937 rng(kSeed);
    std::shuffle(shard_of.begin(), shard_of.end(), rng);
    WriteMetisPartition(shard_of, rq.out_file);
    return 0;
}

std::string MethodNames() {
    std::ostringstream os;
    os << "Random";
    for (const Method& m : kMethods) os << ", " << m.name;
    return os.str();
}

}   // namespace

int main(int argc, const char* argv[]) {
    if (argc < 6 || argc > 7) {
        std::cerr << "Usage ./Partition input-points output-path num-clusters partitioning-method (default|strong) [overlap]" << std::endl;
        std::abort();
    }
    Request rq;
    rq.input = argv[1];
    rq.method = argv[4];
    rq.shards = std::stoi(argv[3]);
    rq.out_file = std::string(argv[2]) + ".k=" + argv[3] + "." + rq.method;
    const std::string strength = argv[5];   // only the graph partitioners read it; the value is still checked
    if (strength != "default" && strength != "strong") throw std::runtime_error("Unknown config: " + strength);
    if (argc == 7) {
        rq.overlap = std::stod(argv[6]);
        rq.out_file += std::string(".o=") + argv[6];
    }
    if (rq.method == "Random") return WriteRandomPartition(rq);

    const Method* chosen = nullptr;
    for (const Method& m : kMethods)
        if (rq.method == m.name) chosen = &m;
    if (chosen == nullptr) {   // before the points are read: a typo should not cost a 50 GB load
        std::cout << "Unsupported partitioning method " << rq.method << " . The supported options are [" << MethodNames() << "]" << std::endl;
        std::abort();
    }
    PointSet points = ReadPoints(rq.input);
    std::cout << "Finished reading points" << std::endl;
    Shards result = chosen->run(points, rq);
    std::cout << "Finished partitioning" << std::endl;
    WriteClusters(result.clusters.empty() ? ConvertPartitionToClusters(result.partition) : result.clusters, rq.out_file);
    return 0;
}
