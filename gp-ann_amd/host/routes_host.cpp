// routes_host.cpp — host mirror of the KMeansTreeRouter part of routes.{h,cpp} and of route_search_combination.{h,cpp}.
// Pure host post-processing of GPU results: replay routes over recorded per-shard hits, the analytic
// QPS model with replicas, the Pareto filter and the .routes / .pareto file formats.
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <tuple>
#include <unordered_set>

#include "gpann_host.h"
#include "host_util.h"

using gpann_host::ParallelFor;

// routes.cpp:11-31
double MaxFirstShardRoutingRecall(const std::vector<std::vector<int>>& buckets_to_probe, const std::vector<NNVec>& ground_truth, int num_neighbors,
                                  const Cover& cover) {
    if (ground_truth.empty()) {
        std::cerr << "Ground truth empty. Max recall calculation during routing not possible (not necessarily an issue).";
        return 555.0;
    }
    size_t hits = 0;
    const size_t nq = buckets_to_probe.size();
    for (size_t q = 0; q < nq; ++q) {
        if (buckets_to_probe[q].empty()) continue;
        const int first = buckets_to_probe[q][0];
        for (int i = 0; i < num_neighbors; ++i) {
            const auto& homes = cover[ground_truth[q][i].second];
            hits += std::find(homes.begin(), homes.end(), first) != homes.end();
        }
    }
    return static_cast<double>(hits) / nq / num_neighbors;
}

namespace {

// routes.cpp:130-157: 8 index budgets, min cluster size 350, 64 centroids (+128 for the two largest budgets)
std::vector<KMeansTreeRouterOptions> RouterConfigGrid(KMeansTreeRouterOptions blueprint) {
    std::vector<KMeansTreeRouterOptions> grid;
    for (int64_t budget : {20'000, 100'000, 200'000, 500'000, 1'000'000, 2'000'000, 5'000'000, 10'000'000}) {
        KMeansTreeRouterOptions o = blueprint;
        o.budget = budget;
        o.min_cluster_size = 350;
        o.num_centroids = 64;
        grid.push_back(o);
        if (budget >= 5'000'000) {
            o.num_centroids = 128;
            grid.push_back(o);
        }
    }
    return grid;
}

}   // namespace

// routes.cpp:164-313, KMeansTreeRouter section.  Routing of the whole query set is one GPU batch; routing_time is the
// wall time of that batch (the reference times one parallel_for over the queries, :195-199).
std::vector<RoutingConfig> IterateRoutingConfigs(PointSet& points, PointSet& queries, const Clusters& clusters, int num_shards,
                                                 KMeansTreeRouterOptions routing_index_options_blueprint, const std::vector<NNVec>& ground_truth,
                                                 int num_neighbors, const std::string& /*routing_index_file*/, const std::string& pyramid_index_file,
                                                 const std::string& our_pyramid_index_file) {
    std::vector<RoutingConfig> routes;
    const Cover cover = ConvertClustersToCover(clusters);
    std::vector<KMeansTreeRouterOptions> grid = RouterConfigGrid(routing_index_options_blueprint);
    if (const char* cap = std::getenv("GPANN_MAX_ROUTER_BUDGET")) {   // optional: skip the largest index sizes
        const int64_t limit = std::atoll(cap);
        grid.erase(std::remove_if(grid.begin(), grid.end(), [&](const auto& o) { return o.budget > limit; }), grid.end());
    }
    for (const KMeansTreeRouterOptions& opts : grid) {
        std::cout << "Train router on " << opts.num_centroids << " centroids " << opts.min_cluster_size << " min cluster size " << opts.budget
                  << " size budget " << std::endl;
        Timer timer;
        timer.Start();
        KMeansTreeRouter router;
        router.Train(points, clusters, opts);
        std::cout << "Training the router took " << timer.Stop() << std::endl;

        {   // standard tree-search routing
            timer.Start();
            std::vector<std::vector<int>> probes = router.QueryBatch(queries, (int)opts.search_budget);
            const double t = timer.Stop();
            std::cout << "KMTR Routing took " << t << " s overall, and " << t / queries.n
                      << " s per query. Max first shard recall = " << MaxFirstShardRoutingRecall(probes, ground_truth, num_neighbors, cover) << std::endl;
            RoutingConfig& r = routes.emplace_back();
            r.routing_algorithm = "KMeansTree";
            r.hnsw_num_voting_neighbors = 0;
            r.routing_time = t;
            r.routing_index_options = opts;
            r.routing_distance_calcs = opts.search_budget;
            r.try_increasing_num_shards = true;
            r.buckets_to_probe = std::move(probes);
        }
        {   // frequency tree-search routing: one walk with 500 voters, replayed for 9 vote counts (:216-244)
            timer.Start();
            std::vector<KMeansTreeRouter::FrequencyQueryData> data = router.FrequencyQueryBatch(queries, (int)opts.search_budget, 500);
            const double t = timer.Stop();
            std::cout << "KMTR Frequency Routing took " << t << " s overall, and " << t / queries.n << " s per query" << std::endl;
            for (size_t voters : {20, 40, 80, 120, 200, 250, 300, 400, 500}) {
                std::vector<std::vector<int>> probes(queries.n);
                ParallelFor(0, queries.n, 0, [&](size_t q) { probes[q] = data[q].Query(num_shards, (int)voters); });
                std::cout << "Num voting points " << voters
                          << " Max first shard recall = " << MaxFirstShardRoutingRecall(probes, ground_truth, num_neighbors, cover) << std::endl;
                RoutingConfig& r = routes.emplace_back();
                r.routing_algorithm = "KMeansTree-Frequency";
                r.hnsw_num_voting_neighbors = voters;
                r.routing_time = t;
                r.routing_index_options = opts;
                r.routing_distance_calcs = opts.search_budget;
                r.try_increasing_num_shards = true;
                r.buckets_to_probe = std::move(probes);
            }
        }
        auto [routing_points, routing_partition] = router.ExtractPoints();
        std::cout << "Extraction finished. Index size = " << routing_points.n << std::endl;
        // The reference continues with HNSW / Pyramid / SPANN routing over these points (:248-311); those need hnswlib,
        // which is not part of the reference checkout, and are out of scope here.
    }
    if (!pyramid_index_file.empty() || !our_pyramid_index_file.empty())
        std::cerr << "Pyramid routing indices are HNSW based and not supported by this build; skipped." << std::endl;
    return routes;
}

// routes.cpp:317-381
std::string RoutingConfig::Serialize() const {
    std::stringstream sb;
    sb << routing_algorithm << " " << index_trainer << " " << hnsw_num_voting_neighbors << " " << hnsw_ef_search << " " << routing_time << " "
       << std::boolalpha << try_increasing_num_shards << std::noboolalpha << " " << buckets_to_probe.size() << " " << routing_index_options.budget << " "
       << routing_index_options.num_centroids << " " << routing_index_options.min_cluster_size << "\n";
    for (const auto& order : buckets_to_probe) {
        for (int b : order) sb << b << " ";
        sb << "\n";
    }
    return sb.str();
}

RoutingConfig RoutingConfig::Deserialize(std::ifstream& in) {
    RoutingConfig r;
    std::string line;
    std::getline(in, line);
    int nq = 0;
    {
        std::istringstream head(line);
        head >> r.routing_algorithm >> r.index_trainer >> r.hnsw_num_voting_neighbors >> r.hnsw_ef_search >> r.routing_time >> std::boolalpha >>
            r.try_increasing_num_shards >> std::noboolalpha >> nq >> r.routing_index_options.budget >> r.routing_index_options.num_centroids >>
            r.routing_index_options.min_cluster_size;
    }
    r.buckets_to_probe.reserve(nq);
    for (int q = 0; q < nq; ++q) {
        std::getline(in, line);
        std::istringstream tokens(line);
        auto& order = r.buckets_to_probe.emplace_back();
        for (int b; tokens >> b;) order.push_back(b);
    }
    return r;
}

void SerializeRoutes(const std::vector<RoutingConfig>& routes, const std::string& output_file) {
    std::ofstream out(output_file);
    out << routes.size() << std::endl;
    for (const auto& r : routes) {
        out << "R" << std::endl;
        out << r.Serialize();
    }
}

std::vector<RoutingConfig> DeserializeRoutes(const std::string& input_file) {
    std::ifstream in(input_file);
    std::string line;
    std::getline(in, line);
    size_t count = 0;
    std::istringstream(line) >> count;
    std::vector<RoutingConfig> routes;
    for (size_t i = 0; i < count; ++i) {
        std::getline(in, line);
        if (line != "R") std::cout << "routing config doesn't start with marker R. Instead: " << line << std::endl;
        routes.push_back(RoutingConfig::Deserialize(in));
    }
    return routes;
}

// ================================================================================================== route_search_combination
// :9-35 — probes are added one rank at a time; hits of a query = min(#distinct recorded neighbours so far, k)
void AttributeRecallAndQueryTimeIncreasingNumProbes(const RoutingConfig& route, const ShardSearch& search, size_t num_queries, size_t num_shards,
                                                    int num_neighbors, std::function<void(EmitResult)>& emit) {
    std::vector<double> work(num_shards, 0.0);
    std::vector<std::unordered_set<uint32_t>> seen(num_queries);
    for (size_t rank = 1; rank <= num_shards; ++rank) {
        for (size_t q = 0; q < num_queries; ++q) {
            const int b = route.buckets_to_probe.at(q).at(rank - 1);
            work[b] += search.time_query_in_shard[b][q];
        }
        std::atomic<size_t> hits{0};
        ParallelFor(0, num_queries, 0, [&](size_t q) {
            const int b = route.buckets_to_probe[q].at(rank - 1);
            for (uint32_t id : search.neighbors[b][q]) seen[q].insert(id);
            hits += std::min<size_t>(seen[q].size(), num_neighbors);
        });
        emit(EmitResult{work, hits.load(), static_cast<double>(rank)});
    }
}

// :37-63 — every query uses its whole (variable-length) probe list
void AttributeRecallAndQueryTimeVariableNumProbes(const RoutingConfig& route, const ShardSearch& search, size_t num_queries, size_t num_shards,
                                                  int num_neighbors, std::function<void(EmitResult)>& emit) {
    std::vector<double> work(num_shards, 0.0);
    std::atomic<size_t> hits{0}, probes{0};
    ParallelFor(0, num_queries, 0, [&](size_t q) {
        std::unordered_set<uint32_t> seen;
        for (int b : route.buckets_to_probe[q])
            for (uint32_t id : search.neighbors[b][q]) seen.insert(id);
        hits += std::min<size_t>(seen.size(), num_neighbors);
        probes += route.buckets_to_probe[q].size();
    });
    for (size_t q = 0; q < num_queries; ++q)
        for (int b : route.buckets_to_probe[q]) work[b] += search.time_query_in_shard[b][q];
    emit(EmitResult{work, hits.load(), double(probes.load()) / num_queries});
}

// :65-83
void MaxShardSearchRecall(const std::vector<ShardSearch>& shard_searches, int num_neighbors, int num_queries, int num_shards, int num_requested_shards) {
    std::cout << "k = " << num_neighbors << " nq = " << num_queries << " num shards = " << num_shards << " num requested = " << num_requested_shards
              << std::endl;
    for (const auto& search : shard_searches) {
        std::atomic<size_t> hits{0};
        ParallelFor(0, (size_t)num_queries, 0, [&](size_t q) {
            std::unordered_set<uint32_t> seen;
            for (const auto& shard : search.neighbors)
                for (uint32_t id : shard[q]) seen.insert(id);
            hits += std::min<size_t>(num_neighbors, seen.size());
        });
        std::cout << "Search with ef_search = " << search.ef_search << " scored " << double(hits.load()) / double(num_queries) / num_neighbors
                  << " total recall" << std::endl;
    }
}

// :85-116
void MaxRoutingRecall(const std::vector<RoutingConfig>& routes, const std::vector<NNVec>& ground_truth, int num_neighbors,
                      const std::vector<int>& partition, int num_shards) {
    for (const auto& route : routes) {
        std::vector<size_t> hits(num_shards, 0);
        for (size_t q = 0; q < route.buckets_to_probe.size(); ++q) {
            std::vector<int> in_shard(num_shards, 0);
            for (int i = 0; i < num_neighbors; ++i) in_shard[partition[ground_truth[q][i].second]]++;
            const int ranks = std::min<int>(num_shards, (int)route.buckets_to_probe[q].size());
            for (int r = 0; r < ranks; ++r) hits[r] += in_shard[route.buckets_to_probe[q][r]];
        }
        std::cout << "Route with " << route.index_trainer << " " << route.routing_algorithm << " " << route.hnsw_num_voting_neighbors << std::endl;
        size_t sum = 0;
        for (int r = 0; r < num_shards; ++r) {
            sum += hits[r];
            if (route.try_increasing_num_shards)
                std::cout << "Nprobes = " << r + 1 << " recall = " << double(sum) / num_neighbors / ground_truth.size() << "\t";
        }
        if (!route.try_increasing_num_shards) std::cout << "recall = " << double(sum) / num_neighbors / ground_truth.size();
        std::cout << std::endl;
    }
}

// :121-226 — simulated cluster: one host per shard plus up to 20 replicas for the slowest shard; Pareto front on (recall, QPS)
void PrintCombinationsOfRoutesAndSearches(const std::vector<RoutingConfig>& routes, const std::vector<ShardSearch>& shard_searches,
                                          const std::string& output_file, int num_neighbors, int num_queries, int num_shards, int num_requested_shards,
                                          const std::string& part_method) {
    const std::string header =
        "partitioning,shard query,routing query,routing index,ef-search-shard,num voting points,routing time,num probes,recall,QPS,QPS per host,"
        "QPS without routing, QPS without routing per host,num hosts,num shards,requested num shards"
        ",routing index size,min cluster size,num centroids"
        "\n";
    struct Row {
        std::string text;
        double recall, qps;
    };
    std::vector<Row> rows;
    for (const auto& route : routes) {
        for (const auto& search : shard_searches) {
            std::function<void(EmitResult)> on_result = [&](const EmitResult& r) {
                const double recall = static_cast<double>(r.total_hits) / static_cast<double>(num_neighbors * num_queries);
                std::vector<double> load = r.local_work;
                std::vector<size_t> replicas(num_shards, 1);
                const size_t max_hosts = std::max(num_requested_shards + 20, num_shards);
                for (size_t hosts = num_shards; hosts <= max_hosts; ++hosts) {
                    const size_t slowest = std::distance(load.begin(), std::max_element(load.begin(), load.end()));
                    const double bottleneck = load[slowest];
                    const double qps_no_routing = num_queries / bottleneck;
                    const double qps = num_queries / (bottleneck + route.routing_time / hosts);
                    std::stringstream s;
                    s << part_method << ",HNSW," << route.routing_algorithm << "," << route.index_trainer << "," << search.ef_search << ","
                      << route.hnsw_num_voting_neighbors << "," << route.routing_time / num_queries << "," << r.n_probes << "," << recall << "," << qps << ","
                      << qps / hosts << "," << qps_no_routing << "," << qps_no_routing / hosts << "," << hosts << "," << num_shards << ","
                      << num_requested_shards << "," << route.routing_index_options.budget << "," << route.routing_index_options.min_cluster_size << ","
                      << route.routing_index_options.num_centroids << "\n";
                    rows.push_back(Row{s.str(), recall, qps});
                    replicas[slowest]++;   // one more replica for the slowest shard
                    load[slowest] = r.local_work[slowest] / replicas[slowest];
                }
            };
            if (route.try_increasing_num_shards)
                AttributeRecallAndQueryTimeIncreasingNumProbes(route, search, num_queries, num_shards, num_neighbors, on_result);
            else
                AttributeRecallAndQueryTimeVariableNumProbes(route, search, num_queries, num_shards, num_neighbors, on_result);
        }
    }
    // Pareto filter (:187-212), same sweep order as the reference so ties resolve identically
    auto dominated_by = [](const Row& l, const Row& r) { return l.recall <= r.recall && l.qps <= r.qps; };
    std::vector<Row> front;
    for (const Row& cand : rows) {
        bool keep = true;
        for (int64_t i = 0; i < (int64_t)front.size(); ++i) {
            if (dominated_by(front[i], cand)) {
                front[i] = std::move(front.back());
                front.pop_back();
                --i;
            } else if (dominated_by(cand, front[i])) {
                keep = false;
                break;
            }
        }
        if (keep) front.push_back(cand);
    }
    std::cout << "Pareto size " << front.size() << std::endl;
    std::sort(front.begin(), front.end(), [](const Row& l, const Row& r) { return l.qps > r.qps; });
    std::ofstream out(output_file + ".pareto");
    out << header;
    for (const Row& r : front) out << r.text;
}
