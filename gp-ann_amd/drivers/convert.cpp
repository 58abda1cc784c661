// Convert — routes + shard searches -> Pareto front.  Contract (the one live branch of the reference's convert.cpp:62-89):
//   argv    routes-file  searches-file  output-file  part-method  query-file
//   output  <output-file>: the CSV PrintCombinationsOfRoutesAndSearches writes (route_search_combination.cpp:121-226), evaluated for k = 10 neighbours
//           and 40 requested hosts; the query file is read only for its point count
// Everything is host arithmetic over two text files (host/routes_host.cpp); no GPU work happens here.
#include <cstdlib>
#include <iostream>
#include <string>

#include "points_io.h"
#include "route_search_combination.h"
#include "routes.h"
#include "shard_searches.h"

namespace {
constexpr int kNeighbours = 10, kRequestedHosts = 40;
struct Inputs {
    std::string routes, searches, output, part_method, queries;
};
}   // namespace

int main(int argc, const char* argv[]) {
    if (argc != 6) {
        std::cerr << "Usage ./Convert routes searches output part-method query-file" << std::endl;
        std::abort();
    }
    const Inputs in{argv[1], argv[2], argv[3], argv[4], argv[5]};
    std::vector<ShardSearch> shard_searches = DeserializeShardSearches(in.searches);
    std::vector<RoutingConfig> routings = DeserializeRoutes(in.routes);
    if (shard_searches.empty() || routings.empty()) {
        std::cerr << "Convert: " << in.routes << " holds " << routings.size() << " routes, " << in.searches << " holds " << shard_searches.size()
                  << " searches; nothing to combine" << std::endl;
        return 1;
    }
    const int shards = (int)shard_searches.front().neighbors.size();
    const int n_queries = (int)ReadPoints(in.queries).n;
    std::cout << "Convert: " << routings.size() << " routes x " << shard_searches.size() << " searches, " << shards << " shards, " << n_queries
              << " queries" << std::endl;
    PrintCombinationsOfRoutesAndSearches(routings, shard_searches, in.output, kNeighbours, n_queries, shards, kRequestedHosts, in.part_method);
    return 0;
}
