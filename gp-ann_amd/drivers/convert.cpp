// convert.cpp — this build's Convert driver: the live branch of the reference's convert.cpp (:62-89): load a .routes and a
// .searches file, replay every route over every shard search and write the Pareto front of (recall, QPS) configurations
// (PrintCombinationsOfRoutesAndSearches, route_search_combination.cpp:121-226).  Same argv, same output format.  The
// reference's other branches are compiled out there (#if false) and are not reproduced.
#include <iostream>

#include "points_io.h"
#include "route_search_combination.h"
#include "routes.h"
#include "shard_searches.h"

int main(int argc, const char* argv[]) {
    if (argc != 6) {
        std::cerr << "Usage ./Convert routes searches output part-method query-file" << std::endl;
        std::abort();
    }
    const std::string routes_file = argv[1], searches_file = argv[2], output_file = argv[3], part_method = argv[4], query_file = argv[5];
    auto searches = DeserializeShardSearches(searches_file);
    std::cout << "Finished loading searches" << std::endl;
    auto routes = DeserializeRoutes(routes_file);
    std::cout << "num routes " << routes.size() << " num searches " << searches.size() << std::endl;
    const int num_actual_shards = (int)searches.front().neighbors.size();
    std::cout << "num actual shards = " << num_actual_shards << std::endl;
    auto queries = ReadPoints(query_file);
    const int num_queries = (int)queries.n;
    PrintCombinationsOfRoutesAndSearches(routes, searches, output_file, 10, num_queries, num_actual_shards, 40, part_method);
    return 0;
}
