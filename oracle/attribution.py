"""CPU restatement of the reference's route/search replay — TEST INFRASTRUCTURE ONLY (see pyoracle.py).

Restates, statement by statement, /root/reference/src/route_search_combination.cpp:
  :9-35    AttributeRecallAndQueryTimeIncreasingNumProbes
  :37-63   AttributeRecallAndQueryTimeVariableNumProbes
  :121-226 PrintCombinationsOfRoutesAndSearches (analytic QPS with the replica loop :149-183, Pareto filter :194-211, QPS sort :215-217)
and the two text formats it consumes: .routes (src/routes.cpp:317-381) and .searches (src/shard_searches.cpp:108-183).

Pure Python (host-side f64 bookkeeping, small cases).  Numbers are formatted like the reference's `std::stringstream << double`,
i.e. printf("%g") with 6 significant digits.  Parity pin: the reference has no test or golden file for this path and the functions
need parlay (absent) to compile — parity unpinned by reference code; the restatement is checked against the host mirror's Convert /
QueryAttribution drivers row for row (tests/test_gpu_dropin.py) and against hand-computed cases (tests/test_attribution.py).
"""

HEADER = ("partitioning,shard query,routing query,routing index,ef-search-shard,num voting points,routing time,num probes,recall,QPS,QPS per host,"
          "QPS without routing, QPS without routing per host,num hosts,num shards,requested num shards"
          ",routing index size,min cluster size,num centroids")


def g(x):
    """`ostream << double` with the default precision: %g"""
    return "%g" % x


def read_routes(path):
    """DeserializeRoutes (src/routes.cpp:363-381) -> list of dicts"""
    lines = open(path).read().split("\n")
    n = int(lines[0])
    pos, routes = 1, []
    for _ in range(n):
        assert lines[pos] == "R"
        h = lines[pos + 1].split()
        nq = int(h[6])
        routes.append(dict(routing_algorithm=h[0], index_trainer=h[1], votes=int(h[2]), ef_search=int(h[3]), routing_time=float(h[4]),
                           try_increasing_num_shards=h[5] == "true", budget=int(h[7]), num_centroids=int(h[8]), min_cluster_size=int(h[9]),
                           buckets_to_probe=[[int(x) for x in lines[pos + 2 + q].split()] for q in range(nq)]))
        pos += 2 + nq
    return routes


def read_searches(path):
    """DeserializeShardSearches (src/shard_searches.cpp:146-183) -> list of dicts: neighbors[b][q] ids, time_query_in_shard[b][q]"""
    lines = open(path).read().split("\n")
    n = int(lines[0])
    pos, out = 1, []
    for _ in range(n):
        assert lines[pos] == "S"
        ef, S, Q = (int(x) for x in lines[pos + 1].split())
        neigh = [[[int(x) for x in lines[pos + 2 + b * Q + q].split()] for q in range(Q)] for b in range(S)]
        times = [[float(x) for x in lines[pos + 2 + S * Q + b].split()] for b in range(S)]
        out.append(dict(ef_search=ef, neighbors=neigh, time_query_in_shard=times))
        pos += 2 + S * Q + S
    return out


def attribute_increasing(route, search, num_queries, num_shards, num_neighbors):
    """:9-35 — yields (local_work copy, total_hits, n_probes) for n_probes = 1..num_shards"""
    local_work = [0.0] * num_shards
    unique = [set() for _ in range(num_queries)]
    for n_probes in range(1, num_shards + 1):
        for q in range(num_queries):
            b = route["buckets_to_probe"][q][n_probes - 1]
            local_work[b] += search["time_query_in_shard"][b][q]
        total_hits = 0
        for q in range(num_queries):
            b = route["buckets_to_probe"][q][n_probes - 1]
            unique[q].update(search["neighbors"][b][q])
            total_hits += min(len(unique[q]), num_neighbors)
        yield list(local_work), total_hits, float(n_probes)


def attribute_variable(route, search, num_queries, num_shards, num_neighbors):
    """:37-63"""
    local_work = [0.0] * num_shards
    total_hits = total_probes = 0
    for q in range(num_queries):
        u = set()
        for b in route["buckets_to_probe"][q]:
            u.update(search["neighbors"][b][q])
        total_hits += min(len(u), num_neighbors)
        total_probes += len(route["buckets_to_probe"][q])
    for q in range(num_queries):
        for b in route["buckets_to_probe"][q]:
            local_work[b] += search["time_query_in_shard"][b][q]
    yield local_work, total_hits, float(total_probes) / num_queries


def combinations(routes, searches, num_neighbors, num_queries, num_shards, num_requested_shards, part_method):
    """:121-190 — every (route, search, probes, hosts) row before the Pareto filter: list of (row string, recall, QPS)"""
    outputs = []
    for route in routes:
        for search in searches:
            gen = attribute_increasing if route["try_increasing_num_shards"] else attribute_variable
            for local_work, total_hits, n_probes in gen(route, search, num_queries, num_shards, num_neighbors):
                recall = float(total_hits) / float(num_neighbors * num_queries)
                lwr = list(local_work)
                assigned = [1] * num_shards
                max_num_hosts = max(num_requested_shards + 20, num_shards)
                num_hosts = num_shards
                while num_hosts <= max_num_hosts:
                    max_shard = lwr.index(max(lwr))        # std::max_element: the first maximum
                    max_latency = lwr[max_shard]
                    qps_wo = num_queries / max_latency if max_latency != 0.0 else float("inf")
                    qps_wo_host = qps_wo / num_hosts
                    total_time = max_latency + route["routing_time"] / num_hosts
                    qps = num_queries / total_time if total_time != 0.0 else float("inf")
                    qps_host = qps / num_hosts
                    row = ",".join([part_method, "HNSW", route["routing_algorithm"], route["index_trainer"], str(search["ef_search"]), str(route["votes"]),
                                    g(route["routing_time"] / num_queries), g(n_probes), g(recall), g(qps), g(qps_host), g(qps_wo), g(qps_wo_host),
                                    str(num_hosts), str(num_shards), str(num_requested_shards), str(route["budget"]), str(route["min_cluster_size"]),
                                    str(route["num_centroids"])])
                    outputs.append((row, recall, qps))
                    assigned[max_shard] += 1
                    lwr[max_shard] = local_work[max_shard] / assigned[max_shard]
                    num_hosts += 1
    return outputs


def pareto_front(outputs):
    """:192-217 — the reference's in-place filter (swap with the back on removal), then sorted by QPS descending"""
    def dominates(l, r):
        return l[1] <= r[1] and l[2] <= r[2]
    pareto = []
    for c in outputs:
        insert_new = True
        i = 0
        while i < len(pareto):
            if dominates(pareto[i], c):
                pareto[i] = pareto[-1]
                pareto.pop()
                continue
            if dominates(c, pareto[i]):
                insert_new = False
                break
            i += 1
        if insert_new:
            pareto.append(c)
    return sorted(pareto, key=lambda t: -t[2])   # (std::sort is unstable: rows of equal QPS may come in any order)


def pareto_rows(routes, searches, num_neighbors, num_queries, num_shards, num_requested_shards, part_method):
    return [r[0] for r in pareto_front(combinations(routes, searches, num_neighbors, num_queries, num_shards, num_requested_shards, part_method))]
