// router_harness — debugging aid for the router's matrix-core bound path: ONE fresh process does what the reference's
// driver does — KMeansTreeRouter::Train, then immediately the first QueryBatch with pageable host queries on the null stream
// (routes.cpp:185-200) — with GPANN_ROUTER_VERIFY=1, i.e. every minimum of the bound path is compared with the canonical evaluation of
// all (query, centroid) pairs inside the library, and then routes the same queries again with GPANN_ROUTER_BOUND=0 and compares the
// probe lists.  tools/run_router_harness.sh starts it N times.  Exit code 0 = clean, 3 = a difference (report on stderr).
//
//   router_harness train <n> <d> <S> <nq> <num_centroids> <min_cluster_size> <index_budget> <search_budget> [seed]
//   router_harness flat  <C_per_root> <d> <S> <nq> <leaf_min> <leaf_max> <search_budget> [seed]      (random tree, no training)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/gpann.h"
#include "../include/gpann_synth.h"

#define CHECK(expr)                                                                                  \
    do {                                                                                             \
        gpann_status _s = (expr);                                                                    \
        if (_s != GPANN_OK) {                                                                        \
            fprintf(stderr, "HARNESS FAIL %s -> %d: %s\n", #expr, (int)_s, gpann_last_error());      \
            return _s == GPANN_ERR_INTERNAL ? 3 : 2;                                                 \
        }                                                                                            \
    } while (0)

static std::vector<float> Mixture(uint64_t seed, uint64_t center_seed, uint32_t n_centers, size_t n, uint32_t d, float offset = 0.f) {
    std::vector<float> out(n * d);
    for (size_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < d; ++j) out[i * d + j] = gpann_synth_coord(seed, center_seed, n_centers, 0.35f, i, j) + offset;
    return out;
}

int main(int argc, char** argv) {
    if (argc < 2) return 64;
    const std::string mode = argv[1];
    setenv("GPANN_ROUTER_VERIFY", "1", 1);
    setenv("GPANN_ROUTER_STRICT", "1", 1);
    gpann_router* router = nullptr;
    uint32_t d = 0, S = 0, nq = 0;
    int32_t search_budget = 0;
    std::vector<float> queries;
    if (mode == "train" && argc >= 10) {
        const size_t n = strtoull(argv[2], nullptr, 10);
        d = atoi(argv[3]); S = atoi(argv[4]); nq = atoi(argv[5]);
        const uint32_t nc = atoi(argv[6]), mcs = atoi(argv[7]);
        const int64_t index_budget = atoll(argv[8]);
        search_budget = atoi(argv[9]);
        const uint64_t seed = argc > 10 ? strtoull(argv[10], nullptr, 10) : 555;
        std::vector<float> pts = Mixture(seed, 7, 32, n, d);
        queries = Mixture(seed + 1, 7, 32, nq, d);
        // shards: nearest of the first S points (any partition will do)
        std::vector<std::vector<uint32_t>> clusters(S);
        for (size_t i = 0; i < n; ++i) {
            uint32_t best = 0;
            float bd = 3e38f;
            for (uint32_t b = 0; b < S; ++b) {
                float s = 0;
                for (uint32_t j = 0; j < d; ++j) { const float t = pts[i * d + j] - pts[(size_t)b * d + j]; s += t * t; }
                if (s < bd) { bd = s; best = b; }
            }
            clusters[best].push_back((uint32_t)i);
        }
        std::vector<uint64_t> offs(S + 1, 0);
        std::vector<uint32_t> ids;
        for (uint32_t b = 0; b < S; ++b) { offs[b + 1] = offs[b] + clusters[b].size(); ids.insert(ids.end(), clusters[b].begin(), clusters[b].end()); }
        gpann_shards* store = nullptr;
        CHECK(gpann_shards_create(pts.data(), n, d, GPANN_METRIC_L2, ids.data(), offs.data(), S, 0, GPANN_STORE_PLAIN, &store));
        CHECK(gpann_router_train(store, n, nc, mcs, index_budget, &router));
        CHECK(gpann_shards_destroy(store));
    } else if (mode == "flat" && argc >= 9) {
        const uint32_t root_c = atoi(argv[2]);
        d = atoi(argv[3]); S = atoi(argv[4]); nq = atoi(argv[5]);
        const uint32_t leaf_min = atoi(argv[6]), leaf_max = atoi(argv[7]);
        search_budget = atoi(argv[8]);
        const uint64_t seed = argc > 9 ? strtoull(argv[9], nullptr, 10) : 555;
        queries = Mixture(seed + 1, 7, 1024, nq, d);
        // per shard: a root with root_c centroids, each with a leaf child of leaf_min..leaf_max centroids (BFS order: root, then its children)
        std::vector<int32_t> node_shard;
        std::vector<uint32_t> cent_first{0}, child_first, child_count, shard_root;
        uint64_t rng = seed * 0x9E3779B97F4A7C15ULL + 1;
        auto next = [&]() { rng = gpann_synth_mix64(rng); return rng; };
        for (uint32_t b = 0; b < S; ++b) {
            const uint32_t root = (uint32_t)node_shard.size();
            shard_root.push_back(root);
            node_shard.push_back((int32_t)b);
            child_first.push_back(root + 1);
            child_count.push_back(root_c);
            cent_first.push_back(cent_first.back() + root_c);
            for (uint32_t i = 0; i < root_c; ++i) {
                node_shard.push_back((int32_t)b);
                child_first.push_back(0);
                child_count.push_back(0);
                cent_first.push_back(cent_first.back() + leaf_min + (uint32_t)(next() % (leaf_max - leaf_min + 1)));
            }
        }
        const size_t C = cent_first.back();
        std::vector<float> cents = Mixture(seed + 2, 7, 1024, C, d);
        CHECK(gpann_router_create(d, GPANN_METRIC_L2, S, (uint32_t)node_shard.size(), node_shard.data(), cent_first.data(), child_first.data(), child_count.data(),
                                  shard_root.data(), cents.data(), 0, &router));
    } else {
        fprintf(stderr, "usage: see the header of tools/router_harness.cpp\n");
        return 64;
    }
    // the first call of the process, immediately after training: pageable host queries, pageable probe array, no min_dist, null stream
    std::vector<int32_t> probes((size_t)nq * S), probes2((size_t)nq * S), probes3((size_t)nq * S);
    CHECK(gpann_router_query_batch(router, queries.data(), nq, search_budget, probes.data(), nullptr, 0, nullptr, nullptr));
    CHECK(gpann_router_query_batch(router, queries.data(), nq, search_budget, probes2.data(), nullptr, 0, nullptr, nullptr));
    gpann_router_stats st{};
    CHECK(gpann_router_get_stats(router, &st));
    // the third call is the CANONICAL route (every pair evaluated exactly).  The handle took its options from the environment when it was created and
    // does not read the environment again: the knob is set on the handle, and read back so that a bound-against-bound comparison cannot pass for a check
    CHECK(gpann_router_set_option(router, "ROUTER_BOUND", "0"));
    {
        char val[16] = {0};
        CHECK(gpann_router_get_option(router, "ROUTER_BOUND", val, sizeof val));
        if (std::strcmp(val, "0") != 0) {
            fprintf(stderr, "HARNESS: ROUTER_BOUND reads back '%s', not 0\n", val);
            return 2;
        }
    }
    std::vector<float> md((size_t)nq * S);
    CHECK(gpann_router_query_batch(router, queries.data(), nq, search_budget, probes3.data(), md.data(), 0, nullptr, nullptr));
    gpann_router_stats st3{};
    CHECK(gpann_router_get_stats(router, &st3));
    if (st3.bound_calls != st.bound_calls) {   // (the reference call must not have gone through the path it is the reference for)
        fprintf(stderr, "HARNESS: the canonical call ran %llu bound blocks\n", (unsigned long long)(st3.bound_calls - st.bound_calls));
        return 2;
    }
    size_t bad1 = 0, bad2 = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        bad1 += std::memcmp(&probes[(size_t)q * S], &probes3[(size_t)q * S], S * 4) != 0;
        bad2 += std::memcmp(&probes2[(size_t)q * S], &probes3[(size_t)q * S], S * 4) != 0;
    }
    uint64_t nc = 0;
    uint32_t n_nodes = 0;
    gpann_router_info(router, nullptr, nullptr, nullptr, &n_nodes, &nc);
    if (bad1 || bad2 || st.bound_selfcheck || st.verify_failures || st.bound_calls == 0) {
        fprintf(stderr, "HARNESS MISMATCH: first call %zu / second call %zu of %u queries differ from the canonical route; bound calls %llu selfcheck %llu verify failures %llu\n",
                bad1, bad2, nq, (unsigned long long)st.bound_calls, (unsigned long long)st.bound_selfcheck, (unsigned long long)st.verify_failures);
        return 3;
    }
    printf("HARNESS OK: %u queries x %u shards, %llu centroids in %u nodes, bound blocks %llu, verified %llu\n", nq, S, (unsigned long long)nc, n_nodes,
           (unsigned long long)st.bound_calls, (unsigned long long)st.verified);
    gpann_router_destroy(router);
    return 0;
}
