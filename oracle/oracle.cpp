// oracle.cpp — CPU restatement of gp-ann's route + shard-scan path.  TEST INFRASTRUCTURE ONLY (see gpann_oracle.h).
//
// Every function cites the reference lines it follows (paths relative to the reference checkout).  The metric is a
// runtime argument here (the reference switches at compile time, src/dist.cpp:75-81): 0 = squared L2, 1 = MIPS.
//
// Build flags matter: this file must be compiled with  -O2 -fno-fast-math -ffp-contract=off  and WITHOUT -mavx so
// that the arithmetic is the reference's scalar branch (src/dist.cpp:17-37) evaluated in strict IEEE binary32.
// Calls whose result depends on the libstdc++ algorithm (std::sort on ties, std::partition, std::sample,
// std::priority_queue) are kept as the very same std:: calls the reference makes.
#include "gpann_oracle.h"
#include "../include/gpann_synth.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <numeric>
#include <queue>
#include <random>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// tiny fork-join helper (the reference uses parlay::parallel_for; every loop parallelised here is order-independent)
template <typename F>
void ParallelFor(size_t begin, size_t end, int nthreads, F f) {
    if (nthreads <= 1 || end - begin <= 1) {
        for (size_t i = begin; i < end; ++i) f(i);
        return;
    }
    size_t n = end - begin;
    size_t T = std::min<size_t>(nthreads, n);
    std::vector<std::thread> threads;
    threads.reserve(T);
    // static blocked partition with small grains claimed through a shared counter
    auto counter = std::make_shared<std::atomic<size_t>>(begin);
    size_t grain = std::max<size_t>(1, n / (T * 8));
    for (size_t t = 0; t < T; ++t) {
        threads.emplace_back([=]() {
            while (true) {
                size_t s = counter->fetch_add(grain);
                if (s >= end) break;
                size_t e = std::min(end, s + grain);
                for (size_t i = s; i < e; ++i) f(i);
            }
        });
    }
    for (auto& th : threads) th.join();
}

// ---------------------------------------------------------------------------------------------------------------
// src/dist.cpp:9-39, scalar (#else) branch
float SqrL2Dist(const float* a, const float* b, unsigned size) {
    float result = 0;
    float diff0, diff1, diff2, diff3;
    const float* last = a + size;
    const float* unroll_group = last - 3;
    while (a < unroll_group) {
        diff0 = a[0] - b[0];
        diff1 = a[1] - b[1];
        diff2 = a[2] - b[2];
        diff3 = a[3] - b[3];
        result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
        a += 4;
        b += 4;
    }
    while (a < last) {
        diff0 = *a++ - *b++;
        result += diff0 * diff0;
    }
    return result;
}

// src/dist.cpp:42-48
float InnerProduct(const float* p, const float* q, unsigned d) {
    float result = 0;
    for (unsigned i = 0; i < d; i++) {
        result += (q[i]) * (p[i]);
    }
    return result;
}

// src/dist.cpp:50-52
float MipsDistance(const float* p, const float* q, unsigned d) { return 1.0f - InnerProduct(p, q, d); }

// src/dist.cpp:54-58
float VecNorm(const float* p, unsigned d) {
    float result = 0.f;
    for (unsigned i = 0; i < d; ++i) result += p[i] * p[i];
    return result;
}

// src/dist.cpp:75-81 (argument order is (point, query) at every reference call site)
inline float Distance(const float* p, const float* q, unsigned d, int metric) {
    return metric == 1 ? MipsDistance(p, q, d) : SqrL2Dist(p, q, d);
}

// ---------------------------------------------------------------------------------------------------------------
// src/topn.h:6-32, verbatim semantics: bounded max-heap on pair<float,uint32_t>, replace top iff top > e.
struct TopN {
    explicit TopN(size_t n) : n(n) {}
    size_t n;
    using value_type = std::pair<float, uint32_t>;
    std::priority_queue<value_type> pq;
    void Add(value_type e) {
        if (pq.size() < n) {
            pq.push(e);
        } else if (pq.top() > e) {
            pq.pop();
            pq.push(e);
        }
    }
    std::vector<value_type> Take() {
        std::vector<value_type> res(pq.size());
        while (!pq.empty()) {
            res[pq.size() - 1] = pq.top();
            pq.pop();
        }
        return res;
    }
    const value_type& Top() const { return pq.top(); }
};

using NNVec = std::vector<std::pair<float, uint32_t>>;

// ---------------------------------------------------------------------------------------------------------------
// src/defs.h:10-21
struct PointSet {
    std::vector<float> coordinates;
    size_t d = 0, n = 0;
    float* GetPoint(size_t i) { return &coordinates[i * d]; }
    const float* GetPoint(size_t i) const { return &coordinates[i * d]; }
};

// src/defs.cpp:11-23
PointSet ExtractPointsInBucket(const std::vector<uint32_t>& bucket, const float* points, size_t d) {
    PointSet ps;
    ps.n = bucket.size();
    ps.d = d;
    ps.coordinates.reserve(ps.n * ps.d);
    for (auto u : bucket) {
        const float* p = points + (size_t)u * d;
        for (size_t j = 0; j < d; ++j) ps.coordinates.push_back(p[j]);
    }
    return ps;
}

// src/defs.cpp:25-37
int NumPartsInPartition(const std::vector<int>& partition) {
    if (partition.empty()) return 0;
    return *std::max_element(partition.begin(), partition.end()) + 1;
}
std::vector<std::vector<uint32_t>> ConvertPartitionToClusters(const std::vector<int>& partition) {
    int num_buckets = NumPartsInPartition(partition);
    std::vector<std::vector<uint32_t>> buckets(num_buckets);
    for (uint32_t u = 0; u < partition.size(); ++u) buckets[partition[u]].push_back(u);
    return buckets;
}

// ---------------------------------------------------------------------------------------------------------------
// src/kmeans.cpp
// :12-23
int Top1Neighbor(const PointSet& P, const float* Q, int metric) {
    int best = -1;
    float best_dist = std::numeric_limits<float>::max();
    for (size_t i = 0; i < P.n; ++i) {
        float new_dist = Distance(P.GetPoint(i), Q, P.d, metric);
        if (new_dist < best_dist) {
            best_dist = new_dist;
            best = i;
        }
    }
    return best;
}

// :25-27
void NearestCenters(const PointSet& P, const PointSet& centroids, std::vector<int>& closest_center, int metric, int nthreads) {
    ParallelFor(0, P.n, nthreads, [&](size_t i) { closest_center[i] = Top1Neighbor(centroids, P.GetPoint(i), metric); });
}

// :29-50
void RemoveEmptyClusters(PointSet& centroids, std::vector<int>& closest_center, const std::vector<size_t>& cluster_size) {
    if (std::any_of(cluster_size.begin(), cluster_size.end(), [](size_t x) { return x == 0; })) {
        std::vector<int> remapped_cluster_ids(centroids.n, -1);
        size_t l = 0;
        for (size_t r = 0; r < centroids.n; ++r) {
            if (cluster_size[r] != 0) {
                if (l != r) {
                    float* L = centroids.GetPoint(l);
                    float* R = centroids.GetPoint(r);
                    for (size_t j = 0; j < centroids.d; ++j) L[j] = R[j];
                }
                remapped_cluster_ids[r] = l;
                l++;
            }
        }
        centroids.n = l;
        centroids.coordinates.resize(centroids.n * centroids.d);
        for (size_t i = 0; i < closest_center.size(); ++i) closest_center[i] = remapped_cluster_ids[closest_center[i]];
    }
}

// :72-86 (MIPS) and :104-115 (L2)
void NormalizeCentroidsIP(PointSet& centroids, const std::vector<size_t>& cluster_size, const std::vector<float>& norm_sums) {
    for (size_t c = 0; c < centroids.n; ++c) {
        float* C = centroids.GetPoint(c);
        if (cluster_size[c] == 0) continue;
        float desired_norm = norm_sums[c] / cluster_size[c];
        float current_norm = VecNorm(C, centroids.d);
        float multiplier = std::sqrt(desired_norm / current_norm);
        for (size_t j = 0; j < centroids.d; ++j) C[j] *= multiplier;
    }
}
void NormalizeCentroidsL2(PointSet& centroids, const std::vector<size_t>& cluster_size) {
    for (size_t c = 0; c < centroids.n; ++c) {
        float* C = centroids.GetPoint(c);
        if (cluster_size[c] == 0) continue;
        for (size_t j = 0; j < centroids.d; ++j) C[j] /= cluster_size[c];
    }
}
// :88-101 (MIPS) and :117-128 (L2)
void SumPointsInClustersIP(const PointSet& P, PointSet& centroids, const std::vector<int>& closest_center, std::vector<size_t>& cluster_size,
                           const std::vector<float>& vector_sqrt_norms, std::vector<float>& norm_sums, size_t start, size_t end) {
    for (size_t i = start; i < end; ++i) {
        int c = closest_center[i];
        cluster_size[c]++;
        float* C = centroids.GetPoint(c);
        const float* Pi = P.GetPoint(i);
        norm_sums[c] += vector_sqrt_norms[i] * vector_sqrt_norms[i];
        float multiplier = 1.0f / vector_sqrt_norms[i];
        for (size_t j = 0; j < P.d; ++j) C[j] += Pi[j] * multiplier;
    }
}
void SumPointsInClustersL2(const PointSet& P, PointSet& centroids, const std::vector<int>& closest_center, std::vector<size_t>& cluster_size,
                           size_t start, size_t end) {
    for (size_t i = start; i < end; ++i) {
        int c = closest_center[i];
        cluster_size[c]++;
        float* C = centroids.GetPoint(c);
        const float* Pi = P.GetPoint(i);
        for (size_t j = 0; j < P.d; ++j) C[j] += Pi[j];
    }
}

// :131-149 AggregateClusters and :151-204 AggregateClustersParallel.
// For P.n <= block_size the reference is the sequential sum below (deterministic).  Above it the reference adds
// per-block partial sums with atomic float adds in scheduler order (non-deterministic); we add the block partials in
// block order, which is one of the reference's possible outcomes (SURVEY.md §7 hard part 7).
std::vector<size_t> AggregateClusters(const PointSet& P, PointSet& centroids, std::vector<int>& closest_center,
                                      const std::vector<float>& vector_sqrt_norms, int metric) {
    size_t block_size = std::max<size_t>(5000000, centroids.coordinates.size() * 200);
    std::vector<size_t> cluster_size(centroids.n, 0);
    std::vector<float> norm_sums(centroids.n, 0.f);
    centroids.coordinates.assign(centroids.coordinates.size(), 0.f);
    if (P.n <= block_size) {
        if (metric == 1) SumPointsInClustersIP(P, centroids, closest_center, cluster_size, vector_sqrt_norms, norm_sums, 0, closest_center.size());
        else SumPointsInClustersL2(P, centroids, closest_center, cluster_size, 0, closest_center.size());
    } else {
        for (size_t start = 0; start < P.n; start += block_size) {
            size_t end = std::min(P.n, start + block_size);
            PointSet b_centroids;
            b_centroids.n = centroids.n;
            b_centroids.d = centroids.d;
            b_centroids.coordinates.assign(centroids.n * centroids.d, 0.f);
            std::vector<size_t> b_cluster_size(centroids.n, 0);
            std::vector<float> b_norm_sums(centroids.n, 0.f);
            if (metric == 1) SumPointsInClustersIP(P, b_centroids, closest_center, b_cluster_size, vector_sqrt_norms, b_norm_sums, start, end);
            else SumPointsInClustersL2(P, b_centroids, closest_center, b_cluster_size, start, end);
            for (size_t i = 0; i < cluster_size.size(); ++i) {
                cluster_size[i] += b_cluster_size[i];
                norm_sums[i] = norm_sums[i] + b_norm_sums[i];
            }
            for (size_t i = 0; i < centroids.coordinates.size(); ++i) centroids.coordinates[i] = centroids.coordinates[i] + b_centroids.coordinates[i];
        }
    }
    if (metric == 1) NormalizeCentroidsIP(centroids, cluster_size, norm_sums);
    else NormalizeCentroidsL2(centroids, cluster_size);
    RemoveEmptyClusters(centroids, closest_center, cluster_size);
    return cluster_size;
}

// :207-226  (std::sample + std::mt19937 exactly as the reference; tail of `sample` stays 0 if n < num_samples)
PointSet RandomSample(const PointSet& points, size_t num_samples, int seed) {
    PointSet centroids;
    centroids.n = num_samples;
    centroids.d = points.d;
    std::vector<int> iota(points.n);
    std::iota(iota.begin(), iota.end(), 0);
    std::mt19937 prng(seed);
    std::vector<int> sample(num_samples);
    std::sample(iota.begin(), iota.end(), sample.begin(), num_samples, prng);
    for (int i : sample) {
        const float* p = points.GetPoint(i);
        for (size_t j = 0; j < points.d; ++j) centroids.coordinates.push_back(p[j]);
    }
    return centroids;
}

// :228-244
std::vector<int> KMeans(const PointSet& P, PointSet& centroids, int metric, int nthreads) {
    if (centroids.n < 1) throw std::runtime_error("KMeans #centroids < 1");
    std::vector<int> closest_center(P.n, -1);
    std::vector<float> vector_sqrt_norms;
    if (metric == 1) {
        vector_sqrt_norms.resize(P.n);
        for (size_t i = 0; i < P.n; ++i) vector_sqrt_norms[i] = std::sqrt(VecNorm(P.GetPoint(i), P.d));
    }
    static constexpr size_t NUM_ROUNDS = 20;
    for (size_t r = 0; r < NUM_ROUNDS; ++r) {
        NearestCenters(P, centroids, closest_center, metric, nthreads);
        AggregateClusters(P, centroids, closest_center, vector_sqrt_norms, metric);
    }
    return closest_center;
}

// ---------------------------------------------------------------------------------------------------------------
// src/kmeans_tree_router.{h,cpp}
struct KMeansTreeRouterOptions {   // kmeans_tree_router.h:6-11
    size_t num_centroids = 64;
    size_t min_cluster_size = 250;
    int64_t budget = 50000;
    int64_t search_budget = 50000;
};

struct TreeNode {   // kmeans_tree_router.h:54-57
    std::vector<TreeNode> children;
    PointSet centroids;
};

// kmeans_tree_router.cpp:80-91
PointSet ReorderCentroids(PointSet& centroids, std::vector<std::pair<size_t, size_t>>& permutation) {
    PointSet re;
    re.d = centroids.d;
    re.n = centroids.n;
    for (const auto& [_, centroid_id] : permutation) {
        float* c = centroids.GetPoint(centroid_id);
        for (size_t j = 0; j < centroids.d; ++j) re.coordinates.push_back(c[j]);
    }
    return re;
}

// kmeans_tree_router.cpp:35-78
void TrainRecursive(PointSet& points, KMeansTreeRouterOptions options, TreeNode& tree_node, int seed, int metric, int nthreads) {
    PointSet centroids = RandomSample(points, std::max(2, std::min<int>(options.num_centroids, options.budget)), seed);
    auto partition = KMeans(points, centroids, metric, nthreads);
    auto buckets = ConvertPartitionToClusters(partition);

    std::vector<std::pair<size_t, size_t>> bucket_size_and_ids(buckets.size());
    for (size_t i = 0; i < buckets.size(); ++i) bucket_size_and_ids[i] = std::make_pair(buckets[i].size(), i);
    size_t num_buckets_in_recursion =
        std::distance(bucket_size_and_ids.begin(), std::partition(bucket_size_and_ids.begin(), bucket_size_and_ids.end(),
                                                                  [&](const auto& pair) { return pair.first > options.min_cluster_size; }));

    tree_node.centroids = ReorderCentroids(centroids, bucket_size_and_ids);

    options.budget -= tree_node.centroids.n;
    if (options.budget <= 0) return;
    if (tree_node.centroids.n == 1) return;

    tree_node.children.resize(num_buckets_in_recursion);

    size_t total_size = 0;
    for (size_t i = 0; i < num_buckets_in_recursion; ++i) total_size += bucket_size_and_ids[i].first;

    for (size_t i = 0; i < num_buckets_in_recursion; ++i) {
        PointSet ps = ExtractPointsInBucket(buckets[bucket_size_and_ids[i].second], points.coordinates.data(), points.d);
        KMeansTreeRouterOptions recursive_options = options;
        recursive_options.budget = double(bucket_size_and_ids[i].first * options.budget) / double(total_size);
        TrainRecursive(ps, recursive_options, tree_node.children[i], seed + i, metric, nthreads);
    }
}

}  // namespace

// flat tree: nodes per shard in BFS order (ExtractPoints order, kmeans_tree_router.cpp:174-201)
struct orc_kmtr {
    uint32_t d = 0, S = 0;
    int metric = 0;
    std::vector<int32_t> node_shard;
    std::vector<uint32_t> node_cent_first;   // n_nodes + 1
    std::vector<uint32_t> node_child_first;
    std::vector<uint32_t> node_child_count;
    std::vector<uint32_t> shard_root;
    std::vector<float> centroids;
    uint32_t n_nodes() const { return (uint32_t)node_shard.size(); }
    uint32_t node_n(uint32_t u) const { return node_cent_first[u + 1] - node_cent_first[u]; }
    const float* centroid(uint32_t u, uint32_t i) const { return &centroids[(size_t)(node_cent_first[u] + i) * d]; }
};

namespace {

void Flatten(const std::vector<TreeNode>& roots, orc_kmtr& t) {
    t.node_cent_first.push_back(0);
    for (size_t b = 0; b < roots.size(); ++b) {
        // BFS exactly as ExtractPoints (kmeans_tree_router.cpp:180-198)
        std::queue<const TreeNode*> queue;
        queue.push(&roots[b]);
        t.shard_root.push_back(t.n_nodes());
        uint32_t next_free = t.n_nodes() + 1;   // node index the next enqueued child will get
        while (!queue.empty()) {
            const TreeNode* u = queue.front();
            queue.pop();
            for (const TreeNode& c : u->children) queue.push(&c);
            if (u->centroids.coordinates.size() != u->centroids.n * u->centroids.d) throw std::runtime_error("C");
            t.node_shard.push_back((int32_t)b);
            t.node_child_first.push_back(u->children.empty() ? 0 : next_free);
            t.node_child_count.push_back((uint32_t)u->children.size());
            next_free += (uint32_t)u->children.size();
            t.centroids.insert(t.centroids.end(), u->centroids.coordinates.begin(), u->centroids.coordinates.end());
            t.node_cent_first.push_back(t.node_cent_first.back() + (uint32_t)u->centroids.n);
        }
    }
}

// kmeans_tree_router.cpp:93-130 / :132-172 share this walk.  The PQ holds node indices instead of TreeNode*; the
// comparator looks at dist only (:98), exactly like the reference, so heap operations are identical.
struct PQEntry {
    float dist = 0.f;
    int shard_id = -1;
    uint32_t node = 0;
    bool operator>(const PQEntry& other) const { return dist > other.dist; }
};

template <typename OnDist>
void BestFirstWalk(const orc_kmtr& t, const float* Q, int budget, std::vector<float>& min_dist, OnDist on_dist) {
    std::priority_queue<PQEntry, std::vector<PQEntry>, std::greater<>> pq;
    min_dist.assign(t.S, std::numeric_limits<float>::max());
    for (int u = 0; u < int(t.S); ++u) {
        float dist = std::numeric_limits<float>::lowest();   // centroids_in_roots is never set (kmeans_tree_router.h:61)
        pq.push(PQEntry{dist, u, t.shard_root[u]});
    }
    while (!pq.empty() && budget > 0) {
        PQEntry top = pq.top();
        pq.pop();
        const uint32_t n = t.node_n(top.node);
        budget -= n;
        for (size_t i = 0; i < n; ++i) {
            float dist = Distance(t.centroid(top.node, i), Q, t.d, t.metric);
            min_dist[top.shard_id] = std::min(min_dist[top.shard_id], dist);
            on_dist(dist, top.shard_id);
            if (i < t.node_child_count[top.node]) {
                pq.push(PQEntry{dist, top.shard_id, t.node_child_first[top.node] + (uint32_t)i});
            }
        }
    }
}

void QueryOne(const orc_kmtr& t, const float* Q, int budget, int32_t* out_probes, float* out_min_dist) {
    std::vector<float> min_dist;
    BestFirstWalk(t, Q, budget, min_dist, [](float, int) {});
    std::vector<int> probes(t.S);
    std::iota(probes.begin(), probes.end(), 0);
    std::sort(probes.begin(), probes.end(), [&](int l, int r) { return min_dist[l] < min_dist[r]; });   // :125-127
    for (uint32_t b = 0; b < t.S; ++b) out_probes[b] = probes[b];
    if (out_min_dist) std::memcpy(out_min_dist, min_dist.data(), sizeof(float) * t.S);
}

}  // namespace

// =================================================================================================================
extern "C" {

float orc_distance(const float* p, const float* q, uint32_t d, int metric) { return Distance(p, q, d, metric); }

void orc_distance_many(const float* points, uint64_t n, uint32_t d, int metric, const float* q, float* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = Distance(points + i * d, q, d, metric);
}

uint32_t orc_topn(const float* dists, const uint32_t* ids, uint64_t n_items, uint32_t n, float* out_dist, uint32_t* out_ids) {
    TopN top(n);
    for (uint64_t i = 0; i < n_items; ++i) top.Add(std::make_pair(dists[i], ids[i]));
    auto res = top.Take();
    for (size_t i = 0; i < res.size(); ++i) {
        out_dist[i] = res[i].first;
        out_ids[i] = res[i].second;
    }
    return (uint32_t)res.size();
}

// src/inverted_index.h:12-41
void orc_ivf_build(const float* points, uint32_t d, const uint32_t* ids_concat, const uint64_t* cluster_offsets, uint32_t S,
                   float* out_clustered, int32_t* out_offsets, uint32_t* out_permutation) {
    out_offsets[0] = 0;
    for (uint32_t b = 0; b < S; ++b) out_offsets[b + 1] = (int32_t)(cluster_offsets[b + 1] - cluster_offsets[b]);
    for (uint32_t i = 2; i < S + 1; ++i) out_offsets[i] += out_offsets[i - 1];
    for (uint32_t b = 0; b < S; ++b) {
        size_t nb = cluster_offsets[b + 1] - cluster_offsets[b];
        for (size_t i_local = 0; i_local < nb; ++i_local) {
            const uint32_t point_id = ids_concat[cluster_offsets[b] + i_local];
            out_permutation[out_offsets[b] + i_local] = point_id;
            float* P = out_clustered + (size_t)(out_offsets[b] + i_local) * d;
            const float* O = points + (size_t)point_id * d;
            for (uint32_t j = 0; j < d; ++j) P[j] = O[j];
        }
    }
}

// src/inverted_index.h:62-75
uint32_t orc_ivf_query_bucket(const float* clustered, uint32_t d, int metric, const int32_t* offsets, const uint32_t* permutation,
                              const float* Q, uint32_t k, int32_t bucket, float* out_dist, uint32_t* out_ids, uint32_t* out_rows) {
    TopN top_k(k);
    for (int i = offsets[bucket]; i < offsets[bucket + 1]; ++i) {
        float new_dist = Distance(clustered + (size_t)i * d, Q, d, metric);
        auto x = std::make_pair(new_dist, (uint32_t)i);
        top_k.Add(x);
    }
    auto result = top_k.Take();
    for (size_t j = 0; j < result.size(); ++j) {
        out_dist[j] = result[j].first;
        if (out_rows) out_rows[j] = result[j].second;
        out_ids[j] = permutation[result[j].second];
    }
    return (uint32_t)result.size();
}

// src/inverted_index.h:43-60
uint32_t orc_ivf_query(const float* clustered, uint32_t d, int metric, const int32_t* offsets, const uint32_t* permutation,
                       const float* Q, uint32_t k, const int32_t* probes, uint32_t nprobe, float* out_dist, uint32_t* out_ids) {
    TopN top_k(k);
    for (size_t j = 0; j < nprobe; ++j) {
        int b = probes[j];
        for (int i = offsets[b]; i < offsets[b + 1]; ++i) {
            float new_dist = Distance(clustered + (size_t)i * d, Q, d, metric);
            auto x = std::make_pair(new_dist, (uint32_t)i);
            top_k.Add(x);
        }
    }
    auto result = top_k.Take();
    for (size_t j = 0; j < result.size(); ++j) {
        out_dist[j] = result[j].first;
        out_ids[j] = permutation[result[j].second];
    }
    return (uint32_t)result.size();
}

void orc_ivf_query_bucket_batch(const float* clustered, uint32_t d, int metric, const int32_t* offsets, const uint32_t* permutation,
                                const float* queries, uint32_t nq, uint32_t k, const int32_t* probes, uint32_t probe_stride,
                                uint32_t nprobe, float* out_dist, uint32_t* out_ids, uint32_t* out_count, int nthreads) {
    ParallelFor(0, (size_t)nq * nprobe, nthreads, [&](size_t t) {
        size_t q = t / nprobe, j = t % nprobe;
        float* od = out_dist + t * k;
        uint32_t* oi = out_ids + t * k;
        for (uint32_t x = 0; x < k; ++x) {
            od[x] = std::numeric_limits<float>::infinity();
            oi[x] = 0xFFFFFFFFu;
        }
        int32_t b = probes[q * probe_stride + j];
        uint32_t c = 0;
        if (b >= 0) c = orc_ivf_query_bucket(clustered, d, metric, offsets, permutation, queries + q * d, k, b, od, oi, nullptr);
        if (out_count) out_count[t] = c;
    });
}

// src/recall.h:25-39
void orc_ground_truth(const float* points, uint64_t n, uint32_t d, int metric, const float* queries, uint32_t nq, uint32_t k,
                      float* out_dist, uint32_t* out_ids, int nthreads) {
    ParallelFor(0, nq, nthreads, [&](size_t i) {
        TopN top_k(k);
        const float* Q = queries + i * d;
        for (uint32_t j = 0; j < n; ++j) {
            const float* P = points + (size_t)j * d;
            float dist = Distance(P, Q, d, metric);
            top_k.Add(std::make_pair(dist, j));
        }
        auto res = top_k.Take();
        std::sort(res.begin(), res.end());
        for (size_t x = 0; x < res.size(); ++x) {
            out_dist[i * k + x] = res[x].first;
            out_ids[i * k + x] = res[x].second;
        }
    });
}

// src/recall.h:10-23
void orc_distance_to_kth(const float* points, uint64_t n, uint32_t d, int metric, const float* queries, uint32_t nq, uint32_t k,
                         float* out_kth, int nthreads) {
    ParallelFor(0, nq, nthreads, [&](size_t i) {
        TopN top_k(k);
        const float* Q = queries + i * d;
        for (uint32_t j = 0; j < n; ++j) {
            float dist = Distance(points + (size_t)j * d, Q, d, metric);
            top_k.Add(std::make_pair(dist, j));
        }
        out_kth[i] = top_k.Top().first;
    });
}

// src/recall.h:74-137 (diagnostic counters other than wrong_sorts omitted)
uint64_t orc_convert_ground_truth(float* gt_dist, uint32_t* gt_ids, uint32_t nq, uint32_t k_stored, uint32_t k, const float* points,
                                  uint32_t d, int metric, const float* queries, float* out_kth) {
    uint64_t wrong_sorts = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        NNVec neighs(k_stored);
        for (uint32_t j = 0; j < k_stored; ++j) neighs[j] = {gt_dist[(size_t)q * k_stored + j], gt_ids[(size_t)q * k_stored + j]};
        auto comp = [](const std::pair<float, uint32_t>& l, const std::pair<float, uint32_t>& r) { return l.first < r.first; };
        const float* Q = queries + (size_t)q * d;
        for (uint32_t j = 0; j < k; ++j) {
            uint32_t point_id = neighs[j].second;
            float true_dist = Distance(points + (size_t)point_id * d, Q, d, metric);
            neighs[j].first = true_dist;
        }
        bool is_sorted = std::is_sorted(neighs.begin(), neighs.begin() + k, comp);
        if (!is_sorted) {
            std::sort(neighs.begin(), neighs.begin() + k, comp);
            wrong_sorts++;
        }
        out_kth[q] = neighs[k - 1].first;
        for (uint32_t j = 0; j < k_stored; ++j) {
            gt_dist[(size_t)q * k_stored + j] = neighs[j].first;
            gt_ids[(size_t)q * k_stored + j] = neighs[j].second;
        }
    }
    return wrong_sorts;
}

// src/recall.h:139-150
double orc_recall(const float* dists, const uint64_t* offsets, uint32_t nq, const float* kth, uint32_t k) {
    size_t hits = 0;
    for (size_t i = 0; i < nq; ++i) {
        for (uint64_t x = offsets[i]; x < offsets[i + 1]; ++x) {
            if (dists[x] <= kth[i]) hits++;
        }
    }
    return static_cast<double>(hits) / static_cast<double>((size_t)nq * k);
}

// small_scale_queries.cpp:15-23
void orc_dedup_neighbors(const float* dists, const uint32_t* ids, const uint64_t* offsets, uint32_t nq, uint32_t k, float* out_dist,
                         uint32_t* out_ids, uint32_t* out_count) {
    for (uint32_t q = 0; q < nq; ++q) {
        NNVec n;
        for (uint64_t x = offsets[q]; x < offsets[q + 1]; ++x) n.emplace_back(dists[x], ids[x]);
        std::sort(n.begin(), n.end(), [](const auto& l, const auto& r) { return l.second < r.second; });
        n.erase(std::unique(n.begin(), n.end()), n.end());
        std::sort(n.begin(), n.end());
        n.resize(std::min<size_t>(n.size(), k));
        for (uint32_t x = 0; x < k; ++x) {
            out_dist[(size_t)q * k + x] = x < n.size() ? n[x].first : std::numeric_limits<float>::infinity();
            out_ids[(size_t)q * k + x] = x < n.size() ? n[x].second : 0xFFFFFFFFu;
        }
        if (out_count) out_count[q] = (uint32_t)n.size();
    }
}

uint32_t orc_kmeans(const float* points, uint64_t n, uint32_t d, int metric, uint32_t k, int seed, int32_t* out_partition,
                    float* out_centroids, int nthreads) {
    PointSet P;
    P.n = n;
    P.d = d;
    P.coordinates.assign(points, points + n * d);
    PointSet centroids = RandomSample(P, k, seed);
    auto part = KMeans(P, centroids, metric, nthreads);
    for (uint64_t i = 0; i < n; ++i) out_partition[i] = part[i];
    if (out_centroids) std::memcpy(out_centroids, centroids.coordinates.data(), sizeof(float) * centroids.n * d);
    return (uint32_t)centroids.n;
}

// kmeans_tree_router.cpp:9-33
orc_kmtr* orc_kmtr_train(const float* points, uint64_t n, uint32_t d, int metric, const uint32_t* ids_concat,
                         const uint64_t* cluster_offsets, uint32_t S, uint64_t num_centroids, uint64_t min_cluster_size, int64_t budget,
                         int nthreads) {
    KMeansTreeRouterOptions options;
    options.num_centroids = num_centroids;
    options.min_cluster_size = min_cluster_size;
    options.budget = budget;
    std::vector<TreeNode> roots(S);
    // shards are independent (seed 555*b); run them in parallel when asked, k-means inside then runs serially
    int outer = std::min<int>(nthreads, S);
    int inner = std::max(1, nthreads / std::max(1, outer));
    ParallelFor(0, S, outer, [&](size_t b) {
        std::vector<uint32_t> cluster(ids_concat + cluster_offsets[b], ids_concat + cluster_offsets[b + 1]);
        PointSet ps = ExtractPointsInBucket(cluster, points, d);
        KMeansTreeRouterOptions recursive_options = options;
        recursive_options.budget = double(cluster.size() * options.budget) / double(n);
        TrainRecursive(ps, recursive_options, roots[b], 555 * (int)b, metric, inner);
    });
    auto* t = new orc_kmtr();
    t->d = d;
    t->S = S;
    t->metric = metric;
    Flatten(roots, *t);
    return t;
}

orc_kmtr* orc_kmtr_from_flat(uint32_t d, int metric, uint32_t S, uint32_t n_nodes, const int32_t* node_shard, const uint32_t* node_cent_first,
                             const uint32_t* node_child_first, const uint32_t* node_child_count, const uint32_t* shard_root,
                             const float* centroids) {
    auto* t = new orc_kmtr();
    t->d = d;
    t->S = S;
    t->metric = metric;
    t->node_shard.assign(node_shard, node_shard + n_nodes);
    t->node_cent_first.assign(node_cent_first, node_cent_first + n_nodes + 1);
    t->node_child_first.assign(node_child_first, node_child_first + n_nodes);
    t->node_child_count.assign(node_child_count, node_child_count + n_nodes);
    t->shard_root.assign(shard_root, shard_root + S);
    t->centroids.assign(centroids, centroids + (size_t)node_cent_first[n_nodes] * d);
    return t;
}

void orc_kmtr_sizes(const orc_kmtr* t, uint32_t* d, uint32_t* S, uint32_t* n_nodes, uint64_t* n_centroids) {
    if (d) *d = t->d;
    if (S) *S = t->S;
    if (n_nodes) *n_nodes = t->n_nodes();
    if (n_centroids) *n_centroids = t->node_cent_first.back();
}

void orc_kmtr_export(const orc_kmtr* t, int32_t* node_shard, uint32_t* node_cent_first, uint32_t* node_child_first,
                     uint32_t* node_child_count, uint32_t* shard_root, float* centroids) {
    std::copy(t->node_shard.begin(), t->node_shard.end(), node_shard);
    std::copy(t->node_cent_first.begin(), t->node_cent_first.end(), node_cent_first);
    std::copy(t->node_child_first.begin(), t->node_child_first.end(), node_child_first);
    std::copy(t->node_child_count.begin(), t->node_child_count.end(), node_child_count);
    std::copy(t->shard_root.begin(), t->shard_root.end(), shard_root);
    std::copy(t->centroids.begin(), t->centroids.end(), centroids);
}

void orc_kmtr_extract_partition(const orc_kmtr* t, int32_t* out) {
    for (uint32_t u = 0; u < t->n_nodes(); ++u)
        for (uint32_t c = t->node_cent_first[u]; c < t->node_cent_first[u + 1]; ++c) out[c] = t->node_shard[u];
}

void orc_kmtr_query(const orc_kmtr* t, const float* Q, int budget, int32_t* out_probes, float* out_min_dist) {
    QueryOne(*t, Q, budget, out_probes, out_min_dist);
}

void orc_kmtr_query_batch(const orc_kmtr* t, const float* queries, uint32_t nq, int budget, int32_t* out_probes, float* out_min_dist,
                          int nthreads) {
    ParallelFor(0, nq, nthreads, [&](size_t q) {
        QueryOne(*t, queries + q * t->d, budget, out_probes + q * t->S, out_min_dist ? out_min_dist + q * t->S : nullptr);
    });
}

// kmeans_tree_router.cpp:132-172
uint32_t orc_kmtr_frequency_query(const orc_kmtr* t, const float* Q, int budget, int num_voting, float* out_min_dist, float* out_nn_dist,
                                  uint32_t* out_nn_shard) {
    TopN top_neighbors(num_voting);
    std::vector<float> min_dist;
    BestFirstWalk(*t, Q, budget, min_dist, [&](float dist, int shard) { top_neighbors.Add(std::make_pair(dist, (uint32_t)shard)); });
    auto near = top_neighbors.Take();
    std::sort(near.begin(), near.end());   // FrequencyQueryData::Init, kmeans_tree_router.h:22
    std::memcpy(out_min_dist, min_dist.data(), sizeof(float) * t->S);
    for (size_t i = 0; i < near.size(); ++i) {
        out_nn_dist[i] = near[i].first;
        out_nn_shard[i] = near[i].second;
    }
    return (uint32_t)near.size();
}

// kmeans_tree_router.h:23-44
void orc_frequency_probes(const float* min_dist, uint32_t S, const uint32_t* nn_shard, uint32_t nn_count, int num_voting,
                          int32_t* out_probes) {
    std::vector<int> frequency(S, 0);
    for (int i = 0; i < std::min<int>(num_voting, nn_count); ++i) frequency[nn_shard[i]]++;
    std::vector<int> probes;
    size_t highest_freq = 0;
    for (size_t b = 1; b < frequency.size(); ++b) {
        if (frequency[b] > frequency[highest_freq]) highest_freq = b;
    }
    probes.push_back(highest_freq);
    for (size_t b = 0; b < frequency.size(); ++b) {
        if (b != highest_freq) probes.push_back(b);
    }
    std::sort(probes.begin() + 1, probes.end(), [&](int l, int r) { return min_dist[l] < min_dist[r]; });
    for (uint32_t b = 0; b < S; ++b) out_probes[b] = probes[b];
}

void orc_kmtr_free(orc_kmtr* t) { delete t; }

uint32_t orc_partition_to_clusters(const int32_t* partition, uint64_t n, uint32_t* out_ids_concat, uint64_t* out_offsets) {
    std::vector<int> p(partition, partition + n);
    auto clusters = ConvertPartitionToClusters(p);
    uint64_t o = 0;
    out_offsets[0] = 0;
    for (size_t b = 0; b < clusters.size(); ++b) {
        for (uint32_t id : clusters[b]) out_ids_concat[o++] = id;
        out_offsets[b + 1] = o;
    }
    return (uint32_t)clusters.size();
}

void orc_synth_rows(uint64_t seed, uint64_t center_seed, uint32_t n_centers, float sigma, uint64_t row0, uint64_t nrows, uint32_t d,
                    float* out, int nthreads) {
    ParallelFor(0, nrows, nthreads, [&](size_t r) {
        for (uint32_t j = 0; j < d; ++j) out[r * d + j] = gpann_synth_coord(seed, center_seed, n_centers, sigma, row0 + r, j);
    });
}

}  // extern "C"
