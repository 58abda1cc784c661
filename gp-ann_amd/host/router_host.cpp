// router_host.cpp — host mirror of k-means (reference src/kmeans.cpp:12-50,72-149,207-244) and of KMeansTreeRouter
// (src/kmeans_tree_router.{h,cpp}).  Train() runs on the GPU (gpann_router_train) unless gpann_host::SetDeviceTraining(false);
// TrainOnHost() below is the host-thread trainer that builds the same tree.  Query / QueryBatch / FrequencyQuery run on the GPU
// through libgpann_hip.so.
//
// The trainer is written against row-index views instead of copying a PointSet per tree node, and emits the flat BFS
// layout directly, but every arithmetic step, every std:: call whose result depends on libstdc++ (std::sample with
// std::mt19937, std::partition) and every iteration order is the reference's, so trees are bit-identical to the
// reference's for shards up to 5M points (above that the reference itself is non-deterministic: kmeans.cpp:153-192).
#include <cmath>
#include <cstring>
#include <iostream>
#include <mutex>
#include <limits>
#include <numeric>
#include <random>

#include "gpann_host.h"
#include "host_util.h"

namespace {

using gpann_host::ParallelFor;

// A set of rows of a row-major matrix: row i of the view is base + rows[i] * d.
struct RowView {
    const float* base;
    size_t d;
    const std::vector<uint32_t>* rows;   // nullptr = identity over n rows
    size_t n;
    const float* row(size_t i) const { return base + (rows ? (size_t)(*rows)[i] : i) * d; }
};

inline float Dist(const float* p, const float* q, unsigned d, bool mips) {
    return mips ? mips_distance(const_cast<float*>(p), const_cast<float*>(q), d) : sqr_l2_dist(p, q, d);
}

// kmeans.cpp:207-226 — same std::sample / std::mt19937 call; when the view holds fewer rows than num_samples the tail of
// the sample stays 0, i.e. row 0 is repeated, exactly like the reference
std::vector<float> SampleRows(const RowView& v, size_t num_samples, int seed) {
    std::vector<int> all(v.n);
    std::iota(all.begin(), all.end(), 0);
    std::mt19937 prng(seed);
    std::vector<int> picked(num_samples);
    std::sample(all.begin(), all.end(), picked.begin(), num_samples, prng);
    std::vector<float> cent(num_samples * v.d);
    for (size_t s = 0; s < num_samples; ++s) std::memcpy(&cent[s * v.d], v.row(picked[s]), v.d * sizeof(float));
    return cent;
}

// kmeans.cpp:228-244 with :25-27 (assignment), :131-149 / :151-204 (update) and :29-50 (empty-cluster removal).
// `cent` holds c*d floats on entry and the surviving centroids on exit; returns the assignment computed in the last round.
std::vector<int> Lloyd(const RowView& v, std::vector<float>& cent, bool mips, int threads) {
    size_t c = cent.size() / v.d;
    if (c < 1) throw std::runtime_error("KMeans #centroids < 1");
    const size_t d = v.d, n = v.n;
    std::vector<int> owner(n, -1);
    std::vector<float> sqrt_norm;
    if (mips) {
        sqrt_norm.resize(n);
        ParallelFor(0, n, threads, [&](size_t i) { sqrt_norm[i] = std::sqrt(vec_norm(const_cast<float*>(v.row(i)), d)); });
    }
    for (int round = 0; round < 20; ++round) {
        // Top1Neighbor (kmeans.cpp:12-23): first strict minimum, -1 if nothing beats FLT_MAX
        ParallelFor(0, n, threads, [&](size_t i) {
            const float* P = v.row(i);
            int best = -1;
            float best_dist = std::numeric_limits<float>::max();
            for (size_t j = 0; j < c; ++j) {
                const float nd = Dist(&cent[j * d], P, d, mips);
                if (nd < best_dist) {
                    best_dist = nd;
                    best = (int)j;
                }
            }
            owner[i] = best;
        });
        // sums in point order; beyond the reference's block size the block partials are added in block order (one of the
        // orders the reference's atomics can produce)
        std::vector<size_t> population(c, 0);
        std::vector<float> norm_sum(c, 0.f);
        std::fill(cent.begin(), cent.end(), 0.f);
        const size_t block = std::max<size_t>(5000000, cent.size() * 200);
        auto accumulate = [&](std::vector<float>& sums, std::vector<size_t>& pop, std::vector<float>& nsum, size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const int o = owner[i];
                pop[o]++;
                float* C = &sums[(size_t)o * d];
                const float* P = v.row(i);
                if (mips) {
                    nsum[o] += sqrt_norm[i] * sqrt_norm[i];
                    const float scale = 1.0f / sqrt_norm[i];
                    for (size_t j = 0; j < d; ++j) C[j] += P[j] * scale;
                } else {
                    for (size_t j = 0; j < d; ++j) C[j] += P[j];
                }
            }
        };
        if (n <= block) {
            accumulate(cent, population, norm_sum, 0, n);
        } else {
            for (size_t lo = 0; lo < n; lo += block) {
                std::vector<float> part(cent.size(), 0.f), part_norm(c, 0.f);
                std::vector<size_t> part_pop(c, 0);
                accumulate(part, part_pop, part_norm, lo, std::min(n, lo + block));
                for (size_t j = 0; j < c; ++j) {
                    population[j] += part_pop[j];
                    norm_sum[j] = norm_sum[j] + part_norm[j];
                }
                for (size_t j = 0; j < cent.size(); ++j) cent[j] = cent[j] + part[j];
            }
        }
        for (size_t j = 0; j < c; ++j) {
            if (population[j] == 0) continue;
            float* C = &cent[j * d];
            if (mips) {   // kmeans.cpp:72-86
                const float want = norm_sum[j] / population[j];
                const float have = vec_norm(C, d);
                const float scale = std::sqrt(want / have);
                for (size_t x = 0; x < d; ++x) C[x] *= scale;
            } else {      // kmeans.cpp:104-115
                for (size_t x = 0; x < d; ++x) C[x] /= population[j];
            }
        }
        // RemoveEmptyClusters
        bool any_empty = false;
        for (size_t j = 0; j < c; ++j) any_empty |= population[j] == 0;
        if (any_empty) {
            std::vector<int> new_id(c, -1);
            size_t kept = 0;
            for (size_t j = 0; j < c; ++j) {
                if (population[j] == 0) continue;
                if (kept != j) std::memmove(&cent[kept * d], &cent[j * d], d * sizeof(float));
                new_id[j] = (int)kept++;
            }
            c = kept;
            cent.resize(c * d);
            for (int& o : owner) o = new_id[o];
        }
    }
    return owner;
}

struct BuildNode {
    int shard;
    std::vector<float> centroids;     // n x d, recursed buckets first
    std::vector<uint32_t> children;   // BuildNode indices, child i <-> centroid i
};

struct Builder {
    const float* base;   // the shard's rows live in the caller's PointSet; views index into it directly
    size_t d;
    bool mips;
    int threads;
    size_t num_centroids, min_cluster_size;
    std::vector<BuildNode> nodes;

    // kmeans_tree_router.cpp:35-78
    uint32_t Grow(const std::vector<uint32_t>& rows, int64_t budget, int seed, int shard) {
        RowView view{base, d, &rows, rows.size()};
        std::vector<float> cent = SampleRows(view, std::max(2, std::min<int>(num_centroids, budget)), seed);
        std::vector<int> owner = Lloyd(view, cent, mips, threads);
        const size_t c = cent.size() / d;
        // ConvertPartitionToClusters: positions ascending inside every bucket
        int parts = 0;
        for (int o : owner) parts = std::max(parts, o + 1);
        std::vector<std::vector<uint32_t>> buckets(parts);
        for (size_t i = 0; i < owner.size(); ++i) buckets[owner[i]].push_back((uint32_t)i);

        std::vector<std::pair<size_t, size_t>> size_and_id(buckets.size());
        for (size_t b = 0; b < buckets.size(); ++b) size_and_id[b] = {buckets[b].size(), b};
        const size_t recursed = std::distance(
            size_and_id.begin(), std::partition(size_and_id.begin(), size_and_id.end(), [&](const auto& e) { return e.first > min_cluster_size; }));

        const uint32_t me = (uint32_t)nodes.size();
        nodes.emplace_back();
        nodes[me].shard = shard;
        // ReorderCentroids (:80-91): `re.n = centroids.n`, one row per entry of the permutation
        nodes[me].centroids.reserve(size_and_id.size() * d);
        for (const auto& e : size_and_id) nodes[me].centroids.insert(nodes[me].centroids.end(), &cent[e.second * d], &cent[e.second * d] + d);
        const size_t n_here = c;   // tree_node.centroids.n

        budget -= (int64_t)n_here;
        if (budget <= 0) return me;
        if (n_here == 1) return me;

        size_t total = 0;
        for (size_t i = 0; i < recursed; ++i) total += size_and_id[i].first;
        std::vector<uint32_t> kids(recursed);
        for (size_t i = 0; i < recursed; ++i) {
            const auto& bucket = buckets[size_and_id[i].second];
            std::vector<uint32_t> sub(bucket.size());
            for (size_t x = 0; x < bucket.size(); ++x) sub[x] = rows[bucket[x]];   // = ExtractPointsInBucket order
            const int64_t child_budget = double(size_and_id[i].first * budget) / double(total);
            kids[i] = Grow(sub, child_budget, seed + (int)i, shard);
        }
        nodes[me].children = std::move(kids);
        return me;
    }
};

}   // namespace

// ================================================================================================== kmeans.h
PointSet RandomSample(PointSet& points, size_t num_samples, int seed) {
    RowView v{points.coordinates.data(), points.d, nullptr, points.n};
    PointSet out;
    out.n = num_samples;
    out.d = points.d;
    out.coordinates = SampleRows(v, num_samples, seed);
    return out;
}

std::vector<int> KMeans(PointSet& points, PointSet& centroids) {
    RowView v{points.coordinates.data(), points.d, nullptr, points.n};
    std::vector<int> owner = Lloyd(v, centroids.coordinates, gpann_host::MetricIsMIPS(), gpann_host::NumThreads());
    centroids.n = centroids.coordinates.size() / centroids.d;
    return owner;
}

// ================================================================================================== KMeansTreeRouter
KMeansTreeRouter::~KMeansTreeRouter() {
    if (device_router_) gpann_router_destroy(device_router_);
}

KMeansTreeRouter::KMeansTreeRouter(KMeansTreeRouter&& other) noexcept : tree_(std::move(other.tree_)), device_router_(other.device_router_) {
    other.device_router_ = nullptr;
}

KMeansTreeRouter& KMeansTreeRouter::operator=(KMeansTreeRouter&& other) noexcept {
    if (this != &other) {
        if (device_router_) gpann_router_destroy(device_router_);
        tree_ = std::move(other.tree_);
        device_router_ = other.device_router_;
        other.device_router_ = nullptr;
    }
    return *this;
}

void KMeansTreeRouter::LoadFlat(Flat tree) {
    if (device_router_) {
        gpann_router_destroy(device_router_);
        device_router_ = nullptr;
    }
    tree_ = std::move(tree);
}

// kmeans_tree_router.cpp:9-33
void KMeansTreeRouter::Train(PointSet& points, const Clusters& clusters, KMeansTreeRouterOptions options) {
    if (!gpann_host::DeviceTraining()) {
        TrainOnHost(points, clusters, options);
        return;
    }
    // a temporary shard store (the gather InvertedIndex would do): shard b = clusters[b] in order, which is all the trainer reads
    std::vector<uint64_t> offsets(clusters.size() + 1, 0);
    for (size_t b = 0; b < clusters.size(); ++b) offsets[b + 1] = offsets[b] + clusters[b].size();
    std::vector<uint32_t> ids;
    ids.reserve(offsets.back());
    for (const auto& c : clusters) ids.insert(ids.end(), c.begin(), c.end());
    gpann_shards* store = nullptr;
    gpann_host::Check(gpann_shards_create(points.coordinates.data(), points.n, (uint32_t)points.d,
                                          gpann_host::MetricIsMIPS() ? GPANN_METRIC_MIPS : GPANN_METRIC_L2, ids.data(), offsets.data(),
                                          (uint32_t)clusters.size(), gpann_host::Device(), GPANN_STORE_PLAIN, &store),
                      "gpann_shards_create");
    try {
        TrainOnDevice(store, points.n, options);
    } catch (...) {
        gpann_shards_destroy(store);
        throw;
    }
    gpann_shards_destroy(store);
}

void KMeansTreeRouter::Train(InvertedIndex& index, size_t num_points, KMeansTreeRouterOptions options) {
    TrainOnDevice(index.device_handle(), num_points, options);
}

void KMeansTreeRouter::TrainOnDevice(gpann_shards* store, size_t num_points, KMeansTreeRouterOptions options) {
    std::cout << "Train. budget = " << options.budget << " (GPU)" << std::endl;
    gpann_router* trained = nullptr;
    gpann_host::Check(gpann_router_train(store, num_points, (uint32_t)options.num_centroids, (uint32_t)options.min_cluster_size, options.budget, &trained),
                      "gpann_router_train");
    Flat flat;
    uint32_t n_nodes = 0;
    uint64_t n_cent = 0;
    int metric = 0;
    gpann_status st = gpann_router_info(trained, &flat.d, &metric, &flat.S, &n_nodes, &n_cent);
    if (st == GPANN_OK) {
        flat.mips = metric == GPANN_METRIC_MIPS;
        flat.node_shard.resize(n_nodes);
        flat.node_cent_first.resize(n_nodes + 1);
        flat.node_child_first.resize(n_nodes);
        flat.node_child_count.resize(n_nodes);
        flat.shard_root.resize(flat.S);
        flat.centroids.resize(n_cent * flat.d);
        st = gpann_router_export(trained, flat.node_shard.data(), flat.node_cent_first.data(), flat.node_child_first.data(),
                                 flat.node_child_count.data(), flat.shard_root.data(), flat.centroids.data());
    }
    if (st != GPANN_OK) {
        gpann_router_destroy(trained);
        gpann_host::Check(st, "gpann_router_export");
    }
    LoadFlat(std::move(flat));
    device_router_ = trained;   // already resident: Query needs no upload
}

void KMeansTreeRouter::TrainOnHost(PointSet& points, const Clusters& clusters, KMeansTreeRouterOptions options) {
    const int S = (int)clusters.size();
    const bool mips = gpann_host::MetricIsMIPS();
    std::cout << "Train. num-shards = " << S << " dim = " << points.d << " budget = " << options.budget << std::endl;

    // shards are independent (seed 555 * b); spread the host threads over shards first, over points inside a shard second
    const int threads = gpann_host::NumThreads();
    const int outer = std::max(1, std::min(threads, S));
    const int inner = std::max(1, threads / outer);
    std::vector<Builder> built(S);
    ParallelFor(0, (size_t)S, outer, [&](size_t b) {
        Builder& B = built[b];
        B.base = points.coordinates.data();
        B.d = points.d;
        B.mips = mips;
        B.threads = inner;
        B.num_centroids = options.num_centroids;
        B.min_cluster_size = options.min_cluster_size;
        const int64_t shard_budget = double(clusters[b].size() * options.budget) / double(points.n);
        B.Grow(clusters[b], shard_budget, 555 * (int)b, (int)b);
    });

    // flatten: per shard, breadth first (the order ExtractPoints walks, :180-198); the children of a node become consecutive
    Flat flat;
    flat.d = (uint32_t)points.d;
    flat.S = (uint32_t)S;
    flat.mips = mips;
    flat.node_cent_first.push_back(0);
    for (int b = 0; b < S; ++b) {
        const auto& nodes = built[b].nodes;
        std::vector<uint32_t> order{0};   // Grow() creates the root first
        flat.shard_root.push_back((uint32_t)flat.node_shard.size());
        const uint32_t first_id = (uint32_t)flat.node_shard.size();
        for (size_t head = 0; head < order.size(); ++head) {
            const BuildNode& u = nodes[order[head]];
            flat.node_shard.push_back(b);
            flat.node_child_count.push_back((uint32_t)u.children.size());
            flat.node_child_first.push_back(u.children.empty() ? 0u : first_id + (uint32_t)order.size());
            for (uint32_t ch : u.children) order.push_back(ch);
            flat.centroids.insert(flat.centroids.end(), u.centroids.begin(), u.centroids.end());
            flat.node_cent_first.push_back(flat.node_cent_first.back() + (uint32_t)(u.centroids.size() / points.d));
        }
    }
    LoadFlat(std::move(flat));
}

void KMeansTreeRouter::EnsureOnDevice() {
    static std::mutex upload_mutex;   // (the reference's callers run Query from many threads: the first of them uploads the tree)
    std::lock_guard<std::mutex> lk(upload_mutex);
    if (device_router_) return;
    if (tree_.S == 0) throw std::runtime_error("KMeansTreeRouter: Query before Train");
    gpann_host::Check(gpann_router_create(tree_.d, tree_.mips ? GPANN_METRIC_MIPS : GPANN_METRIC_L2, tree_.S, (uint32_t)tree_.node_shard.size(),
                                          tree_.node_shard.data(), tree_.node_cent_first.data(), tree_.node_child_first.data(),
                                          tree_.node_child_count.data(), tree_.shard_root.data(), tree_.centroids.data(), gpann_host::Device(),
                                          &device_router_),
                      "gpann_router_create");
}

std::vector<std::vector<int>> KMeansTreeRouter::QueryBatch(PointSet& queries, int budget) {
    EnsureOnDevice();
    const uint32_t nq = (uint32_t)queries.n, S = tree_.S;
    std::vector<int32_t> probes((size_t)nq * S);
    gpann_host::Check(gpann_router_query_batch(device_router_, queries.coordinates.data(), nq, budget, probes.data(), nullptr, 0, nullptr, nullptr),
                      "gpann_router_query_batch");
    std::vector<std::vector<int>> out(nq);
    for (uint32_t q = 0; q < nq; ++q) out[q].assign(probes.begin() + (size_t)q * S, probes.begin() + (size_t)(q + 1) * S);
    return out;
}

void KMeansTreeRouter::MinDistBatch(const float* Q, size_t nq, int budget, float* out) {
    EnsureOnDevice();
    gpann_host::Check(gpann_router_query_batch(device_router_, Q, (uint32_t)nq, budget, nullptr, out, 0, nullptr, nullptr), "gpann_router_query_batch");
}

// kmeans_tree_router.cpp:93-130.  The reference's callers run this from up to 32 threads, one query per call (routes.cpp:194-200): concurrent
// calls are coalesced into device batches (host_util.h Coalescer) — per (router, budget) one gpann_router_query_batch over everything that
// queued up while the previous batch ran.  result = what a batch of one returns (the batch entry points are row-independent).
namespace {
struct RouterQueryReq {
    const float* Q = nullptr;
    int budget = 0;
    gpann_router* handle = nullptr;
    uint32_t d = 0, S = 0;
    std::vector<int32_t> probes;
    bool done = false;
    std::exception_ptr error;
};
gpann_host::Coalescer<RouterQueryReq>& RouterQueue() {
    static gpann_host::Coalescer<RouterQueryReq>* q = new gpann_host::Coalescer<RouterQueryReq>();   // (never destroyed: callers may outlive static destruction order)
    return *q;
}
// one device batch for requests that share (handle, budget); a failed batch of several is retried request by request, so that an error reaches only
// the caller whose request causes it (index_host.cpp RunScanGroup)
void RunRouterGroup(std::vector<RouterQueryReq*>& group) {
    const uint32_t n = (uint32_t)group.size(), d = group[0]->d, S = group[0]->S;
    std::vector<float> q((size_t)n * d);
    for (uint32_t x = 0; x < n; ++x) std::memcpy(&q[(size_t)x * d], group[x]->Q, (size_t)d * sizeof(float));
    std::vector<int32_t> probes((size_t)n * S);
    try {
        gpann_host::Check(gpann_router_query_batch(group[0]->handle, q.data(), n, group[0]->budget, probes.data(), nullptr, 0, nullptr, nullptr), "gpann_router_query_batch");
        for (uint32_t x = 0; x < n; ++x) group[x]->probes.assign(probes.begin() + (size_t)x * S, probes.begin() + (size_t)(x + 1) * S);
    } catch (...) {
        if (n == 1) {
            group[0]->error = std::current_exception();
        } else {
            for (RouterQueryReq* r : group) {
                std::vector<RouterQueryReq*> one{r};
                RunRouterGroup(one);
            }
        }
    }
}
void RunRouterBatch(std::vector<RouterQueryReq*>& batch) {
    std::vector<char> taken(batch.size(), 0);
    for (size_t i = 0; i < batch.size(); ++i) {
        if (taken[i]) continue;
        std::vector<RouterQueryReq*> group;
        for (size_t j = i; j < batch.size(); ++j)
            if (!taken[j] && batch[j]->handle == batch[i]->handle && batch[j]->budget == batch[i]->budget) {
                taken[j] = 1;
                group.push_back(batch[j]);
            }
        RunRouterGroup(group);
    }
}
}   // namespace

extern "C" void gpann_host_router_queue_counters(uint64_t* batches, uint64_t* requests) { RouterQueue().Counters(batches, requests); }

std::vector<int> KMeansTreeRouter::Query(float* Q, int budget) {
    EnsureOnDevice();
    if (!gpann_host::Coalescing()) {   // one query = a batch of one behind the handle's lock
        std::vector<int32_t> probes(tree_.S);
        gpann_host::Check(gpann_router_query_batch(device_router_, Q, 1, budget, probes.data(), nullptr, 0, nullptr, nullptr), "gpann_router_query_batch");
        return std::vector<int>(probes.begin(), probes.end());
    }
    RouterQueryReq req;
    req.Q = Q;
    req.budget = budget;
    req.handle = device_router_;
    req.d = tree_.d;
    req.S = tree_.S;
    RouterQueue().Submit(req, RunRouterBatch);
    return std::vector<int>(req.probes.begin(), req.probes.end());
}

std::vector<KMeansTreeRouter::FrequencyQueryData> KMeansTreeRouter::FrequencyQueryBatch(PointSet& queries, int budget, int num_voting_neighbors) {
    EnsureOnDevice();
    const uint32_t nq = (uint32_t)queries.n, S = tree_.S;
    const uint32_t V = (uint32_t)std::max(0, num_voting_neighbors), Vs = std::max<uint32_t>(V, 1);
    std::vector<float> md((size_t)nq * S), nd((size_t)nq * Vs);
    std::vector<uint32_t> ns((size_t)nq * Vs), count(nq, 0);
    gpann_host::Check(gpann_router_frequency_query_batch(device_router_, queries.coordinates.data(), nq, budget, V, md.data(), nd.data(), ns.data(),
                                                         count.data(), 0, nullptr, nullptr),
                      "gpann_router_frequency_query_batch");
    std::vector<FrequencyQueryData> out(nq);
    for (uint32_t q = 0; q < nq; ++q) {
        out[q].min_dist.assign(md.begin() + (size_t)q * S, md.begin() + (size_t)(q + 1) * S);
        out[q].near_neighbors.resize(count[q]);
        for (uint32_t i = 0; i < count[q]; ++i) out[q].near_neighbors[i] = {nd[(size_t)q * V + i], ns[(size_t)q * V + i]};
        out[q].Init();
    }
    return out;
}

KMeansTreeRouter::FrequencyQueryData KMeansTreeRouter::FrequencyQuery(float* Q, int budget, int num_voting_neighbors) {
    PointSet one;
    one.n = 1;
    one.d = tree_.d;
    one.coordinates.assign(Q, Q + tree_.d);
    return std::move(FrequencyQueryBatch(one, budget, num_voting_neighbors)[0]);
}

// kmeans_tree_router.h:23-44
std::vector<int> KMeansTreeRouter::FrequencyQueryData::Query(int num_shards, int num_voting_neighbors) {
    std::vector<int> votes(num_shards, 0);
    const int voters = std::min<int>(num_voting_neighbors, (int)near_neighbors.size());
    for (int i = 0; i < voters; ++i) votes[near_neighbors[i].second]++;
    size_t winner = 0;
    for (size_t b = 1; b < votes.size(); ++b)
        if (votes[b] > votes[winner]) winner = b;   // strict: the lowest shard id wins a tie
    std::vector<int> probes;
    probes.reserve(num_shards);
    probes.push_back((int)winner);
    for (size_t b = 0; b < votes.size(); ++b)
        if (b != winner) probes.push_back((int)b);
    std::sort(probes.begin() + 1, probes.end(), [&](int l, int r) { return min_dist[l] < min_dist[r]; });
    return probes;
}

// kmeans_tree_router.cpp:174-201: the flat centroid array IS the BFS concatenation
std::pair<PointSet, std::vector<int>> KMeansTreeRouter::ExtractPoints() {
    PointSet points;
    points.d = tree_.d;
    points.n = tree_.node_cent_first.empty() ? 0 : tree_.node_cent_first.back();
    points.coordinates = tree_.centroids;
    if (points.coordinates.size() != points.n * points.d) throw std::runtime_error("C");
    std::vector<int> shard_of(points.n);
    for (size_t u = 0; u < tree_.node_shard.size(); ++u)
        for (uint32_t c = tree_.node_cent_first[u]; c < tree_.node_cent_first[u + 1]; ++c) shard_of[c] = tree_.node_shard[u];
    return {std::move(points), std::move(shard_of)};
}
