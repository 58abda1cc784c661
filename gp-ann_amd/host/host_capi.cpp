// host_capi.cpp — a small extern "C" surface over the C++ host mirror so that Python (tests, bench.py) can drive the
// host-side pieces that have no device counterpart: router training, k-means, the text/binary file formats.
#include <chrono>
#include <cstring>
#include <limits>
#include <mutex>
#include <stdexcept>

#include "gpann_host.h"
#include "host_util.h"

namespace {
Clusters FromCsr(const uint32_t* ids, const uint64_t* offsets, uint32_t S) {
    Clusters c(S);
    for (uint32_t b = 0; b < S; ++b) c[b].assign(ids + offsets[b], ids + offsets[b + 1]);
    return c;
}
thread_local std::string g_err;
}   // namespace

extern "C" {

const char* gpann_host_last_error() { return g_err.c_str(); }

void gpann_host_configure(int mips, int device, int threads) {
    gpann_host::SetMetricMIPS(mips != 0);
    gpann_host::SetDevice(device);
    gpann_host::SetNumThreads(threads);
}

// KMeansTreeRouter::Train on the host; returns an opaque router (the C++ object), or null on error
void* gpann_host_router_train(const float* points, uint64_t n, uint32_t d, const uint32_t* ids, const uint64_t* offsets, uint32_t S,
                              uint64_t num_centroids, uint64_t min_cluster_size, int64_t budget) {
    try {
        PointSet ps;
        ps.n = n;
        ps.d = d;
        ps.coordinates.assign(points, points + n * d);
        KMeansTreeRouterOptions o;
        o.num_centroids = num_centroids;
        o.min_cluster_size = min_cluster_size;
        o.budget = budget;
        auto* r = new KMeansTreeRouter();
        r->TrainOnHost(ps, FromCsr(ids, offsets, S), o);
        return r;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

void gpann_host_router_free(void* r) { delete static_cast<KMeansTreeRouter*>(r); }

void gpann_host_router_sizes(void* r, uint32_t* n_nodes, uint64_t* n_centroids) {
    const auto& f = static_cast<KMeansTreeRouter*>(r)->flat();
    *n_nodes = (uint32_t)f.node_shard.size();
    *n_centroids = f.node_cent_first.empty() ? 0 : f.node_cent_first.back();
}

void gpann_host_router_export(void* r, int32_t* node_shard, uint32_t* node_cent_first, uint32_t* node_child_first, uint32_t* node_child_count,
                              uint32_t* shard_root, float* centroids) {
    const auto& f = static_cast<KMeansTreeRouter*>(r)->flat();
    std::memcpy(node_shard, f.node_shard.data(), f.node_shard.size() * 4);
    std::memcpy(node_cent_first, f.node_cent_first.data(), f.node_cent_first.size() * 4);
    std::memcpy(node_child_first, f.node_child_first.data(), f.node_child_first.size() * 4);
    std::memcpy(node_child_count, f.node_child_count.data(), f.node_child_count.size() * 4);
    std::memcpy(shard_root, f.shard_root.data(), f.shard_root.size() * 4);
    std::memcpy(centroids, f.centroids.data(), f.centroids.size() * 4);
}

// RandomSample(seed) + KMeans; returns the number of centroids left, -1 on error
int64_t gpann_host_kmeans(const float* points, uint64_t n, uint32_t d, uint32_t k, int seed, int32_t* out_partition, float* out_centroids) {
    try {
        PointSet ps;
        ps.n = n;
        ps.d = d;
        ps.coordinates.assign(points, points + n * d);
        PointSet cent = RandomSample(ps, k, seed);
        std::vector<int> part = KMeans(ps, cent);
        for (uint64_t i = 0; i < n; ++i) out_partition[i] = part[i];
        if (out_centroids) std::memcpy(out_centroids, cent.coordinates.data(), cent.coordinates.size() * 4);
        return (int64_t)cent.n;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// the k-means shard builders (partition_host.cpp).  which: 0 = KMeansPartitioning(points, k, eps), 1 = RecursiveKMeansPartitioning(points,
// max_cluster_size, 0, k), 2 = RebalancingKMeansPartitioning(points, max_cluster_size, k); k < 0 = derive it from max_cluster_size.
// device != 0: k-means on the GPU.  Returns the number of parts, -1 on error.
int64_t gpann_host_partition(int which, const float* points, uint64_t n, uint32_t d, int k, double eps, uint64_t max_cluster_size, int device,
                             int32_t* out_partition) {
    const bool before = gpann_host::DeviceTraining();
    try {
        PointSet ps;
        ps.n = n;
        ps.d = d;
        ps.coordinates.assign(points, points + n * d);
        gpann_host::SetDeviceTraining(device != 0);
        Partition part = which == 0   ? KMeansPartitioning(ps, k, eps)
                         : which == 1 ? RecursiveKMeansPartitioning(ps, max_cluster_size, 0, k)
                                      : RebalancingKMeansPartitioning(ps, max_cluster_size, k);
        gpann_host::SetDeviceTraining(before);
        for (uint64_t i = 0; i < n; ++i) out_partition[i] = part[i];
        return n ? NumPartsInPartition(part) : 0;
    } catch (const std::exception& e) {
        gpann_host::SetDeviceTraining(before);
        g_err = e.what();
        return -1;
    }
}

// OverlappingKMeansPartitioningSPANN(points, partition, requested_num_clusters, eps, overlap); the clusters come back as CSR
// (out_ids holds up to `capacity` ids).  Returns the number of clusters, -1 on error.
int64_t gpann_host_overlap_spann(const float* points, uint64_t n, uint32_t d, const int32_t* partition, int requested_num_clusters, double eps,
                                 double overlap, int device, uint32_t* out_ids, uint64_t capacity, uint64_t* out_offsets) {
    const bool before = gpann_host::DeviceTraining();
    try {
        PointSet ps;
        ps.n = n;
        ps.d = d;
        ps.coordinates.assign(points, points + n * d);
        gpann_host::SetDeviceTraining(device != 0);
        Clusters clusters = OverlappingKMeansPartitioningSPANN(ps, Partition(partition, partition + n), requested_num_clusters, eps, overlap);
        gpann_host::SetDeviceTraining(before);
        uint64_t at = 0;
        out_offsets[0] = 0;
        for (size_t b = 0; b < clusters.size(); ++b) {
            if (at + clusters[b].size() > capacity) throw std::runtime_error("gpann_host_overlap_spann: output capacity too small");
            std::memcpy(out_ids + at, clusters[b].data(), clusters[b].size() * 4);
            at += clusters[b].size();
            out_offsets[b + 1] = at;
        }
        return (int64_t)clusters.size();
    } catch (const std::exception& e) {
        gpann_host::SetDeviceTraining(before);
        g_err = e.what();
        return -1;
    }
}

// ---- the file formats of the hot path's callers (points_io.cpp, metis_io.cpp): thin access for tests
// ReadPoints(path, size): header -> *n, *d; coordinates -> out (capacity floats; may be null to query the shape).  Returns 0, -1 on error.
int gpann_host_read_points(const char* path, int64_t size, uint64_t* n, uint32_t* d, float* out, uint64_t capacity) {
    try {
        PointSet ps = ReadPoints(path, size);
        *n = ps.n;
        *d = (uint32_t)ps.d;
        if (out) {
            if (ps.coordinates.size() > capacity) throw std::runtime_error("gpann_host_read_points: output capacity too small");
            std::memcpy(out, ps.coordinates.data(), ps.coordinates.size() * 4);
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

int gpann_host_write_points(const char* path, const float* points, uint64_t n, uint32_t d) {
    try {
        PointSet ps;
        ps.n = n;
        ps.d = d;
        ps.coordinates.assign(points, points + n * d);
        WritePoints(ps, path);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// WriteGroundTruth / ReadGroundTruth round trip: writes [nq][k] (dist, id) pairs, reads them back into out_*
int gpann_host_ground_truth_roundtrip(const char* path, const float* dist, const uint32_t* ids, uint32_t nq, uint32_t k, float* out_dist, uint32_t* out_ids) {
    try {
        std::vector<NNVec> gt(nq, NNVec(k));
        for (uint32_t q = 0; q < nq; ++q)
            for (uint32_t j = 0; j < k; ++j) gt[q][j] = {dist[(size_t)q * k + j], ids[(size_t)q * k + j]};
        WriteGroundTruth(path, gt);
        std::vector<NNVec> back = ReadGroundTruth(path);
        if (back.size() != nq) throw std::runtime_error("ground truth round trip: query count differs");
        for (uint32_t q = 0; q < nq; ++q) {
            if (back[q].size() != k) throw std::runtime_error("ground truth round trip: k differs");
            for (uint32_t j = 0; j < k; ++j) {
                out_dist[(size_t)q * k + j] = back[q][j].first;
                out_ids[(size_t)q * k + j] = back[q][j].second;
            }
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// ---- the reference's calling pattern, for tests and the benchmark of the coalescing queue: `threads` host threads, each calling the
//      one-query interface for its share of the queries (routes.cpp:194-200, shard_searches.cpp:48), with the queue on or off
void gpann_host_router_queue_counters(uint64_t* batches, uint64_t* requests);
void gpann_host_scan_queue_counters(uint64_t* batches, uint64_t* requests);

// tree: the flat arrays of gpann_router_create.  out_probes [nq][S].  Returns 0 / -1; *seconds = wall time of the query loop, *batches = device batches it took
int gpann_host_router_query_threads(uint32_t d, int mips, uint32_t S, uint32_t n_nodes, const int32_t* node_shard, const uint32_t* node_cent_first,
                                    const uint32_t* node_child_first, const uint32_t* node_child_count, const uint32_t* shard_root, const float* centroids,
                                    const float* queries, uint32_t nq, int budget, int threads, int coalesce, int32_t* out_probes, double* seconds, uint64_t* batches) {
    try {
        gpann_host::SetMetricMIPS(mips != 0);
        KMeansTreeRouter::Flat f;
        f.d = d;
        f.S = S;
        f.mips = mips != 0;
        f.node_shard.assign(node_shard, node_shard + n_nodes);
        f.node_cent_first.assign(node_cent_first, node_cent_first + n_nodes + 1);
        f.node_child_first.assign(node_child_first, node_child_first + n_nodes);
        f.node_child_count.assign(node_child_count, node_child_count + n_nodes);
        f.shard_root.assign(shard_root, shard_root + S);
        f.centroids.assign(centroids, centroids + (size_t)f.node_cent_first.back() * d);
        KMeansTreeRouter router;
        router.LoadFlat(std::move(f));
        std::vector<float> q(queries, queries + (size_t)nq * d);
        (void)router.Query(q.data(), budget);   // (the upload and the first call's workspace growth stay out of the clock)
        const bool before = gpann_host::Coalescing();
        gpann_host::SetCoalescing(coalesce != 0);
        uint64_t b0 = 0, r0 = 0, b1 = 0, r1 = 0;
        gpann_host_router_queue_counters(&b0, &r0);
        std::string err;
        std::mutex err_mutex;
        const auto t0 = std::chrono::steady_clock::now();
        gpann_host::ParallelFor(0, nq, threads, [&](size_t i) {
            try {
                const std::vector<int> p = router.Query(&q[i * d], budget);
                for (uint32_t b = 0; b < S; ++b) out_probes[i * S + b] = p[b];
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(err_mutex);
                err = e.what();
            }
        });
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        gpann_host_router_queue_counters(&b1, &r1);
        *batches = coalesce ? b1 - b0 : nq;
        gpann_host::SetCoalescing(before);
        if (!err.empty()) throw std::runtime_error(err);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// InvertedIndex::QueryBucket(Q, k, probes[q]) from `threads` host threads, one call per query: out_ids / out_dist [nq][k]
int gpann_host_index_query_bucket_threads(const float* points, uint64_t n, uint32_t d, int mips, const uint32_t* ids, const uint64_t* offsets, uint32_t S,
                                          const float* queries, uint32_t nq, uint32_t k, const int32_t* bucket, int threads, int coalesce, uint32_t* out_ids,
                                          float* out_dist, double* seconds, uint64_t* batches, uint8_t* failed) {
    // `failed` (may be null): [nq], 1 where THAT query's call threw — the call of another thread must not fail with it; null: any failure fails the run
    try {
        gpann_host::SetMetricMIPS(mips != 0);
        PointSet ps;
        ps.n = n;
        ps.d = d;
        ps.coordinates.assign(points, points + n * d);
        Clusters clusters = FromCsr(ids, offsets, S);
        InvertedIndex index(ps, clusters);
        std::vector<float> q(queries, queries + (size_t)nq * d);
        (void)index.QueryBucket(q.data(), (int)k, bucket[0]);
        const bool before = gpann_host::Coalescing();
        gpann_host::SetCoalescing(coalesce != 0);
        uint64_t b0 = 0, r0 = 0, b1 = 0, r1 = 0;
        gpann_host_scan_queue_counters(&b0, &r0);
        std::string err;
        std::mutex err_mutex;
        const auto t0 = std::chrono::steady_clock::now();
        gpann_host::ParallelFor(0, nq, threads, [&](size_t i) {
            try {
                const NNVec r = index.QueryBucket(&q[i * d], (int)k, bucket[i]);
                for (uint32_t x = 0; x < k; ++x) {
                    out_dist[i * k + x] = x < r.size() ? r[x].first : std::numeric_limits<float>::infinity();
                    out_ids[i * k + x] = x < r.size() ? r[x].second : 0xFFFFFFFFu;
                }
                if (failed) failed[i] = 0;
            } catch (const std::exception& e) {
                if (failed) {
                    failed[i] = 1;
                } else {
                    std::lock_guard<std::mutex> lk(err_mutex);
                    err = e.what();
                }
            }
        });
        *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        gpann_host_scan_queue_counters(&b1, &r1);
        *batches = coalesce ? b1 - b0 : nq;
        gpann_host::SetCoalescing(before);
        if (!err.empty()) throw std::runtime_error(err);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

float gpann_host_distance(const float* p, const float* q, uint32_t d) { return distance(const_cast<float*>(p), const_cast<float*>(q), d); }

}   // extern "C"
