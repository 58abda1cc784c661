// core.cpp — host mirror: containers and conversions (reference src/defs.{h,cpp}), TopN (src/topn.h), the canonical
// host distance functions (src/dist.cpp), binary point / ground-truth files (src/points_io.cpp) and the text formats for
// partitions and clusters (src/metis_io.cpp).  New code; behaviour follows the cited lines.
#include <atomic>

#include "gpann_host.h"
#include "host_util.h"

#include <cmath>
#include <cstring>
#include <iostream>
#include <sstream>

// ============================================================================================== settings
namespace gpann_host {
namespace {
#ifdef MIPS_DISTANCE
bool g_mips = true;
#else
bool g_mips = false;
#endif
int g_device = 0;
int g_threads = 0;
bool g_filtered = true;
bool g_device_training = true;
}   // namespace
void SetDeviceTraining(bool on) { g_device_training = on; }
bool DeviceTraining() { return g_device_training; }
void SetMetricMIPS(bool mips) { g_mips = mips; }
bool MetricIsMIPS() { return g_mips; }
void SetDevice(int device) { g_device = device; }
int Device() { return g_device; }
void SetNumThreads(int n) { g_threads = n; }
void SetFilteredScan(bool on) { g_filtered = on; }
static std::atomic<bool> g_coalesce{true};
void SetCoalescing(bool on) { g_coalesce = on; }
bool Coalescing() { return g_coalesce; }
bool FilteredScan() { return g_filtered; }
int NumThreads() {
    if (g_threads > 0) return g_threads;
    unsigned hw = std::thread::hardware_concurrency();
    return hw ? (int)hw : 1;
}
}   // namespace gpann_host

// ============================================================================================== TopN (topn.h:12-29)
void TopN::Add(value_type e) {
    if (pq.size() < n) {
        pq.push(e);
        return;
    }
    if (pq.top() > e) {   // lexicographic pair order: keeps the n smallest (dist, id)
        pq.pop();
        pq.push(e);
    }
}

std::vector<TopN::value_type> TopN::Take() {
    std::vector<value_type> out(pq.size());
    for (size_t slot = out.size(); slot-- > 0;) {
        out[slot] = pq.top();
        pq.pop();
    }
    return out;
}

// ============================================================================================== defs.cpp
PointSet ExtractPointsInBucket(const std::vector<uint32_t>& bucket, PointSet& points) {
    PointSet out;
    out.d = points.d;
    out.n = bucket.size();
    out.coordinates.resize(out.n * out.d);
    float* dst = out.coordinates.data();
    for (uint32_t id : bucket) {
        std::memcpy(dst, points.GetPoint(id), sizeof(float) * points.d);
        dst += points.d;
    }
    return out;
}

int NumPartsInPartition(const Partition& partition) {
    int top = -1;
    for (int p : partition) top = std::max(top, p);
    return partition.empty() ? 0 : top + 1;
}

Clusters ConvertPartitionToClusters(const Partition& partition) {
    Clusters clusters(NumPartsInPartition(partition));
    for (size_t u = 0; u < partition.size(); ++u) clusters[partition[u]].push_back(static_cast<uint32_t>(u));
    return clusters;
}

Cover ConvertPartitionToCover(const Partition& partition) {
    Cover cover(partition.size());
    for (size_t u = 0; u < partition.size(); ++u) cover[u].assign(1, partition[u]);
    return cover;
}

Cover ConvertClustersToCover(const Clusters& clusters) {
    uint32_t largest = 0;
    for (const auto& c : clusters)
        for (uint32_t id : c) largest = std::max(largest, id);
    Cover cover(static_cast<size_t>(largest) + 1);   // like the reference: one entry even for an empty input
    for (size_t b = 0; b < clusters.size(); ++b)
        for (uint32_t id : clusters[b]) cover[id].push_back(static_cast<int>(b));
    return cover;
}

void RemapPartitionIDs(std::vector<int>& partition) {
    if (partition.empty()) return;
    const int parts = NumPartsInPartition(partition);
    std::vector<size_t> population(parts, 0);
    for (int p : partition) population[p]++;
    bool gap = false;
    for (size_t c : population) gap |= (c == 0);
    if (!gap) return;
    std::vector<int> compact(parts, 0);   // unused slots keep 0, as in the reference
    int next = 0;
    for (int p = 0; p < parts; ++p)
        if (population[p] > 0) compact[p] = next++;
    for (int& p : partition) p = compact[p];
}

NNVec ConvertTopKToNNVec(TopN& top_k) {
    NNVec v = top_k.Take();
    std::reverse(v.begin(), v.end());
    return v;
}

TopN ClosestLeaders(PointSet& points, PointSet& leader_points, uint32_t my_id, int k) {
    TopN best(k);
    float* me = points.GetPoint(my_id);
    for (uint32_t j = 0; j < leader_points.n; ++j) best.Add({distance(leader_points.GetPoint(j), me, points.d), j});
    return best;
}

// ============================================================================================== Timer (defs.h:61-98)
Timepoint Timer::Start() {
    if (running) throw std::runtime_error("Called Start() but timer is already running");
    running = true;
    start = std::chrono::high_resolution_clock::now();
    return start;
}
double Timer::Stop() {
    if (!running) throw std::runtime_error("Timer not running but called Stop()");
    Duration lap = std::chrono::high_resolution_clock::now() - start;
    total_duration += lap;
    running = false;
    return lap.count();
}
double Timer::Restart() {
    if (!running) throw std::runtime_error("Timer not running but called Restart()");
    auto now = std::chrono::high_resolution_clock::now();
    Duration lap = now - start;
    total_duration += lap;
    start = now;
    return lap.count();
}
double Timer::ElapsedRunning() {
    if (!running) throw std::runtime_error("Timer not running but called Restart()");
    Duration lap = std::chrono::high_resolution_clock::now() - start;
    return lap.count();
}

// ============================================================================================== dist.cpp (scalar branch)
// This file is compiled with -ffp-contract=off and without fast-math: every operation below rounds to binary32 on its
// own, which is what the device kernels do too.
float sqr_l2_dist(const float* a, const float* b, unsigned size) {
    float acc = 0.f;
    unsigned i = 0;
    for (; i + 4 <= size; i += 4) {   // dist.cpp:23-31
        const float e0 = a[i] - b[i], e1 = a[i + 1] - b[i + 1], e2 = a[i + 2] - b[i + 2], e3 = a[i + 3] - b[i + 3];
        float group = e0 * e0 + e1 * e1;
        group = group + e2 * e2;
        group = group + e3 * e3;
        acc = acc + group;
    }
    for (; i < size; ++i) {   // dist.cpp:33-36
        const float e = a[i] - b[i];
        acc = acc + e * e;
    }
    return acc;
}

float inner_product(float* p, float* q, unsigned d) {
    float acc = 0.f;
    for (unsigned i = 0; i < d; ++i) acc = acc + q[i] * p[i];
    return acc;
}

float mips_distance(float* p, float* q, unsigned d) { return 1.0f - inner_product(p, q, d); }

float vec_norm(float* p, unsigned d) {
    float acc = 0.f;
    for (unsigned i = 0; i < d; ++i) acc = acc + p[i] * p[i];
    return acc;
}

bool L2Normalize(float* p, unsigned d) {
    const double sq = vec_norm(p, d);
    if (sq < 1e-10) {
        std::cerr << "Vector is fully zero " << std::endl;
        for (unsigned i = 0; i < d; ++i) std::cerr << p[i] << " ";
        std::cerr << std::endl;
        return false;
    }
    const double len = std::sqrt(sq);
    for (unsigned i = 0; i < d; ++i) p[i] /= len;
    return true;
}

float distance(float* p, float* q, unsigned d) { return gpann_host::MetricIsMIPS() ? mips_distance(p, q, d) : sqr_l2_dist(p, q, d); }

float pos_distance(float* p, float* q, unsigned d) {
    if (gpann_host::MetricIsMIPS()) return distance(p, q, d) + 1.0;
    return distance(p, q, d);
}

// ============================================================================================== points_io.cpp
namespace {

bool EndsWith(const std::string& s, const std::string& suffix) {
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

struct BinHeader {
    uint32_t n = 0, d = 0;
};

BinHeader ReadHeader(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    BinHeader h;
    in.read(reinterpret_cast<char*>(&h.n), 4);
    in.read(reinterpret_cast<char*>(&h.d), 4);
    return h;
}

// u32 n, u32 d, n*d elements of T, widened to float (points_io.cpp:14-111); rows are read in parallel slabs
template <typename T>
PointSet ReadMatrix(const std::string& path, int64_t size_override) {
    BinHeader h = ReadHeader(path);
    if (size_override != -1) h.n = static_cast<uint32_t>(size_override);
    std::cout << h.n << " " << h.d << std::endl;
    PointSet points;
    points.n = h.n;
    points.d = h.d;
    points.coordinates.resize(points.n * points.d);
    const size_t total = points.coordinates.size();
    const size_t slabs = std::max<size_t>(1, std::min<size_t>(gpann_host::NumThreads(), 64));
    const size_t per = (total + slabs - 1) / slabs;
    gpann_host::ParallelFor(0, slabs, (int)slabs, [&](size_t s) {
        const size_t lo = std::min(total, s * per), hi = std::min(total, (s + 1) * per);
        if (lo >= hi) return;
        std::ifstream in(path, std::ios::binary);
        in.seekg(8 + lo * sizeof(T));
        if constexpr (std::is_same<T, float>::value) {
            in.read(reinterpret_cast<char*>(&points.coordinates[lo]), (hi - lo) * sizeof(float));
        } else {
            std::vector<T> raw(hi - lo);
            in.read(reinterpret_cast<char*>(raw.data()), raw.size() * sizeof(T));
            for (size_t i = 0; i < raw.size(); ++i) points.coordinates[lo + i] = static_cast<float>(raw[i]);
        }
    });
    return points;
}

}   // namespace

PointSet ReadPoints(const std::string& path, int64_t size) {
    if (EndsWith(path, ".fbin")) return ReadMatrix<float>(path, size);
    if (EndsWith(path, ".u8bin")) return ReadMatrix<uint8_t>(path, -1);   // the reference ignores `size` for byte files
    if (EndsWith(path, ".i8bin")) return ReadMatrix<int8_t>(path, -1);
    throw std::runtime_error("Invalid file ending for the pointset path. Valid options are [.fbin, .u8bin, .i8bin]");
}

void WritePoints(PointSet& points, const std::string& path) {
    std::ofstream out(path, std::ios::binary);
    const uint32_t n = static_cast<uint32_t>(points.n), d = static_cast<uint32_t>(points.d);
    std::cout << n << " " << d << std::endl;
    out.write(reinterpret_cast<const char*>(&n), 4);
    out.write(reinterpret_cast<const char*>(&d), 4);
    out.write(reinterpret_cast<const char*>(points.coordinates.data()), points.coordinates.size() * sizeof(float));
}

// u32 nq, u32 k, nq*k u32 ids, nq*k f32 distances (points_io.cpp:136-179)
std::vector<NNVec> ReadGroundTruth(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    uint32_t nq = 0, k = 0;
    in.read(reinterpret_cast<char*>(&nq), 4);
    in.read(reinterpret_cast<char*>(&k), 4);
    std::cout << "num queries = " << nq << " num neighbors = " << k << std::endl;
    std::vector<uint32_t> ids(static_cast<size_t>(nq) * k);
    std::vector<float> dists(ids.size());
    in.read(reinterpret_cast<char*>(ids.data()), ids.size() * 4);
    in.read(reinterpret_cast<char*>(dists.data()), dists.size() * 4);
    std::vector<NNVec> gt(nq, NNVec(k));
    for (size_t q = 0; q < nq; ++q)
        for (size_t j = 0; j < k; ++j) gt[q][j] = {dists[q * k + j], ids[q * k + j]};
    return gt;
}

void WriteGroundTruth(const std::string& path, const std::vector<NNVec>& ground_truth) {
    const uint32_t nq = static_cast<uint32_t>(ground_truth.size());
    const uint32_t k = static_cast<uint32_t>(ground_truth[0].size());
    std::ofstream out(path, std::ios::binary);
    out.write(reinterpret_cast<const char*>(&nq), 4);
    out.write(reinterpret_cast<const char*>(&k), 4);
    for (const auto& row : ground_truth)
        for (uint32_t j = 0; j < k; ++j) out.write(reinterpret_cast<const char*>(&row[j].second), 4);
    for (const auto& row : ground_truth)
        for (uint32_t j = 0; j < k; ++j) out.write(reinterpret_cast<const char*>(&row[j].first), 4);
}

// ============================================================================================== metis_io.cpp
std::vector<int> ReadMetisPartition(const std::string& path) {
    std::ifstream in(path);
    std::vector<int> partition;
    for (int p; in >> p;) partition.push_back(p);
    RemapPartitionIDs(partition);   // compacts ids when a shard is empty (metis_io.cpp:12-14)
    return partition;
}

void WriteMetisPartition(const std::vector<int>& partition, const std::string& path) {
    std::ofstream out(path);
    for (int p : partition) out << p << "\n";
}

Clusters ReadClusters(const std::string& path) {
    std::ifstream in(path);
    Clusters clusters;
    for (std::string line; std::getline(in, line);) {
        std::istringstream tokens(line);
        auto& shard = clusters.emplace_back();
        for (uint32_t id; tokens >> id;) shard.push_back(id);
    }
    return clusters;
}

void WriteClusters(const Clusters& clusters, const std::string& path) {
    std::ofstream out(path);
    for (const auto& shard : clusters) {
        for (uint32_t id : shard) out << id << " ";
        out << "\n";
    }
}

void WriteMetisGraph(const std::string& path, const AdjGraph& graph) {
    uint64_t endpoints = 0;
    for (const auto& nbrs : graph) endpoints += nbrs.size();
    if (endpoints % 2 != 0) throw std::runtime_error("Number of edges not even");
    std::ofstream out(path);
    out << graph.size() << " " << endpoints / 2 << "\n";
    for (const auto& nbrs : graph) {
        for (int v : nbrs) out << (v + 1) << " ";
        out << "\n";
    }
}
