// index_host.cpp — host mirror of InvertedIndex (reference src/inverted_index.h), the ground-truth / recall helpers
// (src/recall.h), DedupNeighbors (small_scale_queries.cpp:15-23) and RunInShardSearches + ShardSearch (src/shard_searches.*).
// All scans run on the GPU through the C-ABI of libgpann_hip.so.
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <sstream>

#include "gpann_host.h"
#include "host_util.h"

namespace gpann_host {
void Check(gpann_status status, const char* what) {
    if (status == GPANN_OK) return;
    throw std::runtime_error(std::string(what) + " failed (status " + std::to_string((int)status) + "): " + gpann_last_error());
}
}   // namespace gpann_host

using gpann_host::Check;
using gpann_host::ParallelFor;

namespace {

int MetricNow() { return gpann_host::MetricIsMIPS() ? GPANN_METRIC_MIPS : GPANN_METRIC_L2; }

struct Csr {
    std::vector<uint32_t> ids;
    std::vector<uint64_t> offsets;
};

Csr ToCsr(const Clusters& clusters) {
    Csr csr;
    csr.offsets.assign(clusters.size() + 1, 0);
    for (size_t b = 0; b < clusters.size(); ++b) csr.offsets[b + 1] = csr.offsets[b] + clusters[b].size();
    csr.ids.reserve(csr.offsets.back());
    for (const auto& c : clusters) csr.ids.insert(csr.ids.end(), c.begin(), c.end());
    return csr;
}

// [nq][nprobe][k] flat results -> NNVec per (q, j) without the (+inf, NO_ID) padding
NNVec SlotToNNVec(const float* dist, const uint32_t* ids, uint32_t k) {
    NNVec v;
    v.reserve(k);
    for (uint32_t x = 0; x < k && ids[x] != GPANN_NO_ID; ++x) v.emplace_back(dist[x], ids[x]);
    return v;
}

}   // namespace

// ================================================================================================== InvertedIndex
InvertedIndex::InvertedIndex(PointSet& points, const Clusters& clusters) {
    mips_ = gpann_host::MetricIsMIPS();
    const Csr csr = ToCsr(clusters);
    // inverted_index.h:13-21: offsets are ints in the reference
    offsets.assign(clusters.size() + 1, 0);
    for (size_t b = 0; b < clusters.size(); ++b) offsets[b + 1] = offsets[b] + (int)clusters[b].size();
    permutation = csr.ids;   // :32 permutation[offsets[b] + i_local] = clusters[b][i_local]
    clustered_points.d = points.d;
    clustered_points.n = csr.ids.size();   // coordinates stay on the device
    Check(gpann_shards_create(points.coordinates.data(), points.n, (uint32_t)points.d, mips_ ? GPANN_METRIC_MIPS : GPANN_METRIC_L2, csr.ids.data(),
                              csr.offsets.data(), (uint32_t)clusters.size(), gpann_host::Device(), 0, &shards_),
          "gpann_shards_create");
    Check(gpann_shards_set_scan_mode(shards_, gpann_host::FilteredScan() ? 1 : 0), "gpann_shards_set_scan_mode");
    d_ = points.d;
}

InvertedIndex::~InvertedIndex() {
    if (shards_) gpann_shards_destroy(shards_);
}

PointSet InvertedIndex::ExportClusteredPoints() const {
    PointSet out;
    out.d = clustered_points.d;
    out.n = clustered_points.n;
    out.coordinates.resize(out.n * out.d);
    Check(gpann_shards_export(shards_, nullptr, nullptr, out.coordinates.data()), "gpann_shards_export");
    return out;
}

std::vector<std::vector<NNVec>> InvertedIndex::QueryBucketBatch(PointSet& queries, int k, const std::vector<std::vector<int>>& probes, size_t num_probes) {
    const uint32_t nq = (uint32_t)queries.n, np = (uint32_t)num_probes;
    std::vector<int32_t> flat((size_t)nq * np, -1);
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t j = 0; j < np && j < probes[q].size(); ++j) flat[(size_t)q * np + j] = probes[q][j];
    std::vector<uint32_t> ids((size_t)nq * np * k);
    std::vector<float> dist(ids.size());
    float ms = 0.f;
    Check(gpann_scan_buckets(shards_, queries.coordinates.data(), nq, (uint32_t)k, flat.data(), np, np, ids.data(), dist.data(), nullptr, 0, nullptr, &ms),
          "gpann_scan_buckets");
    last_scan_ms = ms;
    std::vector<std::vector<NNVec>> out(nq, std::vector<NNVec>(np));
    ParallelFor(0, nq, 0, [&](size_t q) {
        for (uint32_t j = 0; j < np; ++j) out[q][j] = SlotToNNVec(&dist[(q * np + j) * k], &ids[(q * np + j) * k], (uint32_t)k);
    });
    return out;
}

std::vector<NNVec> InvertedIndex::QueryBatch(PointSet& queries, int k, const std::vector<std::vector<int>>& probes, size_t num_probes) {
    const uint32_t nq = (uint32_t)queries.n, np = (uint32_t)num_probes;
    std::vector<int32_t> flat((size_t)nq * np, -1);
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t j = 0; j < np && j < probes[q].size(); ++j) flat[(size_t)q * np + j] = probes[q][j];
    std::vector<uint32_t> ids((size_t)nq * k);
    std::vector<float> dist(ids.size());
    float ms = 0.f;
    Check(gpann_scan_merged(shards_, queries.coordinates.data(), nq, (uint32_t)k, flat.data(), np, np, ids.data(), dist.data(), 0, nullptr, &ms),
          "gpann_scan_merged");
    last_scan_ms = ms;
    std::vector<NNVec> out(nq);
    for (uint32_t q = 0; q < nq; ++q) out[q] = SlotToNNVec(&dist[(size_t)q * k], &ids[(size_t)q * k], (uint32_t)k);
    return out;
}

// inverted_index.h:43-75 — Query (one TopN over the probed shards) and QueryBucket (one shard).  The reference's callers run them from many
// threads, one query per call (shard_searches.cpp:48, small_scale_queries.cpp:149-155): concurrent calls are coalesced into device batches
// (host_util.h Coalescer) — per (store, kind, k) one scan call over everything that queued up while the previous batch ran; a query's probe
// list is padded with -1 (= skip) to the longest of its batch.  Results are those of a batch of one (the batch entry points are row-independent).
namespace {
struct ScanReq {
    const float* Q = nullptr;
    gpann_shards* handle = nullptr;
    uint32_t d = 0, k = 0;
    bool merged = false;               // InvertedIndex::Query (gpann_scan_merged) / QueryBucket (gpann_scan_buckets, one slot)
    std::vector<int32_t> probes;
    std::vector<uint32_t> ids;
    std::vector<float> dist;
    bool done = false;
    std::exception_ptr error;
};
gpann_host::Coalescer<ScanReq>& ScanQueue() {
    static gpann_host::Coalescer<ScanReq>* q = new gpann_host::Coalescer<ScanReq>();
    return *q;
}
// One device call for a group of requests that share (handle, kind, k).  If the call fails and the group has several members, every member is run
// again AS A BATCH OF ONE: a caller's bad argument (a bucket out of range, k beyond GPANN_MAX_K) then fails that caller alone — what the reference's
// one-query-per-call interface promises — instead of every thread that happened to share its device batch.
void RunScanGroup(std::vector<ScanReq*>& group) {
    {
        const uint32_t n = (uint32_t)group.size(), d = group[0]->d, k = group[0]->k;
        uint32_t np = 0;
        for (ScanReq* r : group) np = std::max<uint32_t>(np, (uint32_t)r->probes.size());
        std::vector<float> q((size_t)n * d);
        std::vector<int32_t> probes((size_t)n * std::max(np, 1u), -1);
        for (uint32_t x = 0; x < n; ++x) {
            std::memcpy(&q[(size_t)x * d], group[x]->Q, (size_t)d * sizeof(float));
            std::copy(group[x]->probes.begin(), group[x]->probes.end(), probes.begin() + (size_t)x * np);
        }
        std::vector<uint32_t> ids((size_t)n * k);
        std::vector<float> dist((size_t)n * k);
        try {
            if (group[0]->merged)
                Check(gpann_scan_merged(group[0]->handle, q.data(), n, k, probes.data(), np, np, ids.data(), dist.data(), 0, nullptr, nullptr), "gpann_scan_merged");
            else
                Check(gpann_scan_buckets(group[0]->handle, q.data(), n, k, probes.data(), 1, 1, ids.data(), dist.data(), nullptr, 0, nullptr, nullptr), "gpann_scan_buckets");
            for (uint32_t x = 0; x < n; ++x) {
                group[x]->ids.assign(ids.begin() + (size_t)x * k, ids.begin() + (size_t)(x + 1) * k);
                group[x]->dist.assign(dist.begin() + (size_t)x * k, dist.begin() + (size_t)(x + 1) * k);
            }
        } catch (...) {
            if (group.size() == 1) {
                group[0]->error = std::current_exception();
            } else {
                for (ScanReq* r : group) {
                    std::vector<ScanReq*> one{r};
                    RunScanGroup(one);
                }
            }
        }
    }
}
void RunScanBatch(std::vector<ScanReq*>& batch) {
    std::vector<char> taken(batch.size(), 0);
    for (size_t i = 0; i < batch.size(); ++i) {
        if (taken[i]) continue;
        std::vector<ScanReq*> group;
        for (size_t j = i; j < batch.size(); ++j)
            if (!taken[j] && batch[j]->handle == batch[i]->handle && batch[j]->merged == batch[i]->merged && batch[j]->k == batch[i]->k) {
                taken[j] = 1;
                group.push_back(batch[j]);
            }
        RunScanGroup(group);
    }
}
}   // namespace

extern "C" void gpann_host_scan_queue_counters(uint64_t* batches, uint64_t* requests) { ScanQueue().Counters(batches, requests); }

NNVec InvertedIndex::Query(float* Q, int k, const std::vector<int>& buckets_to_probe, size_t num_buckets_to_probe) {
    std::vector<int32_t> probes(buckets_to_probe.begin(), buckets_to_probe.begin() + num_buckets_to_probe);
    if (!gpann_host::Coalescing() || probes.empty()) {
        std::vector<uint32_t> ids(k);
        std::vector<float> dist(k);
        Check(gpann_scan_merged(shards_, Q, 1, (uint32_t)k, probes.data(), (uint32_t)probes.size(), (uint32_t)probes.size(), ids.data(), dist.data(), 0, nullptr,
                                nullptr),
              "gpann_scan_merged");
        return SlotToNNVec(dist.data(), ids.data(), (uint32_t)k);
    }
    ScanReq req;
    req.Q = Q;
    req.handle = shards_;
    req.d = (uint32_t)d_;
    req.k = (uint32_t)k;
    req.merged = true;
    req.probes = std::move(probes);
    ScanQueue().Submit(req, RunScanBatch);
    return SlotToNNVec(req.dist.data(), req.ids.data(), (uint32_t)k);
}

// inverted_index.h:62-75
NNVec InvertedIndex::QueryBucket(float* Q, int k, int bucket) {
    if (!gpann_host::Coalescing()) {
        int32_t probe = bucket;
        std::vector<uint32_t> ids(k);
        std::vector<float> dist(k);
        Check(gpann_scan_buckets(shards_, Q, 1, (uint32_t)k, &probe, 1, 1, ids.data(), dist.data(), nullptr, 0, nullptr, nullptr), "gpann_scan_buckets");
        return SlotToNNVec(dist.data(), ids.data(), (uint32_t)k);
    }
    ScanReq req;
    req.Q = Q;
    req.handle = shards_;
    req.d = (uint32_t)d_;
    req.k = (uint32_t)k;
    req.merged = false;
    req.probes.assign(1, (int32_t)bucket);
    ScanQueue().Submit(req, RunScanBatch);
    return SlotToNNVec(req.dist.data(), req.ids.data(), (uint32_t)k);
}

// ================================================================================================== recall.h
std::vector<NNVec> ComputeGroundTruth(PointSet& points, PointSet& queries, int k) {
    const uint32_t nq = (uint32_t)queries.n;
    std::vector<uint32_t> ids((size_t)nq * k);
    std::vector<float> dist(ids.size());
    Check(gpann_ground_truth(nullptr, points.coordinates.data(), points.n, (uint32_t)points.d, MetricNow(), queries.coordinates.data(), nq, (uint32_t)k,
                             ids.data(), dist.data(), gpann_host::Device(), 0, nullptr, nullptr),
          "gpann_ground_truth");
    std::vector<NNVec> gt(nq);
    for (uint32_t q = 0; q < nq; ++q) gt[q] = SlotToNNVec(&dist[(size_t)q * k], &ids[(size_t)q * k], (uint32_t)k);   // ascending (dist, id)
    return gt;
}

std::vector<float> ComputeDistanceToKthNeighbor(PointSet& points, PointSet& queries, int k) {
    auto gt = ComputeGroundTruth(points, queries, k);
    std::vector<float> kth(queries.n);
    for (size_t q = 0; q < queries.n; ++q) kth[q] = gt[q].back().first;   // TopN::Top() = largest kept = k-th smallest
    return kth;
}

// recall.h:41-68
void OracleRecall(const std::vector<NNVec>& ground_truth, const std::vector<int>& partition, int num_neighbors) {
    const int S = NumPartsInPartition(partition);
    std::vector<size_t> hits(S, 0);
    for (const auto& neigh : ground_truth) {
        std::vector<std::pair<int, int>> freq(S);
        for (int b = 0; b < S; ++b) freq[b] = {0, b};
        for (int i = 0; i < num_neighbors; ++i) freq[partition[neigh[i].second]].first++;
        std::sort(freq.begin(), freq.end(), std::greater<>());
        for (int b = 0; b < S; ++b) hits[b] += freq[b].first;
    }
    size_t total = 0;
    for (int b = 0; b < S && hits[b] > 0; ++b) {
        total += hits[b];
        std::cout << "nprobes = " << b + 1 << " oracle recall = " << static_cast<double>(total) / ground_truth.size() / num_neighbors << std::endl;
    }
}

// recall.h:74-137.  The k distances per query are re-evaluated with the canonical host distance() — the same arithmetic
// the scan kernels use — so thresholds and scanned distances are comparable bit for bit.
std::vector<float> ConvertGroundTruthToDistanceToKthNeighbor(std::vector<NNVec>& ground_truth, int k, PointSet& points, PointSet& queries) {
    if (ground_truth.size() != queries.n) {
        std::cout << "Ground truth size and number of queries don't match." << std::endl;
        std::exit(0);
    }
    std::cout << "Convert ground truth. k = " << k << " points.d " << points.d << std::endl;
    std::vector<float> kth(ground_truth.size());
    std::atomic<size_t> mismatches{0}, resorted{0}, unsorted_on_input{0};
    const std::vector<double> epss = {1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9, 1e-10};
    std::vector<std::atomic<size_t>> beyond(epss.size());
    for (auto& b : beyond) b = 0;
    auto by_dist = [](const std::pair<float, uint32_t>& l, const std::pair<float, uint32_t>& r) { return l.first < r.first; };
    ParallelFor(0, queries.n, 0, [&](size_t q) {
        NNVec& neighs = ground_truth[q];
        if (!std::is_sorted(neighs.begin(), neighs.begin() + k, by_dist)) unsorted_on_input++;
        float* Q = queries.GetPoint(q);
        size_t local = 0;
        for (int j = 0; j < k; ++j) {
            const float stored = neighs[j].first;
            const float actual = distance(points.GetPoint(neighs[j].second), Q, (unsigned)points.d);
            if (std::abs(stored - actual) > 1e-8) local++;
            for (size_t r = 0; r < epss.size(); ++r)
                if (std::abs(stored - actual) > epss[r]) beyond[r]++;
            neighs[j].first = actual;
        }
        if (!std::is_sorted(neighs.begin(), neighs.begin() + k, by_dist)) {
            std::sort(neighs.begin(), neighs.begin() + k, by_dist);
            resorted++;
        }
        kth[q] = neighs[k - 1].first;
        mismatches += local;
    });
    std::cout << mismatches << " out of " << ground_truth.size() * ground_truth[0].size() << " distances were wrong" << std::endl;
    std::cout << unsorted_on_input << " out of " << ground_truth.size()
              << " neighbors lists were ordered incorrectly before recomputing distances. And " << resorted << " were ordered incorrectly after recomputing"
              << std::endl;
    for (size_t r = 0; r < epss.size(); ++r)
        std::cout << "For eps = " << epss[r] << " there were " << beyond[r] << " many distances wrong, i.e., |d1 - d2| > eps" << std::endl;
    return kth;
}

// recall.h:139-150
double Recall(const std::vector<NNVec>& neighbors_per_query, const std::vector<float>& distance_to_kth_neighbor, int k) {
    size_t hits = 0;
    for (size_t q = 0; q < neighbors_per_query.size(); ++q)
        for (const auto& e : neighbors_per_query[q]) hits += (e.first <= distance_to_kth_neighbor[q]);
    return static_cast<double>(hits) / static_cast<double>(neighbors_per_query.size() * k);
}

// small_scale_queries.cpp:15-23
void DedupNeighbors(std::vector<NNVec>& neighbors, size_t num_neighbors) {
    ParallelFor(0, neighbors.size(), 0, [&](size_t q) {
        NNVec& n = neighbors[q];
        std::sort(n.begin(), n.end(), [](const auto& l, const auto& r) { return l.second < r.second; });
        n.erase(std::unique(n.begin(), n.end()), n.end());
        std::sort(n.begin(), n.end());
        n.resize(std::min(n.size(), num_neighbors));
    });
}

// ================================================================================================== shard_searches
void ShardSearch::Init(size_t ef, int num_shards, size_t num_queries) {
    ef_search = ef;
    time_query_in_shard.assign(num_shards, std::vector<double>(num_queries, 0.0));
    neighbors.assign(num_shards, std::vector<std::vector<uint32_t>>(num_queries));
}

// shard_searches.cpp:11-104 with the exact GPU scan in place of per-shard HNSW indices.  Kept from the reference:
// every query is searched in every shard; per (shard, query) only hits with distance <= distance_to_kth_neighbor[q] are
// recorded, as global ids in descending-distance order (:78-89); the time of a (shard, query) pair is the median of 5
// timed repetitions of the whole batch in that shard divided by the number of queries (:57-75).
std::vector<ShardSearch> RunInShardSearches(PointSet& points, PointSet& queries, HNSWParameters /*hnsw_parameters*/, int num_neighbors,
                                            const Clusters& clusters, int num_shards, const std::vector<float>& distance_to_kth_neighbor) {
    Timer init_timer;
    init_timer.Start();
    std::vector<ShardSearch> searches(1);
    searches[0].Init(/*ef_search = exact*/ 0, num_shards, queries.n);
    std::cout << "Init search output took " << init_timer.Stop() << std::endl;

    const Csr csr = ToCsr(clusters);
    gpann_shards* shards = nullptr;
    Check(gpann_shards_create(points.coordinates.data(), points.n, (uint32_t)points.d, MetricNow(), csr.ids.data(), csr.offsets.data(),
                              (uint32_t)clusters.size(), gpann_host::Device(), 0, &shards),
          "gpann_shards_create");
    Check(gpann_shards_set_scan_mode(shards, gpann_host::FilteredScan() ? 1 : 0), "gpann_shards_set_scan_mode");
    const uint32_t nq = (uint32_t)queries.n, k = (uint32_t)num_neighbors, S = (uint32_t)num_shards;
    std::vector<uint32_t> hit_ids((size_t)S * nq * k), hit_count((size_t)S * nq);
    gpann_status st = gpann_shard_search_hits(shards, queries.coordinates.data(), nq, k, distance_to_kth_neighbor.data(), hit_ids.data(),
                                              hit_count.data(), 0, nullptr, nullptr);
    if (st != GPANN_OK) {
        gpann_shards_destroy(shards);
        Check(st, "gpann_shard_search_hits");
    }
    size_t total_hits = 0;
    for (uint32_t b = 0; b < S; ++b)
        for (uint32_t q = 0; q < nq; ++q) {
            const uint32_t c = hit_count[(size_t)b * nq + q];
            const uint32_t* src = &hit_ids[((size_t)b * nq + q) * k];
            searches[0].neighbors[b][q].assign(src, src + c);
            total_hits += c;
        }

    // per-shard batch time: all queries against shard b only, 5 repetitions, median (device time of the scan kernels)
    std::vector<int32_t> probe_b(nq);
    std::vector<uint32_t> scratch_ids((size_t)nq * k);
    for (uint32_t b = 0; b < S; ++b) {
        std::fill(probe_b.begin(), probe_b.end(), (int32_t)b);
        std::vector<double> reps;
        for (int r = 0; r < 5; ++r) {
            float ms = 0.f;
            st = gpann_scan_buckets(shards, queries.coordinates.data(), nq, k, probe_b.data(), 1, 1, scratch_ids.data(), nullptr, nullptr, 0, nullptr, &ms);
            if (st != GPANN_OK) {
                gpann_shards_destroy(shards);
                Check(st, "gpann_scan_buckets");
            }
            reps.push_back(ms * 1e-3);
        }
        std::sort(reps.begin(), reps.end());
        const double elapsed = reps[reps.size() / 2];
        for (uint32_t q = 0; q < nq; ++q) searches[0].time_query_in_shard[b][q] = elapsed / nq;
        std::cout << "Shard search (exact scan) in shard " << b << " of size " << clusters[b].size() << " median time " << elapsed << std::endl;
    }
    std::cout << "Exact shard searches: total hits " << total_hits << std::endl;
    gpann_shards_destroy(shards);
    return searches;
}

// shard_searches.cpp:108-183 text format
std::string ShardSearch::Serialize() const {
    std::stringstream out;
    const size_t S = neighbors.size(), Q = neighbors[0].size();
    out << ef_search << " " << S << " " << Q << "\n";
    for (size_t b = 0; b < S; ++b)
        for (size_t q = 0; q < Q; ++q) {
            for (uint32_t id : neighbors[b][q]) out << id << " ";
            out << "\n";
        }
    for (const auto& row : time_query_in_shard) {
        for (double t : row) out << t << " ";
        out << "\n";
    }
    return out.str();
}

ShardSearch ShardSearch::Deserialize(std::ifstream& in) {
    ShardSearch s;
    std::string line;
    std::getline(in, line);
    int S = 0, Q = 0;
    {
        std::istringstream head(line);
        head >> s.ef_search >> S >> Q;
    }
    s.neighbors.assign(S, std::vector<std::vector<uint32_t>>(Q));
    for (int b = 0; b < S; ++b)
        for (int q = 0; q < Q; ++q) {
            std::getline(in, line);
            std::istringstream tokens(line);
            for (int id; tokens >> id;) s.neighbors[b][q].push_back(id);
        }
    s.time_query_in_shard.assign(S, {});
    for (int b = 0; b < S; ++b) {
        std::getline(in, line);
        std::istringstream tokens(line);
        for (double t; tokens >> t;) s.time_query_in_shard[b].push_back(t);
    }
    return s;
}

void SerializeShardSearches(const std::vector<ShardSearch>& shard_searches, const std::string& output_file) {
    std::ofstream out(output_file);
    out << shard_searches.size() << std::endl;
    for (const auto& s : shard_searches) {
        out << "S" << std::endl;
        out << s.Serialize();
    }
}

std::vector<ShardSearch> DeserializeShardSearches(const std::string& input_file) {
    std::ifstream in(input_file);
    std::string line;
    std::getline(in, line);
    size_t count = 0;
    std::istringstream(line) >> count;
    std::vector<ShardSearch> all;
    for (size_t i = 0; i < count; ++i) {
        std::getline(in, line);
        if (line != "S") std::cout << "search config doesn't start with marker S. Instead: " << line << std::endl;
        all.push_back(ShardSearch::Deserialize(in));
    }
    return all;
}
