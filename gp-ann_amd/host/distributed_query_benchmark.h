// distributed_query_benchmark.h — the multi-GPU form of the reference's DistributedQueryBenchmark (src/distributed_query_benchmark.h:13-208),
// same member names and call order (LoadPartition, LoadShardPointSet, BuildInShardIndex, LoadRouter, ProcessQueries), one process per GPU.
//
// The reference sketches an MPI program with ONE shard per rank that routes its slice of the queries, posts every (query, shard) pair to
// the shard's owner through a message queue and gets the neighbours back (:110-208; it does not compile: hnswlib / message-queue /
// MPI are absent and the body refers to undeclared names).  Here, MI355X first:
//   * a rank owns a SET of shards (LPT by rows, gpann_lpt_owner) and keeps only their rows in HBM (gpann_shards_create);
//   * in-shard search = the exact scan of libgpann_hip.so (instead of one HNSW per rank, :90-97);
//   * the router is the KMeansTreeRouter, replicated: every rank trains the subtrees of its own shards and the parts are merged
//     (gpann_comm_merge_router), or a tree file written by an earlier run is loaded (LoadRouter, :99-101);
//   * ProcessQueries is batch-synchronous with fixed shapes — route my slice, ALL-GATHER the probe lists, scan the slots whose shard
//     I own, ALL-GATHER the per-slot lists and select by owner (gpann_comm_gather_slots), merge my slice on the host
//     (gpann_merge_dedup = DedupNeighbors) — three collectives per batch instead of one message per (query, shard).
// Everything below the class boundary is the C-ABI (include/gpann.h, include/gpann_comm.h).
#pragma once

#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <iostream>
#include <memory>
#include <numeric>

#include "../../include/gpann.h"
#include "../../include/gpann_comm.h"
#include "gpann_host.h"

class DistributedQueryBenchmark {
public:
    int rank = 0;
    int comm_size = 1;

    int num_shards = 0;
    int dim = 0;
    int num_neighbors = 10;
    int num_probes = 2;              // distributed_query_benchmark.h:35
    KMeansTreeRouterOptions router_options;

    PointSet shard_points;           // the rows of the shards this rank owns, in point-id order (dropped once they are in HBM)
    std::vector<int> partition;
    std::vector<int32_t> owner;      // shard -> rank
    std::vector<uint32_t> local_to_global;   // local row of shard_points -> point id
    size_t num_points = 0;

    gpann_comm* comm = nullptr;
    gpann_shards* store = nullptr;
    gpann_router* router = nullptr;
    int device = 0;

    // results of the most recent ProcessQueries: for my query slice [slice_begin, slice_end), k merged neighbours each
    size_t slice_begin = 0, slice_end = 0;
    std::vector<float> result_dist;
    std::vector<uint32_t> result_ids;
    double last_route_ms = 0.0, last_scan_ms = 0.0;
    int wait_timeout_s = 600;   // a collective that has not completed by then is given up (gpann_comm_wait; GPANN_COMM_WAIT_S)

    explicit DistributedQueryBenchmark(gpann_comm* c, int device_) : comm(c), device(device_) {
        Check(gpann_comm_info(comm, &rank, &comm_size, &backend_), "gpann_comm_info");
    }
    ~DistributedQueryBenchmark() {
        if (router) gpann_router_destroy(router);
        if (store) gpann_shards_destroy(store);
        for (void* p : {(void*)d_queries_, (void*)d_probes_, (void*)d_all_probes_, (void*)d_ids_, (void*)d_dist_, (void*)d_out_ids_, (void*)d_out_dist_, (void*)d_scratch_,
                        (void*)d_map_, (void*)d_probes2_[0], (void*)d_probes2_[1], (void*)d_all_probes2_[0], (void*)d_all_probes2_[1], (void*)d_ids2_[0], (void*)d_ids2_[1],
                        (void*)d_dist2_[0], (void*)d_dist2_[1]})
            if (p) (void)hipFree(p);
        for (hipStream_t st : {s_route_, s_scan_, s_x_})
            if (st) (void)hipStreamDestroy(st);
    }

    void LoadPartition(const std::string& partition_file) {   // :43-46, + the shard -> rank placement
        partition = ReadMetisPartition(partition_file);
        num_shards = NumPartsInPartition(partition);
        num_points = partition.size();
        std::vector<uint64_t> rows(num_shards, 0);
        for (int p : partition) rows[p]++;
        owner.resize(num_shards);
        std::vector<uint64_t> load(comm_size);
        CheckComm(gpann_lpt_owner(rows.data(), (uint32_t)num_shards, (uint32_t)comm_size, owner.data(), load.data()), "gpann_lpt_owner");
        if (rank == 0) {
            std::cout << "num shards = " << num_shards << ", rows per rank:";
            for (uint64_t l : load) std::cout << " " << l;
            std::cout << std::endl;
        }
    }

    void LoadShardPointSet(const std::string& point_set_file) {   // :48-78: only the rows of my shards are read, run by run
        local_to_global.clear();
        for (size_t i = 0; i < partition.size(); ++i)
            if (owner[partition[i]] == rank) local_to_global.push_back((uint32_t)i);
        uint32_t n = 0, d = 0;
        std::ifstream in(point_set_file, std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + point_set_file);
        in.read(reinterpret_cast<char*>(&n), sizeof(uint32_t));
        in.read(reinterpret_cast<char*>(&d), sizeof(uint32_t));
        if (n != partition.size()) throw std::runtime_error("point file and partition disagree on the number of points");
        shard_points.n = local_to_global.size();
        shard_points.d = d;
        shard_points.coordinates.resize(shard_points.n * shard_points.d);
        dim = (int)d;
        size_t coords_end = 0;
        for (size_t i = 0; i < local_to_global.size();) {
            size_t j = i;
            while (j + 1 < local_to_global.size() && local_to_global[j + 1] == local_to_global[j] + 1) ++j;
            const size_t run = j - i + 1;
            in.seekg(8 + (std::streamoff)local_to_global[i] * d * sizeof(float));
            in.read(reinterpret_cast<char*>(&shard_points.coordinates[coords_end]), (std::streamsize)(run * d * sizeof(float)));
            coords_end += run * d;
            i = j + 1;
        }
    }

    void BuildInShardIndex() {   // :80-97 — the shard store in HBM instead of an HNSW; shards of other ranks are empty here
        std::vector<uint64_t> offsets(num_shards + 1, 0);
        for (uint32_t g : local_to_global) offsets[partition[g] + 1]++;
        for (int b = 0; b < num_shards; ++b) offsets[b + 1] += offsets[b];
        std::vector<uint32_t> ids(local_to_global.size());
        std::vector<uint64_t> at(offsets.begin(), offsets.end() - 1);
        for (uint32_t l = 0; l < local_to_global.size(); ++l) ids[at[partition[local_to_global[l]]]++] = l;   // local rows, ascending inside a shard
        Check(gpann_shards_create(shard_points.coordinates.data(), shard_points.n, (uint32_t)dim, Metric(), ids.data(), offsets.data(), (uint32_t)num_shards,
                                  device, 0, &store),
              "gpann_shards_create");
        Check(gpann_shards_set_scan_mode(store, gpann_host::FilteredScan() ? 1 : 0), "gpann_shards_set_scan_mode");
        shard_points.Drop();
        HipCheck(hipSetDevice(device), "hipSetDevice");
        HipCheck(hipMalloc(&d_map_, std::max<size_t>(local_to_global.size(), 1) * 4), "hipMalloc");
        HipCheck(hipMemcpy(d_map_, local_to_global.data(), local_to_global.size() * 4, hipMemcpyHostToDevice), "hipMemcpy");
    }

    // :99-101.  The tree file is this build's own format (the reference's router file is an hnswlib index).  Missing file: the router
    // is trained — every rank the subtrees of its shards — and rank 0 writes the file.
    void LoadRouter(const std::string& router_file) {
        KMeansTreeRouter::Flat flat;
        int have = 0;
        if (rank == 0) have = ReadTree(router_file, flat) ? 1 : 0;
        CheckComm(gpann_comm_broadcast_host(comm, &have, sizeof have, 0), "broadcast");
        if (have) {
            BroadcastTree(flat);
            Check(gpann_router_create(flat.d, Metric(), flat.S, (uint32_t)flat.node_shard.size(), flat.node_shard.data(), flat.node_cent_first.data(),
                                      flat.node_child_first.data(), flat.node_child_count.data(), flat.shard_root.data(), flat.centroids.data(), device, &router),
                  "gpann_router_create");
            return;
        }
        gpann_router* part = nullptr;
        Check(gpann_router_train(store, num_points, (uint32_t)router_options.num_centroids, (uint32_t)router_options.min_cluster_size, router_options.budget, &part),
              "gpann_router_train");
        const gpann_status st = gpann_comm_merge_router(comm, part, owner.data(), device, &router);
        gpann_router_destroy(part);
        CheckComm(st, "gpann_comm_merge_router");
        if (rank == 0 && !router_file.empty()) WriteTree(router_file);
    }

    // rank 0 holds the query file's contents; the block is broadcast into every rank's HBM (distributed_bench.cpp:39 lets every rank read it)
    void BroadcastQueries(PointSet& queries) {
        uint64_t shape[2] = {queries.n, queries.d};
        CheckComm(gpann_comm_broadcast_host(comm, shape, sizeof shape, 0), "broadcast");
        queries.n = shape[0];
        queries.d = shape[1];
        queries.coordinates.resize(queries.n * queries.d);
        const size_t bytes = queries.coordinates.size() * 4;
        HipCheck(hipSetDevice(device), "hipSetDevice");
        if (d_queries_) (void)hipFree(d_queries_);
        HipCheck(hipMalloc(&d_queries_, std::max<size_t>(bytes, 4)), "hipMalloc");
        if (backend_ == GPANN_COMM_RCCL) {
            if (rank == 0) HipCheck(hipMemcpy(d_queries_, queries.coordinates.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy");
            CheckComm(gpann_comm_broadcast(comm, d_queries_, bytes, 0, nullptr), "gpann_comm_broadcast");
            CheckComm(gpann_comm_wait(comm, nullptr, wait_timeout_s), "gpann_comm_wait");   // (not a bare synchronise: it returns when a rank has left)
            if (rank != 0) HipCheck(hipMemcpy(queries.coordinates.data(), d_queries_, bytes, hipMemcpyDeviceToHost), "hipMemcpy");
        } else {
            CheckComm(gpann_comm_broadcast(comm, queries.coordinates.data(), bytes, 0, nullptr), "gpann_comm_broadcast");
            HipCheck(hipMemcpy(d_queries_, queries.coordinates.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy");
        }
        nq_ = queries.n;
    }

    // :104-108 batched: probe lists of queries [begin, end) of the broadcast block, first num_probes shards each
    std::vector<int32_t> Route(size_t begin, size_t end) {
        const size_t cnt = end - begin;
        std::vector<int32_t> probes(cnt * num_shards), head(cnt * num_probes);
        float ms = 0.f;
        if (cnt)
            Check(gpann_router_query_batch(router, d_queries_ + begin * dim, (uint32_t)cnt, (int32_t)router_options.search_budget, probes.data(), nullptr,
                                           GPANN_QUERIES_ON_DEVICE, nullptr, &ms),
                  "gpann_router_query_batch");
        last_route_ms = ms;
        for (size_t q = 0; q < cnt; ++q) std::memcpy(&head[q * num_probes], &probes[q * num_shards], (size_t)num_probes * 4);
        return head;
    }

    // :110-208.  query_ids: the contiguous slice of the broadcast query block this rank routes and whose results it keeps.
    void ProcessQueries(const std::vector<int>& query_ids, PointSet& queries) {
        const size_t nq = nq_, k = (size_t)num_neighbors, P = (size_t)num_probes;
        if (queries.n != nq) throw std::runtime_error("ProcessQueries: BroadcastQueries first");
        slice_begin = query_ids.empty() ? 0 : (size_t)query_ids.front();
        slice_end = query_ids.empty() ? 0 : (size_t)query_ids.back() + 1;
        const size_t chunk = (nq + comm_size - 1) / comm_size;   // every rank contributes `chunk` probe rows (-1 padded)
        EnsureBuffers(nq, chunk);
        // 1. route my slice
        std::vector<int32_t> mine(chunk * P, -1);
        const std::vector<int32_t> head = Route(slice_begin, slice_end);
        std::copy(head.begin(), head.end(), mine.begin());
        // 2. all-gather of the probe lists
        std::vector<int32_t> all(chunk * P * comm_size);
        if (backend_ == GPANN_COMM_RCCL) {
            HipCheck(hipMemcpyAsync(d_probes_, mine.data(), mine.size() * 4, hipMemcpyHostToDevice, nullptr), "hipMemcpyAsync");
            CheckComm(gpann_comm_allgather(comm, d_probes_, d_all_probes_, mine.size() * 4, nullptr), "gpann_comm_allgather");
            CheckComm(gpann_comm_wait(comm, nullptr, wait_timeout_s), "gpann_comm_wait");
            HipCheck(hipMemcpy(all.data(), d_all_probes_, all.size() * 4, hipMemcpyDeviceToHost), "hipMemcpy");
        } else {
            CheckComm(gpann_comm_allgather(comm, mine.data(), all.data(), mine.size() * 4, nullptr), "gpann_comm_allgather");
        }
        all.resize(nq * P);   // rank r's rows are queries [r * chunk, ...): the gathered block is the probe matrix of the whole batch
        // 3. the slots whose shard I own, scanned on my GPU
        std::vector<int32_t> masked(nq * P);
        CheckComm(gpann_mask_probes(all.data(), (uint32_t)nq, (uint32_t)P, (uint32_t)P, owner.data(), (uint32_t)num_shards, rank, masked.data()), "gpann_mask_probes");
        float ms = 0.f;
        Check(gpann_scan_buckets(store, d_queries_, (uint32_t)nq, (uint32_t)k, masked.data(), (uint32_t)P, (uint32_t)P, d_ids_, d_dist_, nullptr,
                                 // (the slots are only merged, :238: lists cut at this rank's merge bound — same merged neighbours, far shards cost little)
                                 GPANN_QUERIES_ON_DEVICE | GPANN_OUTPUT_ON_DEVICE | GPANN_SCAN_FOR_MERGE, nullptr, &ms),
              "gpann_scan_buckets");
        last_scan_ms = ms;
        // 4. result exchange: local rows -> point ids, all-gather of the per-slot lists, owner selection
        const size_t n = nq * P * k;
        std::vector<float> dist(n);
        std::vector<uint32_t> ids(n);
        if (backend_ == GPANN_COMM_RCCL) {
            CheckComm(gpann_comm_gather_slots(comm, d_dist_, d_ids_, n, d_out_dist_, d_out_ids_, d_scratch_, d_map_, nullptr), "gpann_comm_gather_slots");
            CheckComm(gpann_comm_wait(comm, nullptr, wait_timeout_s), "gpann_comm_wait");
            HipCheck(hipMemcpyAsync(dist.data(), d_out_dist_, n * 4, hipMemcpyDeviceToHost, nullptr), "hipMemcpyAsync");
            HipCheck(hipMemcpy(ids.data(), d_out_ids_, n * 4, hipMemcpyDeviceToHost), "hipMemcpy");
        } else {
            std::vector<float> my_dist(n);
            std::vector<uint32_t> my_ids(n);
            HipCheck(hipMemcpy(my_dist.data(), d_dist_, n * 4, hipMemcpyDeviceToHost), "hipMemcpy");
            HipCheck(hipMemcpy(my_ids.data(), d_ids_, n * 4, hipMemcpyDeviceToHost), "hipMemcpy");
            host_scratch_.resize(gpann_comm_gather_slots_scratch(comm_size, n));
            CheckComm(gpann_comm_gather_slots(comm, my_dist.data(), my_ids.data(), n, dist.data(), ids.data(), host_scratch_.data(), local_to_global.data(), nullptr),
                      "gpann_comm_gather_slots");
        }
        // 5. host merge of my slice (DedupNeighbors, small_scale_queries.cpp:15-23)
        const size_t cnt = slice_end - slice_begin;
        result_dist.assign(cnt * k, 0.f);
        result_ids.assign(cnt * k, 0u);
        if (cnt)
            Check(gpann_merge_dedup(dist.data() + slice_begin * P * k, ids.data() + slice_begin * P * k, (uint32_t)cnt, (uint32_t)P, (uint32_t)P, (uint32_t)k,
                                    result_dist.data(), result_ids.data(), nullptr),
                  "gpann_merge_dedup");
    }

    static void Check(gpann_status s, const char* what) {
        if (s != GPANN_OK) throw std::runtime_error(std::string(what) + " failed: " + gpann_last_error());
    }
    static void CheckComm(gpann_status s, const char* what) {
        if (s != GPANN_OK) throw std::runtime_error(std::string(what) + " failed: " + gpann_comm_last_error());
    }
    static void HipCheck(hipError_t e, const char* what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + " failed: " + hipGetErrorString(e));
    }

    // A STREAM of `batches` batches (each the broadcast query block once more) as a three-stage pipeline — bench.py's N > 1 loop in the driver's
    // own language; ProcessQueries above is one batch at a time, as the reference times it (distributed_bench.cpp:41-54):
    //   route thread   routing of batch i + 2 on its own stream + the all-gather of its probe lists + the wait, through `comm_route` — a SECOND
    //                  communicator (collectives of one communicator must be issued by one thread in one order)
    //   this thread    the scan of batch i + 1 planned and enqueued behind batch i as a DEFERRED call (GPANN_SCAN_DEFER) into the other set of
    //                  device arrays; batch i completed (gpann_scan_finish: an overflowed call is redone BEFORE its lists are exchanged), its
    //                  lists exchanged on a third stream beside the scan of batch i + 1
    //   merge thread   DedupNeighbors of batch i - 1
    // Leaves the last batch's merged neighbours in result_dist / result_ids.
    //
    // Two communicators on one GPU: each is driven by one thread in one order, but their collectives must also be LAUNCHED in the same relative
    // order on every rank — kernels of two communicators that the GPU cannot co-schedule would otherwise wait for each other across ranks for
    // ever (RCCL / NCCL: concurrent communicators need a consistent launch order).  `LaunchOrder` is that order: R j = the probe all-gather of
    // batch j (route thread), X i = the slot exchange of batch i (this thread); sequence R0, R1, then per batch i: R i+2, X i.  A collective
    // takes its place when every earlier one has been launched (not completed: the schedule costs no overlap).  If any stage throws, the
    // schedule and the routing communicator are given up (`Abort`, gpann_comm_abort) BEFORE the futures are destroyed, so a route job parked in
    // the schedule or in an unmatched collective fails at once instead of keeping this rank alive until the communicator's timeout.
    struct LaunchOrder {
        std::mutex mu;
        std::condition_variable cv;
        size_t turn = 0;
        bool dead = false;
        int batches = 0;
        size_t PosR(int j) const { return j < 2 ? (size_t)j : (size_t)(2 * j - 2); }                     // R0 R1 | R2 X0 | R3 X1 | ...
        size_t PosX(int i) const {
            const int routed = std::min(batches, i + 3);   // route collectives launched before X i: R0 .. R(i + 2), as far as they exist
            return (size_t)(routed + i);
        }
        void Enter(size_t pos) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return dead || turn == pos; });
            if (dead) throw std::runtime_error("ProcessQueryStream was given up: another stage of this rank failed");
        }
        void Leave() {
            { std::lock_guard<std::mutex> lk(mu); ++turn; }
            cv.notify_all();
        }
        void Abort() {
            { std::lock_guard<std::mutex> lk(mu); dead = true; }
            cv.notify_all();
        }
    };
    struct InTurn {   // a collective's place in the schedule: taken in the constructor, handed on in the destructor (also when the launch throws)
        LaunchOrder& o;
        InTurn(LaunchOrder& order, size_t pos) : o(order) { o.Enter(pos); }
        ~InTurn() { o.Leave(); }
    };

    void ProcessQueryStream(const std::vector<int>& query_ids, PointSet& queries, int batches, gpann_comm* comm_route) {
        const size_t nq = nq_, k = (size_t)num_neighbors, P = (size_t)num_probes;
        LaunchOrder order;
        order.batches = batches;
        if (queries.n != nq) throw std::runtime_error("ProcessQueryStream: BroadcastQueries first");
        if (batches <= 0) return;
        slice_begin = query_ids.empty() ? 0 : (size_t)query_ids.front();
        slice_end = query_ids.empty() ? 0 : (size_t)query_ids.back() + 1;
        const size_t chunk = (nq + comm_size - 1) / comm_size, cnt = slice_end - slice_begin, n = nq * P * k;
        EnsureBuffers(nq, chunk);
        EnsureStreamBuffers(n, chunk * P);
        auto route_job = [&, this](int j) -> std::vector<int32_t> {   // the probe matrix of batch j (all ranks' slices)
            HipCheck(hipSetDevice(device), "hipSetDevice");
            std::vector<int32_t> mine(chunk * P, -1), probes(cnt * num_shards), all(chunk * P * comm_size);
            float ms = 0.f;
            if (cnt)
                Check(gpann_router_query_batch(router, d_queries_ + slice_begin * dim, (uint32_t)cnt, (int32_t)router_options.search_budget, probes.data(), nullptr,
                                               GPANN_QUERIES_ON_DEVICE, s_route_, &ms),
                      "gpann_router_query_batch");
            for (size_t q = 0; q < cnt; ++q) std::memcpy(&mine[q * P], &probes[q * num_shards], P * 4);
            if (backend_ == GPANN_COMM_RCCL) {
                int32_t *dp = d_probes2_[j & 1], *da = d_all_probes2_[j & 1];
                HipCheck(hipMemcpyAsync(dp, mine.data(), mine.size() * 4, hipMemcpyHostToDevice, s_route_), "hipMemcpyAsync");
                {
                    InTurn turn(order, order.PosR(j));
                    CheckComm(gpann_comm_allgather(comm_route, dp, da, mine.size() * 4, s_route_), "gpann_comm_allgather");
                }
                CheckComm(gpann_comm_wait(comm_route, s_route_, wait_timeout_s), "gpann_comm_wait");
                HipCheck(hipMemcpyAsync(all.data(), da, all.size() * 4, hipMemcpyDeviceToHost, s_route_), "hipMemcpyAsync");
                HipCheck(hipStreamSynchronize(s_route_), "hipStreamSynchronize");
            } else {
                InTurn turn(order, order.PosR(j));
                CheckComm(gpann_comm_allgather(comm_route, mine.data(), all.data(), mine.size() * 4, nullptr), "gpann_comm_allgather");
            }
            all.resize(nq * P);
            return all;
        };
        auto scan_begin = [&, this](const std::vector<int32_t>& all, int set) {
            std::vector<int32_t> masked(nq * P);
            CheckComm(gpann_mask_probes(all.data(), (uint32_t)nq, (uint32_t)P, (uint32_t)P, owner.data(), (uint32_t)num_shards, rank, masked.data()), "gpann_mask_probes");
            float ms = 0.f;
            Check(gpann_scan_buckets(store, d_queries_, (uint32_t)nq, (uint32_t)k, masked.data(), (uint32_t)P, (uint32_t)P, d_ids2_[set], d_dist2_[set], nullptr,
                                     GPANN_QUERIES_ON_DEVICE | GPANN_OUTPUT_ON_DEVICE | GPANN_SCAN_FOR_MERGE | GPANN_SCAN_DEFER | overlap_flag_, s_scan_, &ms),
                  "gpann_scan_buckets");
        };
        std::vector<float> h_dist[2];
        std::vector<uint32_t> h_ids[2];
        for (int b = 0; b < 2; ++b) {
            h_dist[b].resize(n);
            h_ids[b].resize(n);
        }
        auto finish_and_exchange = [&, this](int it) {   // batch `it` -> h_dist[it & 1], h_ids[it & 1]: every slot's list from its owner
            const int set = it & 1;
            float ms = 0.f;
            Check(gpann_scan_finish(store, &ms), "gpann_scan_finish");
            last_scan_ms = ms;
            if (backend_ == GPANN_COMM_RCCL) {
                {
                    InTurn turn(order, order.PosX(it));
                    CheckComm(gpann_comm_gather_slots(comm, d_dist2_[set], d_ids2_[set], n, d_out_dist_, d_out_ids_, d_scratch_, d_map_, s_x_),
                              "gpann_comm_gather_slots");
                }
                CheckComm(gpann_comm_wait(comm, s_x_, wait_timeout_s), "gpann_comm_wait");
                HipCheck(hipMemcpyAsync(h_dist[set].data(), d_out_dist_, n * 4, hipMemcpyDeviceToHost, s_x_), "hipMemcpyAsync");
                HipCheck(hipMemcpyAsync(h_ids[set].data(), d_out_ids_, n * 4, hipMemcpyDeviceToHost, s_x_), "hipMemcpyAsync");
                HipCheck(hipStreamSynchronize(s_x_), "hipStreamSynchronize");
            } else {
                std::vector<float> my_dist(n);
                std::vector<uint32_t> my_ids(n);
                HipCheck(hipMemcpyAsync(my_dist.data(), d_dist2_[set], n * 4, hipMemcpyDeviceToHost, s_x_), "hipMemcpyAsync");
                HipCheck(hipMemcpyAsync(my_ids.data(), d_ids2_[set], n * 4, hipMemcpyDeviceToHost, s_x_), "hipMemcpyAsync");
                HipCheck(hipStreamSynchronize(s_x_), "hipStreamSynchronize");
                host_scratch_.resize(gpann_comm_gather_slots_scratch(comm_size, n));
                InTurn turn(order, order.PosX(it));
                CheckComm(gpann_comm_gather_slots(comm, my_dist.data(), my_ids.data(), n, h_dist[set].data(), h_ids[set].data(), host_scratch_.data(), local_to_global.data(),
                                                  nullptr),
                          "gpann_comm_gather_slots");
            }
        };
        std::vector<float> m_dist(cnt * k);
        std::vector<uint32_t> m_ids(cnt * k);
        auto merge_job = [&, this](int set) {
            if (cnt)
                Check(gpann_merge_dedup(h_dist[set].data() + slice_begin * P * k, h_ids[set].data() + slice_begin * P * k, (uint32_t)cnt, (uint32_t)P, (uint32_t)P, (uint32_t)k,
                                        m_dist.data(), m_ids.data(), nullptr),
                      "gpann_merge_dedup");
        };

        // (declared before the try block: when a stage throws, the catch gives the schedule and the routing communicator up FIRST and only then
        // are these futures destroyed — an std::async future joins its thread in the destructor)
        std::future<std::vector<int32_t>> route_next, route_cur;
        std::future<void> merging;
        try {
            route_cur = std::async(std::launch::async, route_job, 0);
            const std::vector<int32_t> all0 = route_cur.get();
            if (batches > 1) route_next = std::async(std::launch::async, route_job, 1);   // (one route job at a time: they share the router handle and its stream)
            scan_begin(all0, 0);
            for (int it = 0; it < batches; ++it) {
                if (it + 1 < batches) {
                    const std::vector<int32_t> all = route_next.get();
                    if (it + 2 < batches) route_next = std::async(std::launch::async, route_job, it + 2);
                    scan_begin(all, (it + 1) & 1);   // (queued behind batch `it`, which the GPU is still working on)
                }
                if (merging.valid()) merging.get();   // (batch it - 1: its host lists are the set batch it + 1 will fill)
                finish_and_exchange(it);
                merging = std::async(std::launch::async, merge_job, it & 1);
            }
            merging.get();
        } catch (...) {
            order.Abort();
            if (comm_route) (void)gpann_comm_abort(comm_route);
            throw;
        }
        result_dist = m_dist;
        result_ids = m_ids;
    }

private:
    // GPANN_DIST_OVERLAP=0: the deferred scans on ONE stream (round 4); default: GPANN_SCAN_OVERLAP — the front of batch i + 1 beside the rest pass of
    // batch i on the store's own streams (results are in place when gpann_scan_finish returns: the exchange below only starts after it)
    int overlap_flag_ = (std::getenv("GPANN_DIST_OVERLAP") && std::getenv("GPANN_DIST_OVERLAP")[0] == '0') ? 0 : GPANN_SCAN_OVERLAP;
    static int Metric() { return gpann_host::MetricIsMIPS() ? GPANN_METRIC_MIPS : GPANN_METRIC_L2; }

    void EnsureStreamBuffers(size_t n, size_t pw) {
        if (s_route_ && n == sbuf_n_ && pw == sbuf_pw_) return;
        HipCheck(hipSetDevice(device), "hipSetDevice");
        if (!s_route_) {
            HipCheck(hipStreamCreateWithFlags(&s_route_, hipStreamNonBlocking), "hipStreamCreate");
            HipCheck(hipStreamCreateWithFlags(&s_scan_, hipStreamNonBlocking), "hipStreamCreate");
            HipCheck(hipStreamCreateWithFlags(&s_x_, hipStreamNonBlocking), "hipStreamCreate");
        }
        for (int b = 0; b < 2; ++b) {
            for (void** p : {(void**)&d_probes2_[b], (void**)&d_all_probes2_[b], (void**)&d_ids2_[b], (void**)&d_dist2_[b]}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            HipCheck(hipMalloc(&d_probes2_[b], std::max<size_t>(pw, 1) * 4), "hipMalloc");
            HipCheck(hipMalloc(&d_all_probes2_[b], std::max<size_t>(pw * comm_size, 1) * 4), "hipMalloc");
            HipCheck(hipMalloc(&d_ids2_[b], std::max<size_t>(n, 1) * 4), "hipMalloc");
            HipCheck(hipMalloc(&d_dist2_[b], std::max<size_t>(n, 1) * 4), "hipMalloc");
        }
        sbuf_n_ = n;
        sbuf_pw_ = pw;
    }

    void EnsureBuffers(size_t nq, size_t chunk) {
        const size_t n = nq * num_probes * num_neighbors, pw = chunk * num_probes;
        if (n == buf_n_ && pw == buf_pw_) return;
        HipCheck(hipSetDevice(device), "hipSetDevice");
        for (void** p : {(void**)&d_probes_, (void**)&d_all_probes_, (void**)&d_ids_, (void**)&d_dist_, (void**)&d_out_ids_, (void**)&d_out_dist_, (void**)&d_scratch_}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        HipCheck(hipMalloc(&d_probes_, std::max<size_t>(pw, 1) * 4), "hipMalloc");
        HipCheck(hipMalloc(&d_all_probes_, std::max<size_t>(pw * comm_size, 1) * 4), "hipMalloc");
        HipCheck(hipMalloc(&d_ids_, std::max<size_t>(n, 1) * 4), "hipMalloc");
        HipCheck(hipMalloc(&d_dist_, std::max<size_t>(n, 1) * 4), "hipMalloc");
        HipCheck(hipMalloc(&d_out_ids_, std::max<size_t>(n, 1) * 4), "hipMalloc");
        HipCheck(hipMalloc(&d_out_dist_, std::max<size_t>(n, 1) * 4), "hipMalloc");
        HipCheck(hipMalloc(&d_scratch_, std::max<size_t>(gpann_comm_gather_slots_scratch(comm_size, n), 4)), "hipMalloc");
        buf_n_ = n;
        buf_pw_ = pw;
    }

    // tree file: "GPKMTR1\0", u32 d, S, n_nodes, metric, u64 n_centroids, then the flat arrays (gpann_router_create order)
    bool ReadTree(const std::string& path, KMeansTreeRouter::Flat& f) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        char magic[8] = {0};
        uint32_t head[4] = {0, 0, 0, 0};
        uint64_t nc = 0;
        in.read(magic, 8);
        in.read(reinterpret_cast<char*>(head), sizeof head);
        in.read(reinterpret_cast<char*>(&nc), 8);
        if (!in || std::memcmp(magic, "GPKMTR1", 8) != 0) throw std::runtime_error(path + " is not a router tree file of this build");
        f.d = head[0];
        f.S = head[1];
        f.mips = head[3] == GPANN_METRIC_MIPS;
        const uint32_t nn = head[2];
        f.node_shard.resize(nn);
        f.node_cent_first.resize(nn + 1);
        f.node_child_first.resize(nn);
        f.node_child_count.resize(nn);
        f.shard_root.resize(f.S);
        f.centroids.resize(nc * f.d);
        in.read(reinterpret_cast<char*>(f.node_shard.data()), nn * 4);
        in.read(reinterpret_cast<char*>(f.node_cent_first.data()), (nn + 1) * 4);
        in.read(reinterpret_cast<char*>(f.node_child_first.data()), nn * 4);
        in.read(reinterpret_cast<char*>(f.node_child_count.data()), nn * 4);
        in.read(reinterpret_cast<char*>(f.shard_root.data()), f.S * 4);
        in.read(reinterpret_cast<char*>(f.centroids.data()), (std::streamsize)(f.centroids.size() * 4));
        if (!in) throw std::runtime_error(path + " is truncated");
        return true;
    }

    void WriteTree(const std::string& path) {
        uint32_t d = 0, S = 0, nn = 0;
        uint64_t nc = 0;
        int metric = 0;
        Check(gpann_router_info(router, &d, &metric, &S, &nn, &nc), "gpann_router_info");
        std::vector<int32_t> node_shard(nn);
        std::vector<uint32_t> cent_first(nn + 1), child_first(nn), child_count(nn), shard_root(S);
        std::vector<float> cents(nc * d);
        Check(gpann_router_export(router, node_shard.data(), cent_first.data(), child_first.data(), child_count.data(), shard_root.data(), cents.data()),
              "gpann_router_export");
        std::ofstream out(path, std::ios::binary);
        const uint32_t head[4] = {d, S, nn, (uint32_t)metric};
        out.write("GPKMTR1", 8);
        out.write(reinterpret_cast<const char*>(head), sizeof head);
        out.write(reinterpret_cast<const char*>(&nc), 8);
        out.write(reinterpret_cast<const char*>(node_shard.data()), nn * 4);
        out.write(reinterpret_cast<const char*>(cent_first.data()), (nn + 1) * 4);
        out.write(reinterpret_cast<const char*>(child_first.data()), nn * 4);
        out.write(reinterpret_cast<const char*>(child_count.data()), nn * 4);
        out.write(reinterpret_cast<const char*>(shard_root.data()), S * 4);
        out.write(reinterpret_cast<const char*>(cents.data()), (std::streamsize)(cents.size() * 4));
    }

    void BroadcastTree(KMeansTreeRouter::Flat& f) {
        uint64_t head[4] = {f.d, f.S, f.node_shard.size(), f.centroids.size()};
        CheckComm(gpann_comm_broadcast_host(comm, head, sizeof head, 0), "broadcast");
        f.d = (uint32_t)head[0];
        f.S = (uint32_t)head[1];
        f.node_shard.resize(head[2]);
        f.node_cent_first.resize(head[2] + 1);
        f.node_child_first.resize(head[2]);
        f.node_child_count.resize(head[2]);
        f.shard_root.resize(f.S);
        f.centroids.resize(head[3]);
        CheckComm(gpann_comm_broadcast_host(comm, f.node_shard.data(), f.node_shard.size() * 4, 0), "broadcast");
        CheckComm(gpann_comm_broadcast_host(comm, f.node_cent_first.data(), f.node_cent_first.size() * 4, 0), "broadcast");
        CheckComm(gpann_comm_broadcast_host(comm, f.node_child_first.data(), f.node_child_first.size() * 4, 0), "broadcast");
        CheckComm(gpann_comm_broadcast_host(comm, f.node_child_count.data(), f.node_child_count.size() * 4, 0), "broadcast");
        CheckComm(gpann_comm_broadcast_host(comm, f.shard_root.data(), f.shard_root.size() * 4, 0), "broadcast");
        CheckComm(gpann_comm_broadcast_host(comm, f.centroids.data(), f.centroids.size() * 4, 0), "broadcast");
    }

    int backend_ = GPANN_COMM_TCP;
    size_t nq_ = 0, buf_n_ = 0, buf_pw_ = 0;
    float* d_queries_ = nullptr;
    int32_t *d_probes_ = nullptr, *d_all_probes_ = nullptr;
    uint32_t *d_ids_ = nullptr, *d_out_ids_ = nullptr, *d_map_ = nullptr;
    float *d_dist_ = nullptr, *d_out_dist_ = nullptr;
    char* d_scratch_ = nullptr;
    std::vector<char> host_scratch_;
    // ProcessQueryStream: its three streams and the two sets of per-batch device arrays
    hipStream_t s_route_ = nullptr, s_scan_ = nullptr, s_x_ = nullptr;
    size_t sbuf_n_ = 0, sbuf_pw_ = 0;
    int32_t *d_probes2_[2] = {nullptr, nullptr}, *d_all_probes2_[2] = {nullptr, nullptr};
    uint32_t* d_ids2_[2] = {nullptr, nullptr};
    float* d_dist2_[2] = {nullptr, nullptr};
};
