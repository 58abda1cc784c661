// distributed_bench.cpp — this build's DistributedBench: the reference's driver (distributed_bench.cpp:13-62; argv, call order, the
// barrier / max-reduce timing and the "End-to-end time" line are its) over the multi-GPU form of DistributedQueryBenchmark
// (host/distributed_query_benchmark.h): one process per GPU, shards spread over the ranks, RCCL over xGMI for the exchange.
//
//   DistributedBench input-points queries ground-truth-file num_neighbors partition-file router-file
//
// Launch: one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment (what
// `python -m torch.distributed.run --no-python --nproc-per-node N ... DistributedBench ...` exports; the reference uses mpirun).
// Additive environment knobs: GPANN_COMM_BACKEND=rccl|tcp (default rccl; tcp = exchange over host sockets, e.g. several ranks on one
// GPU), GPANN_DIST_PROBES (default 2, distributed_query_benchmark.h:35), GPANN_DIST_REPS (timed repetitions, default 1),
// GPANN_DIST_ROUTER_BUDGET (index size when the router file does not exist yet, default points / shards),
// GPANN_DIST_OUT (rank 0 writes the merged neighbours of all queries there: u32 nq, u32 k, ids, distances — the ground-truth layout),
// GPANN_DIST_PIPELINE=<batches> (after the reference's protocol: the block as a stream of batches through the three-stage pipeline).
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <numeric>

#include "../host/distributed_query_benchmark.h"

namespace {
size_t ComputeChunkSize(size_t a, size_t b) { return (a + b - 1) / b; }
int EnvInt(const char* k, int dflt) {
    const char* v = std::getenv(k);
    return v && *v ? std::atoi(v) : dflt;
}
}   // namespace

int main(int argc, const char* argv[]) {
    if (argc != 7) {
        std::cerr << "Usage ./DistributedBench input-points queries ground-truth-file num_neighbors partition-file router-file" << std::endl;
        std::abort();
    }
    std::string point_file = argv[1];
    std::string query_file = argv[2];
    std::string ground_truth_file = argv[3];
    std::string k_string = argv[4];
    std::string partition_file = argv[5];
    std::string router_file = argv[6];

    const char* be = std::getenv("GPANN_COMM_BACKEND");
    const int backend = be && std::string(be) == "tcp" ? GPANN_COMM_TCP : GPANN_COMM_RCCL;
    const int device = EnvInt("GPANN_DIST_DEVICE", EnvInt("LOCAL_RANK", 0));
    gpann_comm* comm = nullptr;
    if (gpann_comm_init_from_env_on(backend, device, &comm) != GPANN_OK) {   // (MPI_Init; the communicator lives on the device the buffers will)
        std::cerr << "communicator: " << gpann_comm_last_error() << std::endl;
        return 2;
    }
    int rank = 0, comm_size = 1;
    gpann_comm_info(comm, &rank, &comm_size, nullptr);
    int rc = 0;
    try {
        gpann_host::SetDevice(device);
        DistributedQueryBenchmark bench(comm, device);
        if (const char* w = std::getenv("GPANN_COMM_WAIT_S")) bench.wait_timeout_s = std::max(1, std::atoi(w));
        bench.num_neighbors = std::stoi(k_string);
        bench.num_probes = EnvInt("GPANN_DIST_PROBES", 2);
        bench.LoadPartition(partition_file);
        bench.num_probes = std::min(bench.num_probes, bench.num_shards);
        bench.LoadShardPointSet(point_file);
        bench.BuildInShardIndex();
        bench.router_options.budget = EnvInt("GPANN_DIST_ROUTER_BUDGET", (int)(bench.num_points / std::max(1, bench.num_shards)));
        bench.router_options.num_centroids = 64;
        bench.router_options.min_cluster_size = 350;
        bench.LoadRouter(router_file);

        PointSet queries;
        if (rank == 0) queries = ReadPoints(query_file);
        bench.BroadcastQueries(queries);
        std::vector<int> query_ids(queries.n);
        std::iota(query_ids.begin(), query_ids.end(), 0);
        size_t chunk_size = ComputeChunkSize(query_ids.size(), comm_size);
        std::vector<int> my_query_ids(query_ids.begin() + std::min(query_ids.size(), rank * chunk_size),
                                      query_ids.begin() + std::min(query_ids.size(), (rank + 1) * chunk_size));

        const int reps = std::max(1, EnvInt("GPANN_DIST_REPS", 1));
        bench.ProcessQueries(my_query_ids, queries);   // (first call: workspaces are sized; untimed)
        std::vector<double> times;
        for (int rep = 0; rep < reps; ++rep) {
            DistributedQueryBenchmark::CheckComm(gpann_comm_barrier(comm), "barrier");
            Timer timer;
            timer.Start();
            bench.ProcessQueries(my_query_ids, queries);
            double elapsed = timer.Stop();
            DistributedQueryBenchmark::CheckComm(gpann_comm_barrier(comm), "barrier");
            DistributedQueryBenchmark::CheckComm(gpann_comm_max_double(comm, &elapsed), "max");   // (MPI_Reduce MPI_MAX)
            times.push_back(elapsed);
        }
        std::sort(times.begin(), times.end());
        const double max_elapsed = times[times.size() / 2];

        // GPANN_DIST_PIPELINE=<batches>: the same block as a STREAM of batches through the three-stage pipeline (ProcessQueryStream) — what a
        // serving loop does; the reference's protocol above times one batch at a time.  A second communicator (the first one's port + 1) carries the
        // route-ahead thread's all-gathers.
        double stream_per_batch = 0.0;
        int stream_batches = std::max(0, EnvInt("GPANN_DIST_PIPELINE", 0));
        bool stream_identical = true;
        if (stream_batches > 0) {
            stream_batches = std::max(2, stream_batches);
            const std::vector<float> one_dist = bench.result_dist;
            const std::vector<uint32_t> one_ids = bench.result_ids;
            gpann_comm* comm_route = nullptr;
            const char* addr = std::getenv("MASTER_ADDR");
            if (gpann_comm_init(rank, comm_size, addr ? addr : "127.0.0.1", EnvInt("GPANN_COMM_PORT", EnvInt("MASTER_PORT", 29500) + 1) + 1, backend, device, EnvInt("GPANN_COMM_TIMEOUT", 120), &comm_route) != GPANN_OK)
                throw std::runtime_error(std::string("second communicator: ") + gpann_comm_last_error());
            bench.ProcessQueryStream(my_query_ids, queries, 2, comm_route);   // (untimed: streams and buffers)
            DistributedQueryBenchmark::CheckComm(gpann_comm_barrier(comm), "barrier");
            Timer timer;
            timer.Start();
            bench.ProcessQueryStream(my_query_ids, queries, stream_batches, comm_route);
            double elapsed = timer.Stop();
            DistributedQueryBenchmark::CheckComm(gpann_comm_barrier(comm), "barrier");
            DistributedQueryBenchmark::CheckComm(gpann_comm_max_double(comm, &elapsed), "max");
            stream_per_batch = elapsed / stream_batches;
            stream_identical = one_ids == bench.result_ids && one_dist.size() == bench.result_dist.size() &&
                               std::memcmp(one_dist.data(), bench.result_dist.data(), one_dist.size() * 4) == 0;
            gpann_comm_destroy(comm_route);
            if (!stream_identical) throw std::runtime_error("the pipelined stream's neighbours differ from the single batch's");
        }

        // recall of my slice against the ground-truth file (recall.h:74-150), summed over the ranks
        uint64_t hits = 0;
        {
            std::ifstream probe(ground_truth_file, std::ios::binary);
            if (probe) {
                std::vector<NNVec> gt = ReadGroundTruth(ground_truth_file);
                const int k = bench.num_neighbors;
                for (size_t q = bench.slice_begin; q < bench.slice_end; ++q) {
                    float kth = 0.f;
                    for (int j = 0; j < k && j < (int)gt[q].size(); ++j) kth = std::max(kth, gt[q][j].first);
                    for (int j = 0; j < k; ++j) hits += bench.result_dist[(q - bench.slice_begin) * k + j] <= kth;
                }
            }
        }
        std::vector<uint64_t> all_hits(comm_size);
        DistributedQueryBenchmark::CheckComm(gpann_comm_allgather_host(comm, &hits, all_hits.data(), 8), "allgather");
        // optional: all merged neighbours to a file on rank 0 (fixed-size slices, padded)
        if (const char* out_path = std::getenv("GPANN_DIST_OUT")) {
            const size_t k = bench.num_neighbors, words = chunk_size * k;
            std::vector<uint32_t> ids(words, GPANN_NO_ID), all_ids(words * comm_size);
            std::vector<float> dist(words, 0.f), all_dist(words * comm_size);
            std::copy(bench.result_ids.begin(), bench.result_ids.end(), ids.begin());
            std::copy(bench.result_dist.begin(), bench.result_dist.end(), dist.begin());
            DistributedQueryBenchmark::CheckComm(gpann_comm_allgather_host(comm, ids.data(), all_ids.data(), words * 4), "allgather");
            DistributedQueryBenchmark::CheckComm(gpann_comm_allgather_host(comm, dist.data(), all_dist.data(), words * 4), "allgather");
            if (rank == 0) {
                std::ofstream out(out_path, std::ios::binary);
                const uint32_t nq = (uint32_t)queries.n, kk = (uint32_t)k;
                out.write(reinterpret_cast<const char*>(&nq), 4);
                out.write(reinterpret_cast<const char*>(&kk), 4);
                out.write(reinterpret_cast<const char*>(all_ids.data()), (std::streamsize)((size_t)nq * k * 4));
                out.write(reinterpret_cast<const char*>(all_dist.data()), (std::streamsize)((size_t)nq * k * 4));
            }
        }
        if (rank == 0) {
            const uint64_t total = std::accumulate(all_hits.begin(), all_hits.end(), (uint64_t)0);
            std::cout << "backend " << gpann_comm_backend_name(comm) << ", ranks " << comm_size << ", probes " << bench.num_probes << ", recall "
                      << double(total) / double(queries.n) / bench.num_neighbors << ", QPS " << queries.n / max_elapsed << std::endl;
            std::cout << "End-to-end time " << max_elapsed << std::endl;
            if (stream_batches > 0)
                std::cout << "Pipelined stream: " << stream_batches << " batches, " << stream_per_batch << " s per batch, QPS " << queries.n / stream_per_batch
                          << ", neighbours identical to the single batch's " << (stream_identical ? 1 : 0) << std::endl;
        }
    } catch (const std::exception& e) {
        std::cerr << "rank " << rank << ": " << e.what() << std::endl;
        rc = 1;
    }
    // (MPI_Finalize)  A rank that failed leaves WITHOUT waiting for collectives the others may be inside — ncclCommDestroy could block on
    // them — and closes its bootstrap connections: the other ranks' gpann_comm_wait / host collectives return an error and they leave too
    if (rc != 0) gpann_comm_abort(comm);
    else gpann_comm_destroy(comm);
    return rc;
}
