// comm.hip — libgpann_comm.so: the exchange steps of the multi-GPU route + shard-scan path (include/gpann_comm.h).
//
// One process per GPU.  The data path is RCCL (ncclBroadcast for the query block, ncclAllGather for probe lists and per-slot result
// lists, enqueued on the caller's stream next to the scan kernels) plus one small selection kernel; the control path (rendezvous, the
// ncclUniqueId, barrier, max-of-a-double, tree parts at setup) is one TCP connection per rank to rank 0.  xGMI is point to point and the
// payloads are KB..MB, so the exchange is latency bound: fixed shapes, one collective per step, no per-query messages (the reference's
// sketch posts one message per (query, shard), src/distributed_query_benchmark.h:163-185).
// The TCP backend moves the same payloads through rank 0 over the bootstrap sockets: it exists so that the exchange logic can be
// exercised with world size > 1 where there is no GPU, and is never selected implicitly.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpann_comm.h"

namespace {

thread_local std::string g_comm_err;

void SetErr(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_comm_err = buf;
}

#define COMM_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            SetErr("invalid argument: %s (%s:%d)", msg, __FILE__, __LINE__); \
            return GPANN_ERR_INVALID;                             \
        }                                                         \
    } while (0)
#define COMM_HIP(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            SetErr("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
            return GPANN_ERR_HIP;                                                                   \
        }                                                                                           \
    } while (0)
#define COMM_NCCL(expr)                                                                             \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            SetErr("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__);     \
            return GPANN_ERR_HIP;                                                                   \
        }                                                                                           \
    } while (0)
#define COMM_TRY(expr)                     \
    do {                                   \
        gpann_status _s = (expr);          \
        if (_s != GPANN_OK) return _s;     \
    } while (0)

bool SendAll(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
        if (w <= 0) return false;
        c += w;
        n -= (size_t)w;
    }
    return true;
}
bool RecvAll(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        const ssize_t r = ::recv(fd, c, n, 0);
        if (r <= 0) return false;
        c += r;
        n -= (size_t)r;
    }
    return true;
}

}   // namespace

struct gpann_comm {
    int rank = 0, world = 1, backend = GPANN_COMM_TCP, device = 0;
    int listen_fd = -1;
    std::vector<int> peers;   // rank 0: fd of every other rank (index = rank); others: peers[0] = fd to rank 0
    ncclComm_t nccl = nullptr;
    std::string name;
    std::vector<char> host_scratch;
};

namespace {

// ---- control path: star through rank 0 ---------------------------------------------------------------------------------------
gpann_status StarAllgather(gpann_comm* c, const void* send, void* recv, size_t bytes) {
    if (c->world == 1) {
        if (recv != send) std::memcpy(recv, send, bytes);
        return GPANN_OK;
    }
    char* out = static_cast<char*>(recv);
    if (c->rank == 0) {
        std::memmove(out, send, bytes);
        for (int r = 1; r < c->world; ++r)
            if (!RecvAll(c->peers[r], out + (size_t)r * bytes, bytes)) {
                SetErr("allgather: connection to rank %d lost", r);
                return GPANN_ERR_HIP;
            }
        for (int r = 1; r < c->world; ++r)
            if (!SendAll(c->peers[r], out, bytes * c->world)) {
                SetErr("allgather: connection to rank %d lost", r);
                return GPANN_ERR_HIP;
            }
    } else {
        if (!SendAll(c->peers[0], send, bytes) || !RecvAll(c->peers[0], out, bytes * c->world)) {
            SetErr("allgather: connection to rank 0 lost");
            return GPANN_ERR_HIP;
        }
    }
    return GPANN_OK;
}

gpann_status StarBroadcast(gpann_comm* c, void* buf, size_t bytes, int root) {
    if (c->world == 1) return GPANN_OK;
    // root -> rank 0 -> everyone
    if (root != 0) {
        if (c->rank == root && !SendAll(c->peers[0], buf, bytes)) goto lost;
        if (c->rank == 0 && !RecvAll(c->peers[root], buf, bytes)) goto lost;
    }
    if (c->rank == 0) {
        for (int r = 1; r < c->world; ++r)
            if (r != root && !SendAll(c->peers[r], buf, bytes)) goto lost;
    } else if (c->rank != root) {
        if (!RecvAll(c->peers[0], buf, bytes)) goto lost;
    }
    return GPANN_OK;
lost:
    SetErr("broadcast: a bootstrap connection was lost");
    return GPANN_ERR_HIP;
}

// owner selection of the gathered per-slot lists: slot i comes from the one rank where it is not (+inf, NO_ID)
__global__ void SelectSlotsKernel(const float* gathered_dist, const uint32_t* gathered_ids, int world, size_t n, float* out_dist, uint32_t* out_ids) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = INFINITY;
    uint32_t id = GPANN_NO_ID;
    for (int r = 0; r < world; ++r) {
        const uint32_t x = gathered_ids[(size_t)r * n + i];
        if (x != GPANN_NO_ID) {
            id = x;
            d = gathered_dist[(size_t)r * n + i];
        }
    }
    out_dist[i] = d;
    out_ids[i] = id;
}

__global__ void MapIdsKernel(const uint32_t* ids, const uint32_t* id_map, size_t n, uint32_t* out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = ids[i] == GPANN_NO_ID ? GPANN_NO_ID : id_map[ids[i]];
}

int ConnectTo(const char* addr, int port, int timeout_s) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string p = std::to_string(port);
    if (getaddrinfo(addr, p.c_str(), &hints, &res) != 0 || !res) return -1;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
    int fd = -1;
    while (std::chrono::steady_clock::now() < deadline) {
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            // A connect() to a local port nobody listens on yet can CONNECT TO ITSELF when the kernel happens to pick that very port as the
            // source port (TCP simultaneous open; the bootstrap port often lies in the ephemeral range): the socket then looks connected, rank 0
            // can no longer bind the port ("Address already in use") and every rank waits for the others until its timeout.  Seen twice in
            // 110 three-rank runs on one box (round 4, tools/stress_suite_context.sh).  Such a socket is dropped and the connect repeated.
            sockaddr_in self{}, peer{};
            socklen_t ls = sizeof self, lp = sizeof peer;
            const bool self_connected = ::getsockname(fd, reinterpret_cast<sockaddr*>(&self), &ls) == 0 && ::getpeername(fd, reinterpret_cast<sockaddr*>(&peer), &lp) == 0 &&
                                        self.sin_port == peer.sin_port && self.sin_addr.s_addr == peer.sin_addr.s_addr;
            if (!self_connected) break;
        }
        if (fd >= 0) ::close(fd);
        fd = -1;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));   // rank 0 may not be listening yet
    }
    freeaddrinfo(res);
    if (fd >= 0) {
        int one = 1;
        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    }
    return fd;
}

}   // namespace

extern "C" {

const char* gpann_comm_last_error(void) { return g_comm_err.c_str(); }

gpann_status gpann_comm_init(int rank, int world, const char* master_addr, int port, int backend, int device, int timeout_s, gpann_comm** out) {
    COMM_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    COMM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world");
    COMM_REQUIRE(backend == GPANN_COMM_RCCL || backend == GPANN_COMM_TCP, "unknown backend");
    COMM_REQUIRE(world == 1 || (master_addr && port > 0), "master address / port");
    auto* c = new gpann_comm();
    c->rank = rank;
    c->world = world;
    c->backend = backend;
    c->device = device;
    auto fail = [&](gpann_status s) {
        gpann_comm_destroy(c);
        return s;
    };
    if (world > 1) {
        if (rank == 0) {
            c->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
            int one = 1;
            setsockopt(c->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
            sockaddr_in sa{};
            sa.sin_family = AF_INET;
            sa.sin_addr.s_addr = htonl(INADDR_ANY);
            sa.sin_port = htons((uint16_t)port);
            if (c->listen_fd < 0 || ::bind(c->listen_fd, reinterpret_cast<sockaddr*>(&sa), sizeof sa) != 0 || ::listen(c->listen_fd, world) != 0) {
                SetErr("rank 0 cannot listen on port %d: %s", port, strerror(errno));
                return fail(GPANN_ERR_HIP);
            }
            // the deadline covers accept() and the 4-byte hello only: accepted sockets inherit SO_RCVTIMEO on Linux, and a collective that
            // waits longer than that for a slow peer (uneven store builds at 10^8 rows) must not read as "connection lost"
            timeval tv{timeout_s, 0};
            setsockopt(c->listen_fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
            c->peers.assign(world, -1);
            for (int arrived = 1; arrived < world;) {
                const int fd = ::accept(c->listen_fd, nullptr, nullptr);
                if (fd < 0) {
                    if ((errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) && std::chrono::steady_clock::now() < deadline) continue;
                    SetErr("rendezvous: %d of %d ranks arrived within %d s", arrived, world, timeout_s);
                    return fail(GPANN_ERR_HIP);
                }
                int32_t r = -1;
                if (!RecvAll(fd, &r, 4) || r <= 0 || r >= world || c->peers[r] != -1) {
                    ::close(fd);   // a stray connection / a wrong first word: drop it and keep waiting for the ranks
                    if (std::chrono::steady_clock::now() < deadline) continue;
                    SetErr("rendezvous: %d of %d ranks arrived within %d s", arrived, world, timeout_s);
                    return fail(GPANN_ERR_HIP);
                }
                const timeval none{0, 0};
                setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
                int one2 = 1;
                setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one2, sizeof one2);
                c->peers[r] = fd;
                ++arrived;
            }
        } else {
            const int fd = ConnectTo(master_addr, port, timeout_s);
            const int32_t r = rank;
            if (fd < 0 || !SendAll(fd, &r, 4)) {
                SetErr("rank %d cannot reach rank 0 at %s:%d", rank, master_addr, port);
                if (fd >= 0) ::close(fd);
                return fail(GPANN_ERR_HIP);
            }
            c->peers.assign(1, fd);
        }
    }
    if (backend == GPANN_COMM_RCCL) {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device) {
            (void)hipGetLastError();
            SetErr("RCCL backend: HIP device %d is not available (%d devices)", device, n_dev);
            return fail(GPANN_ERR_NO_DEVICE);
        }
        // the communicator is bound to `device`, which STAYS the calling thread's current device (include/gpann_comm.h): a rank process that
        // has not touched HIP yet relies on this call to select its GPU — restoring "the previous device" would silently put its buffers on
        // device 0 while the communicator lives on LOCAL_RANK.  A caller that works on another device calls hipSetDevice itself afterwards.
        int before = -1;
        if (hipGetDevice(&before) != hipSuccess) (void)hipGetLastError();
        if (hipSetDevice(device) != hipSuccess) {
            SetErr("hipSetDevice(%d) failed", device);
            return fail(GPANN_ERR_HIP);
        }
        // (a caller that had chosen ANOTHER device before — its buffers may live there — is told once that its current device has changed: a
        // later allocation would otherwise land on the communicator's device without a word; n_dev > 1: on a one-GPU box there is nothing to mix up)
        if (before >= 0 && before != device && n_dev > 1) {
            static bool warned = false;
            if (!warned) {
                warned = true;
                fprintf(stderr, "[gpann comm] gpann_comm_init: the calling thread's HIP device was %d; it is now %d, the communicator's (include/gpann_comm.h) — "
                                "call hipSetDevice again if your buffers live elsewhere\n", before, device);
            }
        }
        ncclUniqueId id;
        std::memset(&id, 0, sizeof id);
        if (rank == 0) {
            ncclResult_t r = ncclGetUniqueId(&id);
            if (r != ncclSuccess) {
                SetErr("ncclGetUniqueId failed: %s", ncclGetErrorString(r));
                // the other ranks wait for the id: tell them (an all-zero id) before giving up
                (void)StarBroadcast(c, &id, sizeof id, 0);
                return fail(GPANN_ERR_HIP);
            }
        }
        if (StarBroadcast(c, &id, sizeof id, 0) != GPANN_OK) return fail(GPANN_ERR_HIP);
        ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
        if (r != ncclSuccess) {
            SetErr("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, ncclGetErrorString(r));
            c->nccl = nullptr;
            return fail(GPANN_ERR_HIP);
        }
        int v = 0;
        (void)ncclGetVersion(&v);
        c->name = "rccl " + std::to_string(v);
    } else {
        c->name = "tcp";
    }
    *out = c;
    return GPANN_OK;
}

gpann_status gpann_comm_init_from_env(int backend, gpann_comm** out) { return gpann_comm_init_from_env_on(backend, -1, out); }

gpann_status gpann_comm_init_from_env_on(int backend, int device_arg, gpann_comm** out) {
    auto env_int = [](const char* k, int dflt) {
        const char* v = std::getenv(k);
        return v && *v ? std::atoi(v) : dflt;
    };
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), device = device_arg >= 0 ? device_arg : env_int("LOCAL_RANK", 0);
    const char* addr = std::getenv("MASTER_ADDR");
    const int port = env_int("GPANN_COMM_PORT", env_int("MASTER_PORT", 29500) + 1);
    return gpann_comm_init(rank, world, addr && *addr ? addr : "127.0.0.1", port, backend, device, env_int("GPANN_COMM_TIMEOUT", 600), out);
}

gpann_status gpann_comm_destroy(gpann_comm* c) {
    if (!c) return GPANN_OK;
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    for (int fd : c->peers)
        if (fd >= 0) ::close(fd);
    if (c->listen_fd >= 0) ::close(c->listen_fd);
    delete c;
    return GPANN_OK;
}

gpann_status gpann_comm_abort(gpann_comm* c) {
    if (!c) return GPANN_OK;
    if (c->nccl) (void)ncclCommAbort(c->nccl);   // (does not wait for outstanding collectives, unlike ncclCommDestroy)
    for (int fd : c->peers)
        if (fd >= 0) ::close(fd);   // the other ranks' gpann_comm_wait / host collectives see the hang-up
    if (c->listen_fd >= 0) ::close(c->listen_fd);
    delete c;
    return GPANN_OK;
}

// a rank that has hung up: its end of the bootstrap connection reads as closed (0 bytes); data waiting or nothing to read = alive.
// Returns the rank that left (rank 0 sees every peer; the others only see rank 0), or -1.
static int LostPeer(const gpann_comm* c) {
    for (size_t i = 0; i < c->peers.size(); ++i) {
        const int fd = c->peers[i];
        if (fd < 0) continue;
        char b;
        const ssize_t n = ::recv(fd, &b, 1, MSG_PEEK | MSG_DONTWAIT);
        if (n == 0 || (n < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) return c->rank == 0 ? (int)i : 0;
    }
    return -1;
}

gpann_status gpann_comm_check_peers(gpann_comm* c, int* lost_rank) {
    COMM_REQUIRE(c != nullptr, "comm is null");
    const int lost = c->world > 1 ? LostPeer(c) : -1;
    if (lost_rank) *lost_rank = lost;
    if (lost < 0) return GPANN_OK;
    SetErr("rank %d has left (bootstrap connection closed)", lost);
    return GPANN_ERR_INTERNAL;
}

gpann_status gpann_comm_wait(gpann_comm* c, void* stream, int timeout_s) {
    COMM_REQUIRE(c != nullptr, "comm is null");
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(std::max(timeout_s, 1));
    const bool device_work = c->backend == GPANN_COMM_RCCL;
    for (uint64_t spin = 0;; ++spin) {
        const hipError_t q = device_work ? hipStreamQuery(static_cast<hipStream_t>(stream)) : hipSuccess;
        // completed work is a success whatever the peers do afterwards: a rank that finished the same last collective and called
        // gpann_comm_destroy / _abort a few microseconds earlier is not an error of THIS call (gpann_comm_check_peers asks about the peers)
        if (q == hipSuccess) return GPANN_OK;
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            SetErr("hipStreamQuery failed: %s", hipGetErrorString(q));
            return GPANN_ERR_HIP;
        }
        // the work is still pending: is it ever going to complete?
        const char* why = nullptr;
        char buf[96];
        if ((spin & 15) == 15 || spin == 0) {   // (the connections: every 16th poll)
            const int lost = c->world > 1 ? LostPeer(c) : -1;
            if (lost >= 0) {
                snprintf(buf, sizeof buf, "rank %d has left (bootstrap connection closed)", lost);
                why = buf;
            }
            if (!why && c->nccl) {
                ncclResult_t async = ncclSuccess;
                if (ncclCommGetAsyncError(c->nccl, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                    snprintf(buf, sizeof buf, "RCCL reports %s", ncclGetErrorString(async));
                    why = buf;
                }
            }
        }
        if (!why && std::chrono::steady_clock::now() > deadline) {
            snprintf(buf, sizeof buf, "no completion within %d s", timeout_s);
            why = buf;
        }
        if (why) {
            // (a peer that left right after the collective completed on ITS side: our side may be a moment behind — one last look)
            if (device_work && hipStreamQuery(static_cast<hipStream_t>(stream)) == hipSuccess) return GPANN_OK;
            (void)hipGetLastError();
            if (c->nccl) {   // unblock this rank: the collective it waits for will never complete
                (void)ncclCommAbort(c->nccl);
                c->nccl = nullptr;
            }
            SetErr("collective abandoned: %s", why);
            return GPANN_ERR_INTERNAL;
        }
        if (spin > 64) std::this_thread::sleep_for(std::chrono::microseconds(spin > 4096 ? 1000 : 50));
    }
}

gpann_status gpann_comm_info(const gpann_comm* c, int* rank, int* world, int* backend) {
    COMM_REQUIRE(c != nullptr, "comm is null");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (backend) *backend = c->backend;
    return GPANN_OK;
}

const char* gpann_comm_backend_name(const gpann_comm* c) { return c ? c->name.c_str() : ""; }

gpann_status gpann_comm_barrier(gpann_comm* c) {
    COMM_REQUIRE(c != nullptr, "comm is null");
    char token = 1;
    std::vector<char> all((size_t)c->world);
    return StarAllgather(c, &token, all.data(), 1);
}

gpann_status gpann_comm_max_double(gpann_comm* c, double* value) {
    COMM_REQUIRE(c != nullptr && value != nullptr, "null argument");
    std::vector<double> all((size_t)c->world);
    COMM_TRY(StarAllgather(c, value, all.data(), sizeof(double)));
    *value = *std::max_element(all.begin(), all.end());
    return GPANN_OK;
}

gpann_status gpann_comm_allgather_host(gpann_comm* c, const void* send, void* recv, size_t bytes) {
    COMM_REQUIRE(c != nullptr && (bytes == 0 || (send && recv)), "null argument");
    return StarAllgather(c, send, recv, bytes);
}

gpann_status gpann_comm_broadcast_host(gpann_comm* c, void* buf, size_t bytes, int root) {
    COMM_REQUIRE(c != nullptr && (bytes == 0 || buf) && root >= 0 && root < c->world, "argument");
    return StarBroadcast(c, buf, bytes, root);
}

gpann_status gpann_comm_broadcast(gpann_comm* c, void* buf, size_t bytes, int root, void* stream) {
    COMM_REQUIRE(c != nullptr && (bytes == 0 || buf) && root >= 0 && root < c->world, "argument");
    if (c->backend == GPANN_COMM_TCP) return StarBroadcast(c, buf, bytes, root);
    if (bytes == 0) return GPANN_OK;
    COMM_NCCL(ncclBroadcast(buf, buf, bytes, ncclChar, root, c->nccl, static_cast<hipStream_t>(stream)));
    return GPANN_OK;
}

gpann_status gpann_comm_allgather(gpann_comm* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    COMM_REQUIRE(c != nullptr && (bytes_per_rank == 0 || (send && recv)), "null argument");
    if (c->backend == GPANN_COMM_TCP) return StarAllgather(c, send, recv, bytes_per_rank);
    if (bytes_per_rank == 0) return GPANN_OK;
    COMM_NCCL(ncclAllGather(send, recv, bytes_per_rank, ncclChar, c->nccl, static_cast<hipStream_t>(stream)));
    return GPANN_OK;
}

size_t gpann_comm_gather_slots_scratch(int world, size_t n) { return (size_t)2 * (size_t)std::max(world, 1) * n * 4 + 2 * n * 4; }

gpann_status gpann_comm_gather_slots(gpann_comm* c, const float* dist, const uint32_t* ids, size_t n, float* out_dist, uint32_t* out_ids, void* scratch,
                                     const uint32_t* id_map, void* stream) {
    COMM_REQUIRE(c != nullptr && (n == 0 || (dist && ids && out_dist && out_ids && scratch)), "null argument");
    if (n == 0) return GPANN_OK;
    const int W = c->world;
    // one collective: every rank contributes [dist[n] | ids[n]], the gathered block is [rank][dist[n] | ids[n]]
    char* pack = static_cast<char*>(scratch);                   // [2n words] this rank's contribution
    char* gathered = pack + 2 * n * 4;                          // [W][2n words]
    if (c->backend == GPANN_COMM_TCP) {
        std::memcpy(pack, dist, n * 4);
        if (id_map) {
            uint32_t* mapped = reinterpret_cast<uint32_t*>(pack + n * 4);
            for (size_t i = 0; i < n; ++i) mapped[i] = ids[i] == GPANN_NO_ID ? GPANN_NO_ID : id_map[ids[i]];
        } else {
            std::memcpy(pack + n * 4, ids, n * 4);
        }
        COMM_TRY(StarAllgather(c, pack, gathered, 2 * n * 4));
        for (size_t i = 0; i < n; ++i) {
            float d = std::numeric_limits<float>::infinity();
            uint32_t id = GPANN_NO_ID;
            for (int r = 0; r < W; ++r) {
                uint32_t x;
                std::memcpy(&x, gathered + ((size_t)r * 2 * n + n + i) * 4, 4);
                if (x != GPANN_NO_ID) {
                    id = x;
                    std::memcpy(&d, gathered + ((size_t)r * 2 * n + i) * 4, 4);
                }
            }
            out_dist[i] = d;
            out_ids[i] = id;
        }
        return GPANN_OK;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the two lists are gathered as two collectives into [W][n] arrays (no packing copy); grouped, they go out as one launch
    float* g_dist = reinterpret_cast<float*>(gathered);
    uint32_t* g_ids = reinterpret_cast<uint32_t*>(gathered + (size_t)W * n * 4);
    if (id_map) {   // local rows -> point ids (into the head of the scratch area)
        uint32_t* mapped = reinterpret_cast<uint32_t*>(pack);
        hipLaunchKernelGGL(MapIdsKernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, ids, id_map, n, mapped);
        COMM_HIP(hipGetLastError());
        ids = mapped;
    }
    COMM_NCCL(ncclGroupStart());
    COMM_NCCL(ncclAllGather(dist, g_dist, n, ncclFloat, c->nccl, st));
    COMM_NCCL(ncclAllGather(ids, g_ids, n, ncclUint32, c->nccl, st));
    COMM_NCCL(ncclGroupEnd());
    hipLaunchKernelGGL(SelectSlotsKernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, g_dist, g_ids, W, n, out_dist, out_ids);
    COMM_HIP(hipGetLastError());
    return GPANN_OK;
}

gpann_status gpann_comm_debug_select_slots(int world, size_t n, const float* rank_dist, const uint32_t* rank_ids, const uint32_t* id_maps,
                                           const uint64_t* id_map_begin, float* out_dist, uint32_t* out_ids, int device) {
    COMM_REQUIRE(world >= 1 && (n == 0 || (rank_dist && rank_ids && out_dist && out_ids)) && (!id_maps || id_map_begin), "argument");
    if (n == 0) return GPANN_OK;
    int prev_device = -1;   // (a test hook: the caller's current device is put back)
    (void)hipGetDevice(&prev_device);
    struct Restore {
        int dev;
        ~Restore() {
            if (dev >= 0) (void)hipSetDevice(dev);
        }
    } restore{prev_device};
    COMM_HIP(hipSetDevice(device));
    const size_t W = (size_t)world;
    float *g_dist = nullptr, *d_out_dist = nullptr;
    uint32_t *g_ids = nullptr, *d_local = nullptr, *d_maps = nullptr, *d_out_ids = nullptr;
    const size_t map_words = id_maps ? (size_t)id_map_begin[world] : 0;
    gpann_status rc = GPANN_OK;
    auto fail = [&](hipError_t e) {
        if (e == hipSuccess) return false;
        SetErr("debug_select_slots: %s", hipGetErrorString(e));
        rc = GPANN_ERR_HIP;
        return true;
    };
    do {
        if (fail(hipMalloc(&g_dist, W * n * 4)) || fail(hipMalloc(&g_ids, W * n * 4)) || fail(hipMalloc(&d_local, W * n * 4)) ||
            fail(hipMalloc(&d_out_dist, n * 4)) || fail(hipMalloc(&d_out_ids, n * 4)) || (map_words && fail(hipMalloc(&d_maps, map_words * 4))))
            break;
        if (fail(hipMemcpy(g_dist, rank_dist, W * n * 4, hipMemcpyHostToDevice)) || fail(hipMemcpy(d_local, rank_ids, W * n * 4, hipMemcpyHostToDevice)) ||
            (map_words && fail(hipMemcpy(d_maps, id_maps, map_words * 4, hipMemcpyHostToDevice))))
            break;
        for (size_t r = 0; r < W; ++r) {   // every rank maps its own list before the all-gather
            if (id_maps) hipLaunchKernelGGL(MapIdsKernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, d_local + r * n, d_maps + id_map_begin[r], n, g_ids + r * n);
            else if (fail(hipMemcpy(g_ids + r * n, d_local + r * n, n * 4, hipMemcpyDeviceToDevice))) break;
        }
        if (rc != GPANN_OK) break;
        hipLaunchKernelGGL(SelectSlotsKernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, g_dist, g_ids, world, n, d_out_dist, d_out_ids);
        if (fail(hipGetLastError()) || fail(hipMemcpy(out_dist, d_out_dist, n * 4, hipMemcpyDeviceToHost)) ||
            fail(hipMemcpy(out_ids, d_out_ids, n * 4, hipMemcpyDeviceToHost)))
            break;
    } while (false);
    for (void* p : {(void*)g_dist, (void*)g_ids, (void*)d_local, (void*)d_maps, (void*)d_out_dist, (void*)d_out_ids})
        if (p) (void)hipFree(p);
    return rc;
}

// ---- placement ------------------------------------------------------------------------------------------------------------------
gpann_status gpann_lpt_owner(const uint64_t* shard_rows, uint32_t S, uint32_t world, int32_t* owner, uint64_t* load_out) {
    COMM_REQUIRE(world >= 1 && (S == 0 || (shard_rows && owner)), "argument");
    std::vector<uint32_t> order(S);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return shard_rows[a] > shard_rows[b]; });
    std::vector<uint64_t> load(world, 0);
    for (uint32_t b : order) {
        const uint32_t r = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());   // first minimum: the lower rank on ties
        owner[b] = (int32_t)r;
        load[r] += shard_rows[b];
    }
    if (load_out) std::copy(load.begin(), load.end(), load_out);
    return GPANN_OK;
}

gpann_status gpann_mask_probes(const int32_t* probes, uint32_t nq, uint32_t probe_stride, uint32_t nprobe, const int32_t* owner, uint32_t S, int rank,
                               int32_t* out) {
    COMM_REQUIRE(nq == 0 || nprobe == 0 || (probes && owner && out), "null argument");
    COMM_REQUIRE(probe_stride >= nprobe, "probe_stride < nprobe");
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t j = 0; j < nprobe; ++j) {
            const int32_t b = probes[(size_t)q * probe_stride + j];
            out[(size_t)q * nprobe + j] = (b >= 0 && (uint32_t)b < S && owner[b] == rank) ? b : -1;
        }
    return GPANN_OK;
}

// ---- replicas -------------------------------------------------------------------------------------------------------------------
gpann_status gpann_plan_replicas(const uint64_t* shard_rows, const double* shard_work, uint32_t S, uint32_t world, uint64_t capacity_rows,
                                 double max_imbalance, uint32_t* rep_begin, int32_t* rep_rank, double* rank_work_out, uint64_t* rank_rows_out) {
    COMM_REQUIRE(world >= 1 && S >= 1 && shard_rows && shard_work && rep_begin && rep_rank, "argument");
    COMM_REQUIRE(max_imbalance >= 1.0, "max_imbalance < 1");
    std::vector<uint32_t> n_rep(S, 1);
    std::vector<std::vector<int32_t>> where(S);
    std::vector<double> work(world);
    std::vector<uint64_t> rows(world);
    // LPT placement of the current replica multiset: heaviest (work per replica, then rows) first onto the lightest rank that has room
    // and does not hold the shard yet.  Returns false if some replica finds no rank.
    auto place = [&]() -> bool {
        struct Item { uint32_t shard; double w; };
        std::vector<Item> items;
        for (uint32_t b = 0; b < S; ++b)
            for (uint32_t i = 0; i < n_rep[b]; ++i) items.push_back({b, shard_work[b] / n_rep[b]});
        std::stable_sort(items.begin(), items.end(), [&](const Item& l, const Item& r) {
            if (l.w != r.w) return l.w > r.w;
            return shard_rows[l.shard] > shard_rows[r.shard];
        });
        std::fill(work.begin(), work.end(), 0.0);
        std::fill(rows.begin(), rows.end(), 0ull);
        for (auto& w : where) w.clear();
        for (const Item& it : items) {
            int best = -1;
            for (uint32_t r = 0; r < world; ++r) {
                if (rows[r] + shard_rows[it.shard] > capacity_rows) continue;
                if (std::find(where[it.shard].begin(), where[it.shard].end(), (int32_t)r) != where[it.shard].end()) continue;
                if (best < 0 || work[r] < work[best] || (work[r] == work[best] && rows[r] < rows[best])) best = (int)r;
            }
            if (best < 0) return false;
            where[it.shard].push_back(best);
            work[best] += it.w;
            rows[best] += shard_rows[it.shard];
        }
        return true;
    };
    if (!place()) {
        SetErr("the shards do not fit: capacity %llu rows per rank", (unsigned long long)capacity_rows);
        return GPANN_ERR_INVALID;
    }
    std::vector<uint32_t> good_rep = n_rep;
    std::vector<std::vector<int32_t>> good_where = where;
    std::vector<double> good_work = work;
    std::vector<uint64_t> good_rows = rows;
    auto ratio_of = [&](const std::vector<double>& w) {
        const double total = std::accumulate(w.begin(), w.end(), 0.0);
        return total > 0.0 ? *std::max_element(w.begin(), w.end()) / (total / world) : 1.0;
    };
    double best_ratio = ratio_of(work);
    std::vector<uint8_t> frozen(S, 0);
    for (uint32_t round = 0; round < S * world && best_ratio > max_imbalance; ++round) {
        // the shard with the most work per replica that can still get one (route_search_combination.cpp:152,181-182)
        int pick = -1;
        for (uint32_t b = 0; b < S; ++b)
            if (!frozen[b] && n_rep[b] < world && (pick < 0 || shard_work[b] / n_rep[b] > shard_work[pick] / n_rep[pick])) pick = (int)b;
        if (pick < 0) break;
        n_rep[pick]++;
        if (!place()) {   // no room for it: this shard stays as it was
            n_rep[pick]--;
            frozen[pick] = 1;
            continue;
        }
        const double r = ratio_of(work);
        if (r < best_ratio) {
            best_ratio = r;
            good_rep = n_rep;
            good_where = where;
            good_work = work;
            good_rows = rows;
        }
    }
    uint32_t at = 0;
    for (uint32_t b = 0; b < S; ++b) {
        rep_begin[b] = at;
        for (int32_t r : good_where[b]) rep_rank[at++] = r;
    }
    rep_begin[S] = at;
    if (rank_work_out) std::copy(good_work.begin(), good_work.end(), rank_work_out);
    if (rank_rows_out) std::copy(good_rows.begin(), good_rows.end(), rank_rows_out);
    return GPANN_OK;
}

gpann_status gpann_mask_probes_replicated(const int32_t* probes, uint32_t nq, uint32_t probe_stride, uint32_t nprobe, const uint32_t* rep_begin,
                                          const int32_t* rep_rank, uint32_t S, int rank, int32_t* out) {
    COMM_REQUIRE(nq == 0 || nprobe == 0 || (probes && rep_begin && rep_rank && out), "null argument");
    COMM_REQUIRE(probe_stride >= nprobe, "probe_stride < nprobe");
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t j = 0; j < nprobe; ++j) {
            const int32_t b = probes[(size_t)q * probe_stride + j];
            int32_t v = -1;
            if (b >= 0 && (uint32_t)b < S) {
                const uint32_t r = rep_begin[b + 1] - rep_begin[b];
                if (r && rep_rank[rep_begin[b] + q % r] == rank) v = b;
            }
            out[(size_t)q * nprobe + j] = v;
        }
    return GPANN_OK;
}

// ---- router parts ---------------------------------------------------------------------------------------------------------------
gpann_status gpann_comm_merge_router(gpann_comm* c, gpann_router* mine, const int32_t* owner, int device, gpann_router** out) {
    COMM_REQUIRE(c != nullptr && mine != nullptr && owner != nullptr && out != nullptr, "null argument");
    *out = nullptr;
    uint32_t d = 0, S = 0, n_nodes = 0;
    uint64_t n_cent = 0;
    int metric = 0;
    auto hipst = [&](gpann_status s, const char* what) {
        if (s != GPANN_OK) SetErr("%s failed: %s", what, gpann_last_error());
        return s;
    };
    COMM_TRY(hipst(gpann_router_info(mine, &d, &metric, &S, &n_nodes, &n_cent), "gpann_router_info"));
    const int W = c->world;
    // sizes of every rank's part
    uint64_t my_sizes[2] = {n_nodes, n_cent};
    std::vector<uint64_t> sizes((size_t)2 * W);
    COMM_TRY(StarAllgather(c, my_sizes, sizes.data(), sizeof my_sizes));
    uint64_t max_nodes = 0, max_cent = 0;
    for (int r = 0; r < W; ++r) {
        max_nodes = std::max(max_nodes, sizes[2 * r]);
        max_cent = std::max(max_cent, sizes[2 * r + 1]);
    }
    // one fixed-size record per rank: node_shard | cent_first | child_first | child_count | shard_root | centroids
    const size_t rec_words = (size_t)max_nodes * 4 + 1 + S + (size_t)max_cent * d;
    std::vector<uint32_t> rec(rec_words, 0), all((size_t)W * rec_words);
    {
        int32_t* node_shard = reinterpret_cast<int32_t*>(rec.data());
        uint32_t* cent_first = rec.data() + max_nodes;
        uint32_t* child_first = cent_first + max_nodes + 1;
        uint32_t* child_count = child_first + max_nodes;
        uint32_t* shard_root = child_count + max_nodes;
        float* cents = reinterpret_cast<float*>(shard_root + S);
        COMM_TRY(hipst(gpann_router_export(mine, node_shard, cent_first, child_first, child_count, shard_root, cents), "gpann_router_export"));
    }
    COMM_TRY(StarAllgather(c, rec.data(), all.data(), rec_words * 4));
    // shard b's node range comes from rank owner[b]; child and centroid indices are re-based (the nodes of a shard are consecutive)
    std::vector<int32_t> node_shard;
    std::vector<uint32_t> cent_first{0}, child_first, child_count, shard_root;
    std::vector<float> cents;
    for (uint32_t b = 0; b < S; ++b) {
        const int r = owner[b];
        COMM_REQUIRE(r >= 0 && r < W, "owner out of range");
        const uint32_t* part = all.data() + (size_t)r * rec_words;
        const uint32_t nn = (uint32_t)sizes[2 * r];
        const uint32_t* p_cent_first = part + max_nodes;
        const uint32_t* p_child_first = p_cent_first + max_nodes + 1;
        const uint32_t* p_child_count = p_child_first + max_nodes;
        const uint32_t* p_shard_root = p_child_count + max_nodes;
        const float* p_cents = reinterpret_cast<const float*>(p_shard_root + S);
        const uint32_t lo = p_shard_root[b], hi = b + 1 < S ? p_shard_root[b + 1] : nn;
        const uint32_t base = (uint32_t)node_shard.size();
        shard_root.push_back(base);
        for (uint32_t u = lo; u < hi; ++u) {
            node_shard.push_back((int32_t)b);
            child_count.push_back(p_child_count[u]);
            child_first.push_back(p_child_count[u] ? p_child_first[u] - lo + base : 0u);
            cent_first.push_back(cent_first.back() + (p_cent_first[u + 1] - p_cent_first[u]));
        }
        cents.insert(cents.end(), p_cents + (size_t)p_cent_first[lo] * d, p_cents + (size_t)p_cent_first[hi] * d);
    }
    if (cents.empty()) cents.push_back(0.f);
    return hipst(gpann_router_create(d, metric, S, (uint32_t)node_shard.size(), node_shard.data(), cent_first.data(), child_first.data(), child_count.data(),
                                     shard_root.data(), cents.data(), device, out),
                 "gpann_router_create");
}

}   // extern "C"
