// router.hip — KMeansTreeRouter::Query on the GPU (reference src/kmeans_tree_router.cpp:93-130).
//
// Per query the reference runs a best-first walk over S k-means trees: a min-priority queue keyed on centroid distance,
// every popped node has ALL its centroids evaluated (distance(), src/dist.cpp:75-81), `min_dist[shard]` is lowered and
// the children of the node are pushed; the walk stops when the distance-evaluation budget is used up.  The result is
// the shard order by ascending min_dist.
//
// Here: one wavefront per query.
//   * distance evaluation is wave-parallel: the <=64 centroid rows of a node are fetched with coalesced 16-byte loads in
//     32-dim chunks, transposed through a wave-private LDS tile, and lane i accumulates centroid i with the reference's
//     exact unfused sequence (same arithmetic as the scan kernel) against the query row held in LDS.
//   * the priority queue is the reference's std::priority_queue<PQEntry, vector, greater<>>: lane 0 runs libstdc++'s
//     __push_heap / __adjust_heap index arithmetic (bits/stl_heap.h) on a per-wave array, so the pop order is identical
//     even among equal keys (e.g. the S roots, all pushed with key lowest(), kmeans_tree_router.cpp:103-110).
//   * min_dist is returned to the host, where the final std::sort is the same libstdc++ call the reference makes
//     (:125-127), so the order among equal min_dist (unvisited shards = FLT_MAX) is inherited, not imitated.
#include "common.h"
#include "router_bound.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <queue>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

struct gpann_router {
    // The reference's callers invoke Query / FrequencyQuery concurrently, one query per call (routes.cpp:194-200); a handle owns ONE set of
    // device workspaces, so concurrent calls on it take turns (recursive: entry points call each other).
    std::recursive_mutex call_mutex;
    gpann::Options opt;   // the handle's knobs: the environment's GPANN_* at creation, then gpann_router_set_option (common.h)
    int device = 0;
    int metric = 0;
    uint32_t d = 0, ld = 0, S = 0, n_nodes = 0;
    uint64_t n_centroids = 0;
    uint32_t max_children = 0;
    uint32_t max_node = 1;   // largest node (centroids)
    int num_cus = 256;       // cached hipDeviceProp_t::multiProcessorCount
    size_t lds_per_wg = 160u << 10;   // cached hipDeviceProp_t::sharedMemPerBlock: what one workgroup may allocate (gfx950: 160 KiB)
    float* centroids = nullptr;          // [n_centroids][ld]
    int32_t* node_shard = nullptr;
    uint32_t* node_cent_first = nullptr; // [n_nodes+1]
    uint32_t* node_child_first = nullptr;
    uint32_t* node_child_count = nullptr;
    uint32_t* shard_root = nullptr;
    uint32_t* path_cent = nullptr;       // [n_nodes][kPathMax] centroids whose distances are the keys on the path root -> node (threshold replay)
    uint32_t* cent_node = nullptr;       // [n_centroids] node of every centroid
    uint8_t* cent_node8 = nullptr;       // ... as cent_block_node[c >> 8] + cent_node8[c] (ThresholdTree)
    uint32_t* cent_block_node = nullptr;
    uint32_t *lvl_node = nullptr, *lvl_parent = nullptr, *lvl_key = nullptr;   // the nodes level by level (ThresholdTree)
    uint32_t lvl_first[8] = {};   // (kPathMax + 2)
    uint32_t n_levels = 0;
    uint32_t* cent_shard = nullptr;      // [n_centroids] shard of every centroid
    uint32_t* shard_has_cent = nullptr;  // [S] 1 if the shard has at least one centroid (its min_dist cannot stay at FLT_MAX after a full visit)
    char* cent_planes = nullptr;         // optional (d <= 128, finite centroids): bf16 planes of the centroids for RouterBoundKernel
    float* cent_nhn = nullptr;           // [n_centroids + 64] -|c|^2/2
    uint32_t plane_ks = 8;               // k-steps per plane of a centroid tile (8: d <= 128, 16: d <= 256)
    float* cent_mu = nullptr;            // squared L2: [ld] mean centroid — planes, norms and query fragments of the bound path live in the frame c - mu, q - mu
    double cent_rmax = 0.0;              // largest centroid norm
    gpann::DeviceBuffer ws_bound, ws_verify;
    uint64_t stat_bound_calls = 0, stat_bound_selfcheck = 0, stat_bound_overflow = 0, stat_verified = 0, stat_verify_failures = 0;
    uint64_t stat_probed = 0, stat_probe_failures = 0;   // GPANN_ROUTER_PROBE: blocks run twice / blocks whose two runs differed
    uint64_t stat_host_sorted = 0, stat_heap_redo = 0;
    hipEvent_t ev_bound[2] = {nullptr, nullptr};   // around the two RouterBoundKernel launches of the most recent bound block
    double stat_bound_ms = 0.0, stat_bound_flops = 0.0;   // accumulated over the blocks of the most recent call
    uint32_t max_depth = 0;              // deepest non-root level (roots = 0)
    gpann::DeviceBuffer ws_queries, ws_heap, ws_out, ws_dense, ws_redo, ws_pi, ws_frame;
    uint64_t stat_table_blocks = 0, stat_table_queries = 0, stat_table_band_nodes = 0, stat_table_exact = 0, stat_table_band_overflow = 0, stat_table_rescans = 0;
    float* pinned_min = nullptr;         // pinned staging for min_dist (a pageable D2H copy costs several times more)
    size_t pinned_floats = 0;
    int32_t* pinned_probes = nullptr;    // pinned staging for the device-sorted probe lists + one tie flag per query
    size_t pinned_probe_words = 0;
    gpann::DeviceBuffer ws_probes;
    uint32_t root_total = 0, root_min = 0, root_children = 0;   // over the S roots: sum / min of sizes, number of children
    bool leaves_only = false;                                   // every non-root node is a leaf
    hipEvent_t ev[2] = {nullptr, nullptr};
    // host copy of the flat tree: the few queries of a batch whose walk needs the real priority queue are replayed on host threads
    std::vector<int32_t> h_node_shard;
    std::vector<uint32_t> h_node_cent_first, h_node_child_first, h_node_child_count, h_shard_root;
    // a bound block whose counters (candidate slots, unset minima) have not been looked at yet: they travel to the host with the results and
    // are checked after the call's one synchronisation (a bad block is rare; the call is then repeated canonically)
    struct { bool active = false; uint32_t nb = 0, cand_cap = 0; } pending_bound;
    uint32_t* pinned_counters = nullptr;   // [5]: the words of BoundArgs::cand_count
    bool canonical_once = false;           // the repeated call: no bound path
    gpann::DeviceBuffer ws_host_redo;      // packed table rows / results of those queries
    float* pinned_redo = nullptr;          // their table rows on the host
    size_t pinned_redo_floats = 0;
};

namespace gpann {

// Host worker pool for the per-query host steps that stay libstdc++'s (the final std::sort over min_dist, DedupNeighbors):
// threads are created once per process — creating them per call costs more than the work of a 10k-query batch.
class HostPool {
public:
    static HostPool& Get() {
        static HostPool* pool = new HostPool(31);   // intentionally never destroyed (no join during static destruction)
        return *pool;
    }
    // for the walks of the marked queries of a deep-tree batch (HostReplayMarked): ~70 of 10 000 queries, a 1-2 ms walk each, one chunk
    // each — on the 32 threads above they took three rounds.  (A pool of its own: the short merges and sorts do not wake 128 threads.)
    static HostPool& Wide() {
        static HostPool* pool = new HostPool(127);
        return *pool;
    }
    // runs fn(chunk) for chunk in [0, n_chunks) on the pool and the calling thread; returns when all are done
    void Run(uint32_t n_chunks, const std::function<void(uint32_t)>& fn) {
        if (n_chunks <= 1 || workers_.empty()) {
            for (uint32_t c = 0; c < n_chunks; ++c) fn(c);
            return;
        }
        std::unique_lock<std::mutex> run_lock(run_mutex_);   // one batch at a time
        // every batch has its own claim counter: a worker that wakes up late for a finished batch finds that batch exhausted and
        // can never claim (or re-run) a chunk of the batch that follows it
        auto batch = std::make_shared<Batch>();
        batch->fn = &fn;
        batch->n_chunks = n_chunks;
        batch->pending = n_chunks;
        {
            std::lock_guard<std::mutex> lk(m_);
            current_ = batch;
            ++epoch_;
        }
        cv_work_.notify_all();
        Work(*batch);
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return batch->pending == 0; });
    }

private:
    struct Batch {
        const std::function<void(uint32_t)>* fn = nullptr;   // valid until Run returns, i.e. until every chunk has completed
        uint32_t n_chunks = 0, pending = 0;                  // pending: guarded by m_
        std::atomic<uint32_t> next{0};
    };
    explicit HostPool(unsigned max_workers) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned n = std::min(max_workers, hw > 1 ? hw - 1 : 0u);
        for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { Loop(); });
        for (auto& t : workers_) t.detach();
    }
    void Work(Batch& b) {
        for (;;) {
            const uint32_t c = b.next.fetch_add(1);
            if (c >= b.n_chunks) return;
            (*b.fn)(c);
            std::lock_guard<std::mutex> lk(m_);
            if (--b.pending == 0) cv_done_.notify_all();
        }
    }
    void Loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Batch> batch;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                batch = current_;
            }
            Work(*batch);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, run_mutex_;
    std::condition_variable cv_work_, cv_done_;
    std::shared_ptr<Batch> current_;   // guarded by m_
    uint64_t epoch_ = 0;
};

constexpr int kRDC = 32;                 // dims per staged chunk
constexpr int kRStride = kRDC + 4;       // LDS row stride (floats)
constexpr int kRWaves = 4;               // wavefronts (= queries in flight) per workgroup

struct RouterArgs {
    const float* centroids;
    const int32_t* node_shard;
    const uint32_t* node_cent_first;
    const uint32_t* node_child_first;
    const uint32_t* node_child_count;
    const uint32_t* shard_root;
    const float* queries;     // [nq][ld]
    float* heap_key;          // [slots][heap_cap]
    uint32_t* heap_node;      // [slots][heap_cap]
    float* out_min_dist;      // [nq][S]
    float* rec_dist;          // optional [nq][rec_cap]: every evaluated centroid distance, in evaluation order (FrequencyQuery)
    int32_t* rec_shard;       // optional [nq][rec_cap]
    uint32_t* rec_count;      // optional [nq]
    uint64_t rec_cap;
    uint32_t d, ld, S, nq;
    uint64_t heap_cap;
    int32_t budget;
};

// comp(a, b) of std::greater<PQEntry> with PQEntry::operator> comparing dist only (kmeans_tree_router.cpp:98)
__device__ __forceinline__ bool HeapComp(float a, float b) { return a > b; }

// libstdc++ std::__push_heap (bits/stl_heap.h:135-148)
__device__ __forceinline__ void PushHeap(float* key, uint32_t* node, int64_t hole, int64_t top, float vkey, uint32_t vnode) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && HeapComp(key[parent], vkey)) {
        key[hole] = key[parent];
        node[hole] = node[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    key[hole] = vkey;
    node[hole] = vnode;
}

// libstdc++ std::__adjust_heap (bits/stl_heap.h:219-245)
__device__ __forceinline__ void AdjustHeap(float* key, uint32_t* node, int64_t hole, int64_t len, float vkey, uint32_t vnode) {
    const int64_t top = hole;
    int64_t second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (HeapComp(key[second], key[second - 1])) second--;
        key[hole] = key[second];
        node[hole] = node[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        key[hole] = key[second - 1];
        node[hole] = node[second - 1];
        hole = second - 1;
    }
    PushHeap(key, node, hole, top, vkey, vnode);
}

// LDS_HEAP: the priority queue of every wave lives in LDS (a.heap_cap entries per wave, appended to the wave's LDS slice)
// instead of the global workspace.  Same algorithm; lane 0's dependent heap accesses cost ~100 cycles instead of ~1000+.
template <int METRIC, bool LDS_HEAP>
__global__ void __launch_bounds__(64 * kRWaves) RouterWalkKernel(RouterArgs a) {
    extern __shared__ float4 rsmem4[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t d = a.d, ld = a.ld, S = a.S;
    // wave-private LDS: query row [ld] | centroid chunk tile [64][36] | node distances [64] | min_dist [S]
    const size_t per_wave = (size_t)ld + 64 * kRStride + 64 + S + (LDS_HEAP ? 2 * a.heap_cap : 0);
    float* sQ = reinterpret_cast<float*>(rsmem4) + (size_t)wave * ((per_wave + 3) & ~(size_t)3);
    float* sC = sQ + ld;
    float* sDist = sC + 64 * kRStride;
    float* sMin = sDist + 64;

    const uint32_t slot = blockIdx.x * kRWaves + wave;
    const uint32_t n_slots = gridDim.x * kRWaves;
    float* hkey;
    uint32_t* hnode;
    if (LDS_HEAP) {
        hkey = sMin + S;
        hnode = reinterpret_cast<uint32_t*>(hkey + a.heap_cap);
    } else {
        hkey = a.heap_key + (size_t)slot * a.heap_cap;
        hnode = a.heap_node + (size_t)slot * a.heap_cap;
    }
    const uint32_t full_groups = d >> 2;
    const int tail = (int)(d & 3u);
    const uint32_t nchunks = (ld + kRDC - 1) / kRDC;

    for (uint32_t q = slot; q < a.nq; q += n_slots) {
        for (uint32_t c = lane; c < ld; c += 64) sQ[c] = a.queries[(size_t)q * ld + c];
        for (uint32_t b = lane; b < S; b += 64) sMin[b] = FLT_MAX;   // :101
        int64_t size = 0;
        if (lane == 0) {
            for (uint32_t r = 0; r < S; ++r) {   // :103-110, key = numeric_limits<float>::lowest()
                PushHeap(hkey, hnode, size, 0, -FLT_MAX, a.shard_root[r]);
                size++;
            }
        }
        size = S;
        __builtin_amdgcn_wave_barrier();
        int32_t budget = a.budget;
        uint32_t n_rec = 0;
        while (size > 0 && budget > 0) {   // :112
            uint32_t node = 0;
            if (lane == 0) {
                node = hnode[0];   // pq.top()
                if (size > 1) {    // std::pop_heap
                    float vk = hkey[size - 1];
                    uint32_t vn = hnode[size - 1];
                    hkey[size - 1] = hkey[0];
                    hnode[size - 1] = hnode[0];
                    AdjustHeap(hkey, hnode, 0, size - 1, vk, vn);
                }
            }
            size--;
            node = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
            const uint32_t c_first = a.node_cent_first[node];
            const uint32_t n = a.node_cent_first[node + 1] - c_first;
            const uint32_t n_child = a.node_child_count[node];
            const uint32_t child_first = a.node_child_first[node];
            const int shard = a.node_shard[node];
            budget -= (int32_t)n;   // :115
            for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t m = min(64u, n - base);
                float acc = 0.f;
                for (uint32_t ch = 0; ch < nchunks; ++ch) {
                    // stage [m rows][32 dims] of the node's centroids: lane handles float4 #(lane + 64 i) of 512
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int f = lane + 64 * i;
                        const uint32_t row = (uint32_t)(f >> 3);
                        const uint32_t c = ch * kRDC + 4u * (uint32_t)(f & 7);
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (row < m && c < ld) v = *reinterpret_cast<const float4*>(a.centroids + (size_t)(c_first + base + row) * ld + c);
                        *reinterpret_cast<float4*>(sC + row * kRStride + 4 * (f & 7)) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t g0 = ch * (kRDC / 4);
                    for (int g = 0; g < kRDC / 4; ++g) {
                        const uint32_t G = g0 + g;
                        if (G > full_groups || (G == full_groups && tail == 0)) break;
                        const float4 pv = *reinterpret_cast<const float4*>(sC + lane * kRStride + 4 * g);
                        const float4 qv = *reinterpret_cast<const float4*>(sQ + 4 * G);
                        if (G < full_groups) {
                            if (METRIC == GPANN_METRIC_L2) {
                                float d0 = pv.x - qv.x, d1 = pv.y - qv.y, d2 = pv.z - qv.z, d3 = pv.w - qv.w;
                                float s = d0 * d0 + d1 * d1;
                                s = s + d2 * d2;
                                s = s + d3 * d3;
                                acc = acc + s;
                            } else {
                                acc = acc + qv.x * pv.x;
                                acc = acc + qv.y * pv.y;
                                acc = acc + qv.z * pv.z;
                                acc = acc + qv.w * pv.w;
                            }
                        } else {
                            if (tail > 0) acc = TailStep<METRIC>(acc, pv.x, qv.x);
                            if (tail > 1) acc = TailStep<METRIC>(acc, pv.y, qv.y);
                            if (tail > 2) acc = TailStep<METRIC>(acc, pv.z, qv.z);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                float dist = METRIC == GPANN_METRIC_MIPS ? 1.0f - acc : acc;
                if (a.rec_dist && (uint32_t)lane < m) {   // kmeans_tree_router.cpp:158: every (dist, shard) is offered to the vote list
                    a.rec_dist[(size_t)q * a.rec_cap + n_rec + lane] = dist;
                    a.rec_shard[(size_t)q * a.rec_cap + n_rec + lane] = shard;
                }
                n_rec += m;
                if ((uint32_t)lane >= m) dist = FLT_MAX;
                sDist[lane] = dist;
                // min over the node's centroids (:118; std::min keeps the old value on ties and ignores NaN the same way)
                float mn = dist;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    if (mn < sMin[shard]) sMin[shard] = mn;
                    // children of centroids base .. base+m) in centroid order (:119-121)
                    for (uint32_t i = base; i < base + m && i < n_child; ++i) {
                        PushHeap(hkey, hnode, size + (int64_t)(i - base), 0, sDist[i - base], child_first + i);
                    }
                }
                if (n_child > base) size += (int64_t)(min(n_child, base + m) - base);
                __builtin_amdgcn_wave_barrier();
            }
        }
        for (uint32_t b = lane; b < S; b += 64) a.out_min_dist[(size_t)q * S + b] = sMin[b];
        if (a.rec_count && lane == 0) a.rec_count[q] = n_rec;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- dense evaluation + replay ------------------------------------------------------------------------------------------
// When the search budget is of the order of the tree size, the walk visits most nodes anyway and re-fetching every
// node's centroids per query (the kernel above) is bound by L2 traffic (nq x visited centroids x row bytes).  The dense
// path evaluates D[q][c] for ALL tree centroids with the register-tiled exact kernel (each centroid row is fetched once
// per 64 queries) and then REPLAYS the reference's walk on the precomputed distances: same priority queue, same budget
// rule, same pushes — only distance() is a table lookup.  Bit-identical results; chosen by LaunchWalk().
struct DenseArgs {
    const float* centroids;   // [C][ld]
    const float* queries;     // [nq][ld]
    float* out;               // [nq][C]
    uint32_t d, ld, nq, C;
    // MIN_ONLY: instead of the table, the minimum over every shard's centroids as order-preserving bits
    const uint32_t* cent_shard;   // [C]
    uint32_t* out_min_bits;       // [nq][S], initialised to OrderedBits(FLT_MAX)
    uint32_t S;
};

__device__ __forceinline__ uint32_t OrderedBitsD(float f) {   // unsigned order == float order
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// MIN_ONLY is the whole walk when the search budget exceeds the tree (kmeans_tree_router.cpp:112-123 then visits every node):
// min_dist[shard] = min over all of the shard's centroids, taken in the tile's epilogue — no table, no replay.
template <int METRIC, bool MIN_ONLY>
__global__ void __launch_bounds__(256, 3) DenseDistKernel(DenseArgs a) {
    extern __shared__ float4 dsm4[];
    constexpr int kStride = kRDC + 4;
    constexpr int kPairStride = 2 * kRDC + 4;       // centroid rows as pairs (common.h, AccumulatePairs)
    float* sQ = reinterpret_cast<float*>(dsm4);   // [64][36]
    float* sP = sQ + 64 * kStride;                 // [32 pairs][68]
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const uint32_t q0 = blockIdx.y * 64u, c0 = blockIdx.x * 64u;
    const uint32_t d = a.d, ld = a.ld;
    const uint32_t full_groups = d >> 2;
    const int tail = (int)(d & 3u);
    const uint32_t nchunks = (ld + kRDC - 1) / kRDC;
    const int st_row0 = tid >> 3, st_row1 = (tid + 256) >> 3, st_c4 = tid & 7;
    // centroid staging role: pair st_pair = rows st_prow0 and st_prow0 + 16 of the tile, float4 #st_c4 of both
    const int st_pair = tid >> 3;
    const int st_prow0 = (st_pair & 15) + 32 * (st_pair >> 4);
    const float* qrow0 = a.queries + (size_t)min(q0 + st_row0, a.nq - 1) * ld;
    const float* qrow1 = a.queries + (size_t)min(q0 + st_row1, a.nq - 1) * ld;
    const float* prow0 = a.centroids + (size_t)min(c0 + st_prow0, a.C - 1) * ld;
    const float* prow1 = a.centroids + (size_t)min(c0 + st_prow0 + 16, a.C - 1) * ld;
    f32x2_t acc2[4][2];   // acc2[i][pr] = query ty + 16 i x centroids (tx + 32 pr, tx + 32 pr + 16)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) acc2[i][pr] = (f32x2_t){0.f, 0.f};
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gq0, gq1, gp0, gp1;
    {
        const uint32_t c = 4u * st_c4;
        const bool in = c < ld;
        gq0 = in ? *reinterpret_cast<const float4*>(qrow0 + c) : zero4;
        gq1 = in ? *reinterpret_cast<const float4*>(qrow1 + c) : zero4;
        gp0 = in ? *reinterpret_cast<const float4*>(prow0 + c) : zero4;
        gp1 = in ? *reinterpret_cast<const float4*>(prow1 + c) : zero4;
    }
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        __syncthreads();
        *reinterpret_cast<float4*>(sQ + st_row0 * kStride + 4 * st_c4) = gq0;
        *reinterpret_cast<float4*>(sQ + st_row1 * kStride + 4 * st_c4) = gq1;
        // the two rows of the pair, interleaved dimension by dimension
        *reinterpret_cast<float4*>(sP + st_pair * kPairStride + 8 * st_c4) = make_float4(gp0.x, gp1.x, gp0.y, gp1.y);
        *reinterpret_cast<float4*>(sP + st_pair * kPairStride + 8 * st_c4 + 4) = make_float4(gp0.z, gp1.z, gp0.w, gp1.w);
        __syncthreads();
        if (ch + 1 < nchunks) {   // the next chunk's global loads complete under the arithmetic below
            const uint32_t c = (ch + 1) * kRDC + 4u * st_c4;
            const bool in = c < ld;
            gq0 = in ? *reinterpret_cast<const float4*>(qrow0 + c) : zero4;
            gq1 = in ? *reinterpret_cast<const float4*>(qrow1 + c) : zero4;
            gp0 = in ? *reinterpret_cast<const float4*>(prow0 + c) : zero4;
            gp1 = in ? *reinterpret_cast<const float4*>(prow1 + c) : zero4;
        }
        const uint32_t g0 = ch * (kRDC / 4);
        const float* q_base = sQ + ty * kStride;
        const float* p_base = sP + tx * kPairStride;   // pairs tx (centroids tx, tx + 16) and tx + 16 (centroids tx + 32, tx + 48)
        const bool whole = g0 + kRDC / 4 <= full_groups;
#pragma unroll 2
        for (int g = 0; g < kRDC / 4; ++g) {
            const uint32_t G = g0 + g;
            if (!whole && (G > full_groups || (G == full_groups && tail == 0))) break;
            float4 qv[4];
            f32x4_t p0[2], p1[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const float4*>(q_base + 16 * i * kStride + 4 * g);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                p0[pr] = *reinterpret_cast<const f32x4_t*>(p_base + 16 * pr * kPairStride + 8 * g);
                p1[pr] = *reinterpret_cast<const f32x4_t*>(p_base + 16 * pr * kPairStride + 8 * g + 4);
            }
            if (whole || G < full_groups) AccumulatePairs<METRIC>(acc2, qv, p0, p1);
            else AccumulatePairsTail<METRIC>(acc2, qv, p0, p1, tail);
        }
    }
    float acc[4][4];   // acc[i][j]: query ty + 16 i x centroid tx + 16 j
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            acc[i][2 * pr] = acc2[i][pr].x;
            acc[i][2 * pr + 1] = acc2[i][pr].y;
        }
    if (MIN_ONLY) {
        const uint32_t shard_first = a.cent_shard[c0], shard_last = a.cent_shard[min(c0 + 63u, a.C - 1)];
        if (shard_first == shard_last) {   // (centroids are numbered shard by shard) the usual case: one shard per tile
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t q = q0 + ty + 16 * i;
                float m = INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[i][j];
                    if (METRIC == GPANN_METRIC_MIPS) v = 1.0f - v;
                    if (c0 + tx + 16 * j < a.C) m = fminf(m, v);
                }
                // the 16 threads of a query are 16 consecutive lanes
                m = fminf(m, __shfl_xor(m, 8));
                m = fminf(m, __shfl_xor(m, 4));
                m = fminf(m, __shfl_xor(m, 2));
                m = fminf(m, __shfl_xor(m, 1));
                if (tx == 0 && q < a.nq) atomicMin(&a.out_min_bits[(size_t)q * a.S + shard_first], OrderedBitsD(m));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t q = q0 + ty + 16 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t c = c0 + tx + 16 * j;
                    float v = acc[i][j];
                    if (METRIC == GPANN_METRIC_MIPS) v = 1.0f - v;
                    if (q < a.nq && c < a.C) atomicMin(&a.out_min_bits[(size_t)q * a.S + a.cent_shard[c]], OrderedBitsD(v));
                }
            }
        }
        return;
    }
    // the 64 x 64 tile goes out through LDS: one store instruction per table row and wavefront writes 64 consecutive floats (written
    // from the register tile, an instruction wrote 16 floats of 4 different rows)
    constexpr int kOutStride = 65;
    float* sT = reinterpret_cast<float*>(dsm4);   // [64][65] (the staging buffers are done with)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
            if (METRIC == GPANN_METRIC_MIPS) v = 1.0f - v;
            sT[(ty + 16 * i) * kOutStride + tx + 16 * j] = v;
        }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t c = c0 + (uint32_t)lane;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const uint32_t q = q0 + (uint32_t)(16 * wave + r);
        if (q < a.nq && c < a.C) a.out[(size_t)q * a.C + c] = sT[(16 * wave + r) * kOutStride + lane];
    }
}

// min_dist workspace <-> order-preserving bits (DenseDistKernel<.., true>)
__global__ void FillMinBitsKernel(uint32_t* out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = 0xFF7FFFFFu;   // OrderedBits(FLT_MAX)
}
// (query, shard) minima a full visit has left at FLT_MAX although the shard has centroids — counted into *unset (self-check of the
// matrix-core bound path: the caller then evaluates every pair canonically instead)
__global__ void CountUnsetMinBitsKernel(const uint32_t* bits, size_t n, uint32_t S, const uint32_t* shard_has_cent, uint32_t* unset) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && bits[i] == 0xFF7FFFFFu && shard_has_cent[i % S] != 0u) atomicAdd(unset, 1u);
}
// ... the same and, in the same pass, the ordered bits -> floats (the shipping path of the bound: one launch instead of two)
__global__ void CountUnsetAndConvertKernel(uint32_t* io, size_t n, uint32_t S, const uint32_t* shard_has_cent, uint32_t* unset) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = io[i];
    if (o == 0xFF7FFFFFu && shard_has_cent[i % S] != 0u) atomicAdd(unset, 1u);
    io[i] = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
}
__global__ void MinBitsToFloatKernel(uint32_t* io, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t o = io[i];
        io[i] = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    }
}

// FrequencyQuery's TopN(num_voting) on the device: the V lexicographically smallest (distance, shard) pairs of a query's
// evaluated centroids, ascending (= TopN::Take() then FrequencyQueryData::Init(), kmeans_tree_router.h:19-21).  One workgroup per
// query: 8-pass byte-wise radix select of the V-th smallest 64-bit key (order-preserving distance bits << 32 | shard), then the
// keys below it (and as many copies of it as are still missing) are sorted in LDS.  V <= 2048.
__global__ void __launch_bounds__(256) VoteSelectKernel(const float* rec_dist, const int32_t* rec_shard, const uint32_t* rec_count, uint64_t rec_cap,
                                                        uint32_t V, float* out_dist, uint32_t* out_shard, uint32_t* out_count) {
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long buf[2048];
    __shared__ uint32_t s_cursor, s_bin, s_below;
    const uint32_t q = blockIdx.x;
    const uint32_t cnt = rec_count[q];
    const float* dist = rec_dist + (size_t)q * rec_cap;
    const int32_t* shard = rec_shard + (size_t)q * rec_cap;
    const uint32_t keep = min(V, cnt);
    auto key_of = [&](uint32_t i) -> unsigned long long { return ((unsigned long long)OrderedBitsD(dist[i]) << 32) | (uint32_t)shard[i]; };
    uint32_t N = 1;
    while (N < max(keep, 2u)) N <<= 1;
    unsigned long long kth = ~0ull;
    uint32_t below = 0;   // keys strictly below the prefix found so far
    if (cnt > V) {
        unsigned long long prefix = 0, mask = 0;
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            hist[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
                const unsigned long long k = key_of(i);
                if ((k & mask) == prefix) atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t acc = below, b = 0;
                for (; b < 256; ++b) {
                    if (acc + hist[b] >= V) break;
                    acc += hist[b];
                }
                s_bin = b;
                s_below = acc;
            }
            __syncthreads();
            prefix |= (unsigned long long)s_bin << shift;
            mask |= 255ull << shift;
            below = s_below;
            __syncthreads();
        }
        kth = prefix;   // the V-th smallest key; `below` keys are smaller
    }
    if (threadIdx.x == 0) s_cursor = 0;
    for (uint32_t i = threadIdx.x; i < N; i += 256) buf[i] = ~0ull;
    __syncthreads();
    if (cnt > V) {
        for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
            const unsigned long long k = key_of(i);
            if (k < kth) buf[atomicAdd(&s_cursor, 1u)] = k;
        }
        __syncthreads();
        for (uint32_t i = below + threadIdx.x; i < V; i += 256) buf[i] = kth;   // equal pairs are interchangeable
    } else {
        for (uint32_t i = threadIdx.x; i < cnt; i += 256) buf[i] = key_of(i);
    }
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= N; k2 <<= 1)
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += 256) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t p = i | j;
                const bool up = (i & k2) == 0;
                const unsigned long long ei = buf[i], ep = buf[p];
                if ((ei > ep) == up) {
                    buf[i] = ep;
                    buf[p] = ei;
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < keep; i += 256) {
        const unsigned long long k = buf[i];
        const uint32_t o = (uint32_t)(k >> 32);
        out_dist[(size_t)q * V + i] = __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
        out_shard[(size_t)q * V + i] = (uint32_t)k;
    }
    for (uint32_t i = keep + threadIdx.x; i < V; i += 256) {
        out_dist[(size_t)q * V + i] = INFINITY;
        out_shard[(size_t)q * V + i] = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) out_count[q] = keep;
}

struct ReplayArgs {
    const float* dense;       // [nq][C] exact distances
    const int32_t* node_shard;
    const uint32_t* node_cent_first;
    const uint32_t* node_child_first;
    const uint32_t* node_child_count;
    const uint32_t* shard_root;
    float* heap_key;          // global queue workspace (LDS_HEAP == false)
    uint32_t* heap_node;
    float* out_min_dist;      // [nq][S]
    float* rec_dist;          // optional, see RouterArgs
    int32_t* rec_shard;
    uint32_t* rec_count;
    uint64_t rec_cap;
    uint64_t heap_cap;
    uint32_t S, nq, C;
    int32_t budget;
    uint32_t* redo;           // [nq] RouterSortedReplayKernel: 1 = needs the real priority queue; RouterReplayKernel with only_redo:
                              // process only those
    uint32_t only_redo;
    uint32_t* redo_count;     // number of queries RouterSortedReplayKernel marked
    uint32_t root_total;      // sum of the root node sizes
    uint32_t sort_n;          // power of two >= number of root children (frontier capacity of the sorted replay)
    uint32_t n_nodes;
};

__device__ __forceinline__ uint32_t OrderedBits(float f) {   // unsigned order == float order
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float FromOrderedBits(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// ---- sorted replay: trees whose root children are all leaves ------------------------------------------------------------------
// After the S roots have been opened (always first: their keys are lowest()), nothing is pushed any more, so the walk pops the
// root children in ascending key order until the budget is used up: no queue is needed.  One wavefront per query:
//   roots (coalesced reads of D, min per shard, children keys into LDS) -> bitonic sort of the <= 2048 keys in LDS ->
//   prefix sum of node sizes against the budget (:112-115: a node is popped iff the budget BEFORE it is > 0) ->
//   one lane per popped node takes the minimum of its D entries into min_dist[shard] (LDS atomic min on order-preserving bits).
// libstdc++'s tie order among equal keys only matters if an equal key straddles the cut-off; such queries (and budgets too
// small to open every root) are marked in `redo` and done by RouterReplayKernel with the real heap.
__global__ void __launch_bounds__(64 * kRWaves) RouterSortedReplayKernel(ReplayArgs a) {
    extern __shared__ float4 rpsm4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t S = a.S, N = a.sort_n, n_nodes = a.n_nodes;
    // workgroup-shared node tables (the walk's only other global reads would be dependent metadata loads per node)
    uint32_t* sCf = reinterpret_cast<uint32_t*>(rpsm4);            // [n_nodes + 1] first centroid of a node
    uint16_t* sShard = reinterpret_cast<uint16_t*>(sCf + n_nodes + 1);   // [n_nodes]
    const size_t table_words = ((size_t)(n_nodes + 1) + (n_nodes + 1) / 2 + 3) & ~(size_t)3;
    const size_t per_wave = ((size_t)2 * N + S + 3) & ~(size_t)3;
    // frontier entry = (order-preserving key bits << 32) | node: one 64-bit compare sorts by (key, node)
    unsigned long long* fent = reinterpret_cast<unsigned long long*>(reinterpret_cast<float*>(rpsm4) + table_words + (size_t)wave * per_wave);
    uint32_t* sMinOrd = reinterpret_cast<uint32_t*>(fent + N);
    for (uint32_t u = threadIdx.x; u <= n_nodes; u += blockDim.x) sCf[u] = a.node_cent_first[u];
    for (uint32_t u = threadIdx.x; u < n_nodes; u += blockDim.x) sShard[u] = (uint16_t)a.node_shard[u];
    __syncthreads();
    // lane r keeps root r (S <= 64 on this path)
    uint32_t my_root = 0, my_nchild = 0, my_child_first = 0;
    if ((uint32_t)lane < S) {
        my_root = a.shard_root[lane];
        my_nchild = a.node_child_count[my_root];
        my_child_first = a.node_child_first[my_root];
    }
    const uint32_t slot = blockIdx.x * kRWaves + wave, n_slots = gridDim.x * kRWaves;
    for (uint32_t q = slot; q < a.nq; q += n_slots) {
        if (lane == 0) a.redo[q] = 0;
        const float* D = a.dense + (size_t)q * a.C;
        for (uint32_t b = lane; b < S; b += 64) sMinOrd[b] = OrderedBits(FLT_MAX);
        __builtin_amdgcn_wave_barrier();
        uint32_t size = 0, n_rec = 0;
        for (uint32_t r = 0; r < S; ++r) {
            const uint32_t node = (uint32_t)__shfl((int)my_root, r);
            const uint32_t n_child = (uint32_t)__shfl((int)my_nchild, r);
            const uint32_t child_first = (uint32_t)__shfl((int)my_child_first, r);
            const uint32_t c_first = sCf[node];
            const uint32_t n = sCf[node + 1] - c_first;
            const int shard = sShard[node];
            for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t m = min(64u, n - base);
                float dist = FLT_MAX;
                if ((uint32_t)lane < m) {
                    dist = D[c_first + base + lane];
                    if (a.rec_dist) {
                        a.rec_dist[(size_t)q * a.rec_cap + n_rec + lane] = dist;
                        a.rec_shard[(size_t)q * a.rec_cap + n_rec + lane] = shard;
                    }
                    if (base + lane < n_child) fent[size + lane] = ((unsigned long long)OrderedBits(dist) << 32) | (child_first + base + lane);
                }
                n_rec += m;
                if (n_child > base) size += min(n_child, base + m) - base;
                float mn = dist;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
                if (lane == 0) atomicMin(&sMinOrd[shard], OrderedBits(mn));
            }
        }
        for (uint32_t i = size + lane; i < N; i += 64) fent[i] = ~0ull;
        __builtin_amdgcn_wave_barrier();
        const int64_t budget_left = (int64_t)a.budget - (int64_t)a.root_total;
        bool redo = false;
        if (budget_left > 0 && size > 0) {
            // bitonic sort ascending by (key, node)
            for (uint32_t k2 = 2; k2 <= N; k2 <<= 1) {
                for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
                    for (uint32_t t = lane; t < N / 2; t += 64) {
                        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // lower index of the pair
                        const uint32_t p = i | j;
                        const bool up = (i & k2) == 0;
                        const unsigned long long ei = fent[i], ep = fent[p];
                        if ((ei > ep) == up) {
                            fent[i] = ep;
                            fent[p] = ei;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // how many nodes are popped: node i iff budget_left - (sizes of nodes 0..i-1) > 0
            int64_t used = 0;
            uint32_t cut = 0;
            bool open_end = true;
            for (uint32_t base = 0; base < size && open_end; base += 64) {
                const uint32_t i = base + lane;
                uint32_t ni = 0;
                if (i < size) {
                    const uint32_t node = (uint32_t)fent[i];
                    ni = sCf[node + 1] - sCf[node];
                }
                uint32_t incl = ni;   // inclusive scan over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                const bool popped = i < size && budget_left - (used + (int64_t)(incl - ni)) > 0;
                const unsigned long long pm = __ballot(popped);
                cut += (uint32_t)__popcll(pm);
                used += (int64_t)__shfl((int)incl, 63);
                open_end = pm == ~0ull;
            }
            // equal keys across the cut-off: the reference's queue layout decides which of them is popped
            if (cut > 0 && cut < size && (uint32_t)(fent[cut - 1] >> 32) == (uint32_t)(fent[cut] >> 32)) redo = true;
            if (!redo) {
                // minima of the popped nodes: 8 lanes per node, 4 independent loads in flight per lane
                const int grp = lane >> 3, sub = lane & 7;
                for (uint32_t i0 = 0; i0 < cut; i0 += 8) {
                    const uint32_t i = i0 + grp;
                    float mn = FLT_MAX;
                    int shard = 0;
                    if (i < cut) {
                        const uint32_t node = (uint32_t)fent[i];
                        const uint32_t c_first = sCf[node];
                        const uint32_t n = sCf[node + 1] - c_first;
                        shard = sShard[node];
                        const float* row = D + c_first;
                        for (uint32_t e = sub; e < n; e += 32) {
                            const float v0 = row[e];
                            const float v1 = e + 8 < n ? row[e + 8] : FLT_MAX;
                            const float v2 = e + 16 < n ? row[e + 16] : FLT_MAX;
                            const float v3 = e + 24 < n ? row[e + 24] : FLT_MAX;
                            mn = fminf(mn, fminf(fminf(v0, v1), fminf(v2, v3)));
                        }
                    }
                    mn = fminf(mn, __shfl_xor(mn, 1));
                    mn = fminf(mn, __shfl_xor(mn, 2));
                    mn = fminf(mn, __shfl_xor(mn, 4));
                    if (i < cut && sub == 0) atomicMin(&sMinOrd[shard], OrderedBits(mn));
                }
                if (a.rec_dist) {   // FrequencyQuery: every evaluated (distance, shard) pair; the order is irrelevant
                    for (uint32_t i = 0; i < cut; ++i) {
                        const uint32_t node = (uint32_t)fent[i];
                        const uint32_t c_first = sCf[node];
                        const uint32_t n = sCf[node + 1] - c_first;
                        const int shard = sShard[node];
                        for (uint32_t e = lane; e < n; e += 64) {
                            a.rec_dist[(size_t)q * a.rec_cap + n_rec + e] = D[c_first + e];
                            a.rec_shard[(size_t)q * a.rec_cap + n_rec + e] = shard;
                        }
                        n_rec += n;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (redo) {
            if (lane == 0) {
                a.redo[q] = 1;
                atomicAdd(a.redo_count, 1u);
            }
        } else {
            for (uint32_t b = lane; b < S; b += 64) a.out_min_dist[(size_t)q * S + b] = FromOrderedBits(sMinOrd[b]);
            if (a.rec_count && lane == 0) a.rec_count[q] = n_rec;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- threshold replay: any tree -------------------------------------------------------------------------------------------
// The walk pops the queue's minimum, and a node enters the queue when its parent is popped.  Let key(u) be the distance that
// is u's queue key and  pi(u) = max of the keys on the path from u's root child down to u  (roots: lowest()).  A node v
// cannot be popped while a node u with pi(u) < pi(v) is still waiting: when the ancestor of v that carries the key pi(v) was
// popped, the first unpopped node on u's path was in the queue with a smaller key.  So nodes leave the queue in order of
// pi, and the walk — pop while budget > 0 (:112-115) — visits exactly
//     the roots,  every node with pi < x*,  and part of the group pi == x*
// where x* is the pi value at which the node sizes, accumulated in pi order, reach the budget left after the roots.  None of
// this needs a queue: per query, one wavefront
//   1. computes pi(u) for every node from the dense distance row (independent loads along a precomputed path table),
//   2. finds x* with a 4-pass byte-wise radix select over (pi bits, node size) histograms in LDS,
//   3. walks the group pi == x*: its bottleneck node w (key(w) == x*) and, budget permitting, w's descendants with smaller
//      keys, best-first from a small frontier list in LDS (typically a handful of nodes),
//   4. takes per-shard minima over the centroids of all visited nodes (one coalesced pass over the distance row).
// Equal keys make the order libstdc++'s business only if they sit in the cut-off group (two bottlenecks, or equal minima in
// the group's frontier); those queries are marked in `redo` and replayed by RouterReplayKernel with the real heap.
constexpr int kPathMax = 6;
static_assert(kPathMax + 2 == 8, "gpann_router::lvl_first");
constexpr int kBatch = 4;   // independent loop trips a lane of RouterThresholdReplayKernel keeps in flight
constexpr uint32_t kFrontCap = 512;
constexpr uint32_t kNoCent = 0xFFFFFFFFu;

struct ThresholdTree {
    const uint32_t* path_cent;   // [n_nodes][kPathMax]
    const uint32_t* cent_node;   // [C]
    uint32_t* pi;                // [slots][n_nodes] scratch
    uint32_t visit_all;          // the budget exceeds the tree: every node is visited, steps 1-3 are skipped
    // cent_node in a quarter of the bytes (the bounded replay streams it next to the table row): node(c) = cent_block_node[c >> 8] + cent_node8[c]
    // (every node has a centroid, so the 256 centroids of a block belong to at most 256 consecutive nodes)
    const uint8_t* cent_node8;        // [C + 4]
    const uint32_t* cent_block_node;  // [(C >> 8) + 2]
    // the nodes level by level (roots = level 0), ascending node ids inside a level: pi(u) = max(pi(parent), key(u)) needs one table value
    // per node when the parents' level is done — the path table costs six gathers per node, and the texture addresser was the busiest unit
    const uint32_t* lvl_node;     // [n_nodes]
    const uint32_t* lvl_parent;   // [n_nodes] (roots: kNoCent)
    const uint32_t* lvl_key;      // [n_nodes] the centroid whose distance is the node's key (roots: kNoCent)
    uint32_t lvl_first[kPathMax + 2];   // level l = entries lvl_first[l] .. lvl_first[l + 1]
    uint32_t n_levels;
};

__global__ void __launch_bounds__(64 * kRWaves) RouterThresholdReplayKernel(ReplayArgs a, ThresholdTree t) {
    extern __shared__ uint32_t thsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t S = a.S, n_nodes = a.n_nodes, C = a.C;
    const size_t per_wave = ((size_t)256 + S + 2 * kFrontCap + 3) & ~(size_t)3;
    uint32_t* hist = thsm + (size_t)wave * per_wave;
    uint32_t* sMinOrd = hist + 256;
    uint32_t* fkey = sMinOrd + S;
    uint32_t* fnode = fkey + kFrontCap;
    const uint32_t slot = blockIdx.x * kRWaves + wave, n_slots = gridDim.x * kRWaves;
    uint32_t* pi = t.pi + (size_t)slot * n_nodes;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;

    for (uint32_t q = slot; q < a.nq; q += n_slots) {
        const float* D = a.dense + (size_t)q * C;
        for (uint32_t b = lane; b < S; b += 64) sMinOrd[b] = OrderedBits(FLT_MAX);

        // ---- 1. pi (order-preserving bits; 0 = root, later also "visited").  The loops of this kernel are chains of dependent loads
        //      (path table -> distance row, pi -> node sizes, centroid -> node -> pi): four nodes per lane and trip are in flight
        //      together — one at a time the wavefront spent its life waiting, ~1.5 us per trip, 5.7 ms per query.
        for (uint32_t u0 = lane; u0 < n_nodes && !t.visit_all; u0 += 64 * kBatch) {
            uint32_t cidx[kBatch][kPathMax];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const uint32_t u = u0 + 64u * k;
#pragma unroll
                for (int j = 0; j < kPathMax; ++j) cidx[k][j] = u < n_nodes ? t.path_cent[(size_t)u * kPathMax + j] : kNoCent;
            }
            float val[kBatch][kPathMax];
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
#pragma unroll
                for (int j = 0; j < kPathMax; ++j) val[k][j] = D[cidx[k][j] != kNoCent ? cidx[k][j] : 0u];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const uint32_t u = u0 + 64u * k;
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < kPathMax; ++j)
                    if (cidx[k][j] != kNoCent) m = max(m, OrderedBits(val[k][j]));
                if (u < n_nodes) pi[u] = m;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- 2. x*: the smallest pi value at which the accumulated sizes reach the budget left after the roots
        const int64_t budget_left = (int64_t)a.budget - (int64_t)a.root_total;
        bool all = t.visit_all != 0, redo = false;
        uint32_t xs = 1;          // visited <=> pi < xs; 1 = roots only
        uint32_t w_lt = 0;        // sum of the sizes with pi < xs
        if (budget_left > 0 && !all) {
            const uint32_t need = (uint32_t)min(budget_left, (int64_t)0x7FFFFFFF);
            uint32_t prefix = 0, prefix_mask = 0;
            for (int pass = 0; pass < 4 && !all; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int b = lane; b < 256; b += 64) hist[b] = 0;
                __builtin_amdgcn_wave_barrier();
                for (uint32_t u0 = lane; u0 < n_nodes; u0 += 64 * kBatch) {
                    uint32_t p[kBatch], sz[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const uint32_t u = min(u0 + 64u * k, n_nodes - 1);
                        p[k] = u0 + 64u * k < n_nodes ? pi[u] : 0u;
                        sz[k] = a.node_cent_first[u + 1] - a.node_cent_first[u];
                    }
#pragma unroll
                    for (int k = 0; k < kBatch; ++k)
                        if (p[k] != 0 && (p[k] & prefix_mask) == prefix) atomicAdd(&hist[(p[k] >> shift) & 255u], sz[k]);
                }
                __builtin_amdgcn_wave_barrier();
                // lane l owns bins 4l .. 4l+3
                const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const uint32_t mine = h0 + h1 + h2 + h3;
                uint32_t incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                const uint32_t before = w_lt + incl - mine;
                const unsigned long long hit = __ballot(before + mine >= need);
                if (hit == 0) {
                    all = true;   // (only possible in pass 0) the whole tree is visited
                } else {
                    const int l = __builtin_ctzll(hit);
                    uint32_t bin = 0, excl = 0;
                    if (lane == l) {
                        uint32_t c = before;
                        if (c + h0 >= need) { bin = 0; excl = c; }
                        else if ((c += h0) + h1 >= need) { bin = 1; excl = c; }
                        else if ((c += h1) + h2 >= need) { bin = 2; excl = c; }
                        else { c += h2; bin = 3; excl = c; }
                    }
                    bin = (uint32_t)__shfl((int)bin, l) + 4u * (uint32_t)l;
                    w_lt = (uint32_t)__shfl((int)excl, l);
                    prefix |= bin << shift;
                    prefix_mask |= 255u << shift;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!all) {
                xs = prefix;
                // ---- 3. the cut-off group
                int64_t budget_in = (int64_t)need - (int64_t)w_lt;   // > 0
                uint32_t n_bottle = 0, w = 0;
                for (uint32_t u = lane; u < n_nodes; u += 64) {
                    if (pi[u] == xs && OrderedBits(D[t.path_cent[(size_t)u * kPathMax]]) == xs) {
                        ++n_bottle;
                        w = u;
                    }
                }
                const unsigned long long have = __ballot(n_bottle != 0);
                uint32_t total_bottle = n_bottle;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) total_bottle += __shfl_xor(total_bottle, o);
                if (total_bottle != 1) {
                    redo = true;
                } else {
                    w = (uint32_t)__shfl((int)w, __builtin_ctzll(have));
                    uint32_t fcount = 0;
                    uint32_t node = w;
                    while (true) {
                        // visit `node`
                        if (lane == 0) pi[node] = 0;
                        const uint32_t c_first = a.node_cent_first[node];
                        budget_in -= (int64_t)(a.node_cent_first[node + 1] - c_first);
                        if (budget_in <= 0) break;
                        // its children with smaller keys join the group's frontier
                        const uint32_t n_child = a.node_child_count[node], child_first = a.node_child_first[node];
                        for (uint32_t base = 0; base < n_child && !redo; base += 64) {
                            const uint32_t i = base + lane;
                            uint32_t key = 0xFFFFFFFFu;
                            if (i < n_child) key = OrderedBits(D[c_first + i]);
                            const bool in = i < n_child && key < xs;
                            const unsigned long long m = __ballot(in);
                            const uint32_t add = (uint32_t)__popcll(m);
                            if (fcount + add > kFrontCap) {
                                redo = true;
                            } else {
                                if (in) {
                                    const uint32_t at = fcount + (uint32_t)__popcll(m & lanes_below);
                                    fkey[at] = key;
                                    fnode[at] = child_first + i;
                                }
                                fcount += add;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (redo || fcount == 0) break;
                        // next: the frontier's minimum
                        uint32_t best = 0xFFFFFFFFu, best_at = 0;
                        for (uint32_t i = lane; i < fcount; i += 64) {
                            const uint32_t k = fkey[i];
                            if (k < best) {
                                best = k;
                                best_at = i;
                            }
                        }
                        uint32_t mn = best;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
                        uint32_t n_min = 0;
                        for (uint32_t i = lane; i < fcount; i += 64) n_min += fkey[i] == mn ? 1u : 0u;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) n_min += __shfl_xor(n_min, o);
                        if (n_min != 1) {   // equal keys inside the cut-off group
                            redo = true;
                            break;
                        }
                        const int owner = __builtin_ctzll(__ballot(best == mn));
                        best_at = (uint32_t)__shfl((int)best_at, owner);
                        node = fnode[best_at];
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) {
                            fkey[best_at] = fkey[fcount - 1];
                            fnode[best_at] = fnode[fcount - 1];
                        }
                        --fcount;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();

        if (redo) {
            if (lane == 0) {
                a.redo[q] = 1;
                atomicAdd(a.redo_count, 1u);
            }
        } else {
            // ---- 4. minima over the centroids of the visited nodes
            uint32_t n_rec = 0;
            if (!a.rec_dist) {
                // a shard's centroids are one contiguous range (its nodes are, gpann_router_create checks it): every lane keeps a private
                // minimum over its part of the range, one reduction per shard (a reduction per 64 centroids cost 2 172 x 6 shuffles a query)
                for (uint32_t b = 0; b < S; ++b) {
                    const uint32_t c0 = a.node_cent_first[a.shard_root[b]];
                    const uint32_t c1 = b + 1 < S ? a.node_cent_first[a.shard_root[b + 1]] : C;
                    float mn = FLT_MAX;
                    for (uint32_t cb = c0 + lane; cb < c1; cb += 64 * kBatch) {
                        uint32_t node[kBatch];
                        float dv[kBatch];
#pragma unroll
                        for (int k = 0; k < kBatch; ++k) {
                            const uint32_t c = min(cb + 64u * k, c1 - 1);
                            node[k] = t.cent_node[c];
                            dv[k] = D[c];
                        }
                        uint32_t pv[kBatch];
#pragma unroll
                        for (int k = 0; k < kBatch; ++k) pv[k] = pi[node[k]];
#pragma unroll
                        for (int k = 0; k < kBatch; ++k)
                            if (cb + 64u * k < c1 && (all || pv[k] < xs)) mn = fminf(mn, dv[k]);
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
                    if (lane == 0) sMinOrd[b] = min(sMinOrd[b], OrderedBits(mn));
                }
            }
            for (uint32_t base = 0; base < C && a.rec_dist; base += 64) {
                const uint32_t c = base + lane;
                bool take = false;
                uint32_t shard = 0;
                float dist = FLT_MAX;
                if (c < C) {
                    const uint32_t u = t.cent_node[c];
                    take = all || pi[u] < xs;
                    if (take) {
                        shard = (uint32_t)a.node_shard[u];
                        dist = D[c];
                    }
                }
                const unsigned long long tm = __ballot(take);
                if (tm == 0) continue;
                const uint32_t shard0 = (uint32_t)__shfl((int)shard, __builtin_ctzll(tm));
                if (__ballot(take && shard != shard0) == 0) {   // the usual case: one shard for the whole wavefront
                    float mn = dist;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
                    if (lane == 0) atomicMin(&sMinOrd[shard0], OrderedBits(mn));
                } else if (take) {
                    atomicMin(&sMinOrd[shard], OrderedBits(dist));
                }
                if (a.rec_dist) {   // FrequencyQuery: every evaluated (distance, shard) pair; the order is irrelevant
                    if (take) {
                        const uint64_t at = (uint64_t)n_rec + (uint32_t)__popcll(tm & lanes_below);
                        if (at < a.rec_cap) {
                            a.rec_dist[(size_t)q * a.rec_cap + at] = dist;
                            a.rec_shard[(size_t)q * a.rec_cap + at] = (int32_t)shard;
                        }
                    }
                    n_rec += (uint32_t)__popcll(tm);
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t b = lane; b < S; b += 64) a.out_min_dist[(size_t)q * S + b] = FromOrderedBits(sMinOrd[b]);
            if (lane == 0) a.redo[q] = 0;
            if (a.rec_count && lane == 0) a.rec_count[q] = n_rec;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- bounded threshold replay: the same result from a table the MATRIX CORES wrote ------------------------------------------------
// The table above costs nq x C canonical evaluations on the vector ALUs (12.5 of the 26 ms of a step on the 139k-centroid NS tree).  Here
// the table holds  D~ = base - 2 acc  from RouterBoundKernel<.., PASS 3> (bf16 hi/lo products, router_bound.h), and every query knows eps
// with |D~(c) - E(c)| <= eps for the canonical float distance E of EVERY centroid (BoundFrameKernel).  Then pi~ (from D~) is within eps of
// pi for every node, and the cut-off x~* of the pi~ values within eps of x* (shifting every value by <= eps shifts the weighted quantile
// by <= eps).  So
//     pi~ < x~* - 2 eps  =>  pi < x*   visited, whatever the exact values are;      pi~ > x~* + 2 eps  =>  pi > x*   not visited
// and only the BAND in between (about a hundred of 31k nodes) needs exact keys: the canonical distance of the band nodes' path centroids
// (those that can be the path's maximum), then x*, the weight below it and the cut-off group exactly as RouterThresholdReplayKernel
// computes them — on the band entries, which hold every node with pi == x* and every child of one that can enter the group's frontier.
// The per-shard minima are bounded the same way: the minimiser of E over the visited centroids of a shard has D~ <= min D~ + 2 eps, so
// those candidates (one or two per shard) get the canonical evaluation and the minimum of THOSE is the walk's min_dist.
// Everything consumed downstream (min_dist, and through it the probe order) is therefore a canonical value of the reference's sequence;
// the matrix cores only decide which pairs are worth evaluating.  A band that does not fit its LDS list, a non-finite eps and the
// equal-key cases of the threshold replay mark the query: it gets an exact table row and the real queue (HostReplayMarked).
//
// One WORKGROUP per query.  pi~ is kept as a bin number g = 1 + floor((pi~ - h_lo) h_scale) in [1, 2^20] (0: a root) — monotone in pi~,
// 2^20 bins over the interval that holds every table value of the query (BoundFrameKernel) — and all of the above is done in bin space:
// x~* is found as the bin G that holds the weighted quantile (two 1024-bin histograms in LDS: g >> 10, then g & 1023 inside the coarse bin;
// the first is filled while pi~ is computed), the band is  |g - G| <= k  with k = ceil(2 eps h_scale) + 2 bins (pi~ within 2 eps of ANY value of
// bin G lies inside), surely visited is g < G - k.  The visited set becomes a bitmap over the nodes in LDS, so the pass over the query's
// table row (the minima) reads two coalesced streams — the row and cent_node — and nothing that depends on them.
struct BoundedTable {
    const float* table;          // [nq][table_ld]
    uint64_t table_ld;
    const float* eps;            // [nq]
    const float* h_lo;           // [nq]
    const float* h_scale;        // [nq]
    const float* centroids;      // [C][ld]
    const float* queries;        // [nq][ld]
    const uint32_t* cent_shard;  // [C]
    uint32_t d, ld;
    uint32_t band_cap;           // band entries per query (LDS: 3 words each)
    uint32_t* stats;             // [0] band nodes, [1] canonical evaluations, [2] queries whose band overflowed, [3] shard ranges scanned twice
    uint32_t* ticket;            // zeroed: the next unclaimed query
    uint32_t rec_keep;           // FrequencyQuery (ReplayArgs::rec_dist != nullptr): records are needed for the rec_keep smallest pairs only
};
constexpr uint32_t kBandNodeBits = 22;   // a band entry: node | size << 22
constexpr int kBWaves = 4;               // wavefronts per workgroup (= per query) of RouterBoundedReplayKernel
constexpr uint32_t kBinMax = 1u << 20;

__device__ __forceinline__ uint32_t BinOf(float v, float h_lo, float h_scale) {
    const float x = fmaxf((v - h_lo) * h_scale, 0.f);   // (NaN -> 0)
    return 1u + min((uint32_t)fminf(x, 1048575.f), kBinMax - 1u);
}

// exclusive prefix of `mine` over the workgroup's threads (kBWaves wavefronts); `red` = kBWaves words of LDS; returns the workgroup total in `total`
__device__ __forceinline__ uint32_t BlockExclusiveScan(uint32_t mine, uint32_t* red, int lane, int wave, uint32_t& total) {
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    __syncthreads();
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < kBWaves; ++w) {
        const uint32_t v = red[w];
        if (w < wave) before += v;
        total += v;
    }
    return before + incl - mine;
}

template <int METRIC, bool REC>
__global__ void __launch_bounds__(64 * kBWaves, 5) RouterBoundedReplayKernel(ReplayArgs a, ThresholdTree t, BoundedTable bt) {
    extern __shared__ uint32_t bdsm[];
    constexpr int kT = 64 * kBWaves;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t S = a.S, n_nodes = a.n_nodes, C = a.C, cap = bt.band_cap;
    uint32_t* hist = bdsm;                               // 1024
    uint32_t* sMinOrd = hist + 1024;                     // S
    uint32_t* bnode = sMinOrd + ((S + 3u) & ~3u);        // band entries; afterwards the candidate lists of the minima (3 * cap words)
    uint32_t* bkey = bnode + cap;
    uint32_t* bpi = bkey + cap;
    uint32_t* bitmap = bpi + cap;                        // visited nodes
    const uint32_t bm_words = (n_nodes + 31u) / 32u;
    uint32_t* sh = bitmap + ((bm_words + 3u) & ~3u);     // 32 words of scalars
    uint32_t* red = sh + 24;
    enum { N_BAND = 0, W_IN, XS, BELOW, N_BOTTLE, IW, REDO, OVER, SEL_BIN, SEL_BEFORE, SEL_ALL, N_EXACT, N_RESCAN, N_REC, F_ALL };
    uint32_t* g_of = t.pi + (size_t)blockIdx.x * n_nodes;
    const uint32_t node_mask = (1u << kBandNodeBits) - 1u;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;

    for (;;) {
        // (queries are claimed one at a time: their cost varies, and the launch is exactly one resident round of workgroups)
        __syncthreads();
        if (tid == 0) sh[31] = atomicAdd(bt.ticket, 1u);
        __syncthreads();
        const uint32_t q = sh[31];
        if (q >= a.nq) break;
        const float* D = bt.table + (size_t)q * bt.table_ld;
        const float* qrow = bt.queries + (size_t)q * bt.ld;
        const float eps = bt.eps[q], eps2 = 2.f * eps, h_lo = bt.h_lo[q], h_scale = bt.h_scale[q];
        const bool frame_ok = eps < INFINITY && h_scale > 0.f && h_scale < INFINITY && fabsf(h_lo) < INFINITY;
        const uint32_t k_bins = (uint32_t)fminf(eps2 * h_scale, 1048575.f) + 3u;
        for (uint32_t i = tid; i < 1024; i += kT) hist[i] = 0;
        for (uint32_t b = tid; b < S; b += kT) sMinOrd[b] = OrderedBits(FLT_MAX);
        if (tid < 24 + kBWaves) sh[tid] = tid == XS ? 0xFFFFFFFFu : 0u;
        __syncthreads();
        const int64_t budget_left = (int64_t)a.budget - (int64_t)a.root_total;
        bool all = t.visit_all != 0;
        const bool roots_only = budget_left <= 0;
        uint32_t n_exact = 0;
        bool redo = !frame_ok;
        unsigned long long tk[6];
        tk[0] = wall_clock64();

        // ---- 1. g = bin of pi~ for every node, level by level: g(u) = max(g(parent), bin(D~[key(u)])); + the coarse histogram of the node sizes
        if (!all && !roots_only && !redo) {
            for (uint32_t i = tid; i < t.lvl_first[1]; i += kT) g_of[t.lvl_node[i]] = 0u;
            for (uint32_t L = 1; L < t.n_levels; ++L) {
                const uint32_t l_end = t.lvl_first[L + 1];
                for (uint32_t i0 = t.lvl_first[L] + tid; i0 < l_end; i0 += kT * kBatch) {
                    uint32_t u[kBatch], par[kBatch], kc[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const uint32_t i = min(i0 + (uint32_t)kT * k, l_end - 1);
                        u[k] = t.lvl_node[i];
                        par[k] = t.lvl_parent[i];
                        kc[k] = t.lvl_key[i];
                    }
                    float dk[kBatch];
                    uint32_t gp[kBatch], sz[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        dk[k] = D[kc[k]];
                        gp[k] = L > 1 ? g_of[par[k]] : 0u;
                        sz[k] = a.node_cent_first[u[k] + 1] - a.node_cent_first[u[k]];
                    }
#pragma unroll
                    for (int k = 0; k < kBatch; ++k)
                        if (i0 + (uint32_t)kT * k < l_end) {
                            const uint32_t g = max(gp[k], BinOf(dk[k], h_lo, h_scale));
                            g_of[u[k]] = g;
                            atomicAdd(&hist[(g - 1u) >> 10], sz[k]);
                        }
                }
                __syncthreads();   // (the next level reads this level's bins)
            }
        }
        __syncthreads();
        tk[1] = wall_clock64();

        // ---- 2. the bin G of x~*: coarse bin, then the fine bin inside it
        uint32_t lo_g = 1, hi_g = 0;   // surely visited <=> g < lo_g; the band: lo_g <= g <= hi_g
        if (!all && !roots_only && !redo) {
            const uint32_t need = (uint32_t)min(budget_left, (int64_t)0x7FFFFFFF);
            uint32_t need_in = need, G = 0;
            for (int level = 0; level < 2 && !all; ++level) {
                const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const uint32_t mine = h0 + h1 + h2 + h3;
                uint32_t total = 0;
                const uint32_t before = BlockExclusiveScan(mine, red, lane, wave, total);
                if (before < need_in && before + mine >= need_in) {
                    uint32_t c = before, bin;
                    if (c + h0 >= need_in) bin = 0;
                    else if ((c += h0) + h1 >= need_in) bin = 1;
                    else if ((c += h1) + h2 >= need_in) bin = 2;
                    else { c += h2; bin = 3; }
                    sh[SEL_BIN] = 4u * (uint32_t)tid + bin;
                    sh[SEL_BEFORE] = c;
                }
                if (tid == 0 && total < need_in) sh[SEL_ALL] = 1u;   // (level 0 only) the whole tree is visited
                __syncthreads();
                all = sh[SEL_ALL] != 0u;
                const uint32_t bin = sh[SEL_BIN];
                need_in -= sh[SEL_BEFORE];
                G = (G << 10) | bin;
                __syncthreads();
                if (level == 0 && !all) {   // the fine histogram of the coarse bin's nodes
                    for (uint32_t i = tid; i < 1024; i += kT) hist[i] = 0;
                    __syncthreads();
                    for (uint32_t u0 = tid; u0 < n_nodes; u0 += kT * kBatch) {
                        uint32_t g[kBatch];
#pragma unroll
                        for (int k = 0; k < kBatch; ++k) g[k] = u0 + (uint32_t)kT * k < n_nodes ? g_of[u0 + (uint32_t)kT * k] : 0u;
#pragma unroll
                        for (int k = 0; k < kBatch; ++k)
                            if (g[k] != 0 && ((g[k] - 1u) >> 10) == bin) {
                                const uint32_t u = u0 + (uint32_t)kT * k;
                                atomicAdd(&hist[(g[k] - 1u) & 1023u], a.node_cent_first[u + 1] - a.node_cent_first[u]);
                            }
                    }
                    __syncthreads();
                }
            }
            if (!all) {
                G += 1u;   // (bins are numbered from 1)
                lo_g = G > k_bins + 1u ? G - k_bins : 1u;
                hi_g = G + k_bins;
            }
        }

        tk[2] = wall_clock64();
        // ---- 3. one pass over the nodes: the bitmap of the surely visited, the weight of the surely visited non-roots, the band list
        if (all) {
            for (uint32_t i = tid; i < bm_words; i += kT) bitmap[i] = 0xFFFFFFFFu;
        } else if (!redo) {
            uint32_t w_in = 0;
            for (uint32_t base0 = 0; base0 < n_nodes; base0 += kT * kBatch) {
                uint32_t gk[kBatch], sk[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const uint32_t u = base0 + (uint32_t)kT * k + tid;
                    const uint32_t uc = min(u, n_nodes - 1);
                    gk[k] = roots_only ? (u < n_nodes && t.path_cent[(size_t)u * kPathMax] == kNoCent ? 0u : 0xFFFFFFFFu) : (u < n_nodes ? g_of[u] : 0xFFFFFFFFu);
                    sk[k] = a.node_cent_first[uc + 1] - a.node_cent_first[uc];
                }
#pragma unroll
                for (int k = 0; k < kBatch; ++k) {
                    const uint32_t base = base0 + (uint32_t)kT * k, u = base + tid, g = gk[k];
                    const bool vis = g < lo_g;
                    const unsigned long long m = __ballot(vis);
                    if (lane == 0 && base + 64u * wave < n_nodes) {
                        bitmap[(base >> 5) + 2u * wave] = (uint32_t)m;
                        if ((base >> 5) + 2u * wave + 1u < bm_words) bitmap[(base >> 5) + 2u * wave + 1u] = (uint32_t)(m >> 32);
                    }
                    if (g != 0 && g <= hi_g && u < n_nodes) {
                        if (vis) {
                            w_in += sk[k];
                        } else {
                            const uint32_t at = atomicAdd(&sh[N_BAND], 1u);
                            if (at < cap) bnode[at] = u | (sk[k] << kBandNodeBits);
                            else sh[OVER] = 1u;
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) w_in += __shfl_xor(w_in, o);
            if (lane == 0) atomicAdd(&sh[W_IN], w_in);
        }
        __syncthreads();
        const uint32_t n_band = min(sh[N_BAND], cap);
        const bool band_over = sh[OVER] != 0u;
        if (band_over) redo = true;
        tk[3] = wall_clock64();

        if (!all && !roots_only && !redo) {
            const uint32_t need = (uint32_t)min(budget_left, (int64_t)0x7FFFFFFF);
            const uint32_t w_in = sh[W_IN];
            // ---- 4. canonical keys of the band: the node's own key, and every key on its path that can be the path's maximum
            for (uint32_t i = tid; i < n_band; i += kT) {
                const uint32_t u = bnode[i] & node_mask;
                const uint32_t gu = g_of[u];
                uint32_t key = 0, m = 0;
                for (int j = 0; j < kPathMax; ++j) {
                    const uint32_t c = t.path_cent[(size_t)u * kPathMax + j];
                    if (c == kNoCent) break;
                    if (j == 0 || BinOf(D[c], h_lo, h_scale) + k_bins >= gu) {
                        const uint32_t bits = OrderedBits(ExactDistanceVec4<METRIC>(bt.centroids + (size_t)c * bt.ld, qrow, bt.d));
                        if (j == 0) key = bits;
                        m = max(m, bits);
                        ++n_exact;
                    }
                }
                bkey[i] = key;
                bpi[i] = m;
            }
            __syncthreads();
            // ---- 4b. x*: the smallest exact pi in the band at which the accumulated sizes reach the budget left after the roots
            for (uint32_t i = tid; i < n_band; i += kT) {
                const uint32_t mine = bpi[i];
                uint32_t cum = w_in;
                for (uint32_t j = 0; j < n_band; ++j) cum += bpi[j] <= mine ? bnode[j] >> kBandNodeBits : 0u;
                if (cum >= need) atomicMin(&sh[XS], mine);
            }
            __syncthreads();
            const uint32_t xs = sh[XS];
            if (xs == 0xFFFFFFFFu) redo = true;   // (cannot happen while eps holds: x* lies in the band)
            if (!redo) {
                for (uint32_t i = tid; i < n_band; i += kT) {
                    const uint32_t v = bpi[i];
                    if (v < xs) atomicAdd(&sh[BELOW], bnode[i] >> kBandNodeBits);
                    if (v == xs && bkey[i] == xs) {
                        atomicAdd(&sh[N_BOTTLE], 1u);
                        sh[IW] = i;
                    }
                }
                __syncthreads();
                // states from here on: 0 visited, 1 in the group pi == x* (waiting), 2 not visited, 3 in the group's frontier
                for (uint32_t i = tid; i < n_band; i += kT) {
                    const uint32_t v = bpi[i];
                    bpi[i] = v < xs ? 0u : (v == xs ? 1u : 2u);
                }
                __syncthreads();
                // ---- 4c. the cut-off group (RouterThresholdReplayKernel step 3 on the band entries): wavefront 0
                if (wave == 0) {
                    int64_t budget_in = (int64_t)need - (int64_t)(w_in + sh[BELOW]);   // > 0
                    bool bad = sh[N_BOTTLE] != 1u;
                    uint32_t cur = sh[IW];
                    while (!bad) {
                        if (lane == 0) bpi[cur] = 0u;
                        const uint32_t packed = bnode[cur];
                        const uint32_t node = packed & node_mask;
                        budget_in -= (int64_t)(packed >> kBandNodeBits);
                        __builtin_amdgcn_wave_barrier();
                        if (budget_in <= 0) break;
                        // its children with smaller keys join the frontier (a child with pi == x* is in the band)
                        const uint32_t n_child = a.node_child_count[node], child_first = a.node_child_first[node];
                        for (uint32_t i = lane; i < n_band && n_child != 0; i += 64) {
                            const uint32_t u = bnode[i] & node_mask;
                            if (bpi[i] == 1u && u >= child_first && u < child_first + n_child && bkey[i] < xs) bpi[i] = 3u;
                        }
                        __builtin_amdgcn_wave_barrier();
                        uint32_t best = 0xFFFFFFFFu, best_at = 0;
                        for (uint32_t i = lane; i < n_band; i += 64) {
                            if (bpi[i] == 3u && bkey[i] < best) {
                                best = bkey[i];
                                best_at = i;
                            }
                        }
                        uint32_t mn = best;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
                        if (mn == 0xFFFFFFFFu) break;   // the frontier is empty
                        uint32_t n_min = 0;
                        for (uint32_t i = lane; i < n_band; i += 64) n_min += (bpi[i] == 3u && bkey[i] == mn) ? 1u : 0u;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) n_min += __shfl_xor(n_min, o);
                        if (n_min != 1) {   // equal keys inside the cut-off group
                            bad = true;
                            break;
                        }
                        cur = (uint32_t)__shfl((int)best_at, __builtin_ctzll(__ballot(best == mn)));
                    }
                    if (bad && lane == 0) sh[REDO] = 1u;
                }
                __syncthreads();
                if (sh[REDO] != 0u) redo = true;
                if (!redo) {
                    for (uint32_t i = tid; i < n_band; i += kT)
                        if (bpi[i] == 0u) {
                            const uint32_t u = bnode[i] & node_mask;
                            atomicOr(&bitmap[u >> 5], 1u << (u & 31u));
                        }
                }
            }
        }
        __syncthreads();
        tk[4] = wall_clock64();

        if (redo) {
            if (tid == 0) {
                a.redo[q] = 1;
                atomicAdd(a.redo_count, 1u);
            }
        } else {
            // ---- 5. per-shard minima, a wavefront per shard: the two smallest D~ per lane; centroids within 2 eps of the shard's smallest
            //      D~ are the candidates (their own list per wavefront)
            const uint32_t cand_cap = 3u * cap / kBWaves;
            uint32_t* cand = bnode + (size_t)wave * cand_cap;
            uint32_t n_cand = 0, n_rescan = 0;
            bool recording = false;   // FrequencyQuery's records instead of the minima (REC)
            auto flush = [&]() {
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i0 = 0; i0 < n_cand; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    float e = 0.f;
                    uint32_t shard = 0;
                    if (i < n_cand) {
                        const uint32_t c = cand[i];
                        e = ExactDistanceVec4<METRIC>(bt.centroids + (size_t)c * bt.ld, qrow, bt.d);
                        shard = bt.cent_shard[c];
                        if (!REC || !recording) atomicMin(&sMinOrd[shard], OrderedBits(e));
                        ++n_exact;
                    }
                    if (REC && recording) {
                        const uint32_t cnt = min(64u, n_cand - i0);
                        uint32_t at = 0;
                        if (lane == 0) at = atomicAdd(&sh[N_REC], cnt);
                        at = (uint32_t)__shfl((int)at, 0) + (uint32_t)lane;
                        if (i < n_cand && at < a.rec_cap) {
                            a.rec_dist[(size_t)q * a.rec_cap + at] = e;
                            a.rec_shard[(size_t)q * a.rec_cap + at] = (int32_t)shard;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                n_cand = 0;
            };
            auto append = [&](bool take, uint32_t c) {
                const unsigned long long m = __ballot(take);
                if (m == 0) return;
                const uint32_t add = (uint32_t)__popcll(m);
                if (n_cand + add > cand_cap) flush();
                if (take) cand[n_cand + (uint32_t)__popcll(m & lanes_below)] = c;
                n_cand += add;
            };
            for (uint32_t b = wave; b < S; b += kBWaves) {
                const uint32_t c0 = a.node_cent_first[a.shard_root[b]];
                const uint32_t c1 = b + 1 < S ? a.node_cent_first[a.shard_root[b + 1]] : C;
                float v1 = FLT_MAX, v2 = FLT_MAX;
                uint32_t i1 = 0;
                // (16-byte loads: four consecutive centroids per lane, kM trips in flight — the row is 16-byte aligned and padded to table_ld,
                //  cent_node is padded by four entries)
                constexpr int kM = 4;   // 16-byte trips in flight per lane
                for (uint32_t cb = (c0 & ~3u) + 4u * lane; cb < c1; cb += 256 * kM) {
                    uint32_t nd8[kM], nb[kM];
                    float4 dq[kM];
#pragma unroll
                    for (int k = 0; k < kM; ++k) {
                        const uint32_t c = min(cb + 256u * k, (c1 - 1) & ~3u);
                        nd8[k] = *reinterpret_cast<const uint32_t*>(t.cent_node8 + c);
                        nb[k] = t.cent_block_node[c >> 8];
                        dq[k] = *reinterpret_cast<const float4*>(D + c);
                    }
#pragma unroll
                    for (int k = 0; k < kM; ++k) {
                        // (only values below the lane's second smallest matter — after the first trips hardly any: the node, the bitmap and the range
                        //  are looked at for those only; the arithmetic of this loop was half of the kernel's time)
                        if (!(fminf(fminf(dq[k].x, dq[k].y), fminf(dq[k].z, dq[k].w)) < v2)) continue;
                        const uint32_t node[4] = {nb[k] + (nd8[k] & 255u), nb[k] + ((nd8[k] >> 8) & 255u), nb[k] + ((nd8[k] >> 16) & 255u), nb[k] + (nd8[k] >> 24)};
                        const float dv[4] = {dq[k].x, dq[k].y, dq[k].z, dq[k].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t c = cb + 256u * k + e;
                            if (dv[e] < v2 && c >= c0 && c < c1 && ((bitmap[node[e] >> 5] >> (node[e] & 31u)) & 1u)) {
                                if (dv[e] < v1) {
                                    v2 = v1;
                                    v1 = dv[e];
                                    i1 = c;
                                } else if (dv[e] < v2) {
                                    v2 = dv[e];
                                }
                            }
                        }
                    }
                }
                float mn = v1;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
                if (!(mn < FLT_MAX)) continue;   // no visited centroid (a shard without centroids)
                const float thr = mn + eps2;
                if (__any(v2 <= thr)) {   // a lane holds more than one candidate: every visited centroid of the shard within the margin
                    ++n_rescan;
                    for (uint32_t cb = c0; cb < c1; cb += 64) {
                        const uint32_t c = cb + lane;
                        bool take = false;
                        if (c < c1) {
                            const uint32_t node = t.cent_node[c];
                            take = ((bitmap[node >> 5] >> (node & 31u)) & 1u) && D[c] <= thr;
                        }
                        append(take, c);
                    }
                } else {
                    append(v1 <= thr, i1);
                }
            }
            flush();
            if (lane == 0 && n_rescan) atomicAdd(&sh[N_RESCAN], n_rescan);
            __syncthreads();
            for (uint32_t b = tid; b < S; b += kT) a.out_min_dist[(size_t)q * S + b] = FromOrderedBits(sMinOrd[b]);
            if (tid == 0) a.redo[q] = 0;

            // ---- 6. FrequencyQuery (kmeans_tree_router.cpp:132-172): the rec_keep smallest (distance, shard) pairs over the visited centroids.
            //      The V-th smallest table value lies in bin G_t (two histograms of the visited centroids' bins, as for x~*); every pair that can
            //      be among the V smallest canonical ones has its table value within 2 eps of it, i.e. a bin <= G_t + k: those (V + a few dozen)
            //      get the canonical evaluation and are the records VoteSelectKernel selects from.  (All visited centroids if there are <= V.)
            if constexpr (REC) {
                const uint32_t V = max(bt.rec_keep, 1u);
                recording = true;
                uint32_t need_in = V, Gt = 0;
                bool take_all = false;
                // f(c, D~[c]) for every visited centroid c; uniform trip count (the functors vote across the wavefront)
                auto for_visited = [&](auto&& f) {
                    for (uint32_t base = 0; base < C; base += 4u * kT * 2) {
                        uint32_t nd8[2], nbk[2];
                        float4 dq[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const uint32_t c = min(base + 4u * kT * k + 4u * tid, (C - 1) & ~3u);
                            nd8[k] = *reinterpret_cast<const uint32_t*>(t.cent_node8 + c);
                            nbk[k] = t.cent_block_node[c >> 8];
                            dq[k] = *reinterpret_cast<const float4*>(D + c);
                        }
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float dv[4] = {dq[k].x, dq[k].y, dq[k].z, dq[k].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t c = base + 4u * kT * k + 4u * tid + e;
                                const uint32_t node = nbk[k] + ((nd8[k] >> (8 * e)) & 255u);
                                f(c, dv[e], c < C && ((bitmap[min(node, n_nodes - 1) >> 5] >> (node & 31u)) & 1u));
                            }
                        }
                    }
                };
                for (int level = 0; level < 2 && !take_all; ++level) {
                    for (uint32_t i = tid; i < 1024; i += kT) hist[i] = 0;
                    __syncthreads();
                    const uint32_t coarse = Gt;
                    for_visited([&](uint32_t, float dv, bool vis) {
                        if (!vis) return;
                        const uint32_t g = BinOf(dv, h_lo, h_scale) - 1u;
                        if (level == 0) atomicAdd(&hist[g >> 10], 1u);
                        else if ((g >> 10) == coarse) atomicAdd(&hist[g & 1023u], 1u);
                    });
                    __syncthreads();
                    const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                    const uint32_t mine = h0 + h1 + h2 + h3;
                    uint32_t total = 0;
                    const uint32_t before = BlockExclusiveScan(mine, red, lane, wave, total);
                    if (before < need_in && before + mine >= need_in) {
                        uint32_t c = before, bin;
                        if (c + h0 >= need_in) bin = 0;
                        else if ((c += h0) + h1 >= need_in) bin = 1;
                        else if ((c += h1) + h2 >= need_in) bin = 2;
                        else { c += h2; bin = 3; }
                        sh[SEL_BIN] = 4u * (uint32_t)tid + bin;
                        sh[SEL_BEFORE] = c;
                    }
                    if (tid == 0 && total < need_in) sh[F_ALL] = 1u;   // fewer visited centroids than votes: all of them
                    __syncthreads();
                    take_all = sh[F_ALL] != 0u;
                    need_in -= sh[SEL_BEFORE];
                    Gt = (Gt << 10) | sh[SEL_BIN];
                    __syncthreads();
                }
                const uint32_t g_max = take_all ? 0xFFFFFFFFu : Gt + 1u + k_bins;   // (bins are numbered from 1)
                for_visited([&](uint32_t c, float dv, bool vis) { append(vis && BinOf(dv, h_lo, h_scale) <= g_max, c); });
                flush();
                __syncthreads();
                if (tid == 0) {
                    a.rec_count[q] = (uint32_t)min((uint64_t)sh[N_REC], a.rec_cap);
                    if ((uint64_t)sh[N_REC] > a.rec_cap && bt.stats) atomicAdd(bt.stats + 10, 1u);   // (records lost: the caller repeats the block with room for all)
                }
            }
        }
        {
            uint32_t ne = n_exact;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ne += __shfl_xor(ne, o);
            if (lane == 0 && ne) atomicAdd(&sh[N_EXACT], ne);
        }
        __syncthreads();
        if (tid == 0 && bt.stats) {
            atomicAdd(bt.stats + 0, n_band);
            atomicAdd(bt.stats + 1, sh[N_EXACT]);
            if (band_over) atomicAdd(bt.stats + 2, 1u);
            atomicAdd(bt.stats + 3, sh[N_RESCAN]);
            tk[5] = wall_clock64();   // (100 MHz ticks per phase, summed over the queries: GPANN_ROUTER_DEBUG prints them)
            for (int i = 0; i < 5; ++i) atomicAdd(bt.stats + 4 + i, (uint32_t)(tk[i + 1] - tk[i]));
        }
        __syncthreads();
    }
}

// kmeans_tree_router.cpp:93-130 with distance() replaced by the table: one wavefront per query.
template <bool LDS_HEAP>
__global__ void __launch_bounds__(64 * kRWaves) RouterReplayKernel(ReplayArgs a) {
    extern __shared__ float4 rpsm4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t S = a.S;
    const size_t per_wave = ((size_t)64 + S + (LDS_HEAP ? 2 * a.heap_cap : 0) + 3) & ~(size_t)3;
    float* sDist = reinterpret_cast<float*>(rpsm4) + (size_t)wave * per_wave;
    float* sMin = sDist + 64;
    const uint32_t slot = blockIdx.x * kRWaves + wave, n_slots = gridDim.x * kRWaves;
    float* hkey;
    uint32_t* hnode;
    if (LDS_HEAP) {
        hkey = sMin + S;
        hnode = reinterpret_cast<uint32_t*>(hkey + a.heap_cap);
    } else {
        hkey = a.heap_key + (size_t)slot * a.heap_cap;
        hnode = a.heap_node + (size_t)slot * a.heap_cap;
    }
    for (uint32_t q = slot; q < a.nq; q += n_slots) {
        if (a.only_redo && a.redo[q] == 0) continue;
        const float* D = a.dense + (size_t)q * a.C;
        for (uint32_t b = lane; b < S; b += 64) sMin[b] = FLT_MAX;
        int64_t size = 0;
        if (lane == 0) {
            for (uint32_t r = 0; r < S; ++r) {
                PushHeap(hkey, hnode, size, 0, -FLT_MAX, a.shard_root[r]);
                size++;
            }
        }
        size = S;
        __builtin_amdgcn_wave_barrier();
        int32_t budget = a.budget;
        uint32_t n_rec = 0;
        while (size > 0 && budget > 0) {
            uint32_t node = 0;
            if (lane == 0) {
                node = hnode[0];
                if (size > 1) {
                    float vk = hkey[size - 1];
                    uint32_t vn = hnode[size - 1];
                    hkey[size - 1] = hkey[0];
                    hnode[size - 1] = hnode[0];
                    AdjustHeap(hkey, hnode, 0, size - 1, vk, vn);
                }
            }
            size--;
            node = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
            const uint32_t c_first = a.node_cent_first[node];
            const uint32_t n = a.node_cent_first[node + 1] - c_first;
            const uint32_t n_child = a.node_child_count[node];
            const uint32_t child_first = a.node_child_first[node];
            const int shard = a.node_shard[node];
            budget -= (int32_t)n;
            for (uint32_t base = 0; base < n; base += 64) {
                const uint32_t m = min(64u, n - base);
                float dist = FLT_MAX;
                if ((uint32_t)lane < m) {
                    dist = D[c_first + base + lane];
                    if (a.rec_dist) {
                        a.rec_dist[(size_t)q * a.rec_cap + n_rec + lane] = dist;
                        a.rec_shard[(size_t)q * a.rec_cap + n_rec + lane] = shard;
                    }
                }
                n_rec += m;
                sDist[lane] = dist;
                float mn = dist;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    if (mn < sMin[shard]) sMin[shard] = mn;
                    for (uint32_t i = base; i < base + m && i < n_child; ++i)
                        PushHeap(hkey, hnode, size + (int64_t)(i - base), 0, sDist[i - base], child_first + i);
                }
                if (n_child > base) size += (int64_t)(min(n_child, base + m) - base);
                __builtin_amdgcn_wave_barrier();
            }
        }
        for (uint32_t b = lane; b < S; b += 64) a.out_min_dist[(size_t)q * S + b] = sMin[b];
        if (a.rec_count && lane == 0) a.rec_count[q] = n_rec;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace gpann

using namespace gpann;

extern "C" {

gpann_status gpann_router_create(uint32_t d, int metric, uint32_t S, uint32_t n_nodes, const int32_t* node_shard,
                                 const uint32_t* node_cent_first, const uint32_t* node_child_first, const uint32_t* node_child_count,
                                 const uint32_t* shard_root, const float* centroids, int device, gpann_router** out) {
    GPANN_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GPANN_REQUIRE(d > 0 && S > 0 && n_nodes >= S, "d, S must be positive and n_nodes >= S");
    GPANN_REQUIRE(node_shard && node_cent_first && node_child_first && node_child_count && shard_root && centroids, "null tree array");
    GPANN_REQUIRE(metric == GPANN_METRIC_L2 || metric == GPANN_METRIC_MIPS, "unknown metric");
    GPANN_REQUIRE(node_cent_first[0] == 0, "node_cent_first[0] must be 0");
    uint32_t max_children = 0;
    for (uint32_t u = 0; u < n_nodes; ++u) {
        GPANN_REQUIRE(node_cent_first[u] <= node_cent_first[u + 1], "node_cent_first not monotone");
        GPANN_REQUIRE(node_shard[u] >= 0 && (uint32_t)node_shard[u] < S, "node_shard out of range");
        const uint32_t nc = node_cent_first[u + 1] - node_cent_first[u];
        GPANN_REQUIRE(node_child_count[u] <= nc, "a node has more children than centroids");
        GPANN_REQUIRE(node_child_count[u] == 0 || (uint64_t)node_child_first[u] + node_child_count[u] <= n_nodes, "child index out of range");
        max_children = std::max(max_children, node_child_count[u]);
    }
    for (uint32_t b = 0; b < S; ++b) GPANN_REQUIRE(shard_root[b] < n_nodes, "shard_root out of range");
    // the layout the kernels rely on (ExtractPoints order, kmeans_tree_router.cpp:174-201): nodes grouped per shard in ascending shard
    // order, the root of a shard first — tiles of consecutive centroids then hold the shards shard_first..shard_last
    for (uint32_t u = 1; u < n_nodes; ++u) GPANN_REQUIRE(node_shard[u - 1] <= node_shard[u], "nodes must be grouped per shard, shards ascending");
    for (uint32_t b = 0; b < S; ++b) {
        const uint32_t u = shard_root[b];
        GPANN_REQUIRE((uint32_t)node_shard[u] == b && (u == 0 || (uint32_t)node_shard[u - 1] < b), "shard_root[b] must be the first node of shard b");
    }
    GPANN_TRY(EnsureDevice(device));

    auto* r = new gpann_router();
    r->opt.SnapshotEnvironment();   // (the one look at the environment: gpann_router_set_option from here on)
    r->device = device;
    r->metric = metric;
    r->d = d;
    r->ld = RoundUp4(d);
    r->S = S;
    r->n_nodes = n_nodes;
    r->n_centroids = node_cent_first[n_nodes];
    r->max_children = max_children;
    auto fail = [&](gpann_status st) {
        gpann_router_destroy(r);
        return st;
    };
#define TRY_OR_FAIL(expr)                                                                            \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            SetLastError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return fail(_e == hipErrorOutOfMemory ? GPANN_ERR_OOM : GPANN_ERR_HIP);                  \
        }                                                                                            \
    } while (0)
    const size_t nc = r->n_centroids;
    TRY_OR_FAIL(hipMalloc(&r->centroids, std::max<size_t>(nc * r->ld * sizeof(float), 256)));
    TRY_OR_FAIL(hipMalloc(&r->node_shard, n_nodes * 4));
    TRY_OR_FAIL(hipMalloc(&r->node_cent_first, (n_nodes + 1) * 4));
    TRY_OR_FAIL(hipMalloc(&r->node_child_first, n_nodes * 4));
    TRY_OR_FAIL(hipMalloc(&r->node_child_count, n_nodes * 4));
    TRY_OR_FAIL(hipMalloc(&r->shard_root, S * 4));
    if (r->ld != d) TRY_OR_FAIL(hipMemset(r->centroids, 0, nc * r->ld * sizeof(float)));
    if (nc) TRY_OR_FAIL(hipMemcpy2D(r->centroids, (size_t)r->ld * 4, centroids, (size_t)d * 4, (size_t)d * 4, nc, hipMemcpyHostToDevice));
    TRY_OR_FAIL(hipMemcpy(r->node_shard, node_shard, n_nodes * 4, hipMemcpyHostToDevice));
    TRY_OR_FAIL(hipMemcpy(r->node_cent_first, node_cent_first, (n_nodes + 1) * 4, hipMemcpyHostToDevice));
    TRY_OR_FAIL(hipMemcpy(r->node_child_first, node_child_first, n_nodes * 4, hipMemcpyHostToDevice));
    TRY_OR_FAIL(hipMemcpy(r->node_child_count, node_child_count, n_nodes * 4, hipMemcpyHostToDevice));
    TRY_OR_FAIL(hipMemcpy(r->shard_root, shard_root, S * 4, hipMemcpyHostToDevice));
    r->h_node_shard.assign(node_shard, node_shard + n_nodes);
    r->h_node_cent_first.assign(node_cent_first, node_cent_first + n_nodes + 1);
    r->h_node_child_first.assign(node_child_first, node_child_first + n_nodes);
    r->h_node_child_count.assign(node_child_count, node_child_count + n_nodes);
    r->h_shard_root.assign(shard_root, shard_root + S);
    for (auto& e : r->ev) TRY_OR_FAIL(hipEventCreate(&e));
    for (auto& e : r->ev_bound) TRY_OR_FAIL(hipEventCreate(&e));
    {
        hipDeviceProp_t prop;
        TRY_OR_FAIL(hipGetDeviceProperties(&prop, device));
        r->num_cus = prop.multiProcessorCount;
        if (prop.sharedMemPerBlock > 0) r->lds_per_wg = prop.sharedMemPerBlock;
        for (uint32_t u = 0; u < n_nodes; ++u) r->max_node = std::max(r->max_node, node_cent_first[u + 1] - node_cent_first[u]);
        std::vector<uint8_t> is_root(n_nodes, 0);
        r->root_min = 0xFFFFFFFFu;
        for (uint32_t b = 0; b < S; ++b) {
            const uint32_t u = shard_root[b];
            const uint32_t sz = node_cent_first[u + 1] - node_cent_first[u];
            is_root[u] = 1;
            r->root_total += sz;
            r->root_min = std::min(r->root_min, sz);
            r->root_children += node_child_count[u];
        }
        r->leaves_only = true;
        for (uint32_t u = 0; u < n_nodes; ++u)
            if (!is_root[u] && node_child_count[u] != 0) r->leaves_only = false;

        // threshold replay tables: the key centroids along every node's path, and the node of every centroid
        std::vector<uint32_t> parent(n_nodes, kNoCent), key_cent(n_nodes, kNoCent), depth(n_nodes, 0);
        for (uint32_t u = 0; u < n_nodes; ++u)
            for (uint32_t i = 0; i < node_child_count[u]; ++i) {
                parent[node_child_first[u] + i] = u;
                key_cent[node_child_first[u] + i] = node_cent_first[u] + i;
            }
        std::vector<uint32_t> path((size_t)n_nodes * kPathMax, kNoCent), cent_node(std::max<size_t>(nc, 1) + 4, 0);   // (+ 4: 16-byte loads)
        for (uint32_t u = 0; u < n_nodes; ++u) {
            uint32_t v = u, len = 0;
            while (parent[v] != kNoCent && len <= n_nodes) {   // (len bound: malformed input must not hang)
                if (len < (uint32_t)kPathMax) path[(size_t)u * kPathMax + len] = key_cent[v];
                v = parent[v];
                ++len;
            }
            depth[u] = len;
            r->max_depth = std::max(r->max_depth, len);
            for (uint32_t c = node_cent_first[u]; c < node_cent_first[u + 1]; ++c) cent_node[c] = u;
        }
        TRY_OR_FAIL(hipMalloc(&r->path_cent, path.size() * 4));
        {   // the nodes level by level (ThresholdTree::lvl_*), for trees the threshold replays take (depth <= kPathMax)
            const uint32_t n_levels = std::min<uint32_t>(r->max_depth, (uint32_t)kPathMax) + 1;
            std::vector<uint32_t> first(kPathMax + 2, 0), order(n_nodes), lp(n_nodes), lk(n_nodes);
            for (uint32_t u = 0; u < n_nodes; ++u)
                if (depth[u] < n_levels) ++first[depth[u] + 1];
            for (int l = 0; l <= kPathMax; ++l) first[l + 1] += first[l];
            std::vector<uint32_t> at(first.begin(), first.end() - 1);
            for (uint32_t u = 0; u < n_nodes; ++u)
                if (depth[u] < n_levels) {
                    const uint32_t i = at[depth[u]]++;
                    order[i] = u;
                    lp[i] = parent[u];
                    lk[i] = key_cent[u];
                }
            for (int l = 0; l < kPathMax + 2; ++l) r->lvl_first[l] = first[std::min<int>(l, (int)n_levels)];
            r->n_levels = n_levels;
            TRY_OR_FAIL(hipMalloc(&r->lvl_node, (size_t)n_nodes * 4));
            TRY_OR_FAIL(hipMalloc(&r->lvl_parent, (size_t)n_nodes * 4));
            TRY_OR_FAIL(hipMalloc(&r->lvl_key, (size_t)n_nodes * 4));
            TRY_OR_FAIL(hipMemcpy(r->lvl_node, order.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice));
            TRY_OR_FAIL(hipMemcpy(r->lvl_parent, lp.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice));
            TRY_OR_FAIL(hipMemcpy(r->lvl_key, lk.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice));
        }
        std::vector<uint32_t> cent_shard(cent_node.size(), 0);
        for (size_t c = 0; c < nc; ++c) cent_shard[c] = (uint32_t)node_shard[cent_node[c]];
        TRY_OR_FAIL(hipMalloc(&r->cent_shard, cent_shard.size() * 4));
        TRY_OR_FAIL(hipMemcpy(r->cent_shard, cent_shard.data(), cent_shard.size() * 4, hipMemcpyHostToDevice));
        {
            std::vector<uint32_t> has(S, 0);
            for (size_t c = 0; c < std::max<size_t>(nc, 1); ++c)
                if (cent_shard[c] < S) has[cent_shard[c]] = 1;
            TRY_OR_FAIL(hipMalloc(&r->shard_has_cent, (size_t)S * 4));
            TRY_OR_FAIL(hipMemcpy(r->shard_has_cent, has.data(), (size_t)S * 4, hipMemcpyHostToDevice));
        }
        // matrix-core bound for the visit-everything case (router_bound.h).  Squared L2: in the frame of the mean centroid (translation
        // leaves the distances alone; the bound's error terms scale with the norms).
        bool finite = true;
        std::vector<float> mu(r->ld, 0.f);
        const char* centre_env = r->opt.Get("GPANN_FILTER_CENTER");
        const bool centred = metric == GPANN_METRIC_L2 && nc > 0 && !(centre_env && centre_env[0] == '0');
        if (centred) {
            std::vector<double> sum(d, 0.0);
            for (size_t c = 0; c < nc; ++c)
                for (uint32_t x = 0; x < d; ++x) sum[x] += centroids[c * d + x];
            for (uint32_t x = 0; x < d; ++x) mu[x] = (float)(sum[x] / (double)nc);
            for (uint32_t x = 0; x < d; ++x) finite = finite && std::isfinite(mu[x]);
        }
        double rmax2 = 0.0;
        for (size_t c = 0; c < nc && finite; ++c) {
            double s2 = 0.0;
            for (uint32_t x = 0; x < d; ++x) {
                const float raw = centroids[c * d + x];
                finite = finite && std::isfinite(raw);
                const double v = centred ? (double)(raw - mu[x]) : (double)raw;   // (the fp32 difference the planes are split from)
                s2 += v * v;
            }
            rmax2 = std::max(rmax2, s2);
        }
        if (finite && d <= 256u && nc > 0) {
            const uint64_t tiles = (nc + kGTileRows - 1) / kGTileRows + 1;
            r->plane_ks = d <= 128u ? 8u : 16u;
            const size_t tile_bytes = (size_t)r->plane_ks * 2048;
            if (hipMalloc(&r->cent_planes, tiles * tile_bytes) == hipSuccess && hipMalloc(&r->cent_nhn, (nc + 64) * 4) == hipSuccess) {
                TRY_OR_FAIL(hipMemset(r->cent_planes, 0, tiles * tile_bytes));
                TRY_OR_FAIL(hipMemset(r->cent_nhn, 0, (nc + 64) * 4));
                if (centred) {
                    TRY_OR_FAIL(hipMalloc(&r->cent_mu, (size_t)r->ld * 4));
                    TRY_OR_FAIL(hipMemcpy(r->cent_mu, mu.data(), (size_t)r->ld * 4, hipMemcpyHostToDevice));
                }
                hipLaunchKernelGGL(CentroidPlanesKernel, dim3((uint32_t)std::min<uint64_t>((nc * 16 + 255) / 256, 4096)), dim3(256), 0, 0, r->centroids, (uint64_t)nc, d,
                                   r->ld, r->cent_planes, r->cent_nhn, r->cent_mu, r->plane_ks);
                TRY_OR_FAIL(hipGetLastError());
                TRY_OR_FAIL(hipDeviceSynchronize());
                r->cent_rmax = std::sqrt(rmax2) * (1.0 + 1e-9);
            } else {
                (void)hipGetLastError();
                if (r->cent_planes) (void)hipFree(r->cent_planes);
                r->cent_planes = nullptr;
            }
        }
        TRY_OR_FAIL(hipMalloc(&r->cent_node, cent_node.size() * 4));
        {
            std::vector<uint8_t> n8(cent_node.size(), 0);
            std::vector<uint32_t> bn((nc >> 8) + 2, 0);
            for (size_t c = 0; c < nc; ++c) {
                if ((c & 255u) == 0) bn[c >> 8] = cent_node[c];
                n8[c] = (uint8_t)(cent_node[c] - bn[c >> 8]);
                if (cent_node[c] - bn[c >> 8] > 255u) {   // (a node without centroids in between: cannot happen for trees the trainer builds)
                    n8.clear();
                    break;
                }
            }
            if (!n8.empty()) {
                TRY_OR_FAIL(hipMalloc(&r->cent_node8, n8.size()));
                TRY_OR_FAIL(hipMalloc(&r->cent_block_node, bn.size() * 4));
                TRY_OR_FAIL(hipMemcpy(r->cent_node8, n8.data(), n8.size(), hipMemcpyHostToDevice));
                TRY_OR_FAIL(hipMemcpy(r->cent_block_node, bn.data(), bn.size() * 4, hipMemcpyHostToDevice));
            }
        }
        TRY_OR_FAIL(hipMemcpy(r->path_cent, path.data(), path.size() * 4, hipMemcpyHostToDevice));
        TRY_OR_FAIL(hipMemcpy(r->cent_node, cent_node.data(), cent_node.size() * 4, hipMemcpyHostToDevice));
    }
#undef TRY_OR_FAIL
    *out = r;
    return GPANN_OK;
}

gpann_status gpann_router_info(const gpann_router* r, uint32_t* d, int* metric, uint32_t* S, uint32_t* n_nodes, uint64_t* n_centroids) {
    GPANN_REQUIRE(r != nullptr, "router is null");
    if (d) *d = r->d;
    if (metric) *metric = r->metric;
    if (S) *S = r->S;
    if (n_nodes) *n_nodes = r->n_nodes;
    if (n_centroids) *n_centroids = r->n_centroids;
    return GPANN_OK;
}

gpann_status gpann_router_get_stats(const gpann_router* r, gpann_router_stats* out) {
    GPANN_REQUIRE(r != nullptr && out != nullptr, "null argument");
    out->bound_calls = r->stat_bound_calls;
    out->bound_selfcheck = r->stat_bound_selfcheck;
    out->bound_overflow = r->stat_bound_overflow;
    out->verified = r->stat_verified;
    out->verify_failures = r->stat_verify_failures;
    out->host_sorted = r->stat_host_sorted;
    out->heap_redo = r->stat_heap_redo;
    out->last_bound_ms = r->stat_bound_ms;
    out->last_bound_flops = r->stat_bound_flops;
    out->probed = r->stat_probed;
    out->probe_failures = r->stat_probe_failures;
    out->table_blocks = r->stat_table_blocks;
    out->table_queries = r->stat_table_queries;
    out->table_band_nodes = r->stat_table_band_nodes;
    out->table_exact = r->stat_table_exact;
    out->table_band_overflow = r->stat_table_band_overflow;
    out->table_rescans = r->stat_table_rescans;
    return GPANN_OK;
}

gpann_status gpann_router_export(const gpann_router* r, int32_t* node_shard, uint32_t* node_cent_first, uint32_t* node_child_first,
                                 uint32_t* node_child_count, uint32_t* shard_root, float* centroids) {
    GPANN_REQUIRE(r != nullptr, "router is null");
    GPANN_TRY(EnsureDevice(r->device));
    const size_t n = r->n_nodes;
    if (node_shard) GPANN_HIP_TRY(hipMemcpy(node_shard, r->node_shard, n * 4, hipMemcpyDeviceToHost));
    if (node_cent_first) GPANN_HIP_TRY(hipMemcpy(node_cent_first, r->node_cent_first, (n + 1) * 4, hipMemcpyDeviceToHost));
    if (node_child_first) GPANN_HIP_TRY(hipMemcpy(node_child_first, r->node_child_first, n * 4, hipMemcpyDeviceToHost));
    if (node_child_count) GPANN_HIP_TRY(hipMemcpy(node_child_count, r->node_child_count, n * 4, hipMemcpyDeviceToHost));
    if (shard_root) GPANN_HIP_TRY(hipMemcpy(shard_root, r->shard_root, (size_t)r->S * 4, hipMemcpyDeviceToHost));
    if (centroids && r->n_centroids)
        GPANN_HIP_TRY(hipMemcpy2D(centroids, (size_t)r->d * 4, r->centroids, (size_t)r->ld * 4, (size_t)r->d * 4, r->n_centroids, hipMemcpyDeviceToHost));
    return GPANN_OK;
}

gpann_status gpann_router_set_option(gpann_router* r, const char* name, const char* value) {
    GPANN_REQUIRE(r != nullptr, "router handle is null");
    std::lock_guard<std::recursive_mutex> turn(r->call_mutex);
    return SetOption(r->opt, name, value);
}
gpann_status gpann_router_get_option(const gpann_router* r, const char* name, char* out, size_t cap) {
    GPANN_REQUIRE(r != nullptr, "router handle is null");
    return GetOption(r->opt, name, out, cap);
}

gpann_status gpann_router_destroy(gpann_router* r) {
    if (!r) return GPANN_OK;
    (void)hipSetDevice(r->device);
    for (void* p : {(void*)r->centroids, (void*)r->node_shard, (void*)r->node_cent_first, (void*)r->node_child_first,
                    (void*)r->node_child_count, (void*)r->shard_root, (void*)r->path_cent, (void*)r->cent_node, (void*)r->cent_shard, (void*)r->shard_has_cent, (void*)r->cent_planes, (void*)r->cent_nhn, (void*)r->cent_mu, (void*)r->cent_node8, (void*)r->cent_block_node, (void*)r->lvl_node, (void*)r->lvl_parent, (void*)r->lvl_key})
        if (p) (void)hipFree(p);
    r->ws_queries.Release();
    r->ws_heap.Release();
    r->ws_out.Release();
    r->ws_dense.Release();
    r->ws_redo.Release();
    r->ws_pi.Release();
    r->ws_frame.Release();
    r->ws_bound.Release();
    r->ws_verify.Release();
    r->ws_host_redo.Release();
    if (r->pinned_redo) (void)hipHostFree(r->pinned_redo);
    if (r->pinned_counters) (void)hipHostFree(r->pinned_counters);
    if (r->pinned_min) (void)hipHostFree(r->pinned_min);
    if (r->pinned_probes) (void)hipHostFree(r->pinned_probes);
    r->ws_probes.Release();
    for (auto& e : r->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : r->ev_bound)
        if (e) (void)hipEventDestroy(e);
    delete r;
    return GPANN_OK;
}

namespace {

// ---- host replay of the queries a sorted / threshold replay could not settle ---------------------------------------------------------
// Equal keys in the cut-off group: only libstdc++'s own heap order says which of the tied nodes is popped before the budget runs out.
// One wavefront replaying a whole walk takes ~5 us per pop (25k sequential pops on the 139k-centroid NS tree: 136 ms for a batch in which
// 60 of 10 000 queries were marked); a host core does a pop in ~50 ns.  So the marked queries' rows of the distance table go to the host
// (GatherRowsF32Kernel + one copy), host threads run the reference's loop (kmeans_tree_router.cpp:103-123) with the real
// std::priority_queue on them, and the resulting min_dist rows (and FrequencyQuery records) are scattered back.
__global__ void GatherRowsF32Kernel(const float* table, const uint32_t* rows, uint32_t n_rows, uint64_t width, float* out) {
    const uint64_t total = (uint64_t)n_rows * width;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / width, c = i - r * width;
        out[i] = table[(uint64_t)rows[r] * width + c];
    }
}
__global__ void ScatterRowsF32Kernel(const float* packed, const uint32_t* rows, uint32_t n_rows, uint64_t width, uint64_t out_stride, float* out) {
    const uint64_t total = (uint64_t)n_rows * width;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / width, c = i - r * width;
        out[(uint64_t)rows[r] * out_stride + c] = packed[i];
    }
}

// kmeans_tree_router.cpp:93-123 on a row of exact distances (D[c] = distance(centroid c, Q)); records as FrequencyQuery sees them (:155)
// select_smallest: the caller will only look at the smallest pairs (FrequencyQuery's votes, <= rec_cap of them) and rec_cap may be smaller than the
// number of evaluated pairs: the rec_cap lexicographically smallest (distance, shard) pairs are kept instead of the first rec_cap.
void HostWalkOnTable(const gpann_router* r, const float* D, int32_t budget, float* min_dist, float* rec_dist, int32_t* rec_shard, uint32_t* rec_count,
                     uint64_t rec_cap, bool select_smallest = false) {
    struct PQEntry {
        float dist;
        uint32_t node;
        bool operator>(const PQEntry& other) const { return dist > other.dist; }
    };
    std::priority_queue<PQEntry, std::vector<PQEntry>, std::greater<>> pq;
    const uint32_t S = r->S;
    for (uint32_t b = 0; b < S; ++b) min_dist[b] = std::numeric_limits<float>::max();
    for (uint32_t b = 0; b < S; ++b) pq.push(PQEntry{std::numeric_limits<float>::lowest(), r->h_shard_root[b]});
    uint64_t n_rec = 0;
    std::vector<std::pair<float, int32_t>> all;
    while (!pq.empty() && budget > 0) {
        const PQEntry top = pq.top();
        pq.pop();
        const uint32_t first = r->h_node_cent_first[top.node], n = r->h_node_cent_first[top.node + 1] - first;
        const uint32_t shard = (uint32_t)r->h_node_shard[top.node];
        budget -= (int32_t)n;
        for (uint32_t i = 0; i < n; ++i) {
            const float dist = D[first + i];
            min_dist[shard] = std::min(min_dist[shard], dist);
            if (rec_dist && select_smallest) {
                all.emplace_back(dist, (int32_t)shard);
            } else if (rec_dist && n_rec < rec_cap) {
                rec_dist[n_rec] = dist;
                rec_shard[n_rec] = (int32_t)shard;
            }
            ++n_rec;
            if (i < r->h_node_child_count[top.node]) pq.push(PQEntry{dist, r->h_node_child_first[top.node] + i});
        }
    }
    if (rec_dist && select_smallest) {
        if (all.size() > rec_cap) {
            std::nth_element(all.begin(), all.begin() + (ptrdiff_t)rec_cap, all.end());
            all.resize(rec_cap);
        }
        for (size_t i = 0; i < all.size(); ++i) {
            rec_dist[i] = all[i].first;
            rec_shard[i] = all[i].second;
        }
    }
    if (rec_count) *rec_count = (uint32_t)std::min<uint64_t>(n_rec, rec_cap);
}

// Replays the marked queries of a block on the host.  `flags` [nb] (device): 1 = marked.  Returns false (nothing done) if the rows do not
// fit the staging budget — the caller then runs the device heap kernel as before.
// `exact_for` (the bounded replay, whose table is not exact): the marked queries' rows are evaluated canonically here (DenseDistKernel on
// the gathered queries) instead of being gathered from `d_table`.
gpann_status HostReplayMarked(gpann_router* r, const float* d_table, const uint32_t* d_flags, uint32_t nb, uint32_t marked, int32_t budget, float* d_min_dist,
                              float* d_rec_dist, int32_t* d_rec_shard, uint32_t* d_rec_count, uint64_t rec_cap, hipStream_t st, bool* done,
                              const DenseArgs* exact_for = nullptr, bool select_smallest = false) {
    *done = false;
    const uint64_t C = r->n_centroids;
    const uint32_t S = r->S;
    const bool rec = d_rec_dist != nullptr;
    const size_t row_floats = (size_t)C + S + (rec ? 2 * rec_cap + 1 : 0);
    if ((uint64_t)marked * row_floats * 4 > (512ull << 20)) return GPANN_OK;
    std::vector<uint32_t> flags(nb);
    GPANN_HIP_TRY(hipMemcpyAsync(flags.data(), d_flags, (size_t)nb * 4, hipMemcpyDeviceToHost, st));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> list;
    for (uint32_t q = 0; q < nb; ++q)
        if (flags[q]) list.push_back(q);
    const uint32_t m = (uint32_t)list.size();
    if (m == 0) {
        *done = true;
        return GPANN_OK;
    }
    // device: [list m] [rows m x C]; host (pinned): rows m x C | min_dist m x S | records
    const size_t qpack_floats = exact_for ? (((size_t)m + 63) & ~(size_t)63) * exact_for->ld : 0;
    GPANN_TRY(r->ws_host_redo.Reserve((size_t)((m + 15) & ~15u) * 4 + (size_t)m * row_floats * 4 + qpack_floats * 4 + 64));
    uint32_t* d_list = r->ws_host_redo.as<uint32_t>();
    float* d_rows = reinterpret_cast<float*>(d_list + ((m + 15) & ~15u));
    float* d_qpack = d_rows + (size_t)m * row_floats;
    if (r->pinned_redo_floats < (size_t)m * row_floats) {
        if (r->pinned_redo) (void)hipHostFree(r->pinned_redo);
        r->pinned_redo = nullptr;
        r->pinned_redo_floats = 0;
        GPANN_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r->pinned_redo), (size_t)m * row_floats * 4, hipHostMallocDefault));
        r->pinned_redo_floats = (size_t)m * row_floats;
    }
    float* h_rows = r->pinned_redo;
    float* h_min = h_rows + (size_t)m * C;
    float* h_rec_dist = h_min + (size_t)m * S;
    int32_t* h_rec_shard = reinterpret_cast<int32_t*>(h_rec_dist + (rec ? (size_t)m * rec_cap : 0));
    uint32_t* h_rec_count = reinterpret_cast<uint32_t*>(h_rec_shard + (rec ? (size_t)m * rec_cap : 0));
    GPANN_HIP_TRY(hipMemcpyAsync(d_list, list.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    if (exact_for) {
        hipLaunchKernelGGL(GatherRowsF32Kernel, dim3((uint32_t)std::min<uint64_t>(1024, ((uint64_t)m * exact_for->ld + 255) / 256)), dim3(256), 0, st, exact_for->queries,
                           d_list, m, (uint64_t)exact_for->ld, d_qpack);
        DenseArgs da = *exact_for;
        da.queries = d_qpack;
        da.out = d_rows;
        da.nq = m;
        const dim3 dgrid((uint32_t)((C + 63) / 64), (m + 63) / 64);
        const size_t dlds = (size_t)2 * 64 * (kRDC + 4) * sizeof(float);
        if (r->metric == GPANN_METRIC_L2) hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_L2, false>), dgrid, dim3(256), dlds, st, da);
        else hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_MIPS, false>), dgrid, dim3(256), dlds, st, da);
    } else {
        hipLaunchKernelGGL(GatherRowsF32Kernel, dim3((uint32_t)std::min<uint64_t>(4096, ((uint64_t)m * C + 255) / 256)), dim3(256), 0, st, d_table, d_list, m, C, d_rows);
    }
    GPANN_HIP_TRY(hipGetLastError());
    GPANN_HIP_TRY(hipMemcpyAsync(h_rows, d_rows, (size_t)m * C * 4, hipMemcpyDeviceToHost, st));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    const uint32_t chunks = m;
    HostPool::Wide().Run(chunks, [&](uint32_t c) {
        for (uint32_t i = (uint32_t)((uint64_t)m * c / chunks); i < (uint32_t)((uint64_t)m * (c + 1) / chunks); ++i)
            HostWalkOnTable(r, h_rows + (size_t)i * C, budget, h_min + (size_t)i * S, rec ? h_rec_dist + (size_t)i * rec_cap : nullptr,
                            rec ? h_rec_shard + (size_t)i * rec_cap : nullptr, rec ? h_rec_count + i : nullptr, rec_cap, select_smallest);
    });
    // results back: min_dist rows (and the records) of the marked queries
    float* d_min_packed = d_rows;   // (the table rows are no longer needed)
    GPANN_HIP_TRY(hipMemcpyAsync(d_min_packed, h_min, (size_t)m * S * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ScatterRowsF32Kernel, dim3((uint32_t)std::min<uint64_t>(1024, ((uint64_t)m * S + 255) / 256)), dim3(256), 0, st, d_min_packed, d_list, m, (uint64_t)S,
                       (uint64_t)S, d_min_dist);
    if (rec) {
        float* d_pack = d_min_packed + (size_t)m * S;
        GPANN_HIP_TRY(hipMemcpyAsync(d_pack, h_rec_dist, (size_t)m * rec_cap * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(ScatterRowsF32Kernel, dim3(4096), dim3(256), 0, st, d_pack, d_list, m, rec_cap, rec_cap, d_rec_dist);
        float* d_pack2 = d_pack + (size_t)m * rec_cap;
        GPANN_HIP_TRY(hipMemcpyAsync(d_pack2, h_rec_shard, (size_t)m * rec_cap * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(ScatterRowsF32Kernel, dim3(4096), dim3(256), 0, st, d_pack2, d_list, m, rec_cap, rec_cap, reinterpret_cast<float*>(d_rec_shard));
        float* d_pack3 = d_pack2 + (size_t)m * rec_cap;
        GPANN_HIP_TRY(hipMemcpyAsync(d_pack3, h_rec_count, (size_t)m * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(ScatterRowsF32Kernel, dim3(1), dim3(256), 0, st, d_pack3, d_list, m, (uint64_t)1, (uint64_t)1, reinterpret_cast<float*>(d_rec_count));
    }
    GPANN_HIP_TRY(hipGetLastError());
    GPANN_HIP_TRY(hipStreamSynchronize(st));   // (the pinned buffers are reused by the next block)
    *done = true;
    return GPANN_OK;
}

// ---- the matrix-core bound path of the visit-everything case (router_bound.h), with its self-check and the verify mode ------------
__global__ void DiffBitsKernel(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out /* [0] count, [1] first index */) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) {
        atomicAdd(out, 1u);
        atomicMin(out + 1, (uint32_t)i);
    }
}

// The workspace of a bound block in ONE launch (three runtime memsets and the caller's FillMinBitsKernel were four): pass-1 maxima -> 0, the
// candidate list -> (~0, ~0) = unused slot, the counters -> 0, the block's minima -> OrderedBits(FLT_MAX).
__global__ void __launch_bounds__(256) BoundInitKernel(uint32_t* max_bits, size_t nb_s, uint2* cand, size_t cand_cap, uint32_t* counters, uint32_t* min_bits) {
    const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < nb_s; i += stride) {
        max_bits[i] = 0u;
        min_bits[i] = 0xFF7FFFFFu;
    }
    for (size_t i = i0; i < cand_cap; i += stride) cand[i] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    if (i0 < 8) counters[i0] = 0u;
}

// Enqueues pass 1, the margins, pass 2, the canonical evaluation of the candidates and the unset-minima count for the query block of
// `da` (whose out_min_bits are set to OrderedBits(FLT_MAX) here).  On return `ba` describes the workspace that was used.
gpann_status EnqueueBound(gpann_router* r, const DenseArgs& da, uint32_t nb, hipStream_t st, BoundArgs& ba, float*& thr, bool convert_minima = false) {
    const uint32_t S = r->S, ld = r->ld;
    const uint64_t C = r->n_centroids;
    const size_t nb_s = (size_t)nb * S;
    // (slots are handed out in chunks of 256 per wavefront: room for every wavefront's last, partly used chunk)
    const uint32_t wg_queries = r->d > 128 ? 128u : 256u;   // queries per workgroup of RouterBoundKernel
    const uint32_t cand_cap = (uint32_t)std::min<uint64_t>(0x7FFFFFFFull, (uint64_t)nb_s * 4 + (uint64_t)((nb + wg_queries - 1) / wg_queries) * 80 * 8 * 256 + 1024);
    const size_t cand_off = (nb_s * 8 + 15) & ~(size_t)15;   // bytes: max_bits | thr | (pad to 16) | cand | counters
    GPANN_TRY(r->ws_bound.Reserve(cand_off + (size_t)cand_cap * 8 + 64));
    ba.queries = da.queries;
    ba.cent_planes = r->cent_planes;
    ba.cent_nhn = r->cent_nhn;
    ba.cent_shard = r->cent_shard;
    ba.max_bits = r->ws_bound.as<uint32_t>();
    thr = reinterpret_cast<float*>(ba.max_bits + nb_s);
    ba.thr = thr;
    ba.cand = reinterpret_cast<uint2*>(r->ws_bound.as<char>() + cand_off);
    ba.cand_count = reinterpret_cast<uint32_t*>(ba.cand + cand_cap);
    ba.cand_cap = cand_cap;
    ba.nq = nb;
    ba.C = (uint32_t)C;
    ba.S = S;
    ba.ld = ld;
    ba.mu = r->cent_mu;
    ba.plane_ks = r->plane_ks;
    const bool ks16 = r->d > 128;
    const uint32_t group_q = ks16 ? 128u : 256u;   // queries per workgroup (4 / 8 query tiles)
    const uint32_t groups = (nb + group_q - 1) / group_q;
    const uint32_t n_tiles = (uint32_t)((C + kGTileRows - 1) / kGTileRows);
    // one round of workgroups: (query groups) x (centroid segments) <= the CUs (a workgroup per CU, 132 KiB of LDS each) — 40 groups x 8
    // segments = 320 workgroups on 256 CUs ran as 256 + 64, i.e. at the pace of 160
    ba.n_seg = std::max(1u, std::min(std::min(80u, (n_tiles + 7) / 8), std::max(1u, (uint32_t)r->num_cus / groups)));
    // counters: [0] candidate slots, [1] unset minima after pass 2, [2..3] (verify mode), [4] pairs pass 1 left without a maximum / a finite threshold
    if (r->opt.IsSet("GPANN_ROUTER_LEGACY_MEMSET")) {
        // debug: the set-up of rounds 1-3 — three runtime memsets and a fill launch.  The intermittent "minima unset" event of those rounds has
        // not been seen since BoundInitKernel replaced them; tools/stress_suite_context.sh runs this variant under the determinism probe to
        // find out whether the event lives in the runtime's memsets
        GPANN_HIP_TRY(hipMemsetAsync(ba.max_bits, 0, nb_s * 4, st));
        GPANN_HIP_TRY(hipMemsetAsync(ba.cand_count, 0, 32, st));
        GPANN_HIP_TRY(hipMemsetAsync(ba.cand, 0xFF, (size_t)cand_cap * 8, st));
        hipLaunchKernelGGL(FillMinBitsKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s);
    } else {
        hipLaunchKernelGGL(BoundInitKernel, dim3((uint32_t)std::min<uint64_t>(1024, ((uint64_t)cand_cap + 255) / 256)), dim3(256), 0, st, ba.max_bits, nb_s, ba.cand, (size_t)cand_cap,
                           ba.cand_count, da.out_min_bits);
    }
    GPANN_HIP_TRY(hipGetLastError());
    const bool l2 = r->metric == GPANN_METRIC_L2, ks4 = r->d <= 64;
    const size_t blds = (size_t)(ks16 ? 4 * 16 : 8 * (ks4 ? 4 : 8)) * 2048 + 8 * kGTileRows * sizeof(float);
    const void *f1, *f2;
#define GPANN_PICKB(P_)                                                                                                                                          \
    (l2 ? (ks16 ? (const void*)RouterBoundKernel<GPANN_METRIC_L2, 16, P_> : ks4 ? (const void*)RouterBoundKernel<GPANN_METRIC_L2, 4, P_> : (const void*)RouterBoundKernel<GPANN_METRIC_L2, 8, P_>) \
        : (ks16 ? (const void*)RouterBoundKernel<GPANN_METRIC_MIPS, 16, P_> : ks4 ? (const void*)RouterBoundKernel<GPANN_METRIC_MIPS, 4, P_> : (const void*)RouterBoundKernel<GPANN_METRIC_MIPS, 8, P_>))
    f1 = GPANN_PICKB(1);
    f2 = GPANN_PICKB(2);
#undef GPANN_PICKB
    GPANN_HIP_TRY(hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
    GPANN_HIP_TRY(hipFuncSetAttribute(f2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
    void* kargs[] = {&ba};
    GPANN_HIP_TRY(hipEventRecord(r->ev_bound[0], st));
    GPANN_HIP_TRY(hipLaunchKernel(f1, dim3(ba.n_seg, groups), dim3(512), kargs, blds, st));
    // (the self-check's two counters: pass 1's maxima / thresholds here — cand_count + 4 —, the unset minima after pass 2 below — cand_count + 1)
    if (l2) hipLaunchKernelGGL(BoundMarginKernel<GPANN_METRIC_L2>, dim3(nb), dim3(64), 0, st, ba.queries, ba.max_bits, thr, S, r->d, ld, r->cent_rmax, ba.mu, r->shard_has_cent, ba.cand_count + 4);
    else hipLaunchKernelGGL(BoundMarginKernel<GPANN_METRIC_MIPS>, dim3(nb), dim3(64), 0, st, ba.queries, ba.max_bits, thr, S, r->d, ld, r->cent_rmax, ba.mu, r->shard_has_cent, ba.cand_count + 4);
    GPANN_HIP_TRY(hipLaunchKernel(f2, dim3(ba.n_seg, groups), dim3(512), kargs, blds, st));
    GPANN_HIP_TRY(hipEventRecord(r->ev_bound[1], st));
    const uint32_t egrid = (uint32_t)std::min<uint64_t>(4096, ((uint64_t)cand_cap + 255) / 256);
    if (l2) hipLaunchKernelGGL(ExactCandidatesKernel<GPANN_METRIC_L2>, dim3(egrid), dim3(256), 0, st, r->centroids, ba.queries, ba.cand, ba.cand_count, cand_cap,
                               r->cent_shard, da.out_min_bits, S, r->d, ld);
    else hipLaunchKernelGGL(ExactCandidatesKernel<GPANN_METRIC_MIPS>, dim3(egrid), dim3(256), 0, st, r->centroids, ba.queries, ba.cand, ba.cand_count, cand_cap,
                            r->cent_shard, da.out_min_bits, S, r->d, ld);
    GPANN_HIP_TRY(hipGetLastError());
    // self-check: after a full visit every shard with centroids has a minimum.  (cand_count + 1 = the counter)
    // (convert_minima: the minima leave as floats — nobody is going to look at their bits again)
    if (convert_minima) hipLaunchKernelGGL(CountUnsetAndConvertKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s, S, r->shard_has_cent, ba.cand_count + 1);
    else hipLaunchKernelGGL(CountUnsetMinBitsKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s, S, r->shard_has_cent, ba.cand_count + 1);
    GPANN_HIP_TRY(hipGetLastError());
    return GPANN_OK;
}

// Evidence for a bound block that went wrong (a verify difference, a fired self-check, a determinism-probe difference): the block's data
// (run 1), the same pipeline once more on the same inputs (run 2), the canonical minima, and a report that localises the fault — which
// stage's data differs between the runs, and for the first bad (query, shard) pairs whether the true minimiser passes the threshold and
// whether it is in run 1's candidate list.  The text goes to stderr and $GPANN_VERIFY_REPORT; the arrays of BOTH runs (max_bits, thr,
// min_bits, candidate lists, staged queries, canonical minima) go to $GPANN_VERIFY_REPORT.<pid>.<n>.bin (layout: BoundDumpHeader below).
// `da.out_min_bits` must still hold run 1's ORDERED BITS.  Leaves run 2's results in the workspaces.
struct BoundDumpHeader {
    char magic[8];   // "GPANNBD1"
    uint32_t nb, S, d, ld, n_seg, cand_cap, metric, plane_ks;
    uint64_t C;
    uint32_t counters1[8], counters2[8];
    uint32_t n_cand1, n_cand2;
    // then: max_bits1 [nb*S] u32 | thr1 [nb*S] f32 | min_bits1 [nb*S] u32 | queries1 [nb*ld] f32 | cand1 [n_cand1] uint2 |
    //       max_bits2 | thr2 | min_bits2 | queries2 | cand2 [n_cand2] | want (canonical min bits) [nb*S] u32
};
static std::atomic<uint32_t> g_bound_dump_seq{0};

gpann_status BoundEvidence(gpann_router* r, const DenseArgs& da, uint32_t nb, hipStream_t st, BoundArgs& ba, float* thr, const float* caller_queries,
                           bool caller_on_device, const char* why, size_t* bad_minima) {
    const uint32_t S = r->S, ld = r->ld, d = r->d;
    const uint64_t C = r->n_centroids;
    const size_t nb_s = (size_t)nb * S;
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    struct Snap { std::vector<uint32_t> max_bits, min_bits; std::vector<float> thr, queries; std::vector<uint2> cand; uint32_t counters[8]; };
    auto snap = [&](Snap& s) -> hipError_t {
        s.max_bits.resize(nb_s); s.min_bits.resize(nb_s); s.thr.resize(nb_s); s.queries.resize((size_t)nb * ld);
        hipError_t e = hipMemcpy(s.max_bits.data(), ba.max_bits, nb_s * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s.thr.data(), thr, nb_s * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s.min_bits.data(), da.out_min_bits, nb_s * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s.queries.data(), da.queries, (size_t)nb * ld * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s.counters, ba.cand_count, 32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        s.cand.resize(std::min<uint32_t>(s.counters[0], ba.cand_cap));
        if (!s.cand.empty()) e = hipMemcpy(s.cand.data(), ba.cand, s.cand.size() * 8, hipMemcpyDeviceToHost);
        return e;
    };
    Snap s1, s2;
    GPANN_HIP_TRY(snap(s1));
    // the canonical minima of the block
    GPANN_TRY(r->ws_verify.Reserve(nb_s * 4 + 16));
    DenseArgs dv = da;
    dv.out_min_bits = r->ws_verify.as<uint32_t>();
    const dim3 dgrid((uint32_t)((C + 63) / 64), (nb + 63) / 64);
    const size_t dlds = (size_t)2 * 64 * (kRDC + 4) * sizeof(float);
    const uint32_t fgrid = (uint32_t)((nb_s + 255) / 256);
    hipLaunchKernelGGL(FillMinBitsKernel, dim3(fgrid), dim3(256), 0, st, dv.out_min_bits, nb_s);
    if (r->metric == GPANN_METRIC_L2) hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_L2, true>), dgrid, dim3(256), dlds, st, dv);
    else hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_MIPS, true>), dgrid, dim3(256), dlds, st, dv);
    GPANN_HIP_TRY(hipGetLastError());
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> want(nb_s);
    GPANN_HIP_TRY(hipMemcpy(want.data(), dv.out_min_bits, nb_s * 4, hipMemcpyDeviceToHost));
    // run 2: the same pipeline on the same inputs
    hipLaunchKernelGGL(FillMinBitsKernel, dim3(fgrid), dim3(256), 0, st, da.out_min_bits, nb_s);
    GPANN_TRY(EnqueueBound(r, da, nb, st, ba, thr));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    GPANN_HIP_TRY(snap(s2));
    std::vector<float> cents((size_t)C * ld);
    std::vector<uint32_t> cshard(C);
    GPANN_HIP_TRY(hipMemcpy(cents.data(), r->centroids, cents.size() * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(cshard.data(), r->cent_shard, C * 4, hipMemcpyDeviceToHost));
    std::vector<float> nhn(C);
    GPANN_HIP_TRY(hipMemcpy(nhn.data(), r->cent_nhn, C * 4, hipMemcpyDeviceToHost));
    std::vector<float> mu(ld, 0.f);   // the frame of the bound (zero when the router is not centred)
    if (r->cent_mu) GPANN_HIP_TRY(hipMemcpy(mu.data(), r->cent_mu, (size_t)ld * 4, hipMemcpyDeviceToHost));

    auto bits_to_float = [](uint32_t o) { uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o; float f; std::memcpy(&f, &b, 4); return f; };
    size_t bad1 = 0, bad2 = 0, unset1 = 0, max_diff = 0, thr_diff = 0, thr_nan1 = 0, max_zero1 = 0, q_diff = 0, q_vs_caller = 0, min_diff = 0;
    for (size_t i = 0; i < nb_s; ++i) {
        bad1 += s1.min_bits[i] != want[i];
        bad2 += s2.min_bits[i] != want[i];
        min_diff += s1.min_bits[i] != s2.min_bits[i];
        unset1 += s1.min_bits[i] == 0xFF7FFFFFu;
        max_diff += s1.max_bits[i] != s2.max_bits[i];
        thr_diff += std::memcmp(&s1.thr[i], &s2.thr[i], 4) != 0;
        thr_nan1 += std::isnan(s1.thr[i]);
        max_zero1 += s1.max_bits[i] == 0;
    }
    for (size_t i = 0; i < s1.queries.size(); ++i) q_diff += std::memcmp(&s1.queries[i], &s2.queries[i], 4) != 0;
    if (caller_queries && !caller_on_device)
        for (uint32_t q = 0; q < nb; ++q)
            for (uint32_t x = 0; x < d; ++x) q_vs_caller += std::memcmp(&s1.queries[(size_t)q * ld + x], &caller_queries[(size_t)q * d + x], 4) != 0;
    // the candidate SETS of the two runs (slot order depends on which wavefront claimed which chunk)
    auto cand_set = [](const std::vector<uint2>& c) {
        std::vector<uint64_t> v;
        for (const uint2& e : c)
            if (e.x != 0xFFFFFFFFu) v.push_back((uint64_t)e.x << 32 | e.y);
        std::sort(v.begin(), v.end());
        return v;
    };
    const std::vector<uint64_t> set1 = cand_set(s1.cand), set2 = cand_set(s2.cand);
    std::vector<uint64_t> only1, only2;
    std::set_difference(set1.begin(), set1.end(), set2.begin(), set2.end(), std::back_inserter(only1));
    std::set_difference(set2.begin(), set2.end(), set1.begin(), set1.end(), std::back_inserter(only2));
    std::string rep;
    char line[1536];
    snprintf(line, sizeof line,
             "[gpann router EVIDENCE] %s: %zu of %zu minima differ from the canonical evaluation (run 1), %zu (run 2 on the same inputs); run 1: %zu unset, "
             "%zu thr NaN, %zu max_bits zero, candidate slots %u (cap %u), counters pass-1 %u unset-minima %u; between the runs: %zu max_bits differ, %zu thr "
             "differ, %zu min_bits differ, %zu query words differ, candidate sets %zu / %zu entries (%zu only in run 1, %zu only in run 2); device queries vs "
             "caller's host queries: %zu words differ; nq %u S %u C %llu d %u n_seg %u pid %d\n",
             why, bad1, nb_s, bad2, unset1, thr_nan1, max_zero1, s1.counters[0], ba.cand_cap, s1.counters[4], s1.counters[1], max_diff, thr_diff, min_diff, q_diff,
             set1.size(), set2.size(), only1.size(), only2.size(), q_vs_caller, nb, S, (unsigned long long)C, d, ba.n_seg, (int)getpid());
    rep += line;
    {   // where the bad pairs sit: per shard, and per 32-query tile of the block
        std::vector<size_t> per_shard(S, 0), per_qtile((nb + 31) / 32, 0);
        for (size_t i = 0; i < nb_s; ++i)
            if (s1.min_bits[i] != want[i]) {
                per_shard[i % S]++;
                per_qtile[(i / S) / 32]++;
            }
        std::string a = "  bad pairs per shard:", b = "  bad pairs per 32-query tile:";
        for (uint32_t x = 0; x < S && x < 64; ++x) a += " " + std::to_string(per_shard[x]);
        for (size_t x = 0; x < per_qtile.size() && x < 64; ++x) b += " " + std::to_string(per_qtile[x]);
        rep += a + "\n" + b + "\n";
    }
    int shown = 0;
    for (size_t i = 0; i < nb_s && shown < 8; ++i) {
        if (s1.min_bits[i] == want[i] && s1.max_bits[i] == s2.max_bits[i]) continue;
        ++shown;
        const uint32_t q = (uint32_t)(i / S), b = (uint32_t)(i % S);
        // in double: the true acc of every centroid of the shard, the true minimiser
        double best_acc = -1e300, best_dist = 1e300, acc_of_min = 0;
        uint64_t arg_acc = 0, arg_min = 0, n_in = 0;
        const float* qv = &s1.queries[(size_t)q * ld];
        for (uint64_t c = 0; c < C; ++c) {
            if (cshard[c] != b) continue;
            ++n_in;
            double dot = 0, dist = 0;
            for (uint32_t x = 0; x < d; ++x) {
                dot += (double)(qv[x] - mu[x]) * (double)(cents[c * ld + x] - mu[x]);
                const double df = (double)qv[x] - cents[c * ld + x];
                dist += df * df;
            }
            const double acc = r->metric == GPANN_METRIC_L2 ? dot + (double)nhn[c] : dot;
            const double key = r->metric == GPANN_METRIC_L2 ? dist : 1.0 - dot;
            if (acc > best_acc) { best_acc = acc; arg_acc = c; }
            if (key < best_dist) { best_dist = key; arg_min = c; acc_of_min = acc; }
        }
        size_t in_list = 0, list_has_min = 0, in_list2 = 0;
        for (const uint2& qc : s1.cand)
            if (qc.x == q && qc.y != 0xFFFFFFFFu && qc.y < C && cshard[qc.y] == b) { ++in_list; list_has_min += qc.y == arg_min; }
        for (const uint2& qc : s2.cand)
            if (qc.x == q && qc.y != 0xFFFFFFFFu && qc.y < C && cshard[qc.y] == b) ++in_list2;
        snprintf(line, sizeof line,
                 "  (q %u, shard %u, %llu centroids): got %08x (%g) want %08x (%g); run 1 max %g thr %g | run 2 max %g thr %g min %08x; host: max acc %.9g at %llu, "
                 "minimiser %llu dist %.9g acc %.9g (%s thr 1); candidates of the pair in run 1's list: %zu (run 2: %zu), minimiser among them: %zu\n",
                 q, b, (unsigned long long)n_in, s1.min_bits[i], bits_to_float(s1.min_bits[i]), want[i], bits_to_float(want[i]), bits_to_float(s1.max_bits[i]),
                 s1.thr[i], bits_to_float(s2.max_bits[i]), s2.thr[i], s2.min_bits[i], best_acc, (unsigned long long)arg_acc, (unsigned long long)arg_min, best_dist,
                 acc_of_min, acc_of_min >= (double)s1.thr[i] ? ">=" : "<", in_list, in_list2, list_has_min);
        rep += line;
    }
    fputs(rep.c_str(), stderr);
    if (const char* path = std::getenv("GPANN_VERIFY_REPORT")) {
        if (FILE* f = fopen(path, "a")) {
            fputs(rep.c_str(), f);
            fclose(f);
        }
        // the arrays of both runs, for an offline look
        const std::string dump = std::string(path) + "." + std::to_string((int)getpid()) + "." + std::to_string(g_bound_dump_seq.fetch_add(1)) + ".bin";
        if (FILE* f = fopen(dump.c_str(), "wb")) {
            BoundDumpHeader h{};
            std::memcpy(h.magic, "GPANNBD1", 8);
            h.nb = nb; h.S = S; h.d = d; h.ld = ld; h.n_seg = ba.n_seg; h.cand_cap = ba.cand_cap; h.metric = (uint32_t)r->metric; h.plane_ks = r->plane_ks;
            h.C = C;
            std::memcpy(h.counters1, s1.counters, 32);
            std::memcpy(h.counters2, s2.counters, 32);
            h.n_cand1 = (uint32_t)s1.cand.size();
            h.n_cand2 = (uint32_t)s2.cand.size();
            fwrite(&h, sizeof h, 1, f);
            for (const Snap* s : {&s1, &s2}) {
                fwrite(s->max_bits.data(), 4, nb_s, f);
                fwrite(s->thr.data(), 4, nb_s, f);
                fwrite(s->min_bits.data(), 4, nb_s, f);
                fwrite(s->queries.data(), 4, s->queries.size(), f);
                fwrite(s->cand.data(), 8, s->cand.size(), f);
            }
            fwrite(want.data(), 4, nb_s, f);
            fclose(f);
        }
    }
    if (bad_minima) *bad_minima = bad1;
    return GPANN_OK;
}

// GPANN_ROUTER_VERIFY=1: every bound-path result is compared, minimum by minimum, with the canonical evaluation of ALL pairs
// (DenseDistKernel<.., MIN_ONLY>).  A difference ends the call with GPANN_ERR_INTERNAL after a second run of the bound path on the same
// inputs and a report that says which stage's data differed between the two runs (BoundEvidence).
gpann_status VerifyBound(gpann_router* r, const DenseArgs& da, uint32_t nb, hipStream_t st, BoundArgs& ba, float* thr, const float* caller_queries,
                         bool caller_on_device) {
    const uint32_t S = r->S;
    const uint64_t C = r->n_centroids;
    const size_t nb_s = (size_t)nb * S;
    GPANN_TRY(r->ws_verify.Reserve(nb_s * 4 + 16));
    DenseArgs dv = da;
    dv.out_min_bits = r->ws_verify.as<uint32_t>();
    uint32_t* d_diff = ba.cand_count + 2;   // zeroed by EnqueueBound
    const dim3 dgrid((uint32_t)((C + 63) / 64), (nb + 63) / 64);
    const size_t dlds = (size_t)2 * 64 * (kRDC + 4) * sizeof(float);   // (>= the 64 x 65 floats of the output tile)
    const uint32_t fgrid = (uint32_t)((nb_s + 255) / 256);
    hipLaunchKernelGGL(FillMinBitsKernel, dim3(fgrid), dim3(256), 0, st, dv.out_min_bits, nb_s);
    if (r->metric == GPANN_METRIC_L2) hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_L2, true>), dgrid, dim3(256), dlds, st, dv);
    else hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_MIPS, true>), dgrid, dim3(256), dlds, st, dv);
    GPANN_HIP_TRY(hipMemsetAsync(d_diff + 1, 0xFF, 4, st));
    hipLaunchKernelGGL(DiffBitsKernel, dim3(fgrid), dim3(256), 0, st, da.out_min_bits, dv.out_min_bits, nb_s, d_diff);
    GPANN_HIP_TRY(hipGetLastError());
    uint32_t diff[2] = {0, 0};
    GPANN_HIP_TRY(hipMemcpyAsync(diff, d_diff, 8, hipMemcpyDeviceToHost, st));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    ++r->stat_verified;
    if (diff[0] == 0) return GPANN_OK;
    ++r->stat_verify_failures;
    size_t bad1 = 0;
    GPANN_TRY(BoundEvidence(r, da, nb, st, ba, thr, caller_queries, caller_on_device, "bound path differs from the canonical evaluation (GPANN_ROUTER_VERIFY)", &bad1));
    SetLastError("router bound path failed verification: %zu of %zu minima differ from the canonical evaluation (GPANN_ROUTER_VERIFY)", bad1, nb_s);
    return GPANN_ERR_INTERNAL;
}

// GPANN_ROUTER_PROBE=1 (debugging aid; synchronises): the bound pipeline is run a SECOND time on the same block and everything it left —
// pass-1 maxima, thresholds, the candidate SET, the minima — must be bit-equal to the first run's.  A difference is reported through
// BoundEvidence (which runs the pipeline a third time) and ends the call with GPANN_ERR_INTERNAL.  Far more sensitive than the self-check:
// it also sees disturbances that happen not to lose a minimum.
gpann_status ProbeBoundDeterminism(gpann_router* r, const DenseArgs& da, uint32_t nb, hipStream_t st, BoundArgs& ba, float* thr, const float* caller_queries,
                                   bool caller_on_device) {
    const size_t nb_s = (size_t)nb * r->S;
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> max1(nb_s), min1(nb_s), max2(nb_s), min2(nb_s);
    std::vector<float> thr1(nb_s), thr2(nb_s);
    uint32_t c1[8], c2[8];
    auto cands = [&](const uint32_t* counters, std::vector<uint64_t>& out) -> hipError_t {
        std::vector<uint2> c(std::min<uint32_t>(counters[0], ba.cand_cap));
        hipError_t e = c.empty() ? hipSuccess : hipMemcpy(c.data(), ba.cand, c.size() * 8, hipMemcpyDeviceToHost);
        out.clear();
        for (const uint2& x : c)
            if (x.x != 0xFFFFFFFFu) out.push_back((uint64_t)x.x << 32 | x.y);
        std::sort(out.begin(), out.end());
        return e;
    };
    std::vector<uint64_t> set1, set2;
    GPANN_HIP_TRY(hipMemcpy(max1.data(), ba.max_bits, nb_s * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(thr1.data(), thr, nb_s * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(min1.data(), da.out_min_bits, nb_s * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(c1, ba.cand_count, 32, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(cands(c1, set1));
    hipLaunchKernelGGL(FillMinBitsKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s);
    BoundArgs ba2;
    float* thr_b = nullptr;
    GPANN_TRY(EnqueueBound(r, da, nb, st, ba2, thr_b));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    GPANN_HIP_TRY(hipMemcpy(max2.data(), ba2.max_bits, nb_s * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(thr2.data(), thr_b, nb_s * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(min2.data(), da.out_min_bits, nb_s * 4, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(hipMemcpy(c2, ba2.cand_count, 32, hipMemcpyDeviceToHost));
    GPANN_HIP_TRY(cands(c2, set2));
    ++r->stat_probed;
    const bool same = max1 == max2 && min1 == min2 && std::memcmp(thr1.data(), thr2.data(), nb_s * 4) == 0 && set1 == set2 && c1[1] == c2[1] && c1[4] == c2[4];
    if (same) return GPANN_OK;
    ++r->stat_probe_failures;
    size_t dm = 0, dt = 0, dn = 0;
    for (size_t i = 0; i < nb_s; ++i) {
        dm += max1[i] != max2[i];
        dt += std::memcmp(&thr1[i], &thr2[i], 4) != 0;
        dn += min1[i] != min2[i];
    }
    fprintf(stderr, "[gpann router PROBE] two runs of the bound pipeline on the same block differ: %zu max_bits, %zu thr, %zu minima, candidate sets %zu vs %zu, "
                    "counters pass-1 %u / %u, unset %u / %u\n", dm, dt, dn, set1.size(), set2.size(), c1[4], c2[4], c1[1], c2[1]);
    // (BoundEvidence takes its "run 1" from the workspaces, which now hold THIS probe's second run; its "run 2" is a third run.  The
    // first run's arrays are summarised in the line above.)
    size_t bad = 0;
    GPANN_TRY(BoundEvidence(r, da, nb, st, ba2, thr_b, caller_queries, caller_on_device, "determinism probe: runs 2 and 3 of the block (GPANN_ROUTER_PROBE)", &bad));
    ba = ba2;
    SetLastError("router bound path is not deterministic: two runs on the same block differ in %zu max_bits, %zu thr, %zu minima (GPANN_ROUTER_PROBE)", dm, dt, dn);
    return GPANN_ERR_INTERNAL;
}

gpann_status RunBoundPath(gpann_router* r, const DenseArgs& da, uint32_t nb, hipStream_t st, bool* settled, const float* caller_queries, bool caller_on_device,
                          bool may_defer) {
    *settled = false;
    const size_t nb_s = (size_t)nb * r->S;
    BoundArgs ba;
    float* thr = nullptr;
    const bool defer = may_defer && !r->opt.Get("GPANN_ROUTER_VERIFY") && !std::getenv("GPANN_ROUTER_DEBUG") && !r->opt.Get("GPANN_ROUTER_FORCE_RECHECK") &&
                       !r->opt.Get("GPANN_ROUTER_PROBE") && !r->opt.Get("GPANN_ROUTER_FORCE_EVIDENCE");
    GPANN_TRY(EnqueueBound(r, da, nb, st, ba, thr, /*convert_minima=*/defer));
    if (defer) {
        // no round trip in the middle of the call: the counters go to the host with the results (ResolvePendingBound)
        if (!r->pinned_counters) GPANN_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r->pinned_counters), 32, hipHostMallocDefault));
        GPANN_HIP_TRY(hipMemcpyAsync(r->pinned_counters, ba.cand_count, 20, hipMemcpyDeviceToHost, st));
        r->pending_bound.active = true;
        r->pending_bound.nb = nb;
        r->pending_bound.cand_cap = ba.cand_cap;
        *settled = true;
        return GPANN_OK;
    }
    uint32_t counters[5] = {0, 0, 0, 0, 0};   // [0] candidate slots handed out, [1] unset minima, [4] pairs pass 1 left without a maximum / finite threshold
    GPANN_HIP_TRY(hipMemcpyAsync(counters, ba.cand_count, 20, hipMemcpyDeviceToHost, st));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    ++r->stat_bound_calls;
    {   // matrix-core work of this block: both passes run 3 bf16 products over nb x C x d
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r->ev_bound[0], r->ev_bound[1]) == hipSuccess) r->stat_bound_ms += ms;
        r->stat_bound_flops += 2.0 * 2.0 * 3.0 * (double)nb * (double)r->n_centroids * (double)r->d;
    }
    const uint32_t n_cand = counters[0];
    if (std::getenv("GPANN_ROUTER_DEBUG")) fprintf(stderr, "[gpann router] bound: %u candidate slots for %u x %u (query, shard) pairs\n", n_cand, nb, r->S);
    if (r->opt.Get("GPANN_ROUTER_FORCE_RECHECK")) counters[1] = 1;   // test hook: take the canonical re-evaluation
    if (counters[1] != 0 || counters[4] != 0) {
        ++r->stat_bound_selfcheck;
        fprintf(stderr,
                "[gpann router] bound path self-check: pass 1 left %u of %zu (query, shard) pairs without a maximum or with a non-finite threshold, %u minima are unset "
                "after pass 2 (%u candidate slots, cap %u); evaluating every pair canonically instead\n",
                counters[4], nb_s, counters[1], n_cand, ba.cand_cap);
        const bool strict = r->opt.Get("GPANN_ROUTER_STRICT") && !r->opt.Get("GPANN_ROUTER_FORCE_RECHECK");
        if (strict || (std::getenv("GPANN_VERIFY_REPORT") && !r->opt.Get("GPANN_ROUTER_FORCE_RECHECK"))) {
            // the record of a rare event (3 in ~50 test suites of rounds 1-3, none in 7 200 stand-alone runs): the block's arrays, the same block once
            // more on the same inputs, the canonical minima — as text and as a binary dump (BoundEvidence)
            size_t bad = 0;
            BoundArgs ba2 = ba;
            float* thr2 = thr;
            const gpann_status ev = BoundEvidence(r, da, nb, st, ba2, thr2, caller_queries, caller_on_device, "self-check fired", &bad);
            if (strict) {
                SetLastError("router bound path self-check fired: pass 1 %u, pass 2 %u of %zu (query, shard) pairs; %zu minima differ from the canonical "
                             "evaluation%s (GPANN_ROUTER_STRICT)", counters[4], counters[1], nb_s, bad, ev == GPANN_OK ? "" : " [evidence incomplete]");
                return GPANN_ERR_INTERNAL;
            }
        }
        return GPANN_OK;
    }
    if (n_cand > ba.cand_cap) {
        ++r->stat_bound_overflow;
        return GPANN_OK;
    }
    if (r->opt.Get("GPANN_ROUTER_VERIFY")) GPANN_TRY(VerifyBound(r, da, nb, st, ba, thr, caller_queries, caller_on_device));
    if (r->opt.Get("GPANN_ROUTER_PROBE")) GPANN_TRY(ProbeBoundDeterminism(r, da, nb, st, ba, thr, caller_queries, caller_on_device));
    if (r->opt.Get("GPANN_ROUTER_FORCE_EVIDENCE")) {   // test hook: the evidence machinery on a healthy block (its report then says: nothing differs)
        size_t bad = 0;
        GPANN_TRY(BoundEvidence(r, da, nb, st, ba, thr, caller_queries, caller_on_device, "forced (GPANN_ROUTER_FORCE_EVIDENCE, a test hook)", &bad));
    }
    hipLaunchKernelGGL(MinBitsToFloatKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s);
    GPANN_HIP_TRY(hipGetLastError());
    *settled = true;
    return GPANN_OK;
}

// After the call's synchronisation: the deferred look at a bound block's counters.  *repeat = the block has to be evaluated canonically.
gpann_status ResolvePendingBound(gpann_router* r, bool* repeat) {
    *repeat = false;
    if (!r->pending_bound.active) return GPANN_OK;
    r->pending_bound.active = false;
    ++r->stat_bound_calls;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r->ev_bound[0], r->ev_bound[1]) == hipSuccess) r->stat_bound_ms += ms;
    r->stat_bound_flops += 2.0 * 2.0 * 3.0 * (double)r->pending_bound.nb * (double)r->n_centroids * (double)r->d;
    const uint32_t n_cand = r->pinned_counters[0], unset = r->pinned_counters[1], bad1 = r->pinned_counters[4];
    if (unset != 0 || bad1 != 0) {
        ++r->stat_bound_selfcheck;
        fprintf(stderr,
                "[gpann router] bound path self-check: pass 1 left %u of %zu (query, shard) pairs without a maximum or with a non-finite threshold, %u minima are unset "
                "after pass 2 (%u candidate slots); evaluating every pair canonically instead\n",
                bad1, (size_t)r->pending_bound.nb * r->S, unset, n_cand);
        if (r->opt.Get("GPANN_ROUTER_STRICT")) {
            SetLastError("router bound path self-check fired: pass 1 %u, pass 2 %u (query, shard) pairs (GPANN_ROUTER_STRICT)", bad1, unset);
            return GPANN_ERR_INTERNAL;
        }
        *repeat = true;
    } else if (n_cand > r->pending_bound.cand_cap) {
        ++r->stat_bound_overflow;
        *repeat = true;
    }
    return GPANN_OK;
}

// Stages the queries, sizes the queue workspace and launches the walk for queries [0, nq); min_dist lands in r->ws_out.
// rec_* (device, optional) receive every evaluated (distance, shard) pair.
const void* BoundedReplayFn(int metric, bool rec) {
    if (metric == GPANN_METRIC_L2) return rec ? (const void*)RouterBoundedReplayKernel<GPANN_METRIC_L2, true> : (const void*)RouterBoundedReplayKernel<GPANN_METRIC_L2, false>;
    return rec ? (const void*)RouterBoundedReplayKernel<GPANN_METRIC_MIPS, true> : (const void*)RouterBoundedReplayKernel<GPANN_METRIC_MIPS, false>;
}

// The bounded replay's table for a block of queries: BoundFrameKernel (row bases, eps) + RouterBoundKernel<.., PASS 3> (router_bound.h).
gpann_status EnqueueBoundTable(gpann_router* r, const float* d_queries, uint32_t nb, hipStream_t st, float* table, uint64_t table_ld, float* base, float* eps) {
    float *h_lo = eps + nb, *h_scale = eps + 2 * (size_t)nb;   // (the frame: four arrays of nb floats, base first)
    double eps_scale = 1.0;
    if (const char* e = r->opt.Get("GPANN_ROUTER_EPS_SCALE")) eps_scale = std::max(1.0, std::atof(e));   // test knob: a wider band, the same bits
    BoundArgs ba{};
    ba.queries = d_queries;
    ba.cent_planes = r->cent_planes;
    ba.cent_nhn = r->cent_nhn;
    ba.cent_shard = r->cent_shard;
    ba.nq = nb;
    ba.C = (uint32_t)r->n_centroids;
    ba.S = r->S;
    ba.ld = r->ld;
    ba.mu = r->cent_mu;
    ba.plane_ks = r->plane_ks;
    ba.table = table;
    ba.table_ld = table_ld;
    ba.table_base = base;
    const bool l2 = r->metric == GPANN_METRIC_L2, ks16 = r->d > 128, ks4 = r->d <= 64;
    const uint32_t group_q = ks16 ? 128u : 256u;
    const uint32_t groups = (nb + group_q - 1) / group_q;
    const uint32_t n_tiles = (uint32_t)((r->n_centroids + kGTileRows - 1) / kGTileRows);
    // (several rounds of workgroups are fine here: a workgroup's share is hundreds of tiles)
    ba.n_seg = std::max(1u, std::min((n_tiles + 63) / 64, std::max(1u, 2u * (uint32_t)r->num_cus / groups)));
    if (l2) hipLaunchKernelGGL(BoundFrameKernel<GPANN_METRIC_L2>, dim3(nb), dim3(64), 0, st, d_queries, r->d, r->ld, r->cent_rmax, r->cent_mu, base, eps, h_lo, h_scale, eps_scale);
    else hipLaunchKernelGGL(BoundFrameKernel<GPANN_METRIC_MIPS>, dim3(nb), dim3(64), 0, st, d_queries, r->d, r->ld, r->cent_rmax, r->cent_mu, base, eps, h_lo, h_scale, eps_scale);
    GPANN_HIP_TRY(hipGetLastError());
    const size_t blds = (size_t)(ks16 ? 4 * 16 : 8 * (ks4 ? 4 : 8)) * 2048 + 8 * kGTileRows * sizeof(float);
    const void* f3 = l2 ? (ks16 ? (const void*)RouterBoundKernel<GPANN_METRIC_L2, 16, 3> : ks4 ? (const void*)RouterBoundKernel<GPANN_METRIC_L2, 4, 3> : (const void*)RouterBoundKernel<GPANN_METRIC_L2, 8, 3>)
                        : (ks16 ? (const void*)RouterBoundKernel<GPANN_METRIC_MIPS, 16, 3> : ks4 ? (const void*)RouterBoundKernel<GPANN_METRIC_MIPS, 4, 3> : (const void*)RouterBoundKernel<GPANN_METRIC_MIPS, 8, 3>);
    GPANN_HIP_TRY(hipFuncSetAttribute(f3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
    void* kargs[] = {&ba};
    GPANN_HIP_TRY(hipEventRecord(r->ev_bound[0], st));
    GPANN_HIP_TRY(hipLaunchKernel(f3, dim3(ba.n_seg, groups), dim3(512), kargs, blds, st));
    GPANN_HIP_TRY(hipEventRecord(r->ev_bound[1], st));
    return GPANN_OK;
}

// rec_keep (FrequencyQuery with a device-side selection): only the rec_keep smallest pairs of every query will be looked at — the bounded replay
// then records just the pairs that can be among them; 0 = every evaluated pair is needed.
gpann_status LaunchWalk(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, int flags, hipStream_t st, float* rec_dist,
                        int32_t* rec_shard, uint32_t* rec_count, uint64_t rec_cap, uint32_t rec_keep = 0, bool rec_small = false, bool* rec_truncated = nullptr) {
    // rec_small: rec_cap holds the pairs that can be among the rec_keep smallest, not every evaluated pair (the bounded replay records only
    // those; the host's walks of the marked queries keep their rec_cap smallest).  *rec_truncated = some query's records did not fit, or the
    // call would have taken a path that records in evaluation order: the caller repeats it with room for every evaluated pair.
    const uint32_t S = r->S, ld = r->ld;
    const size_t qbytes = (size_t)nq * ld * sizeof(float);
    r->stat_bound_ms = r->stat_bound_flops = 0.0;
    const float* d_q = queries;   // device queries that already have the padded row layout are read in place
    if (!((flags & GPANN_QUERIES_ON_DEVICE) && ld == r->d && (reinterpret_cast<uintptr_t>(queries) & 15u) == 0)) {
        GPANN_TRY(r->ws_queries.Reserve(qbytes));
        if (ld != r->d) GPANN_HIP_TRY(hipMemsetAsync(r->ws_queries.ptr, 0, qbytes, st));
        GPANN_HIP_TRY(hipMemcpy2DAsync(r->ws_queries.ptr, (size_t)ld * 4, queries, (size_t)r->d * 4, (size_t)r->d * 4, nq,
                                       (flags & GPANN_QUERIES_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        d_q = r->ws_queries.as<float>();
    }

    // every node enters the queue at most once (when its parent is popped), so n_nodes bounds the queue; so does
    // S + (#pops <= budget) * max_children
    uint64_t cap = r->n_nodes;
    if (budget > 0) cap = std::min<uint64_t>(cap, (uint64_t)S + (uint64_t)budget * std::max<uint32_t>(1, r->max_children));
    cap = std::max<uint64_t>(cap, S);
    struct { int multiProcessorCount; } prop{r->num_cus};
    GPANN_TRY(r->ws_out.Reserve((size_t)nq * S * 4));
    const char* force = r->opt.Get("GPANN_ROUTER_MODE");   // "walk" | "dense" (default: by size)
    const uint64_t C = r->n_centroids;
    // dense + replay: all distances of a block of queries at once (exact, tiled), then the queue logic on the table
    const uint64_t dense_cap = 6ull << 30;   // bytes of distance table per block of queries
    // (which is cheaper: measured on the NS trees, search budget 50 000 — the walk costs ~1.2 us per (query, evaluated centroid) in its queue and
    // its L2 traffic (252k-centroid tree: 602 ms per 10k queries), the table 0.02 us per (query, centroid) (139k centroids: 26 ms): the table
    // wins until the tree is ~60x the budget.  Rounds 1-2 switched at 4x and sent the 200k and 1M indices of the reference's grid to the walk.)
    const uint64_t dense_factor = (uint64_t)std::max(1, std::atoi(r->opt.Get("GPANN_ROUTER_DENSE_FACTOR") ? r->opt.Get("GPANN_ROUTER_DENSE_FACTOR") : "32"));
    bool dense = budget > 0 && C <= dense_factor * (uint64_t)budget && C < 0xFFFFFFFFull && C * 4 * 64 <= dense_cap;
    if (force && std::string(force) == "walk") dense = false;
    if (force && std::string(force) == "dense" && C * 4 * 64 <= dense_cap) dense = budget > 0;
    if (rec_small && !dense) {
        if (rec_truncated) *rec_truncated = true;
        return GPANN_OK;
    }
    GPANN_HIP_TRY(hipEventRecord(r->ev[0], st));
    if (dense) {
        const char* rmode = r->opt.Get("GPANN_ROUTER_REPLAY");   // "heap" = always the real queue; "threshold" = skip the sorted replay
        const bool force_heap = rmode && std::string(rmode) == "heap";
        const bool all_roots_open = (int64_t)budget > (int64_t)r->root_total - (int64_t)r->root_min;
        uint32_t sort_n = 1;
        while (sort_n < std::max<uint32_t>(r->root_children, 64)) sort_n <<= 1;
        // trees whose root children are all leaves: sorted replay; any other tree: threshold replay.  Both mark the rare query
        // that needs the real heap (equal keys in the cut-off group) and RouterReplayKernel then does only those.
        const bool sorted_ok = r->leaves_only && sort_n <= 2048 && r->n_nodes <= 4096 && S <= 64 && all_roots_open && !force_heap &&
                               !(rmode && std::string(rmode) == "threshold");
        const size_t t_lds = ((((size_t)256 + S + 2 * kFrontCap) + 3) & ~(size_t)3) * kRWaves * sizeof(uint32_t);
        const bool threshold_ok = !sorted_ok && !force_heap && all_roots_open && r->max_depth <= (uint32_t)kPathMax && t_lds <= 64 * 1024 &&
                                  budget < 0x7FFFFFFF;
        // every node is visited iff the budget left after the roots exceeds all non-root sizes together, i.e. budget > C
        const bool min_only = all_roots_open && (uint64_t)budget > C && rec_dist == nullptr && !force_heap;
        // the threshold replay on a table the matrix cores wrote, canonical evaluation of the cut-off band only (RouterBoundedReplayKernel)
        const char* tmode = r->opt.Get("GPANN_ROUTER_BOUND_TABLE");   // "0": the canonical table (DenseDistKernel) for every tree
        const uint64_t table_ld = (C + kGTileRows - 1) / kGTileRows * kGTileRows;
        const bool bounded_able = r->cent_planes != nullptr && !(tmode && tmode[0] == '0') && !r->canonical_once && r->cent_node8 != nullptr &&
                                  r->n_nodes < (1u << kBandNodeBits) && r->max_node < (1u << (32 - kBandNodeBits)) &&
                                  ((size_t)1024 + S + 3 * 4096 + r->n_nodes / 32 + 64) * 4 + (10u << 10) <= r->lds_per_wg;   // (the visited bitmap lives in LDS;
                                                                                       // against the DEVICE's limit, not a constant: a part with less LDS takes the canonical table)
        // FrequencyQuery with a budget above the tree: every centroid is visited, no tree logic at all — the votes are the pairs within the error
        // of the V-th smallest table value (any tree shape: the sorted replay's trees too)
        const bool all_rec = rec_dist != nullptr && rec_keep > 0 && all_roots_open && (uint64_t)budget > C && !force_heap;
        const bool bounded = bounded_able && ((threshold_ok && !min_only && (rec_dist == nullptr || rec_keep > 0) && (uint64_t)budget <= C) || all_rec);
        if (rec_small && !bounded) {
            if (rec_truncated) *rec_truncated = true;
            return GPANN_OK;
        }
        const uint64_t row_floats = bounded ? table_ld : C;
        const uint32_t q_block = (uint32_t)std::min<uint64_t>(nq, std::max<uint64_t>(64, dense_cap / (row_floats * 4) / 64 * 64));
        GPANN_TRY(r->ws_dense.Reserve(std::max<size_t>((size_t)q_block * row_floats * 4, 256)));
        uint32_t band_cap = 512;
        if (const char* e = r->opt.Get("GPANN_ROUTER_BAND_CAP")) band_cap = (uint32_t)std::min(4096, std::max(96, std::atoi(e)));
        for (uint32_t q0 = 0; q0 < nq; q0 += q_block) {
            const uint32_t nb = std::min(q_block, nq - q0);
            DenseArgs da;
            da.centroids = r->centroids;
            da.queries = d_q + (size_t)q0 * ld;
            da.out = r->ws_dense.as<float>();
            da.d = r->d;
            da.ld = ld;
            da.nq = nb;
            da.C = (uint32_t)C;
            const dim3 dgrid((uint32_t)((C + 63) / 64), (nb + 63) / 64);
            const size_t dlds = (size_t)2 * 64 * (kRDC + 4) * sizeof(float);   // (>= the 64 x 65 floats of the output tile)
            da.cent_shard = r->cent_shard;
            da.out_min_bits = reinterpret_cast<uint32_t*>(r->ws_out.as<float>() + (size_t)q0 * S);
            da.S = S;
            if (min_only) {
                // the budget exceeds the tree and nobody asked for the evaluated distances: the per-shard minima are the whole answer
                const size_t nb_s = (size_t)nb * S;
                const char* bmode = r->opt.Get("GPANN_ROUTER_BOUND");   // "0": canonical evaluation of every pair (DenseDistKernel)
                if (r->cent_planes && !(bmode && bmode[0] == '0') && !r->canonical_once) {
                    // matrix-core bound, canonical evaluation of the few pairs that can be the minimum (router_bound.h; its init launch sets the minima to FLT_MAX)
                    bool settled = false;
                    GPANN_TRY(RunBoundPath(r, da, nb, st, &settled, queries + (size_t)q0 * r->d, (flags & GPANN_QUERIES_ON_DEVICE) != 0, /*may_defer=*/nb == nq));
                    if (settled) continue;
                    // (masses of equal distances / self-check) fall through to the canonical evaluation of every pair
                }
                hipLaunchKernelGGL(FillMinBitsKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s);
                if (r->metric == GPANN_METRIC_L2) hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_L2, true>), dgrid, dim3(256), dlds, st, da);
                else hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_MIPS, true>), dgrid, dim3(256), dlds, st, da);
                hipLaunchKernelGGL(MinBitsToFloatKernel, dim3((uint32_t)((nb_s + 255) / 256)), dim3(256), 0, st, da.out_min_bits, nb_s);
                GPANN_HIP_TRY(hipGetLastError());
                continue;
            }
            auto canonical_table = [&]() -> gpann_status {
                if (r->metric == GPANN_METRIC_L2) hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_L2, false>), dgrid, dim3(256), dlds, st, da);
                else hipLaunchKernelGGL((DenseDistKernel<GPANN_METRIC_MIPS, false>), dgrid, dim3(256), dlds, st, da);
                GPANN_HIP_TRY(hipGetLastError());
                return GPANN_OK;
            };
            float *frame_base = nullptr, *frame_eps = nullptr;
            if (bounded) {
                GPANN_TRY(r->ws_frame.Reserve((size_t)4 * nb * 4 + 64));
                frame_base = r->ws_frame.as<float>();
                frame_eps = frame_base + nb;
                GPANN_TRY(EnqueueBoundTable(r, da.queries, nb, st, da.out, table_ld, frame_base, frame_eps));
            } else {
                GPANN_TRY(canonical_table());
            }

            const size_t base_floats = (size_t)64 + S;
            const bool lds_heap = ((((base_floats + 2 * cap) + 3) & ~(size_t)3) * kRWaves * sizeof(float)) <= 40 * 1024;
            const size_t per_wave_lds = ((base_floats + (lds_heap ? 2 * cap : 0)) + 3) & ~(size_t)3;
            const size_t lds = per_wave_lds * kRWaves * sizeof(float);
            uint32_t blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(lds, 1)));
            uint32_t grid = std::min<uint32_t>((nb + kRWaves - 1) / kRWaves, (uint32_t)prop.multiProcessorCount * blocks_per_cu);
            while (!lds_heap && grid > 1 && (uint64_t)grid * kRWaves * cap * 8 > (4ull << 30)) grid /= 2;
            const uint64_t slots = (uint64_t)grid * kRWaves;
            GPANN_TRY(r->ws_heap.Reserve(lds_heap ? 256 : slots * cap * 8));
            ReplayArgs ra;
            ra.dense = r->ws_dense.as<float>();
            ra.node_shard = r->node_shard;
            ra.node_cent_first = r->node_cent_first;
            ra.node_child_first = r->node_child_first;
            ra.node_child_count = r->node_child_count;
            ra.shard_root = r->shard_root;
            ra.heap_key = r->ws_heap.as<float>();
            ra.heap_node = reinterpret_cast<uint32_t*>(ra.heap_key + slots * cap);
            ra.out_min_dist = r->ws_out.as<float>() + (size_t)q0 * S;
            ra.rec_dist = rec_dist ? rec_dist + (size_t)q0 * rec_cap : nullptr;
            ra.rec_shard = rec_shard ? rec_shard + (size_t)q0 * rec_cap : nullptr;
            ra.rec_count = rec_count ? rec_count + q0 : nullptr;
            ra.rec_cap = rec_cap;
            ra.heap_cap = cap;
            ra.S = S;
            ra.nq = nb;
            ra.C = (uint32_t)C;
            ra.budget = budget;
            ra.only_redo = 0;
            ra.redo = nullptr;
            ra.redo_count = nullptr;
            ra.root_total = r->root_total;
            ra.sort_n = sort_n;
            ra.n_nodes = r->n_nodes;
            if (sorted_ok || threshold_ok || bounded) {
                GPANN_TRY(r->ws_redo.Reserve((size_t)nb * 4 + 64));
                ra.redo = r->ws_redo.as<uint32_t>();
                ra.redo_count = ra.redo + nb;   // (+ 1..4: the bounded replay's counters)
                GPANN_HIP_TRY(hipMemsetAsync(ra.redo_count, 0, 56, st));
            }
            if (sorted_ok && !bounded) {
                const size_t table_words = ((size_t)(r->n_nodes + 1) + (r->n_nodes + 1) / 2 + 3) & ~(size_t)3;
                const size_t s_lds = (table_words + ((((size_t)2 * ra.sort_n + S) + 3) & ~(size_t)3) * kRWaves) * sizeof(float);
                const uint32_t s_bpc = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / s_lds));
                const uint32_t s_grid = std::min<uint32_t>((nb + kRWaves - 1) / kRWaves, (uint32_t)prop.multiProcessorCount * s_bpc);
                if (s_lds > 48 * 1024)
                    GPANN_HIP_TRY(hipFuncSetAttribute((const void*)RouterSortedReplayKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s_lds));
                hipLaunchKernelGGL(RouterSortedReplayKernel, dim3(s_grid), dim3(64 * kRWaves), s_lds, st, ra);
                GPANN_HIP_TRY(hipGetLastError());
                ra.only_redo = 1;
            } else if (threshold_ok || bounded) {
                // a wavefront per query while the pi scratch (n_nodes words per wavefront) stays below 2 GiB: the kernel lives on wavefronts in flight
                const uint32_t t_bpc = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / t_lds));
                const uint64_t t_fit = std::max<uint64_t>(1, (2ull << 30) / ((uint64_t)kRWaves * r->n_nodes * 4));
                const uint32_t t_grid = (uint32_t)std::min<uint64_t>((nb + kRWaves - 1) / kRWaves,
                                                                     std::max<uint64_t>((uint64_t)prop.multiProcessorCount * t_bpc, std::min<uint64_t>(t_fit, 1u << 20)));
                // (the bounded replay: a workgroup per query; 1024 + S + 3 band_cap + n_nodes / 32 + 32 words of LDS)
                const size_t b_lds = ((size_t)1024 + ((S + 3u) & ~3u) + 3 * (size_t)band_cap + (((r->n_nodes + 31u) / 32u + 3u) & ~3u) + 32) * sizeof(uint32_t);
                uint32_t b_grid = 1;
                if (bounded) {   // exactly the workgroups that are resident together (registers: 4 per CU today)
                    const void* fb = BoundedReplayFn(r->metric, rec_dist != nullptr);
                    if (b_lds > 48 * 1024) GPANN_HIP_TRY(hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b_lds));
                    int per_cu = 0;
                    GPANN_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fb, 64 * kBWaves, b_lds));
                    b_grid = std::min<uint32_t>(nb, (uint32_t)prop.multiProcessorCount * (uint32_t)std::max(1, per_cu));
                }
                GPANN_TRY(r->ws_pi.Reserve(bounded ? (size_t)b_grid * r->n_nodes * 4 : (size_t)t_grid * kRWaves * r->n_nodes * 4));
                ThresholdTree tt;
                tt.path_cent = r->path_cent;
                tt.cent_node = r->cent_node;
                tt.pi = r->ws_pi.as<uint32_t>();
                tt.cent_node8 = r->cent_node8;
                tt.cent_block_node = r->cent_block_node;
                tt.lvl_node = r->lvl_node;
                tt.lvl_parent = r->lvl_parent;
                tt.lvl_key = r->lvl_key;
                for (int l = 0; l < kPathMax + 2; ++l) tt.lvl_first[l] = r->lvl_first[l];
                tt.n_levels = r->n_levels;
                tt.visit_all = (uint64_t)budget > C ? 1u : 0u;   // then budget_left exceeds the sizes of all non-root nodes together
                if (bounded) {
                    BoundedTable bt;
                    bt.table = da.out;
                    bt.table_ld = table_ld;
                    bt.eps = frame_eps;
                    bt.h_lo = frame_eps + nb;
                    bt.h_scale = frame_eps + 2 * (size_t)nb;
                    bt.centroids = r->centroids;
                    bt.queries = da.queries;
                    bt.cent_shard = r->cent_shard;
                    bt.d = r->d;
                    bt.ld = ld;
                    bt.band_cap = band_cap;
                    bt.stats = ra.redo_count + 1;
                    bt.ticket = ra.redo_count + 12;
                    bt.rec_keep = rec_keep;
                    const void* fb = BoundedReplayFn(r->metric, rec_dist != nullptr);
                    if (b_lds > 48 * 1024) GPANN_HIP_TRY(hipFuncSetAttribute(fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b_lds));
                    void* kargs[] = {&ra, &tt, &bt};
                    GPANN_HIP_TRY(hipLaunchKernel(fb, dim3(b_grid), dim3(64 * kBWaves), kargs, b_lds, st));
                } else {
                    if (t_lds > 48 * 1024)
                        GPANN_HIP_TRY(hipFuncSetAttribute((const void*)RouterThresholdReplayKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t_lds));
                    hipLaunchKernelGGL(RouterThresholdReplayKernel, dim3(t_grid), dim3(64 * kRWaves), t_lds, st, ra, tt);
                }
                GPANN_HIP_TRY(hipGetLastError());
                ra.only_redo = 1;
            }
            // after a sorted / threshold replay the heap kernel is needed only for the marked queries — usually none
            bool need_heap = true;
            uint32_t hgrid = grid;
            uint32_t marked = 0;
            if (ra.only_redo) {
                uint32_t words[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                GPANN_HIP_TRY(hipMemcpyAsync(words, ra.redo_count, sizeof(words), hipMemcpyDeviceToHost, st));
                GPANN_HIP_TRY(hipStreamSynchronize(st));
                marked = words[0];
                need_heap = marked != 0;
                r->stat_heap_redo += marked;
                if (bounded) {
                    ++r->stat_table_blocks;
                    r->stat_table_queries += nb;
                    r->stat_table_band_nodes += words[1];
                    r->stat_table_exact += words[2];
                    r->stat_table_band_overflow += words[3];
                    r->stat_table_rescans += words[4];
                    if (rec_dist && words[11] != 0 && rec_truncated) *rec_truncated = true;
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, r->ev_bound[0], r->ev_bound[1]) == hipSuccess) r->stat_bound_ms += ms;
                    r->stat_bound_flops += 3.0 * 2.0 * (double)nb * (double)C * (double)r->d;
                    if (std::getenv("GPANN_ROUTER_DEBUG"))
                        fprintf(stderr, "[gpann router] bounded table: %u queries x %llu centroids in %.3f ms; band %.1f nodes, %.1f canonical evaluations per query, %u band overflows, %u rescans\n",
                                nb, (unsigned long long)C, ms, (double)words[1] / nb, (double)words[2] / nb, words[3], words[4]);
                    if (std::getenv("GPANN_ROUTER_DEBUG"))
                        fprintf(stderr, "[gpann router] bounded replay, us per query: bins %.1f, select %.1f, band pass %.1f, band keys + group %.1f, minima %.1f\n",
                                words[5] / 100.0 / nb, words[6] / 100.0 / nb, words[7] / 100.0 / nb, words[8] / 100.0 / nb, words[9] / 100.0 / nb);
                }
                if (std::getenv("GPANN_ROUTER_DEBUG"))
                    fprintf(stderr, "[gpann router] %s replay: %u of %u queries need the real priority queue\n", sorted_ok ? "sorted" : "threshold", marked, nb);
                hgrid = std::min<uint32_t>(grid, std::max<uint32_t>(1, (marked + kRWaves - 1) / kRWaves * 4));
            }
            if (need_heap && ra.only_redo && !(r->opt.Get("GPANN_ROUTER_HOST_REDO") && r->opt.Get("GPANN_ROUTER_HOST_REDO")[0] == '0')) {
                // the marked queries go to host threads and the real std::priority_queue (HostReplayMarked); the device heap kernel
                // stays for batches whose marked rows would not fit the staging budget
                bool done = false;
                GPANN_TRY(HostReplayMarked(r, ra.dense, ra.redo, nb, marked, budget, ra.out_min_dist, ra.rec_dist, ra.rec_shard,
                                           ra.rec_count, ra.rec_cap, st, &done, bounded ? &da : nullptr, rec_small));
                need_heap = !done;
            }
            if (need_heap && rec_small) {   // (the device queue records in evaluation order)
                if (rec_truncated) *rec_truncated = true;
                return GPANN_OK;
            }
            if (need_heap && bounded) GPANN_TRY(canonical_table());   // (the device queue walks canonical values: the block's table once more, on the vector ALUs)
            if (need_heap) {
                if (lds_heap) hipLaunchKernelGGL(RouterReplayKernel<true>, dim3(hgrid), dim3(64 * kRWaves), lds, st, ra);
                else hipLaunchKernelGGL(RouterReplayKernel<false>, dim3(hgrid), dim3(64 * kRWaves), lds, st, ra);
                GPANN_HIP_TRY(hipGetLastError());
            }
        }
        GPANN_HIP_TRY(hipEventRecord(r->ev[1], st));
        return GPANN_OK;
    }

    // the queue goes to LDS when it leaves room for 4 workgroups per CU, otherwise to the global workspace
    const size_t base_floats = (size_t)ld + 64 * kRStride + 64 + S;
    const bool lds_heap = ((((base_floats + 2 * cap) + 3) & ~(size_t)3) * kRWaves * sizeof(float)) <= 40 * 1024;
    const size_t per_wave_lds = ((base_floats + (lds_heap ? 2 * cap : 0)) + 3) & ~(size_t)3;
    const size_t lds = per_wave_lds * kRWaves * sizeof(float);
    if (lds > 160 * 1024) {
        SetLastError("router needs %zu bytes of LDS per workgroup (d = %u, S = %u): unsupported", lds, r->d, S);
        return GPANN_ERR_UNSUPPORTED;
    }
    uint32_t blocks_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
    uint32_t grid = std::min<uint32_t>((nq + kRWaves - 1) / kRWaves, (uint32_t)prop.multiProcessorCount * blocks_per_cu);
    // keep the queue workspace bounded (<= 4 GiB)
    while (!lds_heap && grid > 1 && (uint64_t)grid * kRWaves * cap * 8 > (4ull << 30)) grid /= 2;
    const uint64_t slots = (uint64_t)grid * kRWaves;
    GPANN_TRY(r->ws_heap.Reserve(lds_heap ? 256 : slots * cap * 8));

    RouterArgs a;
    a.centroids = r->centroids;
    a.node_shard = r->node_shard;
    a.node_cent_first = r->node_cent_first;
    a.node_child_first = r->node_child_first;
    a.node_child_count = r->node_child_count;
    a.shard_root = r->shard_root;
    a.queries = d_q;
    a.heap_key = r->ws_heap.as<float>();
    a.heap_node = reinterpret_cast<uint32_t*>(a.heap_key + slots * cap);
    a.out_min_dist = r->ws_out.as<float>();
    a.rec_dist = rec_dist;
    a.rec_shard = rec_shard;
    a.rec_count = rec_count;
    a.rec_cap = rec_cap;
    a.d = r->d;
    a.ld = ld;
    a.S = S;
    a.nq = nq;
    a.heap_cap = cap;
    a.budget = budget;
    const dim3 g(grid), b(64 * kRWaves);
    if (lds > 48 * 1024) {   // dynamic LDS beyond the default limit has to be requested per kernel
        const void* fn = r->metric == GPANN_METRIC_L2
                             ? (lds_heap ? (const void*)RouterWalkKernel<GPANN_METRIC_L2, true> : (const void*)RouterWalkKernel<GPANN_METRIC_L2, false>)
                             : (lds_heap ? (const void*)RouterWalkKernel<GPANN_METRIC_MIPS, true> : (const void*)RouterWalkKernel<GPANN_METRIC_MIPS, false>);
        GPANN_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (r->metric == GPANN_METRIC_L2) {
        if (lds_heap) hipLaunchKernelGGL((RouterWalkKernel<GPANN_METRIC_L2, true>), g, b, lds, st, a);
        else hipLaunchKernelGGL((RouterWalkKernel<GPANN_METRIC_L2, false>), g, b, lds, st, a);
    } else {
        if (lds_heap) hipLaunchKernelGGL((RouterWalkKernel<GPANN_METRIC_MIPS, true>), g, b, lds, st, a);
        else hipLaunchKernelGGL((RouterWalkKernel<GPANN_METRIC_MIPS, false>), g, b, lds, st, a);
    }
    GPANN_HIP_TRY(hipGetLastError());
    GPANN_HIP_TRY(hipEventRecord(r->ev[1], st));
    return GPANN_OK;
}


// The final order of KMeansTreeRouter::Query (kmeans_tree_router.cpp:125-127: std::sort of the shard ids by min_dist) on the device,
// one wavefront per query, S <= 64: a bitonic sort of (min_dist, shard).  Where all min_dist values of a query differ, every correct
// sort returns the same permutation as the reference's; a query with two equal values (duplicate centroids, unvisited shards at
// FLT_MAX) is flagged and the host redoes it with the reference's own unstable std::sort call.
__global__ void __launch_bounds__(256) ProbeOrderKernel(const float* min_dist, uint32_t nq, uint32_t S, uint32_t cols, int32_t* probes, uint32_t* tie_flag) {
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nq) return;
    float key = (uint32_t)lane < S ? min_dist[(size_t)q * S + lane] : INFINITY;   // (real values are <= FLT_MAX: the padding sorts last)
    int id = lane;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const float okey = __shfl_xor(key, j);
            const int oid = __shfl_xor(id, j);
            const bool up = (lane & k) == 0;            // this block of k lanes sorts ascending
            const bool lower = (lane & j) == 0;         // this lane keeps the smaller element of the pair when ascending
            const bool other_less = okey < key || (okey == key && oid < id);
            const bool take = (up == lower) ? other_less : !other_less;
            if (take) {
                key = okey;
                id = oid;
            }
        }
    }
    const float next = __shfl_down(key, 1);
    const bool tie = (uint32_t)lane + 1 < S && next == key;
    const unsigned long long any = __ballot(tie);
    if ((uint32_t)lane < cols) probes[(size_t)q * cols + lane] = id;   // (the first `cols` shards of the order)
    if (lane == 0) tie_flag[q] = any != 0ull ? 1u : 0u;
}

}  // namespace

gpann_status gpann_router_query_batch(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, int32_t* out_probes,
                                      float* out_min_dist, int flags, void* stream, float* elapsed_ms) {
    GPANN_REQUIRE(r != nullptr, "router handle is null");
    return gpann_router_query_batch_top(r, queries, nq, budget, r->S, out_probes, out_min_dist, flags, stream, elapsed_ms);
}

static gpann_status QueryBatchTopImpl(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t n_top, int32_t* out_probes,
                                      float* out_min_dist, int flags, void* stream, float* elapsed_ms);

gpann_status gpann_router_query_batch_top(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t n_top, int32_t* out_probes,
                                          float* out_min_dist, int flags, void* stream, float* elapsed_ms) {
    GPANN_REQUIRE(r != nullptr, "router handle is null");
    std::lock_guard<std::recursive_mutex> turn(r->call_mutex);
    const gpann_status st = QueryBatchTopImpl(r, queries, nq, budget, n_top, out_probes, out_min_dist, flags, stream, elapsed_ms);
    // a call that failed between the launch of a bound block and the look at its counters must not leave them to the next call
    if (st != GPANN_OK && r) r->pending_bound.active = false;
    return st;
}

static gpann_status QueryBatchTopImpl(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t n_top, int32_t* out_probes,
                                      float* out_min_dist, int flags, void* stream, float* elapsed_ms) {
    GPANN_REQUIRE(r != nullptr, "router handle is null");
    GPANN_REQUIRE(nq == 0 || queries != nullptr, "queries is null");
    GPANN_REQUIRE(n_top >= 1 && n_top <= r->S, "n_top must be in [1, S]");
    GPANN_TRY(EnsureDevice(r->device));
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (nq == 0) return GPANN_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t S = r->S, cols = n_top;
    GPANN_TRY(LaunchWalk(r, queries, nq, budget, flags, st, nullptr, nullptr, nullptr, 0));

    const size_t words = (size_t)nq * S, pwords = (size_t)nq * cols;   // min_dist words, probe words
    auto pinned = [&](void** p, size_t* have, size_t need_words) {
        if (*have >= need_words) return true;
        if (*p) (void)hipHostFree(*p);
        *p = nullptr;
        *have = 0;
        if (hipHostMalloc(p, need_words * 4, hipHostMallocDefault) == hipSuccess) *have = need_words;
        else (void)hipGetLastError();
        return *have >= need_words;
    };
    // min_dist to the host: always when the caller wants it; for the probe order only if the device could not settle it
    std::vector<float> local_min;
    float* md = nullptr;
    auto fetch_min_dist = [&]() -> gpann_status {
        if (md) return GPANN_OK;
        md = out_min_dist;
        if (!md) {
            local_min.resize(words);
            md = local_min.data();
        }
        float* stage = pinned((void**)&r->pinned_min, &r->pinned_floats, words) ? r->pinned_min : md;
        GPANN_HIP_TRY(hipMemcpyAsync(stage, r->ws_out.ptr, words * 4, hipMemcpyDeviceToHost, st));
        GPANN_HIP_TRY(hipStreamSynchronize(st));
        if (stage != md && out_min_dist) std::memcpy(md, stage, words * 4);
        if (stage != md) md = stage;
        return GPANN_OK;
    };
    // kmeans_tree_router.cpp:125-127 — the reference's own call on the GPU-computed min_dist
    auto host_sort = [&](uint32_t q) {
        const float* min_dist = md + (size_t)q * S;
        int order[4096];
        std::vector<int> big;
        int* probes = order;
        if (S > 4096) {
            big.resize(S);
            probes = big.data();
        }
        std::iota(probes, probes + S, 0);
        std::sort(probes, probes + S, [&](int l, int rr) { return min_dist[l] < min_dist[rr]; });
        std::memcpy(out_probes + (size_t)q * cols, probes, (size_t)cols * sizeof(int));
    };

    if (out_probes && S <= 64 && pinned((void**)&r->pinned_probes, &r->pinned_probe_words, pwords + nq)) {
        // the order on the device (ProbeOrderKernel); the host only redoes the queries with equal min_dist values
        GPANN_TRY(r->ws_probes.Reserve((pwords + nq) * 4));
        int32_t* d_probes = r->ws_probes.as<int32_t>();
        uint32_t* d_flags = reinterpret_cast<uint32_t*>(d_probes + pwords);
        hipLaunchKernelGGL(ProbeOrderKernel, dim3((nq + 3) / 4), dim3(256), 0, st, r->ws_out.as<float>(), nq, S, cols, d_probes, d_flags);
        GPANN_HIP_TRY(hipGetLastError());
        // a caller-provided pinned array receives the lists directly; otherwise they pass through the handle's pinned staging
        hipPointerAttribute_t attr{};
        const bool direct = hipPointerGetAttributes(&attr, out_probes) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!direct) (void)hipGetLastError();
        if (direct) {
            GPANN_HIP_TRY(hipMemcpyAsync(out_probes, d_probes, pwords * 4, hipMemcpyDeviceToHost, st));
            GPANN_HIP_TRY(hipMemcpyAsync(r->pinned_probes + pwords, d_flags, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
        } else {
            GPANN_HIP_TRY(hipMemcpyAsync(r->pinned_probes, d_probes, (pwords + nq) * 4, hipMemcpyDeviceToHost, st));
        }
        if (out_min_dist) GPANN_TRY(fetch_min_dist());
        else GPANN_HIP_TRY(hipStreamSynchronize(st));
        {
            bool repeat = false;
            GPANN_TRY(ResolvePendingBound(r, &repeat));
            if (repeat) {   // (rare) the whole call again without the bound path
                r->canonical_once = true;
                const gpann_status again = gpann_router_query_batch_top(r, queries, nq, budget, n_top, out_probes, out_min_dist, flags, stream, elapsed_ms);
                r->canonical_once = false;
                return again;
            }
        }
        if (elapsed_ms) GPANN_HIP_TRY(hipEventElapsedTime(elapsed_ms, r->ev[0], r->ev[1]));
        if (!direct) std::memcpy(out_probes, r->pinned_probes, pwords * 4);
        const uint32_t* flags = reinterpret_cast<const uint32_t*>(r->pinned_probes + pwords);
        std::vector<uint32_t> redo;
        for (uint32_t q = 0; q < nq; ++q)
            if (flags[q]) redo.push_back(q);
        r->stat_host_sorted += redo.size();
        if (!redo.empty()) {
            GPANN_TRY(fetch_min_dist());
            const uint32_t chunks = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, ((uint64_t)redo.size() * S) / 8192));
            HostPool::Get().Run(chunks, [&](uint32_t c) {
                for (size_t i = redo.size() * c / chunks; i < redo.size() * (c + 1) / chunks; ++i) host_sort(redo[i]);
            });
        }
        return GPANN_OK;
    }

    GPANN_TRY(fetch_min_dist());
    {
        bool repeat = false;
        GPANN_TRY(ResolvePendingBound(r, &repeat));
        if (repeat) {
            r->canonical_once = true;
            const gpann_status again = gpann_router_query_batch_top(r, queries, nq, budget, n_top, out_probes, out_min_dist, flags, stream, elapsed_ms);
            r->canonical_once = false;
            return again;
        }
    }
    if (elapsed_ms) GPANN_HIP_TRY(hipEventElapsedTime(elapsed_ms, r->ev[0], r->ev[1]));
    if (out_probes) {
        // queries are independent, so the batch is spread over a few host threads
        const uint32_t chunks = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, ((uint64_t)nq * S) / 8192));
        HostPool::Get().Run(chunks, [&](uint32_t c) {
            for (uint32_t q = (uint32_t)((uint64_t)nq * c / chunks); q < (uint32_t)((uint64_t)nq * (c + 1) / chunks); ++q) host_sort(q);
        });
    }
    return GPANN_OK;
}

// KMeansTreeRouter::FrequencyQuery (kmeans_tree_router.cpp:132-172): the same walk; additionally the num_voting smallest
// (distance, shard) pairs over ALL evaluated centroids (TopN, then FrequencyQueryData::Init sorts them ascending).
// The walk kernel records every evaluated pair; the selection of the num_voting smallest is done here on the host
// (order-independent: the result is the set of lexicographically smallest pairs).
static gpann_status FrequencyQueryAttempt(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t num_voting, float* out_min_dist,
                                          float* out_nn_dist, uint32_t* out_nn_shard, uint32_t* out_nn_count, int flags, void* stream, float* elapsed_ms,
                                          uint64_t rec_cap, bool rec_small, bool* truncated);

gpann_status gpann_router_frequency_query_batch(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t num_voting,
                                                float* out_min_dist, float* out_nn_dist, uint32_t* out_nn_shard, uint32_t* out_nn_count,
                                                int flags, void* stream, float* elapsed_ms) {
    GPANN_REQUIRE(r != nullptr, "router handle is null");
    std::lock_guard<std::recursive_mutex> turn(r->call_mutex);
    GPANN_REQUIRE(nq == 0 || (queries && out_min_dist && out_nn_dist && out_nn_shard && out_nn_count), "null argument");
    GPANN_TRY(EnsureDevice(r->device));
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (nq == 0) return GPANN_OK;
    // a walk evaluates at most budget + (largest node - 1) centroids after the budget check, and never more than the tree holds
    uint64_t rec_cap = std::min<uint64_t>(r->n_centroids, (uint64_t)std::max(budget, 0) + r->max_node);
    rec_cap = std::max<uint64_t>(rec_cap, 1);
    // The bounded replay records only the pairs that can be among the num_voting smallest (num_voting + a few dozen): room for 4 x num_voting
    // (>= num_voting + 4096) per query lets the whole batch run in one block instead of blocks sized for every evaluated pair (50 000 per query
    // on the NS trees: 8 blocks of 1310 queries).  A query whose records do not fit, or a tree / budget the bounded replay does not take,
    // repeats the call the old way.
    uint64_t small_cap = std::max<uint64_t>(4ull * num_voting, (uint64_t)num_voting + 4096);
    const char* small_env = r->opt.Get("GPANN_ROUTER_FREQ_SMALL");   // "0": never; n > 1 (tests): n records per query (at least num_voting)
    if (small_env && std::atoll(small_env) > 1) small_cap = std::max<uint64_t>((uint64_t)std::atoll(small_env), num_voting);
    if (num_voting >= 1 && num_voting <= 2048 && small_cap < rec_cap && !(small_env && small_env[0] == '0')) {
        bool truncated = false;
        const gpann_status s1 = FrequencyQueryAttempt(r, queries, nq, budget, num_voting, out_min_dist, out_nn_dist, out_nn_shard, out_nn_count, flags, stream, elapsed_ms,
                                                      small_cap, true, &truncated);
        if (s1 != GPANN_OK || !truncated) return s1;
    }
    bool unused = false;
    return FrequencyQueryAttempt(r, queries, nq, budget, num_voting, out_min_dist, out_nn_dist, out_nn_shard, out_nn_count, flags, stream, elapsed_ms, rec_cap, false, &unused);
}

static gpann_status FrequencyQueryAttempt(gpann_router* r, const float* queries, uint32_t nq, int32_t budget, uint32_t num_voting, float* out_min_dist,
                                          float* out_nn_dist, uint32_t* out_nn_shard, uint32_t* out_nn_count, int flags, void* stream, float* elapsed_ms,
                                          uint64_t rec_cap, bool rec_small, bool* truncated) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t S = r->S;
    const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nq, (512ull << 20) / (rec_cap * 8)));
    DeviceBuffer rec;
    GPANN_TRY(rec.Reserve((size_t)batch * rec_cap * 8 + (size_t)batch * 4));
    float* d_rec_dist = rec.as<float>();
    int32_t* d_rec_shard = reinterpret_cast<int32_t*>(d_rec_dist + (size_t)batch * rec_cap);
    uint32_t* d_rec_count = reinterpret_cast<uint32_t*>(d_rec_shard + (size_t)batch * rec_cap);
    const bool device_select = num_voting >= 1 && num_voting <= 2048;
    DeviceBuffer sel;
    float* d_sel_dist = nullptr;
    uint32_t *d_sel_shard = nullptr, *d_sel_count = nullptr;
    if (device_select) {
        GPANN_TRY(sel.Reserve((size_t)batch * num_voting * 8 + (size_t)batch * 4));
        d_sel_dist = sel.as<float>();
        d_sel_shard = reinterpret_cast<uint32_t*>(d_sel_dist + (size_t)batch * num_voting);
        d_sel_count = d_sel_shard + (size_t)batch * num_voting;
    }
    std::vector<float> h_dist(device_select ? 0 : (size_t)batch * rec_cap);
    std::vector<int32_t> h_shard(device_select ? 0 : (size_t)batch * rec_cap);
    std::vector<uint32_t> h_count(batch);
    const size_t qstride = (flags & GPANN_QUERIES_ON_DEVICE) ? r->d : r->d;
    float total_ms = 0.f;
    gpann_status status = GPANN_OK;
    for (uint32_t q0 = 0; q0 < nq && status == GPANN_OK; q0 += batch) {
        const uint32_t m = std::min(batch, nq - q0);
        status = LaunchWalk(r, queries + (size_t)q0 * qstride, m, budget, flags, st, d_rec_dist, d_rec_shard, d_rec_count, rec_cap, device_select ? num_voting : 0u,
                            rec_small, truncated);
        if (status != GPANN_OK) break;
        if (rec_small && *truncated) break;   // (the caller repeats the call with room for every evaluated pair)
        hipError_t e = hipMemcpyAsync(out_min_dist + (size_t)q0 * S, r->ws_out.ptr, (size_t)m * S * 4, hipMemcpyDeviceToHost, st);
        if (device_select) {
            // the votes are selected on the device; only [m][num_voting] pairs travel
            hipLaunchKernelGGL(VoteSelectKernel, dim3(m), dim3(256), 0, st, d_rec_dist, d_rec_shard, d_rec_count, rec_cap, num_voting, d_sel_dist, d_sel_shard,
                               d_sel_count);
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(out_nn_dist + (size_t)q0 * num_voting, d_sel_dist, (size_t)m * num_voting * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(out_nn_shard + (size_t)q0 * num_voting, d_sel_shard, (size_t)m * num_voting * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(out_nn_count + q0, d_sel_count, (size_t)m * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                SetLastError("frequency query transfer failed: %s", hipGetErrorString(e));
                status = GPANN_ERR_HIP;
                break;
            }
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r->ev[0], r->ev[1]);
            total_ms += ms;
            continue;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_count.data(), d_rec_count, (size_t)m * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_dist.data(), d_rec_dist, (size_t)m * rec_cap * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_shard.data(), d_rec_shard, (size_t)m * rec_cap * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            SetLastError("frequency query transfer failed: %s", hipGetErrorString(e));
            status = GPANN_ERR_HIP;
            break;
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r->ev[0], r->ev[1]);
        total_ms += ms;
        std::vector<std::pair<float, uint32_t>> pairs;
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t cnt = h_count[i];
            pairs.resize(cnt);
            for (uint32_t x = 0; x < cnt; ++x) pairs[x] = {h_dist[(size_t)i * rec_cap + x], (uint32_t)h_shard[(size_t)i * rec_cap + x]};
            const uint32_t keep = std::min<uint32_t>(num_voting, cnt);
            std::partial_sort(pairs.begin(), pairs.begin() + keep, pairs.end());   // = TopN(num_voting).Take() then Init()
            for (uint32_t x = 0; x < keep; ++x) {
                out_nn_dist[(size_t)(q0 + i) * num_voting + x] = pairs[x].first;
                out_nn_shard[(size_t)(q0 + i) * num_voting + x] = pairs[x].second;
            }
            out_nn_count[q0 + i] = keep;
        }
    }
    rec.Release();
    sel.Release();
    if (elapsed_ms) *elapsed_ms = total_ms;
    return status;
}

// DedupNeighbors (small_scale_queries.cpp:15-23) over per-slot result lists; host-side, libstdc++ sorts as the reference.
gpann_status gpann_merge_dedup(const float* in_dist, const uint32_t* in_ids, uint32_t nq, uint32_t nprobe, uint32_t nprobe_used, uint32_t k,
                               float* out_dist, uint32_t* out_ids, uint32_t* out_count) {
    GPANN_REQUIRE(nq == 0 || (in_dist && in_ids && out_dist && out_ids), "null argument");
    GPANN_REQUIRE(nprobe_used <= nprobe, "nprobe_used > nprobe");
    // queries are independent: a few host threads for large batches (the per-query work is the reference's four calls)
    auto merge_range = [&](uint32_t q0, uint32_t q1) {
        std::vector<std::pair<float, uint32_t>> n;
        for (uint32_t q = q0; q < q1; ++q) {
            n.clear();
            for (uint32_t j = 0; j < nprobe_used; ++j)
                for (uint32_t x = 0; x < k; ++x) {
                    const size_t idx = ((size_t)q * nprobe + j) * k + x;
                    if (in_ids[idx] != GPANN_NO_ID) n.emplace_back(in_dist[idx], in_ids[idx]);
                }
            // (one probed shard, no equal distances: the list is already what the four calls below would make of it)
            bool settled = true;
            for (size_t i = 1; i < n.size() && settled; ++i) settled = n[i - 1].first < n[i].first;
            if (!settled) {
                std::sort(n.begin(), n.end(), [](const auto& l, const auto& r) { return l.second < r.second; });
                n.erase(std::unique(n.begin(), n.end()), n.end());
                std::sort(n.begin(), n.end());
            }
            n.resize(std::min<size_t>(n.size(), k));
            for (uint32_t x = 0; x < k; ++x) {
                out_dist[(size_t)q * k + x] = x < n.size() ? n[x].first : std::numeric_limits<float>::infinity();
                out_ids[(size_t)q * k + x] = x < n.size() ? n[x].second : GPANN_NO_ID;
            }
            if (out_count) out_count[q] = (uint32_t)n.size();
        }
    };
    const uint32_t chunks = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, ((uint64_t)nq * nprobe_used * k) / 4096));
    HostPool::Get().Run(chunks, [&](uint32_t c) { merge_range((uint32_t)((uint64_t)nq * c / chunks), (uint32_t)((uint64_t)nq * (c + 1) / chunks)); });
    return GPANN_OK;
}

}  // extern "C"
