// kmeans.hip — KMeansTreeRouter training on the GPU (gfx950): level-synchronous batched Lloyd.
//
// Replaces  KMeansTreeRouter::Train / TrainRecursive / ReorderCentroids (src/kmeans_tree_router.cpp:9-91) and the k-means
// they call: RandomSample (src/kmeans.cpp:207-226), KMeans (:228-244), NearestCenters / Top1Neighbor (:12-27),
// AggregateClusters[Parallel] (:131-204), NormalizeCentroids{L2,IP} (:72-86,104-115), RemoveEmptyClusters (:29-50).
//
// The reference recurses node by node; every node is an independent k-means PROBLEM (a set of rows of one shard, an
// initial sample, 20 Lloyd rounds).  Here all problems of one tree level — over all shards — run as ONE batch of kernel
// launches (7 per round), so that thousands of small nodes cost the same number of launches as one large one:
//   AssignKernel     64 rows x 64 centroids per workgroup, the scan kernel's 4x4 register tile with the reference's exact
//                    op sequence; argmin with strict '<' in centroid order (= Top1Neighbor: first minimum wins)
//   CountKernel      per 2048-row chunk: members per cluster (LDS integer atomics)
//   ChunkScanKernel  per (problem, cluster): exclusive prefix over the problem's chunks, cluster size
//   OffsetsKernel    per problem: exclusive prefix of the cluster sizes = start of every cluster's member slab
//   ScatterKernel    per chunk: STABLE counting sort of the rows by cluster — members of a cluster stay in ascending position
//   SumKernel        per cluster: the reference's sequential float sum over its members IN POINT ORDER (batches of rows staged
//                    HBM -> registers -> LDS by the whole workgroup, the next batch in flight while one column per thread does the
//                    adds of the current one in order); beyond the
//                    reference's block size (kmeans.cpp:153) block partials are added in block order — one of the orders its
//                    atomics can produce
//   NormalizeKernel  per problem: divide / rescale, drop empty clusters (compacting the centroid slab)
// Because sums are taken in the reference's order the trained tree is bit-identical to the host trainer's
// (gp-ann_amd/host/router_host.cpp) and hence to the reference's for nodes up to 5M points.
//
// The member slabs of the last round ARE the buckets of ConvertPartitionToClusters (ascending positions), holding row ids:
// the slab array of one level is the row array of the next — no gather, no host round trip of row lists.  The host only
// sees, per level, the surviving centroids and cluster sizes, applies the reference's budget rules / std::partition /
// std::sample (libstdc++ semantics matter, so these calls stay std::) and emits the next level's problem list.
#include "shards.h"
#include "kmeans.h"

#include <algorithm>
#include <atomic>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

namespace gpann {

constexpr int kKT = 64;                 // centroids per tile
constexpr int kKP = 64;                 // rows per tile
constexpr int kKDC = 32;                // dims per staged chunk
constexpr int kKStride = kKDC + 4;
constexpr int kKThreads = 256;
constexpr uint32_t kChunkRows = 2048;   // rows per counting chunk (one wavefront, 32 steps)
constexpr uint32_t kSumFloats = 8192;   // floats of rows staged per SumKernel batch (32 KB of LDS; one register stage of 8 float4 per thread)
constexpr uint32_t kSumMaxRows = 256;
constexpr uint32_t kMaxTrainDim = 1023; // SumKernel: 4 columns per thread, one more column for the IP norm sums

struct LloydProblem {
    uint64_t row_begin;    // first entry of this problem in rows[] / owner[] / member arrays
    uint32_t n;            // rows
    uint32_t cent_begin;   // first slot (centroid row) of this problem
    uint32_t k0;           // slots
    uint32_t chunk_begin;  // first counting chunk
    uint32_t n_chunks;
    uint32_t pad;
};

struct TileRef {
    uint32_t prob, first;
};

struct LloydArgs {
    const float* X;              // [store rows][ld]
    const float* sqrt_norm;      // [store rows] (IP only)
    const LloydProblem* probs;
    const uint32_t* rows;        // [level rows] store row of every problem position
    float* cent;                 // [slots][ld]
    float* csum;                 // [slots][ld]
    float* cnorm;                // [slots]
    uint32_t* cnum;              // [problems] current number of centroids
    uint32_t* csize;             // [slots] members of the cluster in the current round
    uint32_t* coff;              // [slots] start of the member slab inside the problem
    const uint32_t* slot_prob;   // [slots]
    int32_t* owner;              // [level rows]
    uint32_t* members_row;       // [level rows]
    uint32_t* members_pos;       // [level rows]
    const TileRef* tiles;
    const TileRef* chunks;
    uint32_t* chunk_cnt;         // [chunks][kmax]
    uint32_t* chunk_base;        // [chunks][kmax]
    uint32_t n_probs, n_slots, kmax, d, ld;
};

// ---------------------------------------------------------------------------------------------------- AssignKernel
template <int METRIC>
__global__ void __launch_bounds__(kKThreads, 3) AssignKernel(LloydArgs a) {
    __shared__ float sC[kKT * kKStride];
    __shared__ float sP[kKP * kKStride];
    __shared__ float sD[kKT * kKP];

    const uint32_t prob = a.tiles[blockIdx.x].prob;
    const uint32_t first = a.tiles[blockIdx.x].first;
    const uint64_t row_begin = a.probs[prob].row_begin;
    const uint32_t n = a.probs[prob].n;
    const uint32_t cent_begin = a.probs[prob].cent_begin;
    const uint32_t c = a.cnum[prob];
    const uint32_t npts = min((uint32_t)kKP, n - first);

    const int tid = threadIdx.x;
    const int ty = tid >> 4;   // centroid group: centroids ty + 16 i
    const int tx = tid & 15;   // row group:      rows tx + 16 j
    const uint32_t d = a.d, ld = a.ld;
    const uint32_t full_groups = d >> 2;
    const int tail = (int)(d & 3u);
    const uint32_t nchunks = (ld + kKDC - 1) / kKDC;

    const int st_row0 = tid >> 3, st_row1 = (tid + 256) >> 3;
    const int st_c4 = tid & 7;
    const float* prow0 = a.X + (size_t)a.rows[row_begin + first + min((uint32_t)st_row0, npts - 1)] * ld;
    const float* prow1 = a.X + (size_t)a.rows[row_begin + first + min((uint32_t)st_row1, npts - 1)] * ld;

    int best = -1;
    float best_dist = 3.402823466e+38f;   // std::numeric_limits<float>::max(), kmeans.cpp:14
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (uint32_t ct = 0; ct < c; ct += kKT) {
        const float* crow0 = a.cent + (size_t)(cent_begin + min(ct + (uint32_t)st_row0, c - 1)) * ld;
        const float* crow1 = a.cent + (size_t)(cent_begin + min(ct + (uint32_t)st_row1, c - 1)) * ld;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

        float4 gc0, gc1, gp0, gp1;
        {
            const uint32_t col = 4u * st_c4;
            const bool in = col < ld;
            gc0 = in ? *reinterpret_cast<const float4*>(crow0 + col) : zero4;
            gc1 = in ? *reinterpret_cast<const float4*>(crow1 + col) : zero4;
            gp0 = in ? *reinterpret_cast<const float4*>(prow0 + col) : zero4;
            gp1 = in ? *reinterpret_cast<const float4*>(prow1 + col) : zero4;
        }
        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            __syncthreads();
            *reinterpret_cast<float4*>(sC + st_row0 * kKStride + 4 * st_c4) = gc0;
            *reinterpret_cast<float4*>(sC + st_row1 * kKStride + 4 * st_c4) = gc1;
            *reinterpret_cast<float4*>(sP + st_row0 * kKStride + 4 * st_c4) = gp0;
            *reinterpret_cast<float4*>(sP + st_row1 * kKStride + 4 * st_c4) = gp1;
            __syncthreads();
            if (ch + 1 < nchunks) {
                const uint32_t col = (ch + 1) * kKDC + 4u * st_c4;
                const bool in = col < ld;
                gc0 = in ? *reinterpret_cast<const float4*>(crow0 + col) : zero4;
                gc1 = in ? *reinterpret_cast<const float4*>(crow1 + col) : zero4;
                gp0 = in ? *reinterpret_cast<const float4*>(prow0 + col) : zero4;
                gp1 = in ? *reinterpret_cast<const float4*>(prow1 + col) : zero4;
            }
            const uint32_t g0 = ch * (kKDC / 4);
            const float* c_base = sC + ty * kKStride;
            const float* p_base = sP + tx * kKStride;
            if (g0 + kKDC / 4 <= full_groups) {
#pragma unroll 2
                for (int g = 0; g < kKDC / 4; ++g) {
                    float4 cv[4], pv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[i] = *reinterpret_cast<const float4*>(c_base + 16 * i * kKStride + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pv[j] = *reinterpret_cast<const float4*>(p_base + 16 * j * kKStride + 4 * g);
                    AccumulateGroup<METRIC>(acc, cv, pv);
                }
            } else {
                for (int g = 0; g < kKDC / 4; ++g) {
                    const uint32_t G = g0 + g;
                    if (G > full_groups || (G == full_groups && tail == 0)) break;
                    float4 cv[4], pv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) cv[i] = *reinterpret_cast<const float4*>(c_base + 16 * i * kKStride + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pv[j] = *reinterpret_cast<const float4*>(p_base + 16 * j * kKStride + 4 * g);
                    if (G < full_groups) AccumulateGroup<METRIC>(acc, cv, pv);
                    else AccumulateTail<METRIC>(acc, cv, pv, tail);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[i][j];
                if (METRIC == GPANN_METRIC_MIPS) v = 1.0f - v;
                sD[(ty + 16 * i) * kKP + tx + 16 * j] = v;
            }
        __syncthreads();
        if (tid < kKP) {   // Top1Neighbor: centroids in ascending order, strict '<'
            const uint32_t lim = min((uint32_t)kKT, c - ct);
            for (uint32_t jj = 0; jj < lim; ++jj) {
                const float v = sD[jj * kKP + tid];
                if (v < best_dist) {
                    best_dist = v;
                    best = (int)(ct + jj);
                }
            }
        }
    }
    if (tid < (int)npts) a.owner[row_begin + first + tid] = best;
}

// ---------------------------------------------------------------------------------------------------- counting sort
__global__ void __launch_bounds__(64) CountKernel(LloydArgs a) {
    extern __shared__ uint32_t s_cnt[];
    const uint32_t prob = a.chunks[blockIdx.x].prob;
    const uint32_t first = a.chunks[blockIdx.x].first;
    const uint64_t row_begin = a.probs[prob].row_begin;
    const uint32_t n = a.probs[prob].n;
    const int lane = threadIdx.x;
    for (uint32_t j = lane; j < a.kmax; j += 64) s_cnt[j] = 0;
    __syncthreads();
    const uint32_t end = min(n, first + kChunkRows);
    for (uint32_t pos = first + lane; pos < end; pos += 64) {
        const int o = a.owner[row_begin + pos];
        if (o >= 0) atomicAdd(&s_cnt[o], 1u);
    }
    __syncthreads();
    for (uint32_t j = lane; j < a.kmax; j += 64) a.chunk_cnt[(size_t)blockIdx.x * a.kmax + j] = s_cnt[j];
}

// one wavefront per slot (problem, cluster)
__global__ void __launch_bounds__(256) ChunkScanKernel(LloydArgs a) {
    const uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= a.n_slots) return;
    const int lane = threadIdx.x & 63;
    const uint32_t prob = a.slot_prob[slot];
    const uint32_t j = slot - a.probs[prob].cent_begin;
    if (j >= a.cnum[prob]) {
        if (lane == 0) a.csize[slot] = 0;
        return;
    }
    const uint32_t cb = a.probs[prob].chunk_begin, nc = a.probs[prob].n_chunks;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nc; base += 64) {
        const uint32_t idx = base + lane;
        const uint32_t v = idx < nc ? a.chunk_cnt[(size_t)(cb + idx) * a.kmax + j] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (idx < nc) a.chunk_base[(size_t)(cb + idx) * a.kmax + j] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
    if (lane == 0) a.csize[slot] = carry;
}

// one wavefront per problem
__global__ void __launch_bounds__(256) OffsetsKernel(LloydArgs a) {
    const uint32_t prob = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (prob >= a.n_probs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t cb = a.probs[prob].cent_begin, k0 = a.probs[prob].k0;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < k0; base += 64) {
        const uint32_t j = base + lane;
        const uint32_t v = j < k0 ? a.csize[cb + j] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (j < k0) a.coff[cb + j] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}

__global__ void __launch_bounds__(64) ScatterKernel(LloydArgs a) {
    extern __shared__ uint32_t s_run[];
    const uint32_t prob = a.chunks[blockIdx.x].prob;
    const uint32_t first = a.chunks[blockIdx.x].first;
    const uint64_t row_begin = a.probs[prob].row_begin;
    const uint32_t n = a.probs[prob].n;
    const uint32_t cb = a.probs[prob].cent_begin;
    const uint32_t c = a.cnum[prob];
    const int lane = threadIdx.x;
    for (uint32_t j = lane; j < c; j += 64) s_run[j] = a.coff[cb + j] + a.chunk_base[(size_t)blockIdx.x * a.kmax + j];
    __syncthreads();
    const uint32_t end = min(n, first + kChunkRows);
    for (uint32_t base = first; base < end; base += 64) {
        const uint32_t pos = base + lane;
        const int o = pos < end ? a.owner[row_begin + pos] : -1;
        uint32_t dest = 0;
        unsigned long long todo = __ballot(o >= 0);
        while (todo) {   // peel one cluster per iteration: its lanes get consecutive slots in lane (= position) order
            const int leader = __builtin_ctzll(todo);
            const int lo = __shfl(o, leader);
            const unsigned long long mask = __ballot(o == lo);
            if (o == lo) dest = s_run[lo] + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            // (LDS operations of one wavefront execute in program order: the reads above precede the leader's update)
            if (lane == leader) s_run[lo] += (uint32_t)__builtin_popcountll(mask);
            todo &= ~mask;
        }
        if (o >= 0) {
            a.members_row[row_begin + dest] = a.rows[row_begin + pos];
            a.members_pos[row_begin + dest] = pos;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- SumKernel
// One workgroup per slot.  Column `col` of the cluster's sum is accumulated by one thread, strictly in member order.
// Batches of member rows travel HBM -> registers -> LDS: the loads of batch b+1 are in flight while batch b is summed.
constexpr int kSumRegs = 8;   // float4 per thread and batch: 256 threads x 8 x 4 floats = kSumFloats

template <int METRIC>
__global__ void __launch_bounds__(kKThreads) SumKernel(LloydArgs a) {
    extern __shared__ float4 s_sum4[];
    float* sX = reinterpret_cast<float*>(s_sum4);                     // [batch rows][ld]
    uint32_t* sPos = reinterpret_cast<uint32_t*>(sX + kSumFloats);    // [kSumMaxRows]
    float* sMul = reinterpret_cast<float*>(sPos + kSumMaxRows);       // [kSumMaxRows]  1/sqrt_norm (IP)
    float* sSq = sMul + kSumMaxRows;                                  // [kSumMaxRows]  sqrt_norm^2 (IP)

    const uint32_t slot = blockIdx.x;
    const uint32_t prob = a.slot_prob[slot];
    const uint32_t cent_begin = a.probs[prob].cent_begin;
    const uint32_t c = a.cnum[prob];
    if (slot - cent_begin >= c) return;
    const uint64_t base = a.probs[prob].row_begin + a.coff[slot];
    const uint32_t size = a.csize[slot];
    const uint32_t n = a.probs[prob].n;
    const uint32_t d = a.d, ld = a.ld, ld4 = a.ld >> 2;
    const int tid = threadIdx.x;
    const uint32_t batch_rows = min(kSumMaxRows, kSumFloats / ld);

    // kmeans.cpp:153: block_size = max(5000000, centroids.coordinates.size() * 200)
    const uint64_t bs = max((uint64_t)5000000, (uint64_t)c * d * 200ull);
    const bool multi = (uint64_t)n > bs;
    const uint32_t n_blocks = multi ? (uint32_t)(((uint64_t)n + bs - 1) / bs) : 1u;
    uint64_t boundary = bs;
    uint32_t flushed = 0;

    const uint32_t ncols = METRIC == GPANN_METRIC_MIPS ? d + 1 : d;   // column d = norm_sums (kmeans.cpp:95)
    float part[4] = {0.f, 0.f, 0.f, 0.f}, total[4] = {0.f, 0.f, 0.f, 0.f};

    // register stage of one batch: kSumRegs float4 of the rows, and (thread t < rows) the row's position / norm
    float4 stage[kSumRegs];
#pragma unroll
    for (int r = 0; r < kSumRegs; ++r) stage[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t st_pos = 0;
    float st_sn = 1.f;
#define GPANN_SUM_FETCH(B0)                                                                                        \
    do {                                                                                                           \
        const uint32_t nr_ = min(batch_rows, size - (B0));                                                          \
        _Pragma("unroll") for (int r = 0; r < kSumRegs; ++r) {                                                     \
            const uint32_t f = (uint32_t)tid + 256u * r;                                                           \
            const uint32_t row = f / ld4, c4 = f - row * ld4;                                                      \
            if (row < nr_) {                                                                                       \
                const uint32_t srow = a.members_row[base + (B0) + row];                                            \
                stage[r] = *reinterpret_cast<const float4*>(a.X + (size_t)srow * ld + 4u * c4);                    \
            }                                                                                                      \
        }                                                                                                          \
        if ((uint32_t)tid < nr_) {                                                                                 \
            if (multi) st_pos = a.members_pos[base + (B0) + tid];                                                  \
            if (METRIC == GPANN_METRIC_MIPS) st_sn = a.sqrt_norm[a.members_row[base + (B0) + tid]];                \
        }                                                                                                          \
    } while (0)
    if (size) GPANN_SUM_FETCH(0u);

    for (uint32_t b0 = 0; b0 < size; b0 += batch_rows) {
        const uint32_t nr = min(batch_rows, size - b0);
        __syncthreads();   // the previous batch has been summed
#pragma unroll
        for (int r = 0; r < kSumRegs; ++r) {
            const uint32_t f = (uint32_t)tid + 256u * r;
            const uint32_t row = f / ld4, c4 = f - row * ld4;
            if (row < nr) reinterpret_cast<float4*>(sX)[row * ld4 + c4] = stage[r];
        }
        if ((uint32_t)tid < nr) {
            if (multi) sPos[tid] = st_pos;
            if (METRIC == GPANN_METRIC_MIPS) {
                sMul[tid] = 1.0f / st_sn;     // kmeans.cpp:96
                sSq[tid] = st_sn * st_sn;     // kmeans.cpp:95
            }
        }
        __syncthreads();
        if (b0 + batch_rows < size) GPANN_SUM_FETCH(b0 + batch_rows);   // lands while this batch is summed

        if (ncols <= (uint32_t)kKThreads) {
            // the common shape: one column per thread.  Member positions ascend, so a batch crosses a slice boundary of the reference
            // (kmeans.cpp:186-196) at most a few times in the whole cluster: sum the run of rows up to the boundary, flush, go on.
            uint32_t r0 = 0;
            while (r0 < nr) {
                uint32_t r1 = nr;
                if (multi && (uint64_t)sPos[nr - 1] >= boundary) {   // first row of the batch at or past the boundary (uniform search)
                    uint32_t lo = r0, hi = nr - 1;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint64_t)sPos[mid] >= boundary) hi = mid;
                        else lo = mid + 1;
                    }
                    r1 = lo;
                }
                if ((uint32_t)tid < ncols) {
                    float p = part[0];
                    const float* col = sX + tid;
                    if (METRIC == GPANN_METRIC_MIPS) {
                        if ((uint32_t)tid < d) {
#pragma unroll 8
                            for (uint32_t r = r0; r < r1; ++r) p = p + col[r * ld] * sMul[r];
                        } else {
#pragma unroll 8
                            for (uint32_t r = r0; r < r1; ++r) p = p + sSq[r];
                        }
                    } else {
#pragma unroll 8
                        for (uint32_t r = r0; r < r1; ++r) p = p + col[r * ld];
                    }
                    part[0] = p;
                }
                r0 = r1;
                if (r0 < nr) {   // the row at r0 belongs to a later slice: the finished slice's partial joins the total
                    total[0] = total[0] + part[0];
                    part[0] = 0.f;
                    boundary += bs;
                    ++flushed;
                }
            }
            continue;
        }
        for (uint32_t r = 0; r < nr; ++r) {
            if (multi) {
                const uint32_t pos = sPos[r];
                while ((uint64_t)pos >= boundary) {   // the reference's slice is over: add its partial (kmeans.cpp:186-196)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        total[q] = total[q] + part[q];
                        part[q] = 0.f;
                    }
                    boundary += bs;
                    ++flushed;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t col = (uint32_t)tid + 256u * q;
                if (col < ncols) {
                    float v;
                    if (METRIC == GPANN_METRIC_MIPS) v = col < d ? (sX[r * ld + col] * sMul[r]) : sSq[r];
                    else v = sX[r * ld + col];
                    part[q] = part[q] + v;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t col = (uint32_t)tid + 256u * q;
        float result = part[q];
        if (multi) {
            float t = total[q], p = part[q];
            for (uint32_t f = flushed; f < n_blocks; ++f) {
                t = t + p;
                p = 0.f;
            }
            result = t;
        }
        if (col < d) a.csum[(size_t)slot * ld + col] = result;
        else if (METRIC == GPANN_METRIC_MIPS && col == d) a.cnorm[slot] = result;
    }
}

#undef GPANN_SUM_FETCH

// ---------------------------------------------------------------------------------------------------- NormalizeKernel
// One workgroup per problem: kmeans.cpp:72-86 / :104-115, then RemoveEmptyClusters (:29-50) as a compaction of the slab.
template <int METRIC>
__global__ void __launch_bounds__(kKThreads) NormalizeKernel(LloydArgs a) {
    extern __shared__ uint32_t s_new[];                      // [kmax] compacted index, or 0xFFFFFFFF
    float* s_scale = reinterpret_cast<float*>(s_new + a.kmax);   // [kmax]
    const uint32_t prob = blockIdx.x;
    const uint32_t cb = a.probs[prob].cent_begin;
    const uint32_t c = a.cnum[prob];
    const uint32_t d = a.d, ld = a.ld;
    const int tid = threadIdx.x;
    __syncthreads();   // every thread has read cnum before thread 0 overwrites it below
    if (METRIC == GPANN_METRIC_MIPS) {
        for (uint32_t j = tid; j < c; j += kKThreads) {
            const uint32_t sz = a.csize[cb + j];
            float scale = 0.f;
            if (sz != 0) {
                const float want = (a.cnorm[cb + j] / (float)(unsigned long long)sz);
                const float* C = a.csum + (size_t)(cb + j) * ld;
                float have = 0.f;
                for (uint32_t x = 0; x < d; ++x) have = have + (C[x] * C[x]);   // vec_norm, dist.cpp:54-58
                scale = sqrtf((want / have));
            }
            s_scale[j] = scale;
        }
    }
    if (tid == 0) {
        uint32_t kept = 0;
        for (uint32_t j = 0; j < c; ++j) s_new[j] = a.csize[cb + j] != 0 ? kept++ : 0xFFFFFFFFu;
        a.cnum[prob] = kept;
    }
    __syncthreads();
    for (uint32_t j = 0; j < c; ++j) {
        const uint32_t nj = s_new[j];
        if (nj == 0xFFFFFFFFu) continue;
        const float* src = a.csum + (size_t)(cb + j) * ld;
        float* dst = a.cent + (size_t)(cb + nj) * ld;
        for (uint32_t x = tid; x < ld; x += kKThreads) {
            float v = 0.f;
            if (x < d) {
                if (METRIC == GPANN_METRIC_MIPS) v = (src[x] * s_scale[j]);
                else v = (src[x] / (float)(unsigned long long)a.csize[cb + j]);
            }
            dst[x] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- small helpers
__global__ void IotaKernel(uint32_t* out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

// sqrt(vec_norm(row)) per store row (kmeans.cpp:236)
__global__ void SqrtNormKernel(const float* X, uint64_t n, uint32_t d, uint32_t ld, float* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = X + (size_t)i * ld;
    float r = 0.f;
    for (uint32_t x = 0; x < d; ++x) r = r + (p[x] * p[x]);
    out[i] = sqrtf(r);
}

// initial centroids: slot s <- row at position pick[s] of its problem (RandomSample, kmeans.cpp:220-225)
__global__ void GatherInitKernel(LloydArgs a, const uint32_t* pick) {
    const uint32_t slot = blockIdx.x;
    const uint32_t prob = a.slot_prob[slot];
    const float* src = a.X + (size_t)a.rows[a.probs[prob].row_begin + pick[slot]] * a.ld;
    float* dst = a.cent + (size_t)slot * a.ld;
    for (uint32_t x = threadIdx.x; x < a.ld; x += blockDim.x) dst[x] = src[x];
}

template <typename F>
static void ParallelForHost(size_t n, F&& body) {
    const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const unsigned nt = (unsigned)std::min<size_t>(hw, n);
    if (nt <= 1) {
        for (size_t i = 0; i < n; ++i) body(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&] {
            for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) body(i);
        });
    for (auto& th : pool) th.join();
}

struct HostProblem {
    int shard;
    uint64_t row_begin;
    uint32_t n;
    int64_t budget;
    int seed;
    uint32_t node;   // index of the node this problem fills, in its shard's node list
};

struct TrainNode {
    std::vector<float> centroids;     // recursed buckets first (ReorderCentroids)
    std::vector<uint32_t> children;   // node indices in the shard's list
};

// ---------------------------------------------------------------------------------------------------- LloydSession
#define LLOYD_HIP(expr)                                                                               \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            SetLastError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return _e == hipErrorOutOfMemory ? GPANN_ERR_OOM : GPANN_ERR_HIP;                         \
        }                                                                                             \
    } while (0)

LloydSession::~LloydSession() { End(); }

void LloydSession::End() {
    for (DeviceBuffer* b : {&b_rows[0], &b_rows[1], &b_owner, &b_mpos, &b_sqrt, &b_probs, &b_tiles, &b_chunks, &b_ccnt, &b_cbase, &b_cent, &b_csum, &b_cnorm,
                            &b_cnum, &b_csize, &b_coff, &b_slotprob, &b_pick})
        b->Release();
    store_ = nullptr;
}

gpann_status LloydSession::Begin(gpann_shards* s, int metric) {
    End();
    GPANN_REQUIRE(s->d <= kMaxTrainDim, "k-means supports d <= 1023");
    GPANN_REQUIRE(s->rows < (1ull << 32), "k-means needs fewer than 2^32 rows");
    GPANN_TRY(EnsureDevice(s->device));
    store_ = s;
    metric_ = metric;
    cur_ = 0;
    const uint64_t R = std::max<uint64_t>(s->rows, 1);
    hipStream_t st = nullptr;
    GPANN_TRY(b_rows[0].Reserve(R * 4));
    GPANN_TRY(b_rows[1].Reserve(R * 4));
    GPANN_TRY(b_owner.Reserve(R * 4));
    GPANN_TRY(b_mpos.Reserve(R * 4));
    hipLaunchKernelGGL(IotaKernel, dim3((uint32_t)((R + 255) / 256)), dim3(256), 0, st, b_rows[0].as<uint32_t>(), s->rows);
    LLOYD_HIP(hipGetLastError());
    if (metric == GPANN_METRIC_MIPS) {
        GPANN_TRY(b_sqrt.Reserve(R * 4));
        hipLaunchKernelGGL(SqrtNormKernel, dim3((uint32_t)((R + 255) / 256)), dim3(256), 0, st, s->points, s->rows, s->d, s->ld, b_sqrt.as<float>());
        LLOYD_HIP(hipGetLastError());
    }
    return GPANN_OK;
}

const uint32_t* LloydSession::rows() const { return b_rows[cur_].as<uint32_t>(); }

gpann_status LloydSession::RunLevel(const std::vector<LloydTask>& level, const std::vector<uint32_t>& pick, int rounds, LloydResult& out) {
    gpann_shards* s = store_;
    const uint32_t d = s->d, ld = s->ld;
    const bool mips = metric_ == GPANN_METRIC_MIPS;
    hipStream_t st = nullptr;
    const uint32_t P = (uint32_t)level.size();
    std::vector<LloydProblem> probs(P);
    uint32_t n_slots = 0, kmax = 0, n_chunks = 0;
    uint64_t n_tiles = 0;
    for (uint32_t p = 0; p < P; ++p) {
        probs[p].row_begin = level[p].row_begin;
        probs[p].n = level[p].n;
        probs[p].cent_begin = n_slots;
        probs[p].k0 = level[p].k0;
        probs[p].chunk_begin = n_chunks;
        probs[p].n_chunks = (level[p].n + kChunkRows - 1) / kChunkRows;
        probs[p].pad = 0;
        n_slots += level[p].k0;
        kmax = std::max(kmax, level[p].k0);
        n_chunks += probs[p].n_chunks;
        n_tiles += (level[p].n + kKP - 1) / kKP;
    }
    GPANN_REQUIRE(pick.size() == n_slots, "one initial row per centroid slot");
    std::vector<uint32_t> slot_prob(n_slots);
    std::vector<TileRef> tiles, chunks;
    tiles.reserve(n_tiles);
    chunks.reserve(n_chunks);
    for (uint32_t p = 0; p < P; ++p) {
        for (uint32_t x = 0; x < probs[p].k0; ++x) slot_prob[probs[p].cent_begin + x] = p;
        for (uint32_t f = 0; f < probs[p].n; f += kKP) tiles.push_back({p, f});
        for (uint32_t f = 0; f < probs[p].n; f += kChunkRows) chunks.push_back({p, f});
    }
    GPANN_TRY(b_probs.Reserve(std::max<size_t>(P, 1) * sizeof(LloydProblem)));
    GPANN_TRY(b_tiles.Reserve(std::max<size_t>(tiles.size(), 1) * sizeof(TileRef)));
    GPANN_TRY(b_chunks.Reserve(std::max<size_t>(chunks.size(), 1) * sizeof(TileRef)));
    GPANN_TRY(b_ccnt.Reserve(std::max<size_t>((size_t)n_chunks * kmax, 1) * 4));
    GPANN_TRY(b_cbase.Reserve(std::max<size_t>((size_t)n_chunks * kmax, 1) * 4));
    GPANN_TRY(b_cent.Reserve(std::max<size_t>((size_t)n_slots * ld, 1) * 4));
    GPANN_TRY(b_csum.Reserve(std::max<size_t>((size_t)n_slots * ld, 1) * 4));
    GPANN_TRY(b_cnorm.Reserve(std::max<size_t>(n_slots, 1) * 4));
    GPANN_TRY(b_cnum.Reserve(std::max<size_t>(P, 1) * 4));
    GPANN_TRY(b_csize.Reserve(std::max<size_t>(n_slots, 1) * 4));
    GPANN_TRY(b_coff.Reserve(std::max<size_t>(n_slots, 1) * 4));
    GPANN_TRY(b_slotprob.Reserve(std::max<size_t>(n_slots, 1) * 4));
    GPANN_TRY(b_pick.Reserve(std::max<size_t>(n_slots, 1) * 4));
    out.cent_begin.resize(P);
    out.cnum.resize(P);
    for (uint32_t p = 0; p < P; ++p) {
        out.cent_begin[p] = probs[p].cent_begin;
        out.cnum[p] = probs[p].k0;
    }
    out.n_slots = n_slots;
    out.ld = ld;
    if (P == 0 || n_slots == 0) return GPANN_OK;
    LLOYD_HIP(hipMemcpyAsync(b_probs.ptr, probs.data(), P * sizeof(LloydProblem), hipMemcpyHostToDevice, st));
    LLOYD_HIP(hipMemcpyAsync(b_tiles.ptr, tiles.data(), tiles.size() * sizeof(TileRef), hipMemcpyHostToDevice, st));
    LLOYD_HIP(hipMemcpyAsync(b_chunks.ptr, chunks.data(), chunks.size() * sizeof(TileRef), hipMemcpyHostToDevice, st));
    LLOYD_HIP(hipMemcpyAsync(b_cnum.ptr, out.cnum.data(), P * 4, hipMemcpyHostToDevice, st));
    LLOYD_HIP(hipMemcpyAsync(b_slotprob.ptr, slot_prob.data(), n_slots * 4, hipMemcpyHostToDevice, st));
    LLOYD_HIP(hipMemcpyAsync(b_pick.ptr, pick.data(), n_slots * 4, hipMemcpyHostToDevice, st));

    LloydArgs a{};
    a.X = s->points;
    a.sqrt_norm = b_sqrt.as<float>();
    a.probs = b_probs.as<LloydProblem>();
    a.rows = b_rows[cur_].as<uint32_t>();
    a.cent = b_cent.as<float>();
    a.csum = b_csum.as<float>();
    a.cnorm = b_cnorm.as<float>();
    a.cnum = b_cnum.as<uint32_t>();
    a.csize = b_csize.as<uint32_t>();
    a.coff = b_coff.as<uint32_t>();
    a.slot_prob = b_slotprob.as<uint32_t>();
    a.owner = b_owner.as<int32_t>();
    a.members_row = b_rows[cur_ ^ 1].as<uint32_t>();
    a.members_pos = b_mpos.as<uint32_t>();
    a.tiles = b_tiles.as<TileRef>();
    a.chunks = b_chunks.as<TileRef>();
    a.chunk_cnt = b_ccnt.as<uint32_t>();
    a.chunk_base = b_cbase.as<uint32_t>();
    a.n_probs = P;
    a.n_slots = n_slots;
    a.kmax = kmax;
    a.d = d;
    a.ld = ld;

    hipLaunchKernelGGL(GatherInitKernel, dim3(n_slots), dim3(64), 0, st, a, b_pick.as<uint32_t>());
    LLOYD_HIP(hipGetLastError());
    const size_t sum_lds = (size_t)kSumFloats * 4 + (size_t)kSumMaxRows * 12;
    for (int round = 0; round < rounds; ++round) {
        if (mips) hipLaunchKernelGGL(AssignKernel<GPANN_METRIC_MIPS>, dim3((uint32_t)tiles.size()), dim3(kKThreads), 0, st, a);
        else hipLaunchKernelGGL(AssignKernel<GPANN_METRIC_L2>, dim3((uint32_t)tiles.size()), dim3(kKThreads), 0, st, a);
        hipLaunchKernelGGL(CountKernel, dim3(n_chunks), dim3(64), (size_t)kmax * 4, st, a);
        hipLaunchKernelGGL(ChunkScanKernel, dim3((n_slots + 3) / 4), dim3(256), 0, st, a);
        hipLaunchKernelGGL(OffsetsKernel, dim3((P + 3) / 4), dim3(256), 0, st, a);
        hipLaunchKernelGGL(ScatterKernel, dim3(n_chunks), dim3(64), (size_t)kmax * 4, st, a);
        if (mips) {
            hipLaunchKernelGGL(SumKernel<GPANN_METRIC_MIPS>, dim3(n_slots), dim3(kKThreads), sum_lds, st, a);
            hipLaunchKernelGGL(NormalizeKernel<GPANN_METRIC_MIPS>, dim3(P), dim3(kKThreads), (size_t)kmax * 8, st, a);
        } else {
            hipLaunchKernelGGL(SumKernel<GPANN_METRIC_L2>, dim3(n_slots), dim3(kKThreads), sum_lds, st, a);
            hipLaunchKernelGGL(NormalizeKernel<GPANN_METRIC_L2>, dim3(P), dim3(kKThreads), (size_t)kmax * 8, st, a);
        }
        LLOYD_HIP(hipGetLastError());
    }
    out.csize.resize(n_slots);
    out.coff.resize(n_slots);
    out.cent.resize((size_t)n_slots * ld);
    LLOYD_HIP(hipMemcpyAsync(out.cnum.data(), b_cnum.ptr, P * 4, hipMemcpyDeviceToHost, st));
    LLOYD_HIP(hipMemcpyAsync(out.csize.data(), b_csize.ptr, n_slots * 4, hipMemcpyDeviceToHost, st));
    LLOYD_HIP(hipMemcpyAsync(out.coff.data(), b_coff.ptr, n_slots * 4, hipMemcpyDeviceToHost, st));
    LLOYD_HIP(hipMemcpyAsync(out.cent.data(), b_cent.ptr, (size_t)n_slots * ld * 4, hipMemcpyDeviceToHost, st));
    LLOYD_HIP(hipStreamSynchronize(st));
    cur_ ^= 1;   // this level's member slabs are the next level's rows
    return GPANN_OK;
}

}  // namespace gpann

using namespace gpann;

extern "C" gpann_status gpann_router_train(gpann_shards* s, uint64_t n_points, uint32_t num_centroids, uint32_t min_cluster_size, int64_t budget,
                                           gpann_router** out) {
    GPANN_REQUIRE(s != nullptr && out != nullptr, "null handle");
    *out = nullptr;
    GPANN_REQUIRE(n_points > 0, "n_points must be positive");
    GPANN_REQUIRE(num_centroids >= 1 && num_centroids <= 4096, "num_centroids must be in [1, 4096]");
    const uint32_t S = s->S, d = s->d;
    LloydSession session;
    GPANN_TRY(session.Begin(s, s->metric));

    // kmeans_tree_router.cpp:9-33: one root problem per shard, budget share by shard size, seed 555 * b
    std::vector<std::vector<TrainNode>> nodes(S);
    std::vector<HostProblem> level;
    for (uint32_t b = 0; b < S; ++b) {
        nodes[b].emplace_back();
        // additive: an empty shard (the reference would read row 0 of nothing) gets a root without centroids — this is how a
        // rank that holds only some of the shards trains its part of the tree (pygpann/multigpu.py merges the parts)
        if (s->offsets[b + 1] == s->offsets[b]) continue;
        HostProblem p;
        p.shard = (int)b;
        p.row_begin = (uint64_t)s->offsets[b];
        p.n = (uint32_t)(s->offsets[b + 1] - s->offsets[b]);
        p.budget = (int64_t)(double((uint64_t)p.n * (uint64_t)budget) / double(n_points));
        p.seed = 555 * (int)b;
        p.node = 0;
        level.push_back(p);
    }

    while (!level.empty()) {
        const uint32_t P = (uint32_t)level.size();
        // ---- host: problem sizes and the initial samples
        std::vector<LloydTask> tasks(P);
        std::vector<uint32_t> slot_begin(P + 1, 0);
        for (uint32_t p = 0; p < P; ++p) {
            // kmeans_tree_router.cpp:36: std::max(2, std::min<int>(options.num_centroids, options.budget))
            const uint32_t k0 = (uint32_t)std::max(2, std::min<int>((int)num_centroids, (int)level[p].budget));
            tasks[p] = {level[p].row_begin, level[p].n, k0};
            slot_begin[p + 1] = slot_begin[p] + k0;
        }
        std::vector<uint32_t> pick(slot_begin[P], 0);
        ParallelForHost(P, [&](size_t p) {
            // kmeans.cpp:212-217; when the problem has fewer rows than samples the tail of `sample` stays 0
            const uint32_t n = tasks[p].n, k0 = tasks[p].k0;
            std::vector<int> all(n);
            std::iota(all.begin(), all.end(), 0);
            std::mt19937 prng(level[p].seed);
            std::vector<int> sample(k0);
            std::sample(all.begin(), all.end(), sample.begin(), k0, prng);
            for (uint32_t x = 0; x < k0; ++x) pick[slot_begin[p] + x] = (uint32_t)sample[x];
        });
        LloydResult res;
        GPANN_TRY(session.RunLevel(tasks, pick, 20, res));   // kmeans.cpp:239-243: 20 rounds

        // ---- host: what the reference does after KMeans returns (kmeans_tree_router.cpp:38-77)
        const uint32_t ld = res.ld;
        std::vector<HostProblem> next;
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t cb = res.cent_begin[p], k0 = tasks[p].k0;
            // surviving clusters in slot order; the i-th survivor's centroid is row i of the compacted slab
            std::vector<std::pair<size_t, size_t>> size_and_id;
            std::vector<uint32_t> slab;   // survivor -> start of its member slab inside the problem
            for (uint32_t j = 0; j < k0; ++j)
                if (res.csize[cb + j] != 0) {
                    size_and_id.emplace_back((size_t)res.csize[cb + j], size_and_id.size());
                    slab.push_back(res.coff[cb + j]);
                }
            const uint32_t c = res.cnum[p];
            if (size_and_id.size() != c) {
                SetLastError("router training: inconsistent cluster bookkeeping (problem %u: %zu clusters with members, %u centroids)", p,
                             size_and_id.size(), c);
                return GPANN_ERR_INTERNAL;
            }
            const size_t recursed =
                std::distance(size_and_id.begin(), std::partition(size_and_id.begin(), size_and_id.end(),
                                                                  [&](const auto& e) { return e.first > (size_t)min_cluster_size; }));
            std::vector<TrainNode>& shard_nodes = nodes[level[p].shard];
            const uint32_t me = level[p].node;
            shard_nodes[me].centroids.reserve((size_t)c * d);
            for (const auto& e : size_and_id) {
                const float* row = &res.cent[(size_t)(cb + e.second) * ld];
                shard_nodes[me].centroids.insert(shard_nodes[me].centroids.end(), row, row + d);
            }
            int64_t left = level[p].budget - (int64_t)c;
            if (left <= 0 || c == 1) continue;
            size_t total = 0;
            for (size_t i = 0; i < recursed; ++i) total += size_and_id[i].first;
            for (size_t i = 0; i < recursed; ++i) {
                HostProblem ch;
                ch.shard = level[p].shard;
                ch.row_begin = level[p].row_begin + slab[size_and_id[i].second];
                ch.n = (uint32_t)size_and_id[i].first;
                ch.budget = (int64_t)(double(size_and_id[i].first * (size_t)left) / double(total));
                ch.seed = level[p].seed + (int)i;
                ch.node = (uint32_t)shard_nodes.size();
                shard_nodes[me].children.push_back(ch.node);
                shard_nodes.emplace_back();
                next.push_back(ch);
            }
        }
        level.swap(next);
    }
    session.End();

    // flatten: per shard breadth first, the children of a node consecutive (ExtractPoints order, kmeans_tree_router.cpp:180-198)
    std::vector<int32_t> node_shard;
    std::vector<uint32_t> node_cent_first{0}, node_child_first, node_child_count, shard_root;
    std::vector<float> centroids;
    for (uint32_t b = 0; b < S; ++b) {
        const auto& sn = nodes[b];
        std::vector<uint32_t> order{0};
        const uint32_t first_id = (uint32_t)node_shard.size();
        shard_root.push_back(first_id);
        for (size_t head = 0; head < order.size(); ++head) {
            const TrainNode& u = sn[order[head]];
            node_shard.push_back((int32_t)b);
            node_child_count.push_back((uint32_t)u.children.size());
            node_child_first.push_back(u.children.empty() ? 0u : first_id + (uint32_t)order.size());
            for (uint32_t ch : u.children) order.push_back(ch);
            centroids.insert(centroids.end(), u.centroids.begin(), u.centroids.end());
            node_cent_first.push_back(node_cent_first.back() + (uint32_t)(u.centroids.size() / d));
        }
    }
    return gpann_router_create(d, s->metric, S, (uint32_t)node_shard.size(), node_shard.data(), node_cent_first.data(), node_child_first.data(),
                               node_child_count.data(), shard_root.data(), centroids.data(), s->device, out);
}

// RandomSample(points, k, seed) + KMeans(points, centroids) (kmeans.cpp:207-244) over ALL rows of the store as one problem — the
// call every k-means partitioner of the reference starts with (partition.cpp:23-26, partitioning.cpp:19-26, :94-98).
extern "C" gpann_status gpann_kmeans(gpann_shards* s, uint32_t k, int seed, int32_t* out_partition, float* out_centroids, uint32_t* out_k) {
    GPANN_REQUIRE(s != nullptr && out_k != nullptr, "null handle / out_k");
    GPANN_REQUIRE(k >= 1 && k <= 4096, "k must be in [1, 4096]");
    GPANN_REQUIRE(s->rows >= 1, "k-means over an empty point set");
    const uint32_t n = (uint32_t)s->rows, d = s->d;
    LloydSession session;
    GPANN_TRY(session.Begin(s, s->metric));
    std::vector<LloydTask> tasks{{0, n, k}};
    std::vector<uint32_t> pick(k, 0);
    {   // kmeans.cpp:212-217; with fewer rows than samples the tail of `sample` stays 0
        std::vector<int> all(n);
        std::iota(all.begin(), all.end(), 0);
        std::mt19937 prng(seed);
        std::vector<int> sample(k);
        std::sample(all.begin(), all.end(), sample.begin(), k, prng);
        for (uint32_t x = 0; x < k; ++x) pick[x] = (uint32_t)sample[x];
    }
    LloydResult res;
    GPANN_TRY(session.RunLevel(tasks, pick, 20, res));
    // survivors in slot order = the ids RemoveEmptyClusters leaves (kmeans.cpp:29-50); their member slabs are the clusters
    std::vector<uint32_t> rows;
    if (out_partition) {
        rows.resize(n);
        GPANN_TRY(EnsureDevice(s->device));
        hipError_t e = hipMemcpy(rows.data(), session.rows(), (size_t)n * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            SetLastError("hipMemcpy(rows) failed: %s", hipGetErrorString(e));
            return GPANN_ERR_HIP;
        }
    }
    uint32_t alive = 0;
    for (uint32_t j = 0; j < k; ++j) {
        if (res.csize[j] == 0) continue;
        if (out_partition)
            for (uint32_t x = 0; x < res.csize[j]; ++x) out_partition[rows[res.coff[j] + x]] = (int32_t)alive;
        if (out_centroids) std::memcpy(out_centroids + (size_t)alive * d, &res.cent[(size_t)alive * res.ld], d * sizeof(float));
        ++alive;
    }
    if (alive != res.cnum[0]) {
        SetLastError("k-means: inconsistent cluster bookkeeping (%u clusters with members, %u centroids)", alive, res.cnum[0]);
        return GPANN_ERR_INTERNAL;
    }
    *out_k = alive;
    session.End();
    return GPANN_OK;
}
