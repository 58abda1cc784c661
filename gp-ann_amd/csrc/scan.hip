// scan.hip — the per-shard exact distance scan with top-k selection (gfx950).
//
// Replaces the reference's brute-force loops  InvertedIndex::QueryBucket / Query (src/inverted_index.h:43-75) and
// ComputeGroundTruth (src/recall.h:25-39):  for one query, stream every row of a shard, evaluate distance(), keep the
// k lexicographically smallest (distance, row) pairs (TopN, src/topn.h:12-19).
//
// MI355X design:
//   * the caller's (query, probe-slot) pairs are grouped by shard into TASKS; a work UNIT is (<=64 tasks of one shard) x
//     (one contiguous row segment of that shard); one 256-thread workgroup per unit, thousands of units per launch.
//   * a workgroup walks its segment in tiles of 64 rows.  Point rows stream from HBM in 32-dim chunks (coalesced 16-byte
//     loads, 128 contiguous bytes per row), are staged in LDS together with the matching chunk of the 64 query rows, and
//     each thread accumulates a 4x4 (query x point) register tile with the reference's exact op sequence:
//     per 4 dims  acc += ((d0*d0 + d1*d1) + d2*d2) + d3*d3   (unfused: v_sub, v_mul, v_add only — no FMA).
//     Every row loaded from HBM is reused by 64 queries, every LDS operand by 4 pairs.
//   * the 64x64 distance tile goes through LDS to the selection stage: each wavefront owns 16 queries and keeps their
//     running top-k lists in registers, one list entry per lane (k <= 64).  A row enters the insert path only if it beats
//     the current k-th entry (ballot over the 64 rows of the tile); after warm-up that is rare.
//   * a second kernel merges the per-segment lists of a task, maps rows to global ids through `permutation`
//     (inverted_index.h:56-58) and writes the caller's [nq][nprobe][k] arrays.
#include "shards.h"
#include "scan_filter.h"
#include "scan_cells.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <vector>

#define CS(s) ((s)->cs[(s)->cur])   // the per-call host state of the call in progress (shards.h)

// GPANN_SCAN_DEBUG_HOST=1: where the HOST time of a scan call goes (stderr, calls that take more than a millisecond to enqueue)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    char buf[512];
    size_t len = 0;
    HostTrace() {
        const char* v = std::getenv("GPANN_SCAN_DEBUG_HOST");
        on = v && v[0] == '1';
        if (on) t0 = last = std::chrono::steady_clock::now();
    }
    void Mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        if (ms > 0.05 && len + 48 < sizeof buf) len += (size_t)std::snprintf(buf + len, sizeof buf - len, " %s %.2f", what, ms);
    }
    ~HostTrace();
};
static thread_local HostTrace* g_host_trace = nullptr;
#define HT_MARK(what)                                   \
    do {                                                \
        if (g_host_trace) g_host_trace->Mark(what);     \
    } while (0)
HostTrace::~HostTrace() {
        if (g_host_trace == this) g_host_trace = nullptr;
        if (!on) return;
        const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (total > 1.0) fprintf(stderr, "[gpann scan host] %.2f ms:%.*s\n", total, (int)len, buf);
}

namespace gpann {

constexpr int kQT = 64;          // queries (tasks) per unit
constexpr int kPT = 64;          // rows per tile
constexpr int kDC = 32;          // dims per staged chunk
constexpr int kLdsStride = kDC + 4;   // floats; +4 keeps the 16 row addresses of a ds_read_b128 lane group on distinct banks
constexpr int kThreads = 256;
// The 64 point rows of a tile lie in LDS as 32 PAIRS (rows r and r + 16, r in [0,16) or [32,48)): [pair][dim][member], so that one
// ds_read_b128 delivers two dimensions of both rows: the distance sequence runs on natural register pairs — (x_a - q.x, x_b - q.x) and so on
// (common.h AccumulatePairs; scalar fp32 instructions since round 6, the library is built without the packed forms).
// (With one row per float4 the pairs had to be formed across dimensions: 32 v_mov per 192 operations.)
constexpr int kPairStride = 2 * kDC + 4;   // floats per pair row; +4: the 16 pair addresses of a lane group on distinct banks
constexpr int kQueriesPerWave = kQT / 4;

struct ScanArgs {
    const float* points;        // [rows][ld]
    const float* queries;       // [nq][ld]  (repacked, zero padded)
    const ScanUnit* units;
    const uint32_t* task_query; // [tasks]
    const uint32_t* task_part;  // [tasks] index of the task's first partial list
    float* part_dist;           // [lists][k]
    uint32_t* part_row;         // [lists][k]
    uint32_t d, ld, k;
};

// insert candidate (cd, cr) — known to be < entry k-1 — into the sorted list held one entry per lane
__device__ __forceinline__ void ListInsert(float& ld_, uint32_t& lr_, float cd, uint32_t cr, uint32_t k, int lane) {
    bool less = (lane >= (int)k) || PairLess(cd, cr, ld_, lr_);
    unsigned long long m = __ballot(less);
    int pos = __builtin_ctzll(m);
    float up_d = __shfl_up(ld_, 1);
    uint32_t up_r = __shfl_up(lr_, 1);
    if (lane == pos) {
        ld_ = cd;
        lr_ = cr;
    } else if (lane > pos) {
        ld_ = up_d;
        lr_ = up_r;
    }
}

// TAG is a profiling label only: TAG 1 is the very same code under a second symbol, launched for auxiliary passes
// (ground truth, GPANN_AUX_PASS) so that rocprofv3's per-kernel statistics of the hot path are not mixed with them.
// (3 workgroups per CU: 164 VGPRs with the inner loop unrolled by 2.  A 4-workgroup budget — 128 VGPRs, no unrolling, 23 spills — measured
// 6 % slower.)
// (round 6, on scalar fp32 instructions — tools/exact_scan_timing.py, 8M x 128, 2048 queries: 3 workgroups / unroll 2 35.7 ms; 4 / 1 37.5; 3 / 4 35.9;
// 2 / 4 42.2; 2 / 8 35.8; 3 / 1 36.3.  A different kernel altogether — a wavefront owns 16 queries x 64 rows, the query values are SCALAR operands
// (s_load_dwordx4, two blocks of eight in flight), a lane reads only its own row from LDS, no distance tile through LDS: 192 of 240 instructions of its
// inner loop are arithmetic — ran the same store in 37.4 ms, same bits (profiles/r06/exact_kernel/).  Two unlike designs at 42 - 44 T lane-operations/s:
// the limit is not this kernel's LDS mix.)
template <int METRIC, int TAG>
__global__ void __launch_bounds__(kThreads, 3) ScanTileKernel(ScanArgs a) {
    extern __shared__ float4 smem4[];
    float* sQ = reinterpret_cast<float*>(smem4);          // [64][36]
    float* sP = sQ + kQT * kLdsStride;                    // [32 pairs][68]
    float* sD = sP + (kPT / 2) * kPairStride;             // [64][64] distances of the current tile

    // (fields are read one by one: a by-value struct copy ends up in scratch memory)
    struct {
        int64_t row_begin, row_end;
        uint32_t task_begin, task_count, seg;
    } u;
    u.row_begin = a.units[blockIdx.x].row_begin;
    u.row_end = a.units[blockIdx.x].row_end;
    u.task_begin = a.units[blockIdx.x].task_begin;
    u.task_count = a.units[blockIdx.x].task_count;
    u.seg = a.units[blockIdx.x].seg;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ty = tid >> 4;   // query group 0..15 : queries ty + 16 i
    const int tx = tid & 15;   // point group 0..15 : rows    tx + 16 j
    const uint32_t d = a.d, ld = a.ld, k = a.k;
    const uint32_t full_groups = d >> 2;        // groups of 4 complete dims
    const int tail = (int)(d & 3u);
    const uint32_t nchunks = (ld + kDC - 1) / kDC;

    // staging role: float4 #f of a [64 rows][8 float4] chunk, f = tid and tid + 256
    const int st_row0 = tid >> 3, st_row1 = (tid + 256) >> 3;
    const int st_c4 = tid & 7;
    const uint32_t tq0 = u.task_begin + min((uint32_t)st_row0, u.task_count - 1);
    const uint32_t tq1 = u.task_begin + min((uint32_t)st_row1, u.task_count - 1);
    const float* qrow0 = a.queries + (size_t)a.task_query[tq0] * ld;
    const float* qrow1 = a.queries + (size_t)a.task_query[tq1] * ld;

    // running top-k lists of this wave's 16 queries: lane i holds entry i
    float list_d[kQueriesPerWave];
    uint32_t list_r[kQueriesPerWave];
#pragma unroll
    for (int j = 0; j < kQueriesPerWave; ++j) {
        list_d[j] = INFINITY;
        list_r[j] = kNoId;
    }

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // point staging role: pair st_pair = rows st_prow0 and st_prow0 + 16 of the tile, float4 #st_c4 of both
    const int st_pair = tid >> 3;
    const int st_prow0 = (st_pair & 15) + 32 * (st_pair >> 4);
    for (int64_t row0 = u.row_begin; row0 < u.row_end; row0 += kPT) {
        f32x2_t acc[4][2];   // acc[i][pr] = queries ty + 16 i x rows (tx + 32 pr, tx + 32 pr + 16)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) acc[i][pr] = (f32x2_t){0.f, 0.f};

        const int64_t pr0 = min(row0 + st_prow0, u.row_end - 1);
        const int64_t pr1 = min(row0 + st_prow0 + 16, u.row_end - 1);
        const float* prow0 = a.points + (size_t)pr0 * ld;
        const float* prow1 = a.points + (size_t)pr1 * ld;

        // prefetch chunk 0 into registers
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
            __syncthreads();   // previous chunk's LDS reads (and the previous tile's selection reads of sD) are done
            *reinterpret_cast<float4*>(sQ + st_row0 * kLdsStride + 4 * st_c4) = gq0;
            *reinterpret_cast<float4*>(sQ + st_row1 * kLdsStride + 4 * st_c4) = gq1;
            // the two rows of the pair, interleaved dimension by dimension
            *reinterpret_cast<float4*>(sP + st_pair * kPairStride + 8 * st_c4) = make_float4(gp0.x, gp1.x, gp0.y, gp1.y);
            *reinterpret_cast<float4*>(sP + st_pair * kPairStride + 8 * st_c4 + 4) = make_float4(gp0.z, gp1.z, gp0.w, gp1.w);
            __syncthreads();
            if (ch + 1 < nchunks) {   // issue the next chunk's global loads; they complete under the arithmetic below
                const uint32_t c = (ch + 1) * kDC + 4u * st_c4;
                const bool in = c < ld;
                gq0 = in ? *reinterpret_cast<const float4*>(qrow0 + c) : zero4;
                gq1 = in ? *reinterpret_cast<const float4*>(qrow1 + c) : zero4;
                gp0 = in ? *reinterpret_cast<const float4*>(prow0 + c) : zero4;
                gp1 = in ? *reinterpret_cast<const float4*>(prow1 + c) : zero4;
            }
            const uint32_t g0 = ch * (kDC / 4);                        // first group index of this chunk
            const float* q_base = sQ + ty * kLdsStride;
            const float* p_base = sP + tx * kPairStride;               // pairs tx (rows tx, tx + 16) and tx + 16 (rows tx + 32, tx + 48)
            if (g0 + kDC / 4 <= full_groups) {
                // fast path: 8 complete groups
#pragma unroll 2
                for (int g = 0; g < kDC / 4; ++g) {
                    float4 qv[4];
                    f32x4_t p0[2], p1[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const float4*>(q_base + 16 * i * kLdsStride + 4 * g);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        p0[pr] = *reinterpret_cast<const f32x4_t*>(p_base + 16 * pr * kPairStride + 8 * g);
                        p1[pr] = *reinterpret_cast<const f32x4_t*>(p_base + 16 * pr * kPairStride + 8 * g + 4);
                    }
                    AccumulatePairs<METRIC>(acc, qv, p0, p1);
                }
            } else {
                for (int g = 0; g < kDC / 4; ++g) {
                    const uint32_t G = g0 + g;
                    if (G > full_groups || (G == full_groups && tail == 0)) break;
                    float4 qv[4];
                    f32x4_t p0[2], p1[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const float4*>(q_base + 16 * i * kLdsStride + 4 * g);
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        p0[pr] = *reinterpret_cast<const f32x4_t*>(p_base + 16 * pr * kPairStride + 8 * g);
                        p1[pr] = *reinterpret_cast<const f32x4_t*>(p_base + 16 * pr * kPairStride + 8 * g + 4);
                    }
                    if (G < full_groups) AccumulatePairs<METRIC>(acc, qv, p0, p1);
                    else AccumulatePairsTail<METRIC>(acc, qv, p0, p1, tail);
                }
            }
        }

        // publish the 64x64 distance tile: member m of pair pr is row tx + 32 pr + 16 m
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float v = acc[i][pr][m];
                    if (METRIC == GPANN_METRIC_MIPS) v = 1.0f - v;   // src/dist.cpp:50-52
                    sD[(ty + 16 * i) * kPT + tx + 32 * pr + 16 * m] = v;
                }
        __syncthreads();

        // selection: this wave's 16 queries, lane = row of the tile
        const uint32_t my_row = (uint32_t)(row0 + lane);
        const bool row_valid = row0 + lane < u.row_end;
#pragma unroll
        for (int j = 0; j < kQueriesPerWave; ++j) {
            const int ql = wave * kQueriesPerWave + j;
            const float v = sD[ql * kPT + lane];
            float td = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(list_d[j]), k - 1));
            uint32_t tr = (uint32_t)__builtin_amdgcn_readlane((int)list_r[j], k - 1);
            bool pass = row_valid && PairLess(v, my_row, td, tr);
            unsigned long long mask = __ballot(pass);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                const float cd = __shfl(v, l);
                const uint32_t cr = (uint32_t)(row0 + l);
                ListInsert(list_d[j], list_r[j], cd, cr, k, lane);
                td = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(list_d[j]), k - 1));
                tr = (uint32_t)__builtin_amdgcn_readlane((int)list_r[j], k - 1);
                pass = pass && lane > l && PairLess(v, my_row, td, tr);
                mask = __ballot(pass);
            }
        }
        // no barrier needed here: sD is next written after two more __syncthreads of the following tile
    }

    // write this unit's partial lists
#pragma unroll
    for (int j = 0; j < kQueriesPerWave; ++j) {
        const uint32_t ql = wave * kQueriesPerWave + j;
        if (ql < u.task_count && lane < (int)k) {
            const size_t list = (size_t)a.task_part[u.task_begin + ql] + u.seg;
            a.part_dist[list * k + lane] = list_d[j];
            a.part_row[list * k + lane] = list_r[j];
        }
    }
}

// ---- merge ---------------------------------------------------------------------------------------------------------
// One wavefront per output list.  MODE 0: merge the per-segment partial lists of a task -> out[task_out[t]].
// MODE 1: merge the first nprobe per-slot lists of a query (InvertedIndex::Query: one TopN over all probed shards).
struct MergeArgs {
    const float* in_dist;       // lists of k entries, ascending
    const uint32_t* in_row;
    const uint32_t* first_list; // [n_out] index of the first input list (MODE 0) ; unused in MODE 1
    const uint32_t* n_lists;    // [n_out] number of input lists (MODE 0)
    const uint32_t* out_index;  // [n_out] destination list index (MODE 0)
    const uint32_t* permutation;
    float* out_dist;            // may be null
    uint32_t* out_ids;          // may be null
    uint32_t* out_rows;         // may be null
    uint32_t n_out, k, lists_per_out;   // lists_per_out: MODE 1 (nprobe)
};

template <int MODE>
__global__ void __launch_bounds__(256) MergeListsKernel(MergeArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= a.n_out) return;
    const uint32_t k = a.k;
    const size_t first = MODE == 0 ? a.first_list[o] : (size_t)o * a.lists_per_out;
    const uint32_t n = MODE == 0 ? a.n_lists[o] : a.lists_per_out;
    float ld_ = INFINITY;
    uint32_t lr_ = kNoId;
    if (n > 0 && lane < (int)k) {
        ld_ = a.in_dist[first * k + lane];
        lr_ = a.in_row[first * k + lane];
    }
    for (uint32_t s = 1; s < n; ++s) {
        // candidates of list s in ascending order; stop at the first one that does not beat the current k-th entry
        float cd_l = INFINITY;
        uint32_t cr_l = kNoId;
        if (lane < (int)k) {
            cd_l = a.in_dist[(first + s) * k + lane];
            cr_l = a.in_row[(first + s) * k + lane];
        }
        for (uint32_t e = 0; e < k; ++e) {
            const float cd = __shfl(cd_l, e);
            const uint32_t cr = __shfl(cr_l, e);
            const float td = __shfl(ld_, k - 1);
            const uint32_t tr = __shfl(lr_, k - 1);
            if (!PairLess(cd, cr, td, tr)) break;
            ListInsert(ld_, lr_, cd, cr, k, lane);
        }
    }
    if (lane < (int)k) {
        const size_t dst = (size_t)(MODE == 0 && a.out_index ? a.out_index[o] : o) * k + lane;
        if (a.out_dist) a.out_dist[dst] = ld_;
        if (a.out_rows) a.out_rows[dst] = lr_;
        if (a.out_ids) a.out_ids[dst] = lr_ == kNoId ? kNoId : a.permutation[lr_];
    }
}

__global__ void FillKernel(float* dist, uint32_t* ids, uint32_t* rows, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (dist) dist[i] = INFINITY;
        if (ids) ids[i] = kNoId;
        if (rows) rows[i] = kNoId;
    }
}

// GPANN_SCAN_FOR_MERGE on complete per-slot lists [nq][nprobe][k]: tau_q = min over the slots of the slot's k-th distance (+inf where the
// list is short or the slot unused); entries beyond it become (+inf, NO_ID).  One thread per query.  `dist` must not be null.
__global__ void TruncateForMergeKernel(float* dist, uint32_t* ids, uint32_t* rows, uint32_t nq, uint32_t nprobe, uint32_t k) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const size_t base = (size_t)q * nprobe * k;
    float tau = INFINITY;
    for (uint32_t j = 0; j < nprobe; ++j) tau = fminf(tau, dist[base + (size_t)j * k + k - 1]);
    for (size_t i = base; i < base + (size_t)nprobe * k; ++i)
        if (!(dist[i] <= tau)) {
            dist[i] = INFINITY;
            if (ids) ids[i] = kNoId;
            if (rows) rows[i] = kNoId;
        }
}

// RunInShardSearches hit lists: from per-(q, shard) ascending lists keep entries with dist <= kth[q], emit ids descending.
__global__ void ShardHitsKernel(const float* dist, const uint32_t* ids, const float* kth, uint32_t nq, uint32_t S, uint32_t k,
                                uint32_t* out_ids, uint32_t* out_count) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;   // t = q * S + b
    if (t >= (size_t)nq * S) return;
    const uint32_t q = (uint32_t)(t / S), b = (uint32_t)(t % S);
    const float thr = kth[q];
    uint32_t c = 0;
    for (uint32_t i = 0; i < k; ++i) {
        if (ids[t * k + i] != kNoId && dist[t * k + i] <= thr) c++;
        else break;
    }
    uint32_t* o = out_ids + ((size_t)b * nq + q) * k;
    for (uint32_t i = 0; i < k; ++i) o[i] = i < c ? ids[t * k + (c - 1 - i)] : kNoId;
    out_count[(size_t)b * nq + q] = c;
}

// ---- host orchestration -------------------------------------------------------------------------------------------
// host -> device copy of a small per-call array through the handle's pinned staging area (asynchronous; pageable fallback if it is full)
static hipError_t UploadAsync(gpann_shards* s, void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    void* p = CS(s).stage.Take(bytes);
    if (!p) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    std::memcpy(p, src, bytes);
    return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st);
}

// The small host -> device arrays of a scan call travel TOGETHER: staged back to back in the handle's pinned area and copied with one
// hipMemcpyAsync into a device mirror of that area (Flush); Add returns the address an array has there.  (One copy per array was a dozen
// launches per call, each with its own few microseconds of stream time and host work between them.)  Without room in the stage an array goes to
// its own destination at once, as before.
// Copies by a KERNEL between device memory and PINNED host memory (through the host allocation's device address).  GPANN_SCAN_OVERLAP uses them
// instead of hipMemcpyAsync: a runtime copy enqueued on a stream that is waiting for another stream's event was seen to block the HOST until that
// event had fired (5 - 8 ms inside the upload of a call whose front stream waited for an earlier rest pass, and inside the copies of the lists to the
// host on the tail stream) — the hand-over between the streams is the whole point of that mode.  16-byte words; n16 words.
__global__ void CopyWords16Kernel(uint4* dst, const uint4* src, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void CopyWords4Kernel(uint32_t* dst, const uint32_t* src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the device address of a pinned host allocation (nullptr: not device-accessible — the caller falls back to the runtime's copy)
static void* DeviceAddressOfPinned(const void* host) {
    void* dp = nullptr;
    if (host == nullptr || hipHostGetDevicePointer(&dp, const_cast<void*>(host), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dp;
}
static hipError_t KernelCopy(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15u) == 0) {
        const size_t n16 = bytes / 16;
        hipLaunchKernelGGL(CopyWords16Kernel, dim3((uint32_t)std::min<size_t>(256, (n16 + 255) / 256)), dim3(256), 0, st, static_cast<uint4*>(dst), static_cast<const uint4*>(src), n16);
    } else {
        const size_t n = bytes / 4;   // (every array that travels this way is made of 4-byte words)
        hipLaunchKernelGGL(CopyWords4Kernel, dim3((uint32_t)std::min<size_t>(256, (n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t*>(dst), static_cast<const uint32_t*>(src), n);
    }
    return hipGetLastError();
}

struct UploadBatch {
    gpann_shards* s;
    hipStream_t st;
    bool by_kernel = false;   // GPANN_SCAN_OVERLAP: the staged arrays reach the device mirror through CopyWords16Kernel (above)
    size_t first = SIZE_MAX, last = 0;
    hipError_t err = hipSuccess;
    template <typename T>
    T* Add(T* own_dst, const void* src, size_t bytes) {
        if (bytes == 0) return own_dst;
        void* p = CS(s).ws_stage_dev.ptr ? CS(s).stage.Take(bytes) : nullptr;
        if (!p || (size_t)(static_cast<char*>(p) - CS(s).stage.base) + bytes > CS(s).ws_stage_dev.bytes) {
            const hipError_t e = UploadAsync(s, own_dst, src, bytes, st);
            if (err == hipSuccess) err = e;
            return own_dst;
        }
        std::memcpy(p, src, bytes);
        const size_t off = (size_t)(static_cast<char*>(p) - CS(s).stage.base);
        first = std::min(first, off);
        last = std::max(last, off + bytes);
        return reinterpret_cast<T*>(CS(s).ws_stage_dev.as<char>() + off);
    }
    hipError_t Flush() {
        hipError_t e = err;
        if (first < last) {
            hipError_t c;
            char* src_dev = by_kernel ? static_cast<char*>(DeviceAddressOfPinned(CS(s).stage.base)) : nullptr;
            if (src_dev) {   // (stage offsets are 64-byte aligned: PinnedStage::Take)
                const size_t a = first & ~(size_t)15, b = (last + 15) & ~(size_t)15;
                c = KernelCopy(CS(s).ws_stage_dev.as<char>() + a, src_dev + a, std::min(b, CS(s).ws_stage_dev.bytes) - a, st);
            } else {
                c = hipMemcpyAsync(CS(s).ws_stage_dev.as<char>() + first, CS(s).stage.base + first, last - first, hipMemcpyHostToDevice, st);
            }
            if (e == hipSuccess) e = c;
        }
        first = SIZE_MAX;
        last = 0;
        err = hipSuccess;
        return e;
    }
};


// Group the caller's (query, slot) pairs by shard (counting sort): the TASKS of this call.
static gpann_status BuildTasks(const gpann_shards* s, uint32_t nq, const int32_t* probes, uint32_t probe_stride, uint32_t nprobe, ScanPlan& plan) {
    const uint32_t S = s->S;
    std::vector<uint32_t>& count = plan.shard_task_begin;
    count.assign(S + 1, 0);
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t j = 0; j < nprobe; ++j) {
            int32_t b = probes[(size_t)q * probe_stride + j];
            if (b < 0) continue;
            GPANN_REQUIRE((uint32_t)b < S, "probe shard id out of range");
            count[b + 1]++;
        }
    for (uint32_t b = 0; b < S; ++b) count[b + 1] += count[b];
    const uint32_t T = count[S];
    plan.task_query.resize(T);
    plan.task_out.resize(T);
    plan.task_shard.resize(T);
    std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t j = 0; j < nprobe; ++j) {
            int32_t b = probes[(size_t)q * probe_stride + j];
            if (b < 0) continue;
            uint32_t t = cursor[b]++;
            plan.task_query[t] = q;
            plan.task_out[t] = q * nprobe + j;
            plan.task_shard[t] = (uint32_t)b;
        }
    plan.pairs = 0;
    for (uint32_t b = 0; b < S; ++b) plan.pairs += (uint64_t)(count[b + 1] - count[b]) * (uint64_t)(s->offsets[b + 1] - s->offsets[b]);
    return GPANN_OK;
}

// Units of the exact kernel for the row range [lo[b], hi[b]) of every shard: (<= 64 tasks) x (one row segment).  Segments
// are sized so that the launch has enough workgroups to fill 256 CUs several times over while each unit still amortises
// the warm-up of its top-k lists.
static gpann_status BuildExactUnits(const gpann_shards* s, ScanPlan& plan, uint32_t S, const std::vector<int64_t>& lo, const std::vector<int64_t>& hi) {
    const std::vector<uint32_t>& count = plan.shard_task_begin;
    const uint32_t T = count[S];
    plan.task_part.assign(T, 0);
    plan.task_nseg.assign(T, 0);
    plan.units.clear();
    uint64_t tile_rows = 0;
    for (uint32_t b = 0; b < S; ++b) tile_rows += (uint64_t)((count[b + 1] - count[b] + kQT - 1) / kQT) * (uint64_t)std::max<int64_t>(0, hi[b] - lo[b]);
    const uint32_t target_units = std::max<uint32_t>(1, s->opt.U32("GPANN_SCAN_TARGET_UNITS", 8192));
    const uint32_t min_seg_rows = std::max<uint32_t>(64, s->opt.U32("GPANN_SCAN_MIN_SEG_ROWS", 1024));
    uint64_t seg_rows = std::max<uint64_t>(min_seg_rows, (tile_rows + target_units - 1) / std::max<uint32_t>(1, target_units));
    // short scans (stage 0 of the filtered pipeline with few tasks): rather 256-row segments than fewer than ~500 workgroups
    if (tile_rows / seg_rows < 512) seg_rows = std::max<uint64_t>(256, std::min<uint64_t>(seg_rows, tile_rows / 1024));
    seg_rows = (seg_rows + kPT - 1) / kPT * kPT;
    uint64_t lists = 0;
    for (uint32_t b = 0; b < S; ++b) {
        const uint32_t tb = count[b + 1] - count[b];
        if (tb == 0) continue;
        const int64_t r0 = lo[b], r1 = hi[b];
        const uint64_t nb = (uint64_t)std::max<int64_t>(0, r1 - r0);
        const uint32_t nseg = nb == 0 ? 0 : (uint32_t)((nb + seg_rows - 1) / seg_rows);
        for (uint32_t t = count[b]; t < count[b + 1]; ++t) {
            plan.task_part[t] = (uint32_t)lists;
            plan.task_nseg[t] = nseg;
            lists += nseg;
        }
        GPANN_REQUIRE(lists < 0xFFFFFFFFull, "too many partial lists");
        if (nseg == 0) continue;
        // equal segments (multiples of the tile height) rather than seg_rows + remainder
        const uint64_t per = ((nb + nseg - 1) / nseg + kPT - 1) / kPT * kPT;
        const uint32_t ntiles_q = (tb + kQT - 1) / kQT;
        for (uint32_t qt = 0; qt < ntiles_q; ++qt) {
            // balanced query tiles: e.g. 65 tasks -> 33 + 32 instead of 64 + 1
            const uint32_t tbeg = count[b] + (uint32_t)((uint64_t)tb * qt / ntiles_q);
            const uint32_t tend = count[b] + (uint32_t)((uint64_t)tb * (qt + 1) / ntiles_q);
            for (uint32_t sg = 0; sg < nseg; ++sg) {
                ScanUnit u;
                u.row_begin = r0 + (int64_t)(sg * per);
                u.row_end = std::min<int64_t>(r1, u.row_begin + (int64_t)per);
                if (u.row_begin >= u.row_end) u.row_begin = u.row_end = r1;
                u.task_begin = tbeg;
                u.task_count = tend - tbeg;
                u.seg = sg;
                u.pad = 0;
                plan.units.push_back(u);
                plan.padded_pairs += (uint64_t)kQT * (uint64_t)(((u.row_end - u.row_begin) + kPT - 1) / kPT * kPT);
            }
        }
    }
    plan.n_lists = lists;
    return GPANN_OK;
}

// queries (host or device, stride d) -> device workspace with stride ld, zero padded
static gpann_status StageQueries(gpann_shards* s, const float* queries, uint32_t nq, bool on_device, hipStream_t st, const float** out) {
    if (on_device && s->ld == s->d && (reinterpret_cast<uintptr_t>(queries) & 15u) == 0) {   // already in the layout the kernels read
        *out = queries;
        return GPANN_OK;
    }
    const size_t bytes = (size_t)nq * s->ld * sizeof(float);
    GPANN_TRY(CS(s).ws_queries.Reserve(std::max<size_t>(bytes, 256)));
    float* dst = CS(s).ws_queries.as<float>();
    if (s->ld != s->d) GPANN_HIP_TRY(hipMemsetAsync(dst, 0, bytes, st));
    GPANN_HIP_TRY(hipMemcpy2DAsync(dst, (size_t)s->ld * sizeof(float), queries, (size_t)s->d * sizeof(float), (size_t)s->d * sizeof(float), nq,
                                   on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    *out = dst;
    return GPANN_OK;
}

struct DeviceLists {   // per-(query, slot) result lists on the device: [nq*nprobe][k]
    float* dist = nullptr;
    uint32_t* ids = nullptr;
    uint32_t* rows = nullptr;
};

struct TaskArrays {   // device copies of the shard-sorted task arrays
    uint32_t *query, *out, *part, *nseg, *shard;
};

// One exact stage: ScanTileKernel over plan.units, then MergeListsKernel into `m` (destination chosen by the caller).
static gpann_status RunExactStage(gpann_shards* s, ScanPlan& plan, const TaskArrays& ta, const float* d_queries, uint32_t k, int flags, hipStream_t st,
                                  MergeArgs m) {
    const uint32_t T = (uint32_t)plan.task_query.size();
    GPANN_TRY(CS(s).ws_units.Reserve(std::max<size_t>(plan.units.size() * sizeof(ScanUnit), 256)));
    GPANN_TRY(CS(s).ws_partials.Reserve(std::max<size_t>(plan.n_lists * k * 8, 256)));
    float* d_part_dist = CS(s).ws_partials.as<float>();
    uint32_t* d_part_row = reinterpret_cast<uint32_t*>(d_part_dist + plan.n_lists * k);
    if (T) {
        GPANN_HIP_TRY(UploadAsync(s, ta.part, plan.task_part.data(), T * 4, st));
        GPANN_HIP_TRY(UploadAsync(s, ta.nseg, plan.task_nseg.data(), T * 4, st));
    }
    if (!plan.units.empty()) {
        GPANN_HIP_TRY(UploadAsync(s, CS(s).ws_units.ptr, plan.units.data(), plan.units.size() * sizeof(ScanUnit), st));
        ScanArgs a;
        a.points = s->points;
        a.queries = d_queries;
        a.units = CS(s).ws_units.as<ScanUnit>();
        a.task_query = ta.query;
        a.task_part = ta.part;
        a.part_dist = d_part_dist;
        a.part_row = d_part_row;
        a.d = s->d;
        a.ld = s->ld;
        a.k = k;
        const size_t lds = (size_t)(kQT * kLdsStride + (kPT / 2) * kPairStride + kQT * kPT) * sizeof(float);
        const dim3 grid((uint32_t)plan.units.size());
        const bool aux = (flags & GPANN_AUX_PASS) != 0;
        if (s->metric == GPANN_METRIC_L2) {
            if (aux) hipLaunchKernelGGL((ScanTileKernel<GPANN_METRIC_L2, 1>), grid, dim3(kThreads), lds, st, a);
            else hipLaunchKernelGGL((ScanTileKernel<GPANN_METRIC_L2, 0>), grid, dim3(kThreads), lds, st, a);
        } else {
            if (aux) hipLaunchKernelGGL((ScanTileKernel<GPANN_METRIC_MIPS, 1>), grid, dim3(kThreads), lds, st, a);
            else hipLaunchKernelGGL((ScanTileKernel<GPANN_METRIC_MIPS, 0>), grid, dim3(kThreads), lds, st, a);
        }
        GPANN_HIP_TRY(hipGetLastError());
    }
    GPANN_HIP_TRY(hipEventRecord(CS(s).ev[1], st));
    if (T) {
        m.in_dist = d_part_dist;
        m.in_row = d_part_row;
        m.first_list = ta.part;
        m.n_lists = ta.nseg;
        m.n_out = T;
        m.k = k;
        m.lists_per_out = 0;
        hipLaunchKernelGGL(MergeListsKernel<0>, dim3((T + 3) / 4), dim3(256), 0, st, m);
        GPANN_HIP_TRY(hipGetLastError());
    }
    return GPANN_OK;
}

// Splits the tb tasks of a shard into filter groups of <= group_tasks (256; 128 beyond 128 dimensions) tasks, sized in whole 32-task query tiles (only a shard's last
// group can end on a partly used tile).
struct FilterGroup {
    uint32_t task_off, task_count;
};
static std::vector<FilterGroup> PlanFilterGroups(uint32_t tb, uint32_t group_tasks) {
    std::vector<FilterGroup> out;
    const uint32_t nqb = (tb + group_tasks - 1) / group_tasks;
    const uint32_t tiles = (tb + 31) / 32;
    uint32_t done = 0;
    for (uint32_t qb = 0; qb < nqb; ++qb) {
        const uint32_t p0 = (uint32_t)((uint64_t)tiles * qb / nqb), p1 = (uint32_t)((uint64_t)tiles * (qb + 1) / nqb);
        const uint32_t cnt = std::min(tb - done, (p1 - p0) * 32u);
        out.push_back({done, cnt});
        done += cnt;
    }
    return out;
}

// The FilterKernelCells instances a store's passes launch (its k-steps and metric pick them): both bf16 planes / hi plane only, and — a store with an
// 8-bit plane — the rest pass, the sample pass under a symbol of its own (TAG 1: the rest pass keeps rocprofv3's per-kernel statistics to itself; one row
// tile per unit: its items are the 20-odd tiles of a cell's front) and the rest pass of a call that runs BESIDE other calls (TAG 2, GPANN_SCAN_OVERLAP:
// the same code under a third symbol, so that the statistics keep launches that share the chip apart from the ones that have it to themselves).
struct FilterKernelSet {
    const void *fn = nullptr, *fn_hi = nullptr, *fn_q8 = nullptr, *fn_q8_sample = nullptr, *fn_q8_beside = nullptr;
};
static bool PickFilterKernels(bool l2, uint32_t KS, bool use_q8, uint32_t KS8, FilterKernelSet* out) {
    FilterKernelSet f;
    switch (KS) {
#define GPANN_PICK(KS_)                                                                                                                                   \
    case KS_:                                                                                                                                             \
        f.fn_hi = l2 ? (const void*)FilterKernelCells<GPANN_METRIC_L2, 0, KS_, 1> : (const void*)FilterKernelCells<GPANN_METRIC_MIPS, 0, KS_, 1>;    \
        if constexpr (KS_ <= 8) f.fn = l2 ? (const void*)FilterKernelCells<GPANN_METRIC_L2, 0, KS_, 0> : (const void*)FilterKernelCells<GPANN_METRIC_MIPS, 0, KS_, 0>; \
        else f.fn = f.fn_hi;                                                                                                                              \
        break;
        GPANN_PICK(2) GPANN_PICK(4) GPANN_PICK(6) GPANN_PICK(7) GPANN_PICK(8) GPANN_PICK(10) GPANN_PICK(13) GPANN_PICK(16)
#undef GPANN_PICK
        default: return false;
    }
    if (use_q8) {
        switch (KS8) {
#define GPANN_PICK8(KS_)                                                                                                                       \
    case KS_:                                                                                                                                  \
        f.fn_q8 = l2 ? (const void*)FilterKernelCells<GPANN_METRIC_L2, 0, KS_, 2, 8> : (const void*)FilterKernelCells<GPANN_METRIC_MIPS, 0, KS_, 2, 8>; \
        f.fn_q8_sample = l2 ? (const void*)FilterKernelCells<GPANN_METRIC_L2, 1, KS_, 2, 8, 1> : (const void*)FilterKernelCells<GPANN_METRIC_MIPS, 1, KS_, 2, 8, 1>; \
        f.fn_q8_beside = l2 ? (const void*)FilterKernelCells<GPANN_METRIC_L2, 2, KS_, 2, 8> : (const void*)FilterKernelCells<GPANN_METRIC_MIPS, 2, KS_, 2, 8>; \
        break;
            GPANN_PICK8(1) GPANN_PICK8(2) GPANN_PICK8(3) GPANN_PICK8(4) GPANN_PICK8(5) GPANN_PICK8(6) GPANN_PICK8(7) GPANN_PICK8(8)
#undef GPANN_PICK8
            default: break;
        }
    }
    *out = f;
    return true;
}
static bool StoreUsesQ8(const gpann_shards* s) {
    const auto& C = s->cells;
    return C.q8_planes != nullptr && C.q8_ks >= 1 && C.q8_ks <= 8 && (C.hi_compact || s->opt.U32("GPANN_FILTER_Q8", 1) != 0);
}

// GPANN_SCAN_OVERLAP rests on one property of the BUILT kernels: every matrix-core kernel is allocated all 256 architectural vector registers, so two of
// its wavefronts fill a SIMD's register file and no wavefront of another launch is ever resident beside them (common.h TakeWholeRegisterFile: a
// wavefront that shared a SIMD with another kernel's v_mfma_i32_32x32x32_i8 wavefront was measured to end with wrong registers,
// profiles/r05/overlap_race.md).  tests/test_capi_exports.py checks the code objects of the shipped library; this is the same question asked of the
// RUNTIME about the very kernels this store launches, once per store, before its first overlapped call: a library built by another compiler, or with the
// claim edited away, runs its calls one after the other on the caller's stream instead (same results, no sharing of compute units) and says so once.
static bool OverlapKernelsOwnTheirSimds(gpann_shards* s) {
    if (s->overlap_kernels_state == 0) {
        FilterKernelSet f;
        bool ok = PickFilterKernels(s->metric == GPANN_METRIC_L2, s->cells.plane_ks, StoreUsesQ8(s), s->cells.q8_ks, &f);
        int least = 256;
        if (ok) {
            for (const void* fn : {f.fn, f.fn_hi, f.fn_q8, f.fn_q8_sample, f.fn_q8_beside}) {
                if (!fn) continue;
                hipFuncAttributes attr{};
                if (hipFuncGetAttributes(&attr, fn) != hipSuccess) {
                    (void)hipGetLastError();
                    ok = false;
                    break;
                }
                least = std::min(least, attr.numRegs);
            }
        }
        s->overlap_kernels_regs = ok ? least : 0;
        s->overlap_kernels_state = ok && least >= 256 ? 1 : 2;
    }
    // (GPANN_OVERLAP_ASSUME_SHARED_SIMDS: test hook — behave as if a kernel of this build left room beside itself)
    const bool ok = s->overlap_kernels_state == 1 && s->opt.U32("GPANN_OVERLAP_ASSUME_SHARED_SIMDS", 0) == 0;
    if (!ok && !s->overlap_refusal_said) {
        s->overlap_refusal_said = true;
        fprintf(stderr, "[gpann] GPANN_SCAN_OVERLAP is ignored for this store: a matrix-core kernel of this build is allocated %d vector registers, not 256 — "
                        "wavefronts of other launches could run beside it (csrc/common.h TakeWholeRegisterFile); its calls run one after the other instead\n",
                s->overlap_kernels_regs);
    }
    return ok;
}

// The MFMA-filtered pipeline (cells.hip, scan_cells.h, scan_filter.h):
//   tasks of a shard ordered by nearest cell -> stage 0 (exact, first rows of the shard in store order: a random sample of it)
//   -> head pass (the first tiles of every task's nearest cell) -> main pass (every cell the bound cannot exclude), each followed
//   by the exact re-scoring of its candidates.  `ta.query` / `ta.out` are re-uploaded in the new task order.
// `st`: the stream of the call's FRONT — everything before the rest pass's filter launch; `st_rest`: the stream of that launch; `st_tail`: what follows
// it.  The same stream for a plain call; GPANN_SCAN_OVERLAP: the handle's three streams (each waits for the stage before it: CallState::ev_x).
// *st_end = the stream the call's last launch went to.
static gpann_status RunCellFilteredScan(gpann_shards* s, ScanPlan& plan, const TaskArrays& ta_in, UploadBatch& up, const float* d_queries, uint32_t nq, uint32_t nprobe_of_call,
                                        uint32_t k, int flags, hipStream_t st, hipStream_t st_rest, hipStream_t st_tail, hipStream_t* st_end, DeviceLists out) {
    *st_end = st;
    const bool overlap = st_rest != st;
    TaskArrays ta = ta_in;   // (query / out point to the re-ordered copies once the tasks are grouped by nearest cell; the caller's arrays keep the host's order)
    CS(s).filter_passes_timed = 0;
    const uint32_t S = s->S, T = (uint32_t)plan.task_query.size();
    if (T == 0) {   // no task: every slot reads (+inf, NO_ID)
        const size_t n = (size_t)nq * nprobe_of_call * k;
        if (n) hipLaunchKernelGGL(FillKernel, dim3((uint32_t)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, out.dist, out.ids, out.rows, n);
        GPANN_HIP_TRY(hipGetLastError());
        return GPANN_OK;
    }
    const auto& C = s->cells;
    const uint32_t kmax = C.kmax;
    const int64_t m0 = (int64_t)s->opt.U32("GPANN_FILTER_EXACT_ROWS", 4096) / kPT * kPT;
    // candidate bucket per task: what its rows cost is the candidates actually written, so the size is only bounded by memory (6 GiB);
    // the pass structure below is laid out for a fill of about a quarter of min(cap, 4096)
    const uint32_t cap = s->opt.U32("GPANN_FILTER_CAND_CAP", (uint32_t)std::min<uint64_t>(16384, std::max<uint64_t>(1024, (6144ull << 20) / (4ull * T))));
    const uint32_t cap_plan = std::min<uint32_t>(cap, 4096);
    // Three filter passes keep the candidate buckets small whatever the geometry: rows inside a cell are in store order, i.e.
    // a random sample of the cell, so after a pass over a fraction 1/f of the needed rows only about f * k rows of the rest can
    // still beat the k-th distance.  Pass 0 = the head of each task's nearest cell (at most half a bucket of rows), pass 1 =
    // the first 1/16 of every needed cell, pass 2 = the rest of the cells still needed under the tightened k-th distance.
    // (GPANN_FILTER_HEAD_ROWS=0: no head pass — the seed lists go straight to the sample pass)
    // (the head pass multiplies both planes: it cannot reach past the rows that have a lo tile)
    // Small shards (on average fewer than GPANN_FILTER_SMALL_SHARD_ROWS rows; default 262144 = the shard size below which DefaultCellRows no longer
    // shrinks the cells with the shard): no shard-sample and no head pass unless their knobs ask for them.  Both exist to tighten the k-th distances
    // cheaply BEFORE the passes that stream the store, and each costs a set-up, a filter and a re-scoring launch — about 0.1 ms whatever the store's
    // size.  On a store whose sample + rest passes take 0.3 ms together they cost more than they save (C2, 1M x 128 in 8 shards, 10k queries:
    // 1.19 -> 1.02 ms per step, candidates 0.62M -> 0.94M; C3, 10M x 96 in 16 shards, is slower without them: profiles/r06/sweeps/small_store_passes.txt).
    // ... where the cells prune: on a cloud without structure (one blob, overlapping components) every row reaches the matrix cores whatever the thresholds, the
    // two passes' tighter k-th distances then save more re-scoring than their launches cost (C2-size store, one blob: 2.33M QPS without them, 2.73M with;
    // sigma 0.8: 4.86M / 5.11M), so the rule applies only while the previous filtered call of the store sent at most GPANN_FILTER_SMALL_SHARD_MAX_SHARE per
    // cent (default 50) of the (task, row) pairs it was asked for to the matrix cores — C2: 17 %, the blob: 104 % (pairs of all passes over the pairs asked
    // for).  A store's first call has no history and takes the rule.  (Like the CU split of overlapped calls, this is a choice between pass structures
    // that return the same bits.)
    const bool cells_prune = s->filter_pair_share < 0.0 || s->filter_pair_share * 100.0 <= (double)s->opt.U32("GPANN_FILTER_SMALL_SHARD_MAX_SHARE", 50);
    const bool small_shards = cells_prune && (uint64_t)s->rows < (uint64_t)s->opt.U32("GPANN_FILTER_SMALL_SHARD_ROWS", 262144) * (uint64_t)std::max<uint32_t>(1, s->S);
    const uint32_t head_tiles = std::min<uint32_t>(std::min<uint32_t>(std::min<uint32_t>(s->opt.U32("GPANN_FILTER_HEAD_ROWS", small_shards ? 0 : 512), cap_plan / 2) / kGTileRows,
                                                                      C.plane_ks > 8 ? 0xFFFFFFFFu : C.lo_head_tiles),
                                                   C.hi_compact ? 16u : 0xFFFFFFFFu);   // (compact hi planes keep 16 tiles of every cell at least)
    int64_t max_rows = 1;
    for (uint32_t b = 0; b < S; ++b) max_rows = std::max<int64_t>(max_rows, s->offsets[b + 1] - s->offsets[b]);
    // the sample pass expects (rows / div) * k / m0 candidates per task: keep that near a quarter of a bucket
    const uint64_t div_auto = ((uint64_t)max_rows * k * 4 + (uint64_t)m0 * cap_plan - 1) / ((uint64_t)std::max<int64_t>(m0, 1) * cap_plan);
    // (compact hi planes: the sample segment has to stay inside the tiles that kept a bf16 hi tile)
    // (round 3, buckets of up to 16k candidates: half of that share is enough — NS 10 -> 8: 0.1 ms of 4.3, C5 18 -> 9; profiles/r03/sweep_sample_div.md)
    const uint32_t sample_div = std::max<uint32_t>(std::max<uint32_t>(2, C.hi_keep_div), s->opt.U32("GPANN_FILTER_SAMPLE_DIV", (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(8, div_auto / 2))));
    if (std::getenv("GPANN_CELLS_DEBUG")) fprintf(stderr, "[gpann cells] sample_div %u (auto %llu)\n", sample_div, (unsigned long long)div_auto);
    const bool l2 = s->metric == GPANN_METRIC_L2;
    (void)flags;
    const uint32_t KS = C.plane_ks, QT = (uint32_t)FilterQueryTiles((int)KS);
    const bool wide = KS > 8;   // beyond 128 dimensions: hi-plane passes only, 4 query tiles per group
    // the rest pass on the 8-bit plane (scan_cells.h MODE 2) when the store has one: a quarter of the fp32 bytes per row instead of half
    // (a store with compact hi planes has no bf16 tiles for the rest pass: the knob is ignored there)
    const bool use_q8 = StoreUsesQ8(s);
    const uint32_t KS8 = C.q8_ks, QT8 = 8;
    // (never the head pass: an 8-bit pass gives a task candidates only from the cells NeedKernel marked for it — its per-(task, cell) thresholds —
    // while the later passes skip a segment for a whole query tile once the tile has run on it; the head pass marks one cell per task, the bf16
    // kernel's per-task threshold tests every task of the tile against it all the same.  Measured with the clamp lifted: wrong lists.)
    const int q8_from_pass = (int)std::max<uint32_t>(1, std::min<uint32_t>(2, s->opt.U32("GPANN_FILTER_Q8_FROM_PASS", 1)));
    const uint32_t group_tasks = 32u * QT;   // the query image of a group has to fit LDS

    // ---- workspace
    const bool for_merge = (flags & GPANN_SCAN_FOR_MERGE) != 0;
    // dirty tasks (bucket overflow): at most dirty_cap of them are redone exactly on the device; more than that = the data defeats the
    // filter (masses of equal distances) and the whole call falls back to the plain exact scan
    const uint32_t dirty_cap = std::max<uint32_t>(16, std::min<uint32_t>(256, T / 100));
    // ... each shard cut into segments of about 512 rows, one wavefront each (a fixed 512 segments left a wavefront 6 500 rows of a 3.3M-row C5
    // shard to walk, and DirtyMergeKernel walked the partial lists one dependent load after the other: ONE dirty task cost 2.4 ms of a 9 ms step)
    uint32_t range_segs = 512;
    {
        int64_t largest = 0;
        for (uint32_t b = 0; b < s->S; ++b) largest = std::max<int64_t>(largest, s->offsets[b + 1] - s->offsets[b]);
        const uint64_t fit = (256ull << 20) / ((uint64_t)dirty_cap * k * 8);   // partial lists: at most 256 MiB
        range_segs = (uint32_t)std::max<uint64_t>(512, std::min<uint64_t>(std::min<uint64_t>(8192, fit), (uint64_t)largest / 512));
    }
    GPANN_TRY(CS(s).ws_lists.Reserve((size_t)T * k * 8 + (size_t)T * 4 + 256 + 64 + (size_t)nq * 4 + (size_t)T * 8 + (size_t)dirty_cap * 8 + 64));
    GPANN_TRY(CS(s).ws_partials.Reserve((size_t)dirty_cap * range_segs * k * 8 + 256));
    float* list_dist = CS(s).ws_lists.as<float>();
    uint32_t* list_row = reinterpret_cast<uint32_t*>(list_dist + (size_t)T * k);
    float* task_U = reinterpret_cast<float*>(list_row + (size_t)T * k);
    uint32_t* d_overflow = reinterpret_cast<uint32_t*>(task_U + T + ((T & 1) ? 1 : 0));   // [0] overflow count, [2..3] candidate total (u64)
    unsigned long long* d_cand_total = reinterpret_cast<unsigned long long*>(d_overflow + 2);
    // GPANN_SCAN_FOR_MERGE: per query the smallest k-th distance any of its tasks has reached (OrderedKey; RescoreKernel lowers it, the
    // next pass's thresholds and cell tests of ALL the query's tasks read it)
    uint32_t* d_query_tau = for_merge ? d_overflow + 16 : nullptr;
    uint32_t* d_dirty_flag = d_overflow + 16 + nq;   // [T] | dirty_list [dirty_cap] uint2 (8-byte aligned below)
    uint2* d_dirty_list = reinterpret_cast<uint2*>(d_dirty_flag + (size_t)T + ((reinterpret_cast<uintptr_t>(d_dirty_flag + (size_t)T) & 4) ? 1 : 0));
    GPANN_TRY(CS(s).ws_cand.Reserve((size_t)T * 4 + (size_t)T * cap * 4));
    uint32_t* cand_cnt = CS(s).ws_cand.as<uint32_t>();
    uint32_t* cand_row = cand_cnt + T;

    // groups of <= 256 tasks per shard; distance-table tiles of <= 64 tasks
    std::vector<CellGroup> groups;
    std::vector<uint32_t> task_group(T);
    std::vector<CellTileRef> dtiles;
    for (uint32_t b = 0; b < S; ++b) {
        const uint32_t t0 = plan.shard_task_begin[b], tb = plan.shard_task_begin[b + 1] - t0;
        if (tb == 0) continue;
        for (const FilterGroup& g : PlanFilterGroups(tb, group_tasks)) {
            for (uint32_t x = 0; x < g.task_count; ++x) task_group[t0 + g.task_off + x] = (uint32_t)groups.size();
            groups.push_back({t0 + g.task_off, g.task_count, b, 0});
        }
        for (uint32_t f = 0; f < tb; f += 64) dtiles.push_back({b, t0 + f, std::min(64u, tb - f), 0});
    }
    const uint32_t G = (uint32_t)groups.size();
    const size_t need_words = (size_t)G * kmax;
    // (the 8-bit passes of a squared-L2 store whose plane has its own norm words bound the quantisation error in the frame of the cell: |q - c|_1 per (task, cell))
    // (GPANN_FILTER_Q8_L1=0: no table — |q - c|_1 <= sqrt(d) |q - c|_2 from the table of squared distances instead, ~25 % looser)
    const bool cell_l1 = use_q8 && C.q8_norm != nullptr && s->opt.U32("GPANN_FILTER_Q8_L1", 1) != 0;
    GPANN_TRY(CS(s).ws_celld.Reserve((size_t)T * 16 + (size_t)T * 8 + (size_t)T * kmax * 4 * (cell_l1 ? 2 : 1) + (size_t)T * 20 + dtiles.size() * sizeof(CellTileRef) + (size_t)(S + 1) * 4 + 256));
    double* d_task_nq = CS(s).ws_celld.as<double>();   // |q|^2 of every task (old task order, like the table and the nearest cells)
    double* d_task_q1 = d_task_nq + T;              // |q~|_1 (old order) ...
    float* d_task_tscale = reinterpret_cast<float*>(d_task_q1 + T);   // ... the 8-bit query step (old order) ...
    float* d_task_qscale = d_task_tscale + T;                          // ... and step / 254 in the NEW order (QueryImageQ8Kernel)
    float* cell_dist = d_task_qscale + T;
    float* cell_l1_tab = cell_l1 ? cell_dist + (size_t)T * kmax : nullptr;
    uint32_t* d_nearest = reinterpret_cast<uint32_t*>(cell_dist + (size_t)T * kmax * (cell_l1 ? 2 : 1));
    uint32_t* d_task_group_own = d_nearest + T;
    CellTileRef* d_dtiles_own = reinterpret_cast<CellTileRef*>(d_task_group_own + T + (T & 1));
    // work items of a pass: pieces of at most seg_tiles row tiles (small stores: small pieces, so that every CU gets some)
    const uint32_t seg_tiles = std::max<uint32_t>(8, s->opt.U32("GPANN_FILTER_SEG_TILES", (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(16, C.n_tiles / 2048))));
    GPANN_REQUIRE(((uint64_t)C.sample_tile_base + C.n_sample_tiles + 2) * kGTileRows < 0xFFFFFFFFull, "plane rows beyond 32 bits");
    // one item list per task group: group_first[g] = where it starts (capacity: every cell of the group's shard cut into pieces)
    uint64_t items_cap = 0;
    std::vector<uint32_t> group_first(G + 1);
    for (uint32_t g = 0; g < G; ++g) {
        GPANN_REQUIRE(items_cap < 0xFFFFFFFFull, "too many filter work items");
        group_first[g] = (uint32_t)items_cap;
        for (uint32_t c = C.shard_cell_begin[groups[g].shard]; c < C.shard_cell_begin[groups[g].shard + 1]; ++c)
            items_cap += (C.cell_tile_begin[c + 1] - C.cell_tile_begin[c]) / seg_tiles + 1;
    }
    GPANN_REQUIRE(items_cap < 0xFFFFFFFFull, "too many filter work items");
    group_first[G] = (uint32_t)items_cap;
    const size_t gtab_words = ((size_t)3 * G + 1 + 3) / 4 * 4;   // group_first [G+1] | group_count [G] | group_cursor [G], 16-byte multiple
    const size_t ticket_words = 0;
    GPANN_TRY(CS(s).ws_need.Reserve((4 * need_words + ticket_words) * 4 + (size_t)G * sizeof(CellGroup) + 96 + gtab_words * 4 + (size_t)items_cap * sizeof(CellItem)));
    uint32_t* need_pass[3];
    need_pass[0] = CS(s).ws_need.as<uint32_t>();
    need_pass[1] = need_pass[0] + need_words;
    need_pass[2] = need_pass[1] + need_words;
    uint32_t* done_head = need_pass[2] + need_words;   // need_pass[0] | need_pass[1]
    uint32_t* d_tickets = done_head + need_words;      // (zeroed with the masks)
    CellGroup* d_groups_own = reinterpret_cast<CellGroup*>(d_tickets + ticket_words);
    unsigned long long* d_work = reinterpret_cast<unsigned long long*>(d_groups_own + G);   // [0..2] work counters (scan_cells.h), [3] = [1] before the rest pass
    uint32_t* d_group_first_own = reinterpret_cast<uint32_t*>(d_work + 8);   // (8 counters: [4] / [5] = bf16 / int8 matrix instructions issued, [6] = [5] before the rest pass, [7] spare)
    uint32_t* d_group_count = d_group_first_own + G + 1;   // (+ the claim cursors: zeroed together by NeedKernel)
    CellItem* d_items = reinterpret_cast<CellItem*>(d_group_first_own + gtab_words);
    // the call's tables join the task arrays in the upload batch (one copy for all of them, flushed below)
    const CellTileRef* d_dtiles = up.Add(d_dtiles_own, dtiles.data(), dtiles.size() * sizeof(CellTileRef));
    const CellGroup* d_groups = up.Add(d_groups_own, groups.data(), (size_t)G * sizeof(CellGroup));
    const uint32_t* d_group_first = up.Add(d_group_first_own, group_first.data(), (size_t)(G + 1) * 4);
    const uint32_t* d_task_group = up.Add(d_task_group_own, task_group.data(), (size_t)T * 4);
    // the shard-sample pass's work items (host-built: every group x its shard's sample tiles) and their list tables travel with them
    std::vector<CellItem> sitems;
    std::vector<uint32_t> stab((size_t)3 * G + 4, 0u);   // group_first [G] | group_count [G] | group_cursor [G] of that pass's lists
    uint32_t* d_stab = nullptr;
    CellItem* d_sitems = nullptr;
    const bool shard_sample_pass = C.n_sample_tiles != 0 && s->opt.U32("GPANN_FILTER_SEED_ROWS", 128) != 0 && s->opt.U32("GPANN_FILTER_SHARD_SAMPLE", small_shards ? 0 : 1) != 0;
    if (shard_sample_pass) {
        const uint32_t piece = 32;   // row tiles per work item
        for (uint32_t g = 0; g < G; ++g) {
            stab[g] = (uint32_t)sitems.size();
            const uint64_t t0 = C.sample_tile_base + C.sample_tile_begin[groups[g].shard], t1 = C.sample_tile_base + C.sample_tile_begin[groups[g].shard + 1];
            const uint32_t qtiles = (groups[g].task_count + 31) / 32, mask = (1u << qtiles) - 1u;
            for (uint64_t t = t0; t < t1; t += piece) {
                CellItem it;
                it.t_begin = (uint32_t)t;
                it.t_end = (uint32_t)std::min<uint64_t>(t1, t + piece);
                it.t_a = it.t_b = it.t_begin;   // no head / sample segment: every query tile runs on every row tile
                it.task_begin = groups[g].task_begin;
                it.lo_shift = (uint32_t)(C.sample_tile_base - C.lo_sample_begin);   // sample tile j: row tile sample_tile_base + j, lo tile lo_sample_begin + j ...
                it.hi_shift = (uint32_t)(C.sample_tile_base - C.hi_sample_begin);   // ... hi tile hi_sample_begin + j
                it.block = 0;
                it.cell_scale = 0.f;
                it.pad1 = 0;
                it.acts = mask | mask << 8 | mask << 16 | (groups[g].task_count - 1u) << 24;
                it.group = g;
                sitems.push_back(it);
            }
            stab[G + g] = (uint32_t)sitems.size() - stab[g];
        }
        if (!sitems.empty()) {
            const size_t stab_words = ((size_t)3 * G + 3) / 4 * 4;
            GPANN_TRY(CS(s).ws_misc.Reserve(sitems.size() * sizeof(CellItem) + stab_words * 4 + 64));
            d_stab = up.Add(CS(s).ws_misc.as<uint32_t>(), stab.data(), (size_t)3 * G * 4);
            d_sitems = up.Add(reinterpret_cast<CellItem*>(CS(s).ws_misc.as<uint32_t>() + stab_words), sitems.data(), sitems.size() * sizeof(CellItem));
        }
    }

    CellDistArgs cd;
    cd.queries = d_queries;
    cd.task_query = ta.query;
    cd.center = C.d_center;
    cd.shard_cell_begin = C.d_shard_cell_begin;
    cd.tiles = d_dtiles;
    cd.out = cell_dist;
    cd.kmax = kmax;
    cd.d = s->d;
    cd.ld = s->ld;
    cd.out_l1 = cell_l1_tab;
    auto cell_table = [&](bool with_qnorm) -> gpann_status {
        const dim3 cgrid((uint32_t)dtiles.size(), std::max<uint32_t>(1, (kmax + 63) / 64));
        if (cell_l1) hipLaunchKernelGGL(CellDistKernel<true>, cgrid, dim3(256), 0, st, cd);
        else hipLaunchKernelGGL(CellDistKernel<false>, cgrid, dim3(256), 0, st, cd);
        hipLaunchKernelGGL(NearestCellKernel, dim3((T + 15) / 16), dim3(256), 0, st, cell_dist, ta.shard, C.d_shard_cell_begin, d_queries, ta.query, T, kmax,
                           s->d, s->ld, d_nearest, with_qnorm ? d_task_nq : nullptr, C.d_mu, use_q8 ? d_task_q1 : nullptr, use_q8 ? d_task_tscale : nullptr);
        GPANN_HIP_TRY(hipGetLastError());
        return GPANN_OK;
    };

    // ---- order the tasks of every shard by their nearest cell (on the device; the host copies of task_query / task_out are
    //      not consulted again in this call).  The table, the nearest cells and |q|^2 keep the old order: d_task_src maps.
    const uint32_t* d_stb = nullptr;
    {
        uint32_t* stb_own = reinterpret_cast<uint32_t*>(d_dtiles_own + dtiles.size()) + 3 * (size_t)T;   // (behind tmp_q | tmp_o | task_src, below)
        d_stb = up.Add(stb_own, plan.shard_task_begin.data(), (size_t)(S + 1) * 4);
    }
    HT_MARK("plan+reserve");
    GPANN_HIP_TRY(up.Flush());   // task arrays, tables, the shard-sample items: one copy
    HT_MARK("flush");
    GPANN_TRY(cell_table(true));
    uint32_t* d_task_src = nullptr;
    {
        uint32_t* tmp_q = reinterpret_cast<uint32_t*>(d_dtiles_own + dtiles.size());
        uint32_t* tmp_o = tmp_q + T;
        d_task_src = tmp_o + T;
        hipLaunchKernelGGL(TaskReorderKernel, dim3(S), dim3(256), (size_t)(kmax + 1) * 4, st, d_stb, d_nearest, C.d_shard_cell_begin, ta.query, ta.out, tmp_q, tmp_o,
                           d_task_src);
        GPANN_HIP_TRY(hipGetLastError());
        ta.query = tmp_q;
        ta.out = tmp_o;
    }

    // ---- the groups' query images (bf16 fragments in MFMA B-operand order), once per call
    GPANN_TRY(CS(s).ws_image.Reserve((size_t)G * QT * KS * 2048 + 256));
    {
        char* img = CS(s).ws_image.as<char>();
        switch (KS) {
#define GPANN_IMAGE(KS_)                                                                                                                              \
    case KS_:                                                                                                                                         \
        hipLaunchKernelGGL((QueryImageKernel<KS_, FilterQueryTiles(KS_)>), dim3(G * QT), dim3(64), 0, st, d_queries, ta.query, d_groups, s->ld, img, C.d_mu); \
        break;
            GPANN_IMAGE(2) GPANN_IMAGE(4) GPANN_IMAGE(6) GPANN_IMAGE(7) GPANN_IMAGE(8) GPANN_IMAGE(10) GPANN_IMAGE(13) GPANN_IMAGE(16)
#undef GPANN_IMAGE
            default: GPANN_REQUIRE(false, "no filter kernel for this number of k-steps");
        }
        GPANN_HIP_TRY(hipGetLastError());
    }
    // the 8-bit images and the (task, cell) threshold table of the rest pass
    char* q8_image = nullptr;
    float* q8_B = nullptr;
    size_t q8_B_words = 0;
    if (use_q8) {
        const size_t img8 = (size_t)G * QT8 * KS8 * 2048, b_words = (size_t)G * kmax * 32u * QT8;
        // (one threshold table PER 8-bit pass, both set to +inf — "this task does not need this cell" — by FilterInitKernel below: a pass's set-up
        // launch then writes the few needed (task, cell) slots only.  With one table every slot had to be rewritten in every pass: 20 000 scattered
        // 4-byte stores per task group)
        GPANN_TRY(CS(s).ws_q8.Reserve(img8 + 2 * b_words * 4 + 256));
        q8_image = CS(s).ws_q8.as<char>();
        q8_B = reinterpret_cast<float*>(q8_image + img8);
        q8_B_words = b_words;
        switch (KS8) {
#define GPANN_IMAGE8(KS_)                                                                                                                                   \
    case KS_:                                                                                                                                               \
        hipLaunchKernelGGL((QueryImageQ8Kernel<KS_, 8>), dim3(G * QT8), dim3(64), 0, st, d_queries, ta.query, d_groups, s->ld, q8_image, C.d_mu, d_task_tscale, \
                           d_task_src, d_task_qscale);                                                                                                      \
        break;
            GPANN_IMAGE8(1) GPANN_IMAGE8(2) GPANN_IMAGE8(3) GPANN_IMAGE8(4) GPANN_IMAGE8(5) GPANN_IMAGE8(6) GPANN_IMAGE8(7) GPANN_IMAGE8(8)
#undef GPANN_IMAGE8
            default: GPANN_REQUIRE(false, "no 8-bit filter kernel for this number of k-steps");
        }
        GPANN_HIP_TRY(hipGetLastError());
    }

    {   // every counter / table of the call in one launch (FilterInitKernel)
        FilterInitArgs ia{};
        ia.zero[0] = d_overflow;               // overflow / dirty counters, candidate total (16 words)
        ia.zero_words[0] = 16;
        ia.zero[1] = d_dirty_flag;             // [T]
        ia.zero_words[1] = T;
        ia.zero[2] = need_pass[0];             // the passes' need / done masks
        ia.zero_words[2] = 4 * (uint64_t)need_words + ticket_words;
        ia.zero[4] = reinterpret_cast<uint32_t*>(d_work);   // work counters (8 x u64)
        ia.zero_words[4] = 16;
        ia.zero[3] = cand_cnt;                 // [T]
        ia.zero_words[3] = T;
        ia.query_tau = d_query_tau;
        ia.nq = nq;
        ia.inf_floats = reinterpret_cast<uint32_t*>(q8_B);
        ia.inf_words = 2 * q8_B_words;
        ia.fill_dist = out.dist;   // unused slots (skipped probes, shards with < k rows) read as (+inf, NO_ID): FillKernel's job, in this launch
        ia.fill_ids = out.ids;
        ia.fill_rows = out.rows;
        ia.fill_n = (uint64_t)nq * nprobe_of_call * k;
        const uint64_t most = std::max<uint64_t>(std::max<uint64_t>(std::max<uint64_t>(ia.zero_words[2], 2 * q8_B_words), T), ia.fill_n);
        hipLaunchKernelGGL(FilterInitKernel, dim3((uint32_t)std::min<uint64_t>(2048, (most + 255) / 256)), dim3(256), 0, st, ia);
        GPANN_HIP_TRY(hipGetLastError());
    }

    // ---- stage 0 -> the first lists.  Default: the canonical distances of the first seed_rows rows of every task's NEAREST cell
    //      (rows inside a cell are in store order, i.e. a random sample of the cell): k-th of 64 .. 128 such rows is as tight a first
    //      threshold as the k-th of 4096 random rows of the whole shard, for 1/64 of the evaluations (0.77 -> 0.1 ms per 10k tasks on
    //      the NS workload).  GPANN_FILTER_SEED_ROWS=0: the earlier exact scan of the first m0 rows of the shard (ScanTileKernel).
    // (at least 6 k rows: the k-th of the seed rows has to be a quantile of them, not their maximum)
    const uint32_t seed_rows_env = s->opt.U32("GPANN_FILTER_SEED_ROWS", 128);   // (NS: 128 rows 1 % faster end to end than 64, 10 % fewer candidates)
    const uint32_t seed_rows = seed_rows_env == 0 ? 0u : std::max<uint32_t>(seed_rows_env, (6 * k + 15) / 16 * 16);
    plan.padded_pairs = 0;
    if (seed_rows == 0) {
        std::vector<int64_t> lo(S), hi(S);
        for (uint32_t b = 0; b < S; ++b) {
            lo[b] = s->offsets[b];
            hi[b] = std::min<int64_t>(s->offsets[b + 1], s->offsets[b] + m0);
        }
        GPANN_TRY(BuildExactUnits(s, plan, S, lo, hi));
        MergeArgs m{};
        m.out_index = nullptr;   // identity: list t
        m.permutation = s->permutation;
        m.out_dist = list_dist;
        m.out_rows = list_row;
        m.out_ids = nullptr;
        GPANN_TRY(RunExactStage(s, plan, ta, d_queries, k, flags, st, m));
    }

    ThresholdArgs th;
    th.list_dist = list_dist;
    th.queries = d_queries;
    th.task_query = ta.query;
    th.task_shard = ta.shard;
    th.shard_rmax = s->d_shard_rmax;
    th.task_U = task_U;
    th.tasks = T;
    th.k = k;
    th.d = s->d;
    th.ld = s->ld;
    th.hi_only = 0;
    th.task_nq = d_task_nq;
    th.query_tau = d_query_tau;
    th.task_src = nullptr;   // (set once the tasks are reordered)
    NeedArgs na;
    na.cell_dist = cell_dist;
    na.task_shard = ta.shard;
    na.task_group = d_task_group;
    na.groups = d_groups;
    na.shard_cell_begin = C.d_shard_cell_begin;
    na.radius = C.d_radius;
    na.cell_cnorm = C.d_cnorm;
    na.shard_rmax = s->d_shard_rmax;
    na.task_nq = d_task_nq;
    na.list_dist = list_dist;
    na.nearest = d_nearest;
    na.task_src = d_task_src;
    na.task_U = task_U;
    na.cand_cnt = cand_cnt;
    na.group_zero = d_group_count;
    na.n_groups = G;
    na.query_tau = d_query_tau;
    na.task_query = ta.query;
    na.q8_B = nullptr;   // (set for the pass that runs on the 8-bit plane)
    na.dirty_flag = nullptr;   // (set below, with the re-scoring arguments)
    na.q8_scale = C.d_q8_scale;
    na.q8_cnorm = C.d_q8_cnorm;
    na.task_q1 = d_task_q1;
    na.task_tscale = d_task_tscale;
    na.group_tasks = 32u * QT8;
    na.cell_l1 = cell_l1_tab;
    na.q8_own_norm = (use_q8 && C.q8_norm != nullptr) ? 1u : 0u;
    th.task_src = d_task_src;
    na.tasks = T;
    na.kmax = kmax;
    na.k = k;
    na.d = s->d;
    CellFilterArgs fa;
    fa.queries = d_queries;
    fa.task_query = ta.query;
    fa.task_U = task_U;
    fa.cand_cnt = cand_cnt;
    fa.cand_row = cand_row;
    fa.cand_cap = cap;
    fa.ld = s->ld;
    fa.frag_planes = s->frag_planes;
    fa.plane_tile_bytes = C.plane_ks * 1024u;
    fa.plane_lo_offset = (uint64_t)C.lo_offset_tiles * fa.plane_tile_bytes;
    fa.plane_norm = C.plane_norm;
    fa.plane_perm = C.plane_perm;
    fa.groups = d_groups;
    fa.shard_cell_begin = C.d_shard_cell_begin;
    fa.cell_tile_begin = C.d_cell_tile_begin;
    fa.cell_lo_begin = C.d_cell_lo_begin;
    fa.cell_hi_begin = C.d_cell_hi_begin;
    fa.kmax = kmax;
    fa.head_tiles = head_tiles;
    fa.sample_div = sample_div;
    fa.work = d_work;
    fa.n_groups = G;
    fa.items = d_items;
    fa.group_first = d_group_first;
    fa.group_count = d_group_count;
    fa.group_cursor = d_group_count + G;
    fa.image = CS(s).ws_image.as<char>();
    fa.seg_tiles = seg_tiles;
    fa.q8_planes = C.q8_planes;
    fa.q8_norm = C.q8_norm ? C.q8_norm : C.plane_norm;
    fa.tile_bound = C.d_tile_bound;
    fa.q8_B = q8_B;
    fa.task_qscale = d_task_qscale;
    fa.q8_scale = C.d_q8_scale;
    fa.round_tiles = kFWaves;
    RescoreArgs ra;
    ra.points = s->points;
    ra.queries = d_queries;
    ra.task_query = ta.query;
    ra.cand_cnt = cand_cnt;
    ra.cand_row = cand_row;
    ra.list_dist = list_dist;
    ra.list_row = list_row;
    ra.task_shard = ta.shard;
    ra.shard_offsets = s->d_offsets;
    ra.tasks = T;
    ra.k = k;
    ra.d = s->d;
    ra.ld = s->ld;
    ra.cand_cap = cap;
    ra.dedupe = 1;
    ra.dirty_flag = d_dirty_flag;
    na.dirty_flag = d_dirty_flag;
    ra.dirty_list = d_dirty_list;
    ra.dirty_count = d_overflow + 1;
    ra.dirty_cap = dirty_cap;
    ra.range_segs = 0;
    ra.parts = 1;
    ra.part_dist = CS(s).ws_partials.as<float>();
    ra.part_row = reinterpret_cast<uint32_t*>(ra.part_dist + (size_t)dirty_cap * range_segs * k);
    ra.seed_rows = 0;
    ra.nearest = d_nearest;
    ra.task_src = d_task_src;
    ra.shard_cell_begin = C.d_shard_cell_begin;
    ra.cell_tile_begin = C.d_cell_tile_begin;
    ra.plane_perm = C.plane_perm;
    ra.query_tau = d_query_tau;
    ra.seed_first_slot_of = 0;
    ra.seed_catch_up = 0;
    ra.task_out = ta.out;
    ra.seed_task_U = nullptr;
    ra.task_nq = d_task_nq;
    ra.shard_rmax = s->d_shard_rmax;
    ra.seed_hi_only = 0;
    // the kernels for this store's k-steps (PickFilterKernels): both planes (the shard-sample and head passes, up to 8 k-steps) and hi plane only
    FilterKernelSet fks;
    GPANN_REQUIRE(PickFilterKernels(l2, KS, use_q8, KS8, &fks), "no filter kernel for this number of k-steps");
    const void *fn = fks.fn, *fn_hi = fks.fn_hi;
    // the sample and rest passes (all but the small head pass) read only the hi plane of the points: half the HBM bytes and two of
    // the three products for a slightly looser bound (rest pass 10.1 -> 6.1 ms on the NS workload, 40% more candidates).  A store
    // built with GPANN_FILTER_LO_ROWS=all has a lo tile for every tile: GPANN_FILTER_HI_ONLY=0 then keeps all three products everywhere,
    // GPANN_FILTER_HI_FROM_PASS chooses the first hi-plane pass.  Beyond 128 dimensions every pass is a hi-plane pass.
    const bool hi_last = wide || !C.lo_full || (s->opt.U32("GPANN_FILTER_HI_ONLY", 1) != 0 && cap >= 2048);   // (small buckets of huge task counts cannot take the extra candidates)
    const int hi_from_pass = wide ? 0 : (C.lo_full ? (int)s->opt.U32("GPANN_FILTER_HI_FROM_PASS", 1) : (int)std::min<uint32_t>(1, s->opt.U32("GPANN_FILTER_HI_FROM_PASS", 1)));
    const size_t glds = FilterLdsBytes((size_t)QT * KS * 2048);
    GPANN_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
    GPANN_HIP_TRY(hipFuncSetAttribute(fn_hi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
    // (the 8-bit passes: the sample pass under a symbol of its own with one row tile per unit — three rounds of 8 tiles instead of two of 16, the second
    // a fifth full — and the overlapped rest pass under a third: FilterKernelSet)
    const void *fn_q8 = fks.fn_q8, *fn_q8_sample = fks.fn_q8_sample, *fn_q8_beside = fks.fn_q8_beside;
    uint32_t q8_unit_tiles = (uint32_t)FilterUnitTiles(2, (int)KS8), q8_waves = (uint32_t)FilterWaves(2, (int)KS8);
    size_t glds8 = 0;
    if (use_q8) {
        // (16 wavefronts of 128 registers per workgroup, one row tile per unit, were measured on the 128-dimension kernel: no faster on the NS rest
        // pass nor on the uniform cloud's — profiles/r03/q8_kernel_variants.md)
        glds8 = FilterLdsBytes((size_t)QT8 * KS8 * 2048, (int)q8_waves);
        GPANN_HIP_TRY(hipFuncSetAttribute(fn_q8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds8));
        GPANN_HIP_TRY(hipFuncSetAttribute(fn_q8_sample, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds8));
        GPANN_HIP_TRY(hipFuncSetAttribute(fn_q8_beside, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds8));
    }
    const size_t rlds = (size_t)kRescoreTasksPerWG * (l2 ? RescoreLdsFloats<GPANN_METRIC_L2>(s->ld) : RescoreLdsFloats<GPANN_METRIC_MIPS>(s->ld)) * sizeof(float);
    const bool wide_rows = s->ld > 128u;   // (RescoreKernel's WIDE instances: further chunks of 32 groups per row)
    const void* fn_rescore = wide_rows ? (l2 ? (const void*)RescoreKernel<GPANN_METRIC_L2, false, true> : (const void*)RescoreKernel<GPANN_METRIC_MIPS, false, true>)
                                       : (l2 ? (const void*)RescoreKernel<GPANN_METRIC_L2, false, false> : (const void*)RescoreKernel<GPANN_METRIC_MIPS, false, false>);
    const void* fn_rescore_ranged = wide_rows ? (l2 ? (const void*)RescoreKernel<GPANN_METRIC_L2, true, true> : (const void*)RescoreKernel<GPANN_METRIC_MIPS, true, true>)
                                              : (l2 ? (const void*)RescoreKernel<GPANN_METRIC_L2, true, false> : (const void*)RescoreKernel<GPANN_METRIC_MIPS, true, false>);
    if (rlds > (48u << 10)) {
        GPANN_HIP_TRY(hipFuncSetAttribute(fn_rescore, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
        GPANN_HIP_TRY(hipFuncSetAttribute(fn_rescore_ranged, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds));
    }
    // re-scoring launches: `parts` wavefronts per task (heavy buckets are split, RescoreArgs::parts)
    // — as many as keep the launch near 128k wavefronts: every (task, part) is a workgroup, most of which find nothing to do and return, and
    // at 40 000 tasks (C3) 8 parts cost more in workgroups started than the split saves (2, 4, 8 parts: 2.28, 2.33, 2.38 ms of scan kernels; NS with
    // 10 000 tasks: 4.44, 4.19, 4.15)
    const uint32_t split_parts = std::max<uint32_t>(1, std::min<uint32_t>(8, s->opt.U32("GPANN_RESCORE_PARTS", std::max<uint32_t>(2, std::min<uint32_t>(8, 131072u / std::max<uint32_t>(T, 1u))))));
    GPANN_TRY(CS(s).ws_split.Reserve((size_t)T * split_parts * k * 8 + 256));
    ra.split_min = std::max<uint32_t>(64, s->opt.U32("GPANN_RESCORE_SPLIT_MIN", 128));
    ra.stat_overflowed = d_overflow;
    ra.stat_candidates = d_cand_total;
    ra.split_dist = CS(s).ws_split.as<float>();
    ra.split_row = reinterpret_cast<uint32_t*>(ra.split_dist + (size_t)T * split_parts * k);
    auto launch_rescore = [&](uint32_t parts) {
        ra.parts = parts;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(0x7FFFFFFFull, (uint64_t)parts * T);
        void* kargs[] = {&ra};
        (void)hipLaunchKernel(fn_rescore, dim3(grid), dim3(64 * kRescoreTasksPerWG), kargs, rlds, st);
        // (the pieces of split buckets are merged — and the pass's statistics kept — by the NEXT launch that reads the lists: PassSetupKernel of
        // the following pass, FinalMergeKernel after the last one)
    };
    bool buckets_pending = false;   // a bucket re-scoring has run and the pieces of its split buckets are not merged yet
    int n_cus = 256;
    (void)hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, s->device);
    // GPANN_SCAN_OVERLAP: the front of this call shares the chip with the rest pass of the call before it (and its own rest pass with the front of
    // the next).  A filter launch takes whole CUs (one 512-thread workgroup with ~all of a CU's registers and most of its LDS), so the two sides are
    // kept apart by their GRID sizes: the front's filter launches start at most front_cus workgroups; the rest pass rest_cus PERSISTENT ones — the
    // product's default, GPANN_OVERLAP_REST_ITEMS=0 — or, GPANN_OVERLAP_REST_ITEMS=n, as many workgroups as its items need, each leaving after n
    // items (CellFilterArgs::wg_item_budget): the rest stream has the lower priority, so a CU it gives up goes to the front when the front has a
    // workgroup waiting and back to the rest pass when it has not.  (Measured no better than the fixed split, profiles/r05/NOTES.md; the budget path is
    // kept covered by tests/test_gpu_filter.py's rest_items cases and tools/overlap_race_probe.py --rest-items.)
    // Defaults (NS sweeps, profiles/r05/NOTES.md): a store whose rest pass streams gigabytes — 160 persistent rest workgroups, 96 for a front's filter
    // launches (the rest pass is HBM-bound: on 160 of 256 CUs it loses a fifth of its speed and the fronts get what they need); a small store (its
    // planes below 4 GB: the step is its launches, not its bytes) — no partition at all, every launch may take the whole chip.
    // ... and no partition either where the rest pass is bound by its PRODUCTS, not by HBM (overlapping clusters, data without structure: every row
    // tile meets many query tiles) — there it needs every CU it can get, and a split costs exactly what it takes away.  Known from the previous
    // completed call's counters: (query tile, row tile) products per row tile of the rest pass (NS 1.7, sigma 0.8 ~8, one blob ~9); unknown: no split.
    const double ppt_max = (double)s->opt.U32("GPANN_OVERLAP_SPLIT_MAX_PRODUCTS", 3);
    const bool big_store = (uint64_t)C.n_tiles * kGTileRows * (use_q8 ? 32u * KS8 + 4u : 32u * KS + 4u) >= (4ull << 30) && s->rest_products_per_tile > 0.0 &&
                           s->rest_products_per_tile <= ppt_max;
    // (a call with nothing else in flight — the first of a stream of batches — has the chip to itself until its own rest pass: no cap on its front)
    bool others_in_flight = false;
    for (int i = 1; i < gpann_shards::kCallSlots; ++i) others_in_flight = others_in_flight || s->cs[(s->cur + i) % gpann_shards::kCallSlots].pending;
    const uint32_t front_cus = overlap && others_in_flight ? std::max<uint32_t>(8, std::min<uint32_t>((uint32_t)n_cus, s->opt.U32("GPANN_OVERLAP_FRONT_CUS", big_store ? 96 : (uint32_t)n_cus))) : (uint32_t)n_cus;
    const uint32_t rest_items_env = overlap ? s->opt.U32("GPANN_OVERLAP_REST_ITEMS", 0) : 0u;
    const uint32_t rest_items = rest_items_env == 0 ? 0u : std::max<uint32_t>(2, rest_items_env);   // (a workgroup's first two claims are made before anything is counted)
    const uint32_t rest_cus = overlap && rest_items == 0 ? std::max<uint32_t>(8, std::min<uint32_t>((uint32_t)n_cus, s->opt.U32("GPANN_OVERLAP_REST_CUS", big_store ? 160 : (uint32_t)n_cus))) : (uint32_t)n_cus;
    fa.wg_item_budget = 0;

    HT_MARK("tables+images");
    bool seed_wrote_U = false;
    if (seed_rows != 0) {
        ra.seed_rows = seed_rows;
        // (under the merge bound only the first slot of a query is seeded: RescoreArgs::seed_first_slot_of)
        ra.seed_first_slot_of = for_merge && nprobe_of_call > 1 && s->opt.U32("GPANN_FILTER_SEED_ALL", 0) == 0 ? nprobe_of_call : 0u;
        ra.seed_catch_up = 0;
        // every task seeded: the seed launch also writes the shard-sample pass's thresholds (one launch less); under the merge bound with several
        // probes only the first slots are seeded and the thresholds need the queries' bounds: ThresholdKernel, below
        const bool seed_sets_U = ra.seed_first_slot_of == 0 && shard_sample_pass && !sitems.empty();
        ra.seed_task_U = seed_sets_U ? task_U : nullptr;
        ra.seed_hi_only = wide ? 1u : 0u;
        launch_rescore(1);   // (seed mode: a fixed number of rows per task)
        ra.seed_task_U = nullptr;
        seed_wrote_U = seed_sets_U;
        if (ra.seed_first_slot_of != 0) {
            ra.seed_catch_up = 1;
            launch_rescore(1);
            ra.seed_catch_up = 0;
        }
        ra.seed_first_slot_of = 0;
        GPANN_HIP_TRY(hipGetLastError());
        ra.seed_rows = 0;
        plan.padded_pairs = (uint64_t)T * seed_rows;
    }
    // ---- the shard-sample pass (with the seed lists): the first rows of every shard (store order: a random sample of the shard; appended
    //      to the cell tiles at store build) go through the matrix cores against the seed thresholds, the candidates are re-scored.
    //      Afterwards every list's k-th distance is at most the k-th of ~4096 random rows of the shard — what bounds the number of rows
    //      the later passes can still find below it (and so the bucket fill) WHATEVER the query's position; the nearest-cell seed alone
    //      gives no such bound for a query whose neighbours live elsewhere (one overflowing bucket costs a whole-shard re-read:
    //      with the head pass run BEFORE this one the NS step took 475 ms instead of 6.5 — same bits).
    //      (Run inside the head pass's launch instead — the tiles share the arrays — it saved no kernel time and let 18 % more
    //      candidates through: the head pass then works with the looser seed threshold.)
    auto debug_candidate_counts = [&](const char* what) {   // GPANN_CELLS_DEBUG: how the candidates of a pass spread over the tasks
        std::vector<uint32_t> h_cnt(T);
        if (hipMemcpy(h_cnt.data(), cand_cnt, (size_t)T * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
        std::sort(h_cnt.begin(), h_cnt.end());
        uint64_t sum = 0;
        for (uint32_t v : h_cnt) sum += v;
        fprintf(stderr, "[gpann cells] %s candidates per task: mean %.1f  median %u  p90 %u  p99 %u  p99.9 %u  max %u\n", what, (double)sum / T, h_cnt[T / 2],
                h_cnt[(size_t)T * 9 / 10], h_cnt[(size_t)T * 99 / 100], h_cnt[(size_t)T * 999 / 1000], h_cnt[T - 1]);
    };
    if (seed_rows != 0 && shard_sample_pass) {
        if (!sitems.empty()) {
            th.hi_only = wide ? 1u : 0u;   // (cand_cnt: zeroed by FilterInitKernel)
            if (!seed_wrote_U) {
                if (l2) hipLaunchKernelGGL(ThresholdKernel<GPANN_METRIC_L2>, dim3((T + 255) / 256), dim3(256), 0, st, th);
                else hipLaunchKernelGGL(ThresholdKernel<GPANN_METRIC_MIPS>, dim3((T + 255) / 256), dim3(256), 0, st, th);
            }
            CellFilterArgs fs = fa;
            fs.items = d_sitems;
            fs.group_first = d_stab;
            fs.group_count = d_stab + G;
            fs.group_cursor = d_stab + 2 * G;
            fs.pass = 0;
            fs.need = nullptr;
            fs.done0 = fs.done0b = fs.done1 = nullptr;
            void* sargs[] = {&fs};
            GPANN_HIP_TRY(hipLaunchKernel(fn, dim3((uint32_t)std::min<size_t>((size_t)front_cus, sitems.size())), dim3(64 * kFWaves), sargs, glds, st));
            launch_rescore(split_parts);
            buckets_pending = true;
            GPANN_HIP_TRY(hipGetLastError());
            if (std::getenv("GPANN_CELLS_DEBUG")) debug_candidate_counts("shard-sample pass");
        }
    }
    for (int pass = head_tiles ? 0 : 1; pass < 3; ++pass) {
        GPANN_REQUIRE(PassSetupLdsBytes(kmax) <= (64u << 10), "too many cells in one shard for the pass set-up kernel");
        HT_MARK(pass == 0 ? "seed+sample" : pass == 1 ? "head" : "sample");
        const uint32_t pass_cus = pass == 2 ? rest_cus : front_cus;
        if (CS(s).filter_passes_timed < 2) GPANN_HIP_TRY(hipEventRecord(CS(s).ev_filter[2 * CS(s).filter_passes_timed], st));
        const bool hi_pass = hi_last && pass >= hi_from_pass;
        // NeedKernel also sets the pass up: thresholds from the current lists, candidate counters and the item cursor back to zero
        // the sample and rest passes (GPANN_FILTER_Q8_FROM_PASS=2: the rest pass only) on the 8-bit plane; their thresholds come from Q8Threshold, not from task_U
        // (NeedKernel writes every (task, cell) slot of the table: a threshold or "nothing passes")
        const bool q8_pass = use_q8 && pass >= q8_from_pass;
        na.hi_only = hi_pass ? 1u : 0u;
        na.head = pass == 0 ? 1u : 0u;
        na.need = need_pass[pass];
        float* const pass_B = q8_pass ? q8_B + (pass == 2 ? q8_B_words : 0) : nullptr;   // (this pass's own table)
        na.q8_B = pass_B;
        fa.q8_B = pass_B;
        fa.need = na.need;
        fa.pass = (uint32_t)pass;
        // (the 8-bit sample pass: one row tile per unit of work by default — GPANN_FILTER_SAMPLE_UNIT=2: the rest pass's kernel, two tiles per unit)
        const bool sample_wide = q8_pass && pass == 1 && s->opt.U32("GPANN_FILTER_SAMPLE_UNIT", 1) >= 2 && q8_unit_tiles >= 2;
        fa.round_tiles = q8_pass ? q8_waves * ((pass == 2 || sample_wide) ? q8_unit_tiles : 1u) : (uint32_t)kFWaves;
        fa.done0 = pass == 0 ? nullptr : need_pass[0];
        fa.done0b = pass == 2 ? need_pass[1] : nullptr;
        fa.done1 = pass == 2 ? need_pass[1] : nullptr;
        {   // one launch sets the pass up: split buckets of the previous pass merged, thresholds, cell tests, work items (PassSetupKernel)
            PassSetupArgs ps;
            ps.need = na;
            ps.filt = fa;
            ps.merge = ra;
            ps.merge.parts = split_parts;
            ps.merge_splits = buckets_pending ? 1u : 0u;
            const size_t slds = PassSetupLdsBytes(kmax);
            if (l2) hipLaunchKernelGGL(PassSetupKernel<GPANN_METRIC_L2>, dim3(kSetupChunks, G), dim3(kSetupThreads), slds, st, ps);
            else hipLaunchKernelGGL(PassSetupKernel<GPANN_METRIC_MIPS>, dim3(kSetupChunks, G), dim3(kSetupThreads), slds, st, ps);
            hipLaunchKernelGGL(PassItemsKernel, dim3(G), dim3(kItemsThreads), 0, st, ps);
            buckets_pending = false;
        }
        if (pass == 2 && overlap) {   // the front ends here (the rest pass's set-up was its last launch): the rest stream carries nothing but the calls'
            GPANN_HIP_TRY(hipEventRecord(CS(s).ev_x[1], st));   // rest-pass launches, back to back
            GPANN_HIP_TRY(hipStreamWaitEvent(st_rest, CS(s).ev_x[1], 0));
            st = st_rest;
        }
        if (pass == 2 && overlap) {   // (fires when the previous call's rest pass has ended and this call's front is done: this rest pass starts)
            GPANN_HIP_TRY(hipEventRecord(CS(s).ev_x[3], st));
            CS(s).rest_begin_recorded = true;
        }
        if (pass == 2) {   // the big launch on its own: row tiles streamed before it (FilterItemsKernel keeps the count, work[3]), and its begin / end
            GPANN_HIP_TRY(hipEventRecord(CS(s).ev_filter[4], st));
            plan.filter_rest_hi_only = hi_pass;
        }
        if (q8_pass) {
            // (measurement aid, WRONG RESULTS: GPANN_DEBUG_Q8_NO_CAND=1 raises every threshold to +inf after the items are built — the launch streams
            // and multiplies the same tiles but finds no candidate: what the candidate hand-off costs)
            if (s->opt.U32("GPANN_DEBUG_Q8_NO_CAND", 0) != 0)
                GPANN_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pass_B), 0x7F800000, (size_t)G * kmax * 32u * QT8, st));
            CellFilterArgs f8 = fa;
            f8.image = q8_image;
            // (the rest pass beside another call's front: workgroups that leave after rest_items items, as many as the item capacity can need)
            // (at least one workgroup: a call whose tasks all sit on shards WITHOUT cells — an empty or a three-row shard probed by a single query — has
            // no work item at all, and a launch of zero workgroups is an invalid configuration; the one workgroup finds every list empty and leaves)
            uint64_t grid8 = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pass_cus, items_cap));
            if (pass == 2 && rest_items != 0) {
                f8.wg_item_budget = rest_items;
                grid8 = std::max<uint64_t>(1, std::min<uint64_t>(items_cap, (items_cap + rest_items - 1) / rest_items + (uint64_t)n_cus));
            }
            void* kargs8[] = {&f8};
            GPANN_HIP_TRY(hipLaunchKernel(pass == 2 && overlap ? fn_q8_beside : (pass == 2 || sample_wide) ? fn_q8 : fn_q8_sample, dim3((uint32_t)grid8), dim3(64 * q8_waves),
                                          kargs8, glds8, st));
            if (pass == 2) plan.filter_rest_row_bytes = 32u * KS8 + 4u;
        } else {
            CellFilterArgs fb = fa;
            uint64_t gridb = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)pass_cus, items_cap));
            if (pass == 2 && rest_items != 0 && hi_pass) {   // (the two-plane kernel has no item budget: persistent workgroups, all CUs — a store without hi-only passes)
                fb.wg_item_budget = rest_items;
                gridb = std::max<uint64_t>(1, std::min<uint64_t>(items_cap, (items_cap + rest_items - 1) / rest_items + (uint64_t)n_cus));
            }
            void* kargsb[] = {&fb};
            GPANN_HIP_TRY(hipLaunchKernel(hi_pass ? fn_hi : fn, dim3((uint32_t)gridb), dim3(64 * kFWaves), kargsb, glds, st));
            if (pass == 2) plan.filter_rest_row_bytes = (hi_pass ? 32u : 64u) * KS + 4u;
        }
        GPANN_HIP_TRY(hipGetLastError());
        if (pass == 2) GPANN_HIP_TRY(hipEventRecord(CS(s).ev_filter[5], st));
        if (pass == 2 && overlap) {   // ... and what follows the launch — re-scoring, final merge, the lists' way out — goes to the tail stream: the NEXT call's rest pass starts at once
            GPANN_HIP_TRY(hipEventRecord(CS(s).ev_x[2], st));
            GPANN_HIP_TRY(hipStreamWaitEvent(st_tail, CS(s).ev_x[2], 0));
            st = st_tail;
            *st_end = st_tail;
        }
        // (the handle times two filter launches: the head pass and, as one interval, the sample and rest passes)
        if (pass == 0 || pass == 2) {
            GPANN_HIP_TRY(hipEventRecord(CS(s).ev_filter[2 * CS(s).filter_passes_timed + 1], st));
            CS(s).filter_passes_timed++;
        }
        // Overflowed buckets are counted on the device (cumulatively over the passes) and looked at once, after the last pass: a
        // handful (the tail of the candidate-count distribution) is re-read exactly inside RescoreKernel; once the count passes
        // the limit RescoreKernel stops doing that — the call is going to fall back to the plain exact scan anyway.
        launch_rescore(split_parts);
        buckets_pending = true;
        GPANN_HIP_TRY(hipGetLastError());
        plan.filter_launches++;
        if (std::getenv("GPANN_CELLS_DEBUG")) {
            uint32_t h_words[4] = {0, 0, 0, 0};
            unsigned long long h_work[2] = {0, 0};
            GPANN_HIP_TRY(hipMemcpyAsync(h_words, d_overflow, 16, hipMemcpyDeviceToHost, st));
            GPANN_HIP_TRY(hipMemcpyAsync(h_work, d_work, 16, hipMemcpyDeviceToHost, st));
            GPANN_HIP_TRY(hipStreamSynchronize(st));
            debug_candidate_counts(pass == 0 ? "head pass" : pass == 1 ? "sample pass" : "rest pass");
            fprintf(stderr, "[gpann cells] pass %d: T=%u G=%u kmax=%u cap=%u  pairs so far %.3e (all rows x tasks %.3e)  row tiles so far %llu  candidates so far %llu  overflowed so far %u\n",
                    pass, T, G, kmax, cap, (double)h_work[0] * 1024.0, (double)plan.pairs, (unsigned long long)h_work[1],
                    (unsigned long long)((uint64_t)h_words[2] | ((uint64_t)h_words[3] << 32)), h_words[0]);
        }
    }
    HT_MARK("rest");
    {   // the dirty tasks: their whole shards exactly, range_segs segments each on a wavefront of its own, merged into their lists
        // (launched whatever the count — no host round trip: without dirty tasks both launches end at once)
        RescoreArgs rr = ra;
        rr.range_segs = range_segs;
        void* kargs[] = {&rr};
        (void)hipLaunchKernel(fn_rescore_ranged, dim3(8192), dim3(64 * kRescoreTasksPerWG), kargs, rlds, st);
        // ... and, in the same launch, the pieces of the last pass's split buckets into their lists (FinalMergeKernel)
        rr.parts = split_parts;
        const uint32_t merge_blocks = buckets_pending ? (T + 4 * kSplitMergeTasksPerWave - 1) / (4 * kSplitMergeTasksPerWave) : 0u;
        hipLaunchKernelGGL(FinalMergeKernel, dim3(merge_blocks + dirty_cap), dim3(256), 0, st, rr, merge_blocks);
        buckets_pending = false;
        GPANN_HIP_TRY(hipGetLastError());
    }
    {
        // the pipeline's counters travel to the call's pinned words; they are looked at when the call is completed (CollectFilterCounters: right
        // after this function in a plain call, in gpann_scan_finish for a deferred one) — nothing here waits for the device
        uint32_t* pin = CS(s).pin_words;
        uint32_t* pin_dev = overlap ? static_cast<uint32_t*>(DeviceAddressOfPinned(pin)) : nullptr;
        if (pin_dev) {   // (GPANN_SCAN_OVERLAP: no runtime copies on the call's streams — KernelCopy)
            GPANN_HIP_TRY(KernelCopy(pin_dev, d_overflow, 16, st));
            GPANN_HIP_TRY(KernelCopy(pin_dev + 4, d_work, 56, st));
        } else {
            GPANN_HIP_TRY(hipMemcpyAsync(pin, d_overflow, 16, hipMemcpyDeviceToHost, st));
            GPANN_HIP_TRY(hipMemcpyAsync(pin + 4, d_work, 56, hipMemcpyDeviceToHost, st));
        }
        GPANN_HIP_TRY(hipEventRecord(CS(s).ev[1], st));
        const size_t n = (size_t)T * k;
        hipLaunchKernelGGL(ListOutputKernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, list_dist, list_row, ta.out, s->permutation, T, k,
                           out.dist, out.ids, out.rows, (const uint32_t*)d_query_tau, (const uint32_t*)ta.query);
        GPANN_HIP_TRY(hipGetLastError());
        CS(s).dirty_cap = dirty_cap;
    }
    return GPANN_OK;
}

// After the call's device work has completed: the filtered pipeline's counters (pin_words) -> the plan's statistics; *overflow = more dirty tasks
// than the pipeline redoes itself: the data defeats the filter, the call has to be redone with the plain exact scan.
static void CollectFilterCounters(gpann_shards::CallState& c, bool* overflow) {
    const uint32_t* h_words = c.pin_words;
    unsigned long long h_work[7];
    std::memcpy(h_work, c.pin_words + 4, 56);
    ScanPlan& plan = c.plan;
    plan.filter_rest_rows = (uint64_t)(h_work[1] - h_work[3]) * kGTileRows;
    plan.filter_candidates = (uint64_t)h_words[2] | ((uint64_t)h_words[3] << 32);
    plan.filter_pairs = (uint64_t)h_work[0] * 1024ull;
    plan.filter_rows = (uint64_t)(h_work[1] + h_work[2]) * kGTileRows;
    plan.filter_two_plane_rows = (uint64_t)h_work[2] * kGTileRows;
    plan.filter_mfma_bf16 = h_work[4];
    plan.filter_mfma_i8 = h_work[5];
    plan.filter_rest_mfma_i8 = h_work[5] - h_work[6];
    plan.filter_dirty_tasks = h_words[1];
    *overflow = h_words[1] > c.dirty_cap;
    if (std::getenv("GPANN_CELLS_DEBUG")) fprintf(stderr, "[gpann cells] buckets overflowed (task x pass) %u, dirty tasks redone exactly %u (cap %u)\n", h_words[0], h_words[1], c.dirty_cap);
}

static bool FilterWanted(const gpann_shards* s, uint32_t k) {
    if (!s->filter_ok || s->cells.n_cells == 0 || s->frag_planes == nullptr || s->d > 256u || k > GPANN_MAX_K) return false;
    const char* mode = s->opt.Get("GPANN_SCAN_MODE");
    if (mode && *mode) return std::string(mode) == "filter";
    return s->scan_mode == 1;
}

// The plain exact scan of the call in progress (CS(s)): every task against its whole shard, lists to `out` — the path of a store / call without
// filter, and the redo of a call the filtered pipeline gave up on.  `ta` = the call's device task arrays (query / out / shard filled or queued in `up`).
static gpann_status RunExactPath(gpann_shards* s, const TaskArrays& ta, UploadBatch& up, const float* d_queries, uint32_t nq, uint32_t nprobe, uint32_t k, int flags,
                                 hipStream_t st, DeviceLists out) {
    ScanPlan& plan = CS(s).plan;
    GPANN_HIP_TRY(up.Flush());
    std::vector<int64_t> lo(s->offsets.begin(), s->offsets.end() - 1), hi(s->offsets.begin() + 1, s->offsets.end());
    plan.padded_pairs = 0;
    GPANN_TRY(BuildExactUnits(s, plan, s->S, lo, hi));
    MergeArgs m{};
    m.out_index = ta.out;
    m.permutation = s->permutation;
    m.out_dist = out.dist;
    m.out_ids = out.ids;
    m.out_rows = out.rows;
    GPANN_TRY(RunExactStage(s, plan, ta, d_queries, k, flags, st, m));
    // GPANN_SCAN_FOR_MERGE in exact mode: the full lists are there — cut them at the query's merge bound, as the filtered pipeline does
    if ((flags & GPANN_SCAN_FOR_MERGE) && nq && nprobe)
        hipLaunchKernelGGL(TruncateForMergeKernel, dim3((nq + 255) / 256), dim3(256), 0, st, out.dist, out.ids, out.rows, nq, nprobe, k);
    GPANN_HIP_TRY(hipGetLastError());
    return GPANN_OK;
}

static gpann_status CompleteCall(gpann_shards* s, int slot, float* elapsed_ms);

// Plans the call and ENQUEUES its device work (tasks + scan + merge; results land in the device arrays `out`): nothing here waits for the
// device.  The call is completed — counters read, an overflowed filter call redone exactly, statistics, timing — by CompleteCall, which the
// entry points run before they return unless the caller asked for GPANN_SCAN_DEFER.
// A new call takes the slot of the call before last (completing it first if it still waits for its gpann_scan_finish) and makes it the call in
// progress: from here on CS(s) — host state AND device workspaces — is this call's.  Every entry point does this before it touches a workspace.
static gpann_status AcquireSlot(gpann_shards* s, hipStream_t st) {
    const int n = (s->cur + 1) % gpann_shards::kCallSlots;   // the oldest slot
    if (s->cs[n].pending) GPANN_TRY(CompleteCall(s, n, nullptr));
    // (rounds 1-4: the two slots shared the device workspaces, and a call on another stream than the most recent one had to wait for it here.
    // The workspaces are per slot now; what is left to order is this slot's OWN previous call — ScanToDevice waits for its ev[3] — and, for
    // callers that pass device buffers of their own to both calls, their own streams.)
    (void)st;
    s->cur = n;
    return GPANN_OK;
}

// GPANN_SCAN_OVERLAP: the handle's two streams (created at the first such call).  Priorities: the front high — short launches on a chain of
// dependencies —, the rest pass low: it takes the compute units the fronts leave.
static gpann_status OverlapStreams(gpann_shards* s) {
    if (s->st_rest) return GPANN_OK;
    int least = 0, greatest = 0;
    GPANN_HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (hipStream_t& st : s->st_front)
        if (!st) GPANN_HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
    if (!s->st_tail) GPANN_HIP_TRY(hipStreamCreateWithPriority(&s->st_tail, hipStreamNonBlocking, greatest));
    if (!s->st_rest) GPANN_HIP_TRY(hipStreamCreateWithPriority(&s->st_rest, hipStreamNonBlocking, least));
    // The runtime sets a stream's queues and copy paths up when they are FIRST used — 5 to 30 ms each, on the host, in the middle of whichever call
    // comes first (measured: an 11 ms D2H enqueue on the tail stream, 5 ms H2D flushes on the front streams, scattered over the first calls of a
    // timed loop).  Touch every stream once here: a copy to the device, a launch, a copy back.
    {
        uint32_t *pin = nullptr, *d_tmp = nullptr;
        GPANN_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&pin), 64, hipHostMallocDefault));
        std::memset(pin, 0, 64);
        GPANN_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_tmp), 256));
        hipStream_t all[gpann_shards::kCallSlots + 2];
        int n = 0;
        for (hipStream_t st : s->st_front) all[n++] = st;
        all[n++] = s->st_rest;
        all[n++] = s->st_tail;
        for (int i = 0; i < n; ++i) {
            (void)hipMemcpyAsync(d_tmp, pin, 64, hipMemcpyHostToDevice, all[i]);
            hipLaunchKernelGGL(FillKernel, dim3(1), dim3(64), 0, all[i], reinterpret_cast<float*>(d_tmp), d_tmp + 16, d_tmp + 32, (size_t)8);
            (void)hipMemcpyAsync(pin, d_tmp, 64, hipMemcpyDeviceToHost, all[i]);
            (void)hipStreamSynchronize(all[i]);
        }
        (void)hipFree(d_tmp);
        (void)hipHostFree(pin);
        (void)hipGetLastError();
    }
    return GPANN_OK;
}

static gpann_status ScanToDevice(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const int32_t* probes,
                                 uint32_t probe_stride, uint32_t nprobe, int flags, hipStream_t st_caller, DeviceLists out) {
    HostTrace local_trace;
    if (!g_host_trace) g_host_trace = &local_trace;   // (gpann_scan_buckets traces the whole entry point; the other entry points from here)
    HostTrace& ht = *g_host_trace;
    ScanPlan& plan = CS(s).plan;
    // GPANN_SCAN_OVERLAP (a deferred call in filter mode whose queries the kernels read in place — the conditions under which the call stays
    // deferred): the call's work goes to the handle's own streams; the caller's stream only orders its start
    hipStream_t st = st_caller, st_rest = st_caller, st_tail = st_caller;
    if ((flags & GPANN_SCAN_OVERLAP) && (flags & GPANN_SCAN_DEFER) && (flags & GPANN_QUERIES_ON_DEVICE) && FilterWanted(s, k) && s->ld == s->d &&
        (reinterpret_cast<uintptr_t>(queries) & 15u) == 0 && OverlapKernelsOwnTheirSimds(s)) {
        GPANN_TRY(OverlapStreams(s));
        GPANN_HIP_TRY(hipEventRecord(CS(s).ev_x[0], st_caller));
        // (a front stream per slot: the fronts of the calls in flight run side by side; 1: one after the other)
        const uint32_t n_front = std::max<uint32_t>(1, std::min<uint32_t>(gpann_shards::kCallSlots, s->opt.U32("GPANN_OVERLAP_FRONT_STREAMS", 3)));
        st = s->st_front[(uint32_t)s->cur % n_front];
        GPANN_HIP_TRY(hipStreamWaitEvent(st, CS(s).ev_x[0], 0));
        // Pacing: this front does not start before the rest pass of the call TWO before it has started (= the call before that has left the rest
        // stream).  Without it three fronts start together after a cold start, finish together and leave the rest stream idle in between — a convoy
        // that takes ten steps to dissolve; with it the fronts are one rest pass apart from the first step on.
        gpann_shards::CallState& two_before = s->cs[(s->cur + 1) % gpann_shards::kCallSlots];
        if (two_before.pending && two_before.rest_begin_recorded && s->opt.U32("GPANN_OVERLAP_PACE", 1) != 0)
            GPANN_HIP_TRY(hipStreamWaitEvent(st, two_before.ev_x[3], 0));
        st_rest = s->st_rest;
        st_tail = s->st_tail;
    }
    ht.Mark("streams");
    // The pinned staging area and the plan's host arrays of this slot are the sources of an earlier call's asynchronous uploads: that call may
    // have returned with them still queued (device output, no timing — nothing made it synchronise), so wait for the end of its work first.
    if (CS(s).stage_in_flight) {
        GPANN_HIP_TRY(hipEventSynchronize(CS(s).ev[3]));
        CS(s).stage_in_flight = false;
    }
    ht.Mark("stage_wait");
    plan.Clear();
    GPANN_TRY(BuildTasks(s, nq, probes, probe_stride, nprobe, plan));
    ht.Mark("tasks");
    const uint32_t T = (uint32_t)plan.task_query.size(), S = s->S;
    // (task arrays, group / tile tables, unit lists: ~32 bytes per task + per-shard tables; exact-mode unit lists fall back to pageable copies if larger)
    CS(s).stage.Reserve(std::min<size_t>((size_t)T * 40 + (size_t)S * 64 + (256u << 10), 256u << 20));
    if (CS(s).ws_stage_dev.Reserve(std::max<size_t>(CS(s).stage.cap, 256)) != GPANN_OK) (void)hipGetLastError();   // (no mirror: every array on its own)
    UploadBatch up{s, st};
    up.by_kernel = st != st_caller;   // (the call runs on the store's own streams)
    const size_t n_out = (size_t)nq * nprobe;

    ht.Mark("stage_reserve");
    const float* d_queries = nullptr;
    GPANN_TRY(StageQueries(s, queries, nq, (flags & GPANN_QUERIES_ON_DEVICE) != 0, st, &d_queries));

    // task arrays: query | out | part | nseg | shard
    GPANN_TRY(CS(s).ws_tasks.Reserve(std::max<size_t>((size_t)T * 5 * sizeof(uint32_t), 256)));
    TaskArrays ta;
    ta.query = CS(s).ws_tasks.as<uint32_t>();
    ta.out = ta.query + T;
    ta.part = ta.out + T;
    ta.nseg = ta.part + T;
    ta.shard = ta.nseg + T;
    if (T) {   // (flushed with the filtered pipeline's tables, or before the exact stage)
        ta.query = up.Add(ta.query, plan.task_query.data(), (size_t)T * 4);
        ta.out = up.Add(ta.out, plan.task_out.data(), (size_t)T * 4);
        ta.shard = up.Add(ta.shard, plan.task_shard.data(), (size_t)T * 4);
    }
    // (the plan's host arrays live in the slot until the call after next, so the async uploads stay valid)

    const bool want_filter = FilterWanted(s, k);
    if (!want_filter) {   // unused slots (skipped probes, shards with < k rows) read as (+inf, NO_ID); the filtered pipeline's FilterInitKernel does it in its own launch
        size_t n = n_out * k;
        int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
        if (n) hipLaunchKernelGGL(FillKernel, dim3(grid), dim3(256), 0, st, out.dist, out.ids, out.rows, n);
    }

    GPANN_HIP_TRY(hipEventRecord(CS(s).ev[0], st));
    GPANN_HIP_TRY(hipEventRecord(CS(s).ev[1], st));   // (re-recorded where the scan kernels end; a call without tasks ends here)
    gpann_shards::CallState& c = CS(s);
    c.filter_passes_timed = 0;
    c.rest_begin_recorded = false;
    c.filtered = false;
    c.overlapped = st_rest != st_caller;   // (the call runs on the store's own streams)
    c.dirty_cap = 0;
    c.d_queries = d_queries;
    c.nq = nq;
    c.k = k;
    c.nprobe = nprobe;
    c.T = T;
    c.flags = flags;
    c.stream = st;
    c.out_dist = out.dist;
    c.out_ids = out.ids;
    c.out_rows = out.rows;
    c.host_dist = nullptr;
    c.host_ids = c.host_rows = nullptr;
    ht.Mark("pre");
    if (want_filter) {
        hipStream_t st_end = st;
        GPANN_TRY(RunCellFilteredScan(s, plan, ta, up, d_queries, nq, nprobe, k, flags, st, st_rest, st_tail, &st_end, out));
        c.filtered = T != 0;
        st = st_end;
        c.stream = st_end;   // (what follows the call — the copies to the host, a redo, the caller's own merge — is queued behind its LAST launch)
    } else {
        GPANN_TRY(RunExactPath(s, ta, up, d_queries, nq, nprobe, k, flags, st, out));
    }
    ht.Mark("pipeline");
    GPANN_HIP_TRY(hipEventRecord(c.ev[2], st));
    GPANN_HIP_TRY(hipEventRecord(c.ev[3], st));   // the end of the device work of this call so far (uploads AND kernels: a later call on this slot waits for it; an entry point that enqueues more records it again)
    ht.Mark("end_events");
    c.stage_in_flight = true;
    c.pending = true;
    return GPANN_OK;
}

static void FillStats(gpann_shards* s, const gpann_shards::CallState& c, bool filtered) {
    const ScanPlan& plan = c.plan;
    s->stats = gpann_scan_stats{};
    s->stats.filtered = filtered ? 1 : 0;
    s->stats.overlapped = filtered && c.overlapped ? 1u : 0u;
    s->stats.pairs = plan.pairs;
    s->stats.algorithmic_bytes = plan.pairs * s->d * 4ull + (uint64_t)c.T * c.k * 8ull;
    s->stats.tasks = c.T;
    s->stats.units = plan.units.size();
    s->stats.padded_pairs = plan.padded_pairs;
    s->stats.filter_pairs = plan.filter_pairs;
    s->stats.filter_rows = plan.filter_rows;
    s->stats.filter_rest_rows = plan.filter_rest_rows;
    s->stats.filter_rest_hi_only = plan.filter_rest_hi_only ? 1u : 0u;
    s->stats.filter_candidates = plan.filter_candidates;
    s->stats.filter_plane_ksteps = filtered ? s->cells.plane_ks : 0u;
    s->stats.filter_two_plane_rows = plan.filter_two_plane_rows;
    s->stats.filter_mfma_bf16 = plan.filter_mfma_bf16;
    s->stats.filter_mfma_i8 = plan.filter_mfma_i8;
    s->stats.filter_rest_mfma_i8 = plan.filter_rest_mfma_i8;
    s->stats.filter_dirty_tasks = plan.filter_dirty_tasks;
    s->stats.filter_rest_row_bytes = filtered ? plan.filter_rest_row_bytes : 0u;
    s->stats.filter_seed_rows = filtered ? plan.padded_pairs : 0;
}

static gpann_status FinishTiming(gpann_shards* s, gpann_shards::CallState& c, float* elapsed_ms) {
    GPANN_HIP_TRY(hipEventSynchronize(c.ev[2]));
    float a = 0.f, b = 0.f;
    GPANN_HIP_TRY(hipEventElapsedTime(&a, c.ev[0], c.ev[1]));
    GPANN_HIP_TRY(hipEventElapsedTime(&b, c.ev[1], c.ev[2]));
    s->stats.scan_ms = a;
    s->stats.merge_ms = b;
    s->stats.filter_ms = 0.f;
    if (s->stats.filtered) {
        for (int p = 0; p < c.filter_passes_timed; ++p) {
            float f = 0.f;
            GPANN_HIP_TRY(hipEventElapsedTime(&f, c.ev_filter[2 * p], c.ev_filter[2 * p + 1]));
            s->stats.filter_ms += f;
        }
        s->stats.filter_rest_ms = 0.f;
        if (c.filter_passes_timed >= 1) GPANN_HIP_TRY(hipEventElapsedTime(&s->stats.filter_rest_ms, c.ev_filter[4], c.ev_filter[5]));
    }
    if (elapsed_ms) *elapsed_ms = a + b;
    return GPANN_OK;
}

// host destinations of a call without GPANN_OUTPUT_ON_DEVICE: the copies are queued behind the call's kernels (and again after a redo)
static gpann_status EnqueueHostCopies(gpann_shards* s, gpann_shards::CallState& c) {
    hipStream_t st = static_cast<hipStream_t>(c.stream);
    const size_t n = (size_t)c.nq * c.nprobe * c.k;
    // (GPANN_SCAN_OVERLAP and pinned, device-accessible destinations: by kernel — see KernelCopy; anything else through the runtime)
    auto to_host = [&](void* host, const void* dev) -> gpann_status {
        if (!host) return GPANN_OK;
        void* host_dev = (c.flags & GPANN_SCAN_OVERLAP) && st != nullptr && (st == s->st_tail) ? DeviceAddressOfPinned(host) : nullptr;
        if (host_dev) GPANN_HIP_TRY(KernelCopy(host_dev, dev, n * 4, st));
        else GPANN_HIP_TRY(hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, st));
        return GPANN_OK;
    };
    GPANN_TRY(to_host(c.host_ids, c.out_ids));
    GPANN_TRY(to_host(c.host_dist, c.out_dist));
    GPANN_TRY(to_host(c.host_rows, c.out_rows));
    GPANN_HIP_TRY(hipEventRecord(c.ev[3], st));
    return GPANN_OK;
}

// Completes the call of `slot`: waits for its device work, reads the filtered pipeline's counters, redoes the call with the plain exact scan
// if the data defeated the filter (more dirty tasks than the pipeline redoes itself), fills the statistics and the timing.
static gpann_status CompleteCall(gpann_shards* s, int slot, float* elapsed_ms) {
    gpann_shards::CallState& c = s->cs[slot];
    if (elapsed_ms) *elapsed_ms = 0.f;
    if (!c.pending) return GPANN_OK;
    c.pending = false;
    GPANN_HIP_TRY(hipEventSynchronize(c.ev[3]));
    c.stage_in_flight = false;
    bool overflow = false;
    if (c.filtered) CollectFilterCounters(c, &overflow);
    if (c.filtered && !overflow && c.plan.pairs != 0)   // (what the next call's small-shard rule goes by: RunCellFilteredScan)
        s->filter_pair_share = (double)c.plan.filter_pairs / (double)c.plan.pairs;
    if (c.filtered && !overflow && c.plan.filter_rest_rows != 0)   // (what the next overlapped call's CU split goes by: RunCellFilteredScan)
        s->rest_products_per_tile = ((double)c.plan.filter_pairs / 1024.0) / ((double)c.plan.filter_rest_rows / (double)kGTileRows);
    if (overflow) {
        // the redo runs as "the call in progress" of its own slot (its plan's host arrays are intact, its pinned stage is free again); a later
        // call that is already queued on the stream simply runs before it
        const int keep = s->cur;
        s->cur = slot;
        hipStream_t st = static_cast<hipStream_t>(c.stream);
        ScanPlan& plan = c.plan;
        const uint32_t T = c.T;
        c.stage.used = 0;
        UploadBatch up{s, st};
        gpann_status rc = CS(s).ws_tasks.Reserve(std::max<size_t>((size_t)T * 5 * sizeof(uint32_t), 256));
        if (rc == GPANN_OK) {
            TaskArrays ta;
            ta.query = CS(s).ws_tasks.as<uint32_t>();
            ta.out = ta.query + T;
            ta.part = ta.out + T;
            ta.nseg = ta.part + T;
            ta.shard = ta.nseg + T;
            ta.query = up.Add(ta.query, plan.task_query.data(), (size_t)T * 4);
            ta.out = up.Add(ta.out, plan.task_out.data(), (size_t)T * 4);
            ta.shard = up.Add(ta.shard, plan.task_shard.data(), (size_t)T * 4);
            if (c.host_ids) {   // a call with host outputs: its device lists live in ws_out, which a later call may have re-grown (same layout)
                const size_t n = (size_t)c.nq * c.nprobe * c.k;
                c.out_dist = CS(s).ws_out.as<float>();
                c.out_ids = reinterpret_cast<uint32_t*>(c.out_dist + n);
                c.out_rows = c.out_ids + n;
            }
            DeviceLists out;
            out.dist = c.out_dist;
            out.ids = c.out_ids;
            out.rows = c.out_rows;
            rc = RunExactPath(s, ta, up, c.d_queries, c.nq, c.nprobe, c.k, c.flags, st, out);
        }
        if (rc == GPANN_OK && hipEventRecord(c.ev[2], st) != hipSuccess) rc = GPANN_ERR_HIP;
        if (rc == GPANN_OK) rc = EnqueueHostCopies(s, c);
        if (rc == GPANN_OK && hipStreamSynchronize(st) != hipSuccess) rc = GPANN_ERR_HIP;
        s->cur = keep;
        if (rc != GPANN_OK) return rc;
    }
    FillStats(s, c, c.filtered && !overflow);
    return FinishTiming(s, c, elapsed_ms);
}

static gpann_status CheckScanArgs(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const int32_t* probes, uint32_t nprobe,
                                  uint32_t probe_stride) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    GPANN_REQUIRE(k >= 1, "k must be >= 1");
    if (k > GPANN_MAX_K) {
        SetLastError("k = %u exceeds GPANN_MAX_K = %u", k, GPANN_MAX_K);
        return GPANN_ERR_UNSUPPORTED;
    }
    GPANN_REQUIRE(nq == 0 || queries != nullptr, "queries is null");
    GPANN_REQUIRE(nq == 0 || nprobe == 0 || probes != nullptr, "probes is null");
    GPANN_REQUIRE(probe_stride >= nprobe, "probe_stride < nprobe");
    GPANN_REQUIRE((uint64_t)nq * nprobe < 0xFFFFFFFFull, "nq * nprobe must fit 32 bits");
    return EnsureDevice(s->device);
}

}  // namespace gpann

using namespace gpann;

extern "C" {

gpann_status gpann_scan_buckets(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const int32_t* probes, uint32_t probe_stride,
                                uint32_t nprobe, uint32_t* out_ids, float* out_dist, uint32_t* out_rows, int flags, void* stream,
                                float* elapsed_ms) {
    HostTrace entry_trace;
    g_host_trace = &entry_trace;
    GPANN_TRY(CheckScanArgs(s, queries, nq, k, probes, nprobe, probe_stride));
    HT_MARK("args+device");
    std::lock_guard<std::recursive_mutex> turn(s->call_mutex);   // (concurrent calls on one handle take turns: shards.h)
    HT_MARK("lock");
    GPANN_REQUIRE(out_ids != nullptr, "out_ids is null");
    GPANN_REQUIRE(!(flags & GPANN_SCAN_FOR_MERGE) || !(flags & GPANN_OUTPUT_ON_DEVICE) || out_dist != nullptr, "GPANN_SCAN_FOR_MERGE with device output needs out_dist");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)nq * nprobe * k;
    if (n == 0) {
        if (elapsed_ms) *elapsed_ms = 0.f;
        return GPANN_OK;
    }
    GPANN_TRY(AcquireSlot(s, st));
    HT_MARK("slot");
    DeviceLists dl;
    const bool dev_out = (flags & GPANN_OUTPUT_ON_DEVICE) != 0;
    if (dev_out) {
        dl.ids = out_ids;
        dl.dist = out_dist;
        dl.rows = out_rows;
    } else {
        GPANN_TRY(CS(s).ws_out.Reserve(n * 12));
        dl.dist = CS(s).ws_out.as<float>();
        dl.ids = reinterpret_cast<uint32_t*>(dl.dist + n);
        dl.rows = dl.ids + n;
    }
    GPANN_TRY(ScanToDevice(s, queries, nq, k, probes, probe_stride, nprobe, flags, st, dl));
    gpann_shards::CallState& c = CS(s);
    if (!dev_out) {   // the lists travel to the caller's host arrays behind the kernels
        c.host_ids = out_ids;
        c.host_dist = out_dist;
        c.host_rows = out_rows;
        GPANN_TRY(EnqueueHostCopies(s, c));
        HT_MARK("host_copies");
    }
    // GPANN_SCAN_DEFER: the call is completed by gpann_scan_finish — provided a redo could still find its inputs: device queries read in place
    if ((flags & GPANN_SCAN_DEFER) && c.d_queries == queries) return GPANN_OK;
    // a plain exact call with device output and no timing stays asynchronous, as documented (nothing to collect)
    if (dev_out && !elapsed_ms && !c.filtered) {
        c.pending = false;
        FillStats(s, c, false);
        return GPANN_OK;
    }
    return CompleteCall(s, s->cur, elapsed_ms);
}

gpann_status gpann_scan_finish(gpann_shards* s, float* elapsed_ms) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    std::lock_guard<std::recursive_mutex> turn(s->call_mutex);
    GPANN_TRY(EnsureDevice(s->device));
    if (elapsed_ms) *elapsed_ms = 0.f;
    for (int age = gpann_shards::kCallSlots - 1; age > 0; --age) {   // the oldest deferred call first
        const int slot = (s->cur + gpann_shards::kCallSlots - age) % gpann_shards::kCallSlots;
        if (s->cs[slot].pending) return CompleteCall(s, slot, elapsed_ms);
    }
    return CompleteCall(s, s->cur, elapsed_ms);
}

gpann_status gpann_scan_merged(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const int32_t* probes, uint32_t probe_stride,
                               uint32_t nprobe, uint32_t* out_ids, float* out_dist, int flags, void* stream, float* elapsed_ms) {
    GPANN_TRY(CheckScanArgs(s, queries, nq, k, probes, nprobe, probe_stride));
    std::lock_guard<std::recursive_mutex> turn(s->call_mutex);
    GPANN_REQUIRE(out_ids != nullptr, "out_ids is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n_slot = (size_t)nq * nprobe * k, n = (size_t)nq * k;
    if (n == 0) {
        if (elapsed_ms) *elapsed_ms = 0.f;
        return GPANN_OK;
    }
    // per-slot lists (dist,rows) in workspace, then one more merge across the slots of a query
    GPANN_TRY(AcquireSlot(s, st));
    if (CS(s).stage_in_flight) {   // (this slot's previous call may still read ws_out: an untimed device-output merge)
        GPANN_HIP_TRY(hipEventSynchronize(CS(s).ev[3]));
        CS(s).stage_in_flight = false;
    }
    GPANN_TRY(CS(s).ws_out.Reserve(n_slot * 8 + n * 8 + 256));
    DeviceLists slot;
    slot.dist = CS(s).ws_out.as<float>();
    slot.rows = reinterpret_cast<uint32_t*>(slot.dist + n_slot);
    slot.ids = nullptr;
    float* d_out_dist = reinterpret_cast<float*>(slot.rows + n_slot);
    uint32_t* d_out_ids = reinterpret_cast<uint32_t*>(d_out_dist + n);
    const bool dev_out = (flags & GPANN_OUTPUT_ON_DEVICE) != 0;
    if (nprobe == 0) {
        hipLaunchKernelGGL(FillKernel, dim3(64), dim3(256), 0, st, dev_out ? out_dist : d_out_dist, dev_out ? out_ids : d_out_ids,
                           (uint32_t*)nullptr, n);
    } else {
        // (one TopN over the slots: entries beyond the smallest k-th distance of the query's slots cannot be in it — GPANN_SCAN_FOR_MERGE)
        GPANN_TRY(ScanToDevice(s, queries, nq, k, probes, probe_stride, nprobe, (flags & ~GPANN_SCAN_DEFER) | GPANN_SCAN_FOR_MERGE, st, slot));
        float scan_ms = 0.f;
        GPANN_TRY(CompleteCall(s, s->cur, &scan_ms));   // (the slot lists are final — an overflowed filter call has been redone — before they are merged)
        MergeArgs m{};
        m.in_dist = slot.dist;
        m.in_row = slot.rows;
        m.permutation = s->permutation;
        m.out_dist = dev_out ? out_dist : d_out_dist;
        m.out_ids = dev_out ? out_ids : d_out_ids;
        m.out_rows = nullptr;
        m.n_out = nq;
        m.k = k;
        m.lists_per_out = nprobe;
        hipLaunchKernelGGL(MergeListsKernel<1>, dim3((nq + 3) / 4), dim3(256), 0, st, m);
        GPANN_HIP_TRY(hipGetLastError());
        GPANN_HIP_TRY(hipEventRecord(CS(s).ev[2], st));
        // ev[3] = the end of ALL device work of this call on the handle (the merge reads the slot lists in ws_out: a following call on this
        // slot waits for it before it touches the stage; the NEXT call on the handle — another thread, another stream — must not overwrite
        // ws_out under the merge either: it is made to wait below when this call does not synchronise itself)
        GPANN_HIP_TRY(hipEventRecord(CS(s).ev[3], st));
        CS(s).stage_in_flight = true;
    }
    if (!dev_out) {
        GPANN_HIP_TRY(hipMemcpyAsync(out_ids, d_out_ids, n * 4, hipMemcpyDeviceToHost, st));
        if (out_dist) GPANN_HIP_TRY(hipMemcpyAsync(out_dist, d_out_dist, n * 4, hipMemcpyDeviceToHost, st));
        GPANN_HIP_TRY(hipStreamSynchronize(st));
    }
    if (nprobe && (!dev_out || elapsed_ms)) GPANN_TRY(FinishTiming(s, CS(s), elapsed_ms));
    else if (nprobe) GPANN_HIP_TRY(hipEventSynchronize(CS(s).ev[3]));   // (device output, untimed: the merge kernel still reads ws_out — the next call on the handle may not start under it)
    return GPANN_OK;
}

gpann_status gpann_shard_search_hits(gpann_shards* s, const float* queries, uint32_t nq, uint32_t k, const float* distance_to_kth,
                                     uint32_t* out_ids, uint32_t* out_count, int flags, void* stream, float* elapsed_ms) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    std::lock_guard<std::recursive_mutex> turn(s->call_mutex);
    const uint32_t S = s->S;
    std::vector<int32_t> probes((size_t)nq * S);
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t b = 0; b < S; ++b) probes[(size_t)q * S + b] = (int32_t)b;
    GPANN_TRY(CheckScanArgs(s, queries, nq, k, probes.data(), S, S));
    GPANN_REQUIRE(out_ids != nullptr && out_count != nullptr && distance_to_kth != nullptr, "null output / threshold pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)nq * S * k;
    if (n == 0) return GPANN_OK;
    // workspace: slot dist | slot ids | hits ids | hits count | kth
    GPANN_TRY(AcquireSlot(s, st));
    if (CS(s).stage_in_flight) {
        GPANN_HIP_TRY(hipEventSynchronize(CS(s).ev[3]));
        CS(s).stage_in_flight = false;
    }
    GPANN_TRY(CS(s).ws_out.Reserve(n * 12 + (size_t)nq * S * 4 + (size_t)nq * 4 + 256));
    DeviceLists slot;
    slot.dist = CS(s).ws_out.as<float>();
    slot.ids = reinterpret_cast<uint32_t*>(slot.dist + n);
    slot.rows = nullptr;
    uint32_t* d_hits = slot.ids + n;
    uint32_t* d_count = d_hits + n;
    float* d_kth = reinterpret_cast<float*>(d_count + (size_t)nq * S);
    GPANN_HIP_TRY(hipMemcpyAsync(d_kth, distance_to_kth, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    GPANN_TRY(ScanToDevice(s, queries, nq, k, probes.data(), S, S, flags & (GPANN_QUERIES_ON_DEVICE | GPANN_AUX_PASS), st, slot));
    GPANN_TRY(CompleteCall(s, s->cur, nullptr));
    const size_t tt = (size_t)nq * S;
    hipLaunchKernelGGL(ShardHitsKernel, dim3((uint32_t)((tt + 255) / 256)), dim3(256), 0, st, slot.dist, slot.ids, d_kth, nq, S, k, d_hits, d_count);
    GPANN_HIP_TRY(hipGetLastError());
    GPANN_HIP_TRY(hipMemcpyAsync(out_ids, d_hits, n * 4, hipMemcpyDeviceToHost, st));
    GPANN_HIP_TRY(hipMemcpyAsync(out_count, d_count, tt * 4, hipMemcpyDeviceToHost, st));
    GPANN_HIP_TRY(hipStreamSynchronize(st));
    GPANN_TRY(FinishTiming(s, CS(s), elapsed_ms));
    return GPANN_OK;
}

gpann_status gpann_ground_truth(gpann_shards* s_all, const float* points, uint64_t n, uint32_t d, int metric, const float* queries,
                                uint32_t nq, uint32_t k, uint32_t* out_ids, float* out_dist, int device, int flags, void* stream,
                                float* elapsed_ms) {
    gpann_shards* s = s_all;
    bool temp = false;
    if (!s) {
        GPANN_REQUIRE(points != nullptr, "points is null and no store given");
        GPANN_REQUIRE(n < 0xFFFFFFFFull, "n must fit 32 bits");
        std::vector<uint32_t> ids(n);
        for (uint64_t i = 0; i < n; ++i) ids[i] = (uint32_t)i;
        uint64_t offs[2] = {0, n};
        GPANN_TRY(gpann_shards_create(points, n, d, metric, ids.data(), offs, 1, device, flags & GPANN_POINTS_ON_DEVICE, &s));
        temp = true;
    } else {
        GPANN_REQUIRE(s->S == 1, "ground truth needs a 1-shard store over all points");
        // (rows in another order would break ties by ROW where ComputeGroundTruth, recall.h:25-39, breaks them by point id)
        GPANN_REQUIRE(s->ids_in_order, "ground truth needs a store whose row i is point i (ids 0 .. n-1 in order)");
    }
    std::vector<int32_t> probes(nq, 0);
    gpann_status st = gpann_scan_buckets(s, queries, nq, k, probes.data(), 1, 1, out_ids, out_dist, nullptr,
                                         (flags & (GPANN_QUERIES_ON_DEVICE | GPANN_OUTPUT_ON_DEVICE)) | GPANN_AUX_PASS, stream, elapsed_ms);
    if (temp) {
        if (st == GPANN_OK && (flags & GPANN_OUTPUT_ON_DEVICE)) (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
        gpann_shards_destroy(s);
    }
    return st;
}

gpann_status gpann_scan_last_stats(const gpann_shards* s, gpann_scan_stats* out) {
    GPANN_REQUIRE(s != nullptr && out != nullptr, "null argument");
    *out = s->stats;
    return GPANN_OK;
}

// (the struct only ever grows at its end: a caller compiled against an older or newer header passes ITS sizeof and gets the fields both sides know)
gpann_status gpann_scan_last_stats_sized(const gpann_shards* s, void* out, size_t out_size) {
    GPANN_REQUIRE(s != nullptr && out != nullptr, "null argument");
    std::memset(out, 0, out_size);
    std::memcpy(out, &s->stats, std::min(out_size, sizeof(gpann_scan_stats)));
    return GPANN_OK;
}

}  // extern "C"
