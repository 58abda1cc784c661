// scan_filter.h — shared pieces of the filtered exact shard scan (included by scan.hip and cells.hip): the bf16 split, the
// plane layout constants, the threshold of a filter pass and the exact re-scoring of its candidates.  The pipeline and the MFMA
// kernel are in scan_cells.h.
//
// The exact scan is bound by FP32 VALU issue: 3 unfused lane-ops per (pair, dimension).  Only the k best rows of a shard
// matter, so almost every one of those exact evaluations is spent on rows that are nowhere near the current k-th distance.
// The filter replaces them by a matrix-core bound:
//   * per task an exact list with k-th distance tau exists from the rows evaluated so far;
//   * v_mfma_f32_32x32x16_bf16 evaluates   acc = q.p - |p|^2/2  (L2)   /   acc = q.p  (MIPS)   with q and p each split into two bf16
//     planes (hi + lo; products hi*hi, hi*lo, lo*hi, f32 accumulate);
//   * a row is kept as a CANDIDATE iff acc >= U[task], where U is derived from tau and a rigorous bound on
//     |acc - true value| + |canonical float distance - true value| (ThresholdKernel) — every row whose canonical distance is
//     <= tau passes, plus a few percent of near misses;
//   * RescoreKernel evaluates the canonical binary32 distance (the reference's exact sequence) of the candidates only and
//     inserts them into the task's exact top-k list.
// The result lists are the exact lists: the filter only decides which rows get the exact evaluation and never drops a row that
// could belong to the top-k.  If the fixed-size candidate buckets overflow widely (adversarial inputs: masses of equal
// distances) the caller falls back to the plain exact scan for the whole call.
#pragma once

#include "shards.h"

#include <type_traits>

namespace gpann {


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kFKS = 8;                   // k-steps of 16 dims held in registers (d <= 128)

// two floats -> packed bf16 hi pair and bf16 lo pair (round to nearest even; x - float(hi) is exact in binary32)
__device__ __forceinline__ void SplitBf16(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    f32x2 x = {x0, x1};
    bf16x2 h = __builtin_convertvector(x, bf16x2);
    f32x2 back = __builtin_convertvector(h, f32x2);
    f32x2 rest = x - back;
    bf16x2 l = __builtin_convertvector(rest, bf16x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

union Frag {   // 8 bf16 = 4 VGPRs, the A/B operand of v_mfma_f32_32x32x16_bf16
    bf16x8 v;
    uint32_t u[4];
    uint4 q;
};

// The points' bf16 planes live in HBM in MFMA A-operand order (scan_cells.h): per 32-row tile and k-step one contiguous KiB per
// plane (lane (c, h) holds row 32 T + c, dims 16 ks + 8 h .. + 7), 8 KiB per tile and plane.  The shard store keeps all hi
// tiles in one array and all lo tiles in a second one (the big passes stream only the hi array, sequentially); the router's
// centroid planes keep hi and lo of a tile side by side (kGTileBytes).
constexpr int kGTileRows = 32;
// k-steps (16 dimensions each) a row of d dimensions is stored with: ceil(d / 16), rounded up to the next count FilterKernelCells /
// QueryImageKernel are instantiated for.  A plane row is 32 * k-steps bytes: d = 96 -> 6 (192 B), 100 -> 7, 128 -> 8, 200 -> 13.
// (Rounds 1-2 knew 4 / 8 / 16 only: every d between paid for the zeros up to the next power of two in HBM bytes and MFMA issue.)
__host__ __device__ constexpr uint32_t PlaneKSteps(uint32_t d) {
    const uint32_t need = (d + 15u) / 16u;
    return need <= 2 ? 2u : need <= 4 ? 4u : need <= 6 ? 6u : need <= 7 ? 7u : need <= 8 ? 8u : need <= 10 ? 10u : need <= 13 ? 13u : 16u;
}
// The front of a cell: its first max(16, tiles / 8 + 1) row tiles, in slab order (a random sample of the cell).  The head pass (512 rows) and the
// sample pass (at most 1 / 8 of a cell by default) stay inside it; behind it the rows are ordered by norm word (cells.hip, RowsByNorm) and,
// when the store has the 8-bit plane, keep no bf16 tile.
constexpr uint32_t kFilterFrontDiv = 8;
__host__ __device__ constexpr uint64_t FilterFrontTiles(uint64_t tiles) {
    return (tiles + kFilterFrontDiv - 1) / kFilterFrontDiv + 1 > 16 ? (tiles + kFilterFrontDiv - 1) / kFilterFrontDiv + 1 : 16;
}
constexpr int kGPlaneBytes = kFKS * 1024;      // one plane of a tile: [ks][lane][16 B]
constexpr int kGTileBytes = 2 * kGPlaneBytes;  // [plane hi|lo][ks][lane][16 B]: 16 KiB per 32 rows (512 B per row)
constexpr int kGTasks = 256;                   // tasks per filter group: 8 query tiles of 32

// One 32-query tile of the LDS query image (MFMA B-operand order: [ks][plane hi|lo][lane][16 B], KS * 2 KiB), built by one
// wavefront: lane (c, h) converts dims 16 ks + 8 h .. + 7 of query row `qrow` (its row c of the tile; zeros if !real).  All
// global loads are issued before the first conversion: per-k-step load -> convert -> store chains cost ~2 us each and made the
// image (rebuilt for every work item) the dominant cost of small items.
// `mu` (optional, [ld], pad entries zero): the centre the store's planes were built around (squared L2 is translation invariant: both
// sides of the bound live in the frame p - mu, q - mu, whose norms — and with them the bound's error terms — are as small as the data allows)
template <int KS>
__device__ __forceinline__ void BuildQueryTile(char* tile_lds, const float* qrow, bool real, uint32_t ld, int lane, const float* mu = nullptr) {
    const int h = lane >> 5;
    float4 x0[KS], x1[KS];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const uint32_t k0 = 16u * ks + 8u * h;
        // (addresses are clamped into the row, the values masked afterwards: no branch between the loads)
        float4 a0 = *reinterpret_cast<const float4*>(qrow + min(k0, ld - 4));
        float4 a1 = *reinterpret_cast<const float4*>(qrow + min(k0 + 4, ld - 4));
        if (mu) {
            const float4 m0 = *reinterpret_cast<const float4*>(mu + min(k0, ld - 4));
            const float4 m1 = *reinterpret_cast<const float4*>(mu + min(k0 + 4, ld - 4));
            a0 = make_float4(a0.x - m0.x, a0.y - m0.y, a0.z - m0.z, a0.w - m0.w);
            a1 = make_float4(a1.x - m1.x, a1.y - m1.y, a1.z - m1.z, a1.w - m1.w);
        }
        x0[ks] = (real && k0 < ld) ? a0 : zero4;
        x1[ks] = (real && k0 + 4 < ld) ? a1 : zero4;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        Frag fh, fl;
        SplitBf16(x0[ks].x, x0[ks].y, fh.u[0], fl.u[0]);
        SplitBf16(x0[ks].z, x0[ks].w, fh.u[1], fl.u[1]);
        SplitBf16(x1[ks].x, x1[ks].y, fh.u[2], fl.u[2]);
        SplitBf16(x1[ks].z, x1[ks].w, fh.u[3], fl.u[3]);
        *reinterpret_cast<uint4*>(tile_lds + (ks * 2 + 0) * 1024 + lane * 16) = fh.q;
        *reinterpret_cast<uint4*>(tile_lds + (ks * 2 + 1) * 1024 + lane * 16) = fl.q;
    }
}

// ---- thresholds ------------------------------------------------------------------------------------------------------
// For task t with exact k-th distance tau (its list's last entry; +inf while the list is not full):  U[t] such that every
// row whose CANONICAL binary32 distance E satisfies E <= tau has acc >= U.  All in double, rounded towards -inf.
//
// Notation: u = 2^-24, D = exact real squared distance, R = max |p| over the shard, nq = |q|^2.
//  L2:   E <= tau  =>  D <= tau (1 + 2 g1),  g1 = (d + 32) 2^-23           (rounding of the reference's sum of squares)
//        true acc* = q.p - |p|^2/2 = -(D - nq)/2
//        |acc - acc*| <= e2/2, e2 = c2 (R + |q|)^2, c2 = 2 * 2^-16 + 8 (d + 16) 2^-24 + 2^-20
//              2 * 2^-16: the three dropped bf16 split terms (lo*lo and the two residuals, each <= 2^-16 |q_i p_i| with RNE
//                         splits): 2 * 3 * 2^-16 |q||p| <= 1.5 * 2^-16 (R + |q|)^2
//              8 (d+16) u: accumulation of the 3 d products and the norm term, any order, rounded OR truncated adds
//                         (2 u per add), over sum |terms| <= (R + |q|)^2 / 2, times 2
//              2^-20: rounding of the stored norm and slack
//        U = -( tau (1 + 2 g1) - nq (1 - 2^-40) + e2 ) / 2
//  MIPS: E = fl(1 - fl(q.p));  E <= tau  =>  q.p >= 1 - tau - e1,  e1 = (d + 4) 2^-23 (|q| R + 1) + |tau| 2^-22
//        |acc - q.p| <= e2' = (4 * 2^-16 + 8 (d + 16) 2^-24) |q| R
//        U = 1 - tau - e1 - e2'
struct ThresholdArgs {
    const float* list_dist;       // [tasks][k]
    const float* queries;         // [nq][ld]
    const uint32_t* task_query;
    const uint32_t* task_shard;
    const double* shard_rmax;     // [S]
    float* task_U;
    uint32_t tasks, k, d, ld;
    uint32_t hi_only;             // the filter pass reads only the hi plane of the points: the dropped q.p_lo term joins the error bound
    const double* task_nq;        // optional [tasks]: |q|^2 of every task (NearestCellKernel), indexed through task_src; else summed here
    const uint32_t* task_src;
    const uint32_t* query_tau;    // optional [nq] (GPANN_SCAN_FOR_MERGE): OrderedKey of the query's merge bound, which caps every task's tau
};

// nq = |q|^2 (double, summed in the query's dimension order), R = max |p| over the shard
template <int METRIC>
__device__ __forceinline__ float ThresholdOf(double tau, double nq, double R, double dd, bool hi_only) {
    if (!(tau < (double)INFINITY)) return -INFINITY;   // list not full yet (or NaN): everything is a candidate
    const double qn = sqrt(nq) * (1.0 + 1e-12);
    double U;
    if (METRIC == GPANN_METRIC_L2) {
        const double g1 = (dd + 32.0) * 0x1p-23;
        // (+ 2^-21: the rounding of the centred coordinates fl(p - mu), fl(q - mu): |D_centred - D| <= 2^-22 (R + |q|)^2)
        const double c2 = 2.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24 + 0x1p-20 + 0x1p-21;
        // hi plane only: |(q_h + q_l).(p_l + r_p) + r_q.p| <= (2^-9 * 1.01 + 2^-17) |q||p| of the dot product is missing, twice that in D
        const double e_hi = hi_only ? 2.0 * 1.02 * 0x1p-9 * qn * R : 0.0;
        const double e2 = c2 * (R + qn) * (R + qn) + e_hi + 1e-30;
        U = -(tau * (1.0 + 2.0 * g1) - nq * (1.0 - 0x1p-40) + e2) * 0.5;
    } else {
        const double e1 = (dd + 4.0) * 0x1p-23 * (qn * R + 1.0) + fabs(tau) * 0x1p-22;
        const double e_hi = hi_only ? 1.02 * 0x1p-9 * qn * R : 0.0;
        const double e2 = (4.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24) * qn * R + e_hi + 1e-30;
        U = 1.0 - tau - e1 - e2;
    }
    float Uf = (float)U;
    if ((double)Uf > U) Uf = nextafterf(Uf, -INFINITY);
    return Uf;
}

template <int METRIC>
__global__ void ThresholdKernel(ThresholdArgs a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.tasks) return;
    double tau = (double)a.list_dist[(size_t)t * a.k + a.k - 1];
    if (a.query_tau) tau = fmin(tau, (double)OrderedKeyToFloat(a.query_tau[a.task_query[t]]));
    double nq;
    if (a.task_nq) {
        nq = a.task_nq[a.task_src ? a.task_src[t] : t];
    } else {
        const float* q = a.queries + (size_t)a.task_query[t] * a.ld;
        nq = 0.0;
        for (uint32_t i = 0; i < a.d; ++i) nq += (double)q[i] * (double)q[i];
    }
    a.task_U[t] = ThresholdOf<METRIC>(tau, nq, a.shard_rmax[a.task_shard[t]], (double)a.d, a.hi_only != 0);
}

// ---- rescoring ---------------------------------------------------------------------------------------------------------
struct RescoreArgs {
    const float* points;
    const float* queries;
    const uint32_t* task_query;
    const uint32_t* cand_cnt;
    const uint32_t* cand_row;
    float* list_dist;             // [tasks][k] in/out, ascending (dist, row)
    uint32_t* list_row;
    const uint32_t* task_shard;
    const int64_t* shard_offsets;
    uint32_t tasks, k, d, ld, cand_cap;
    uint32_t dedupe;              // 1: skip candidates already in the list (cell passes)
    // A task whose bucket overflows in some pass is DIRTY: its bucket is not re-scored (its list, made of real rows, stays a valid if
    // looser bound for the later passes) and after the last pass its whole shard is evaluated exactly by many wavefronts (RANGE mode
    // below + DirtyMergeKernel).  (Until round 3 the one wavefront of the task re-read the shard inside this kernel: 1.6 s for a 7.5M-row shard.)
    uint32_t* dirty_flag;         // [tasks] zeroed per call
    uint2* dirty_list;            // [dirty_cap] (task, unused)
    uint32_t* dirty_count;        // dirty tasks so far (may exceed dirty_cap: the call then falls back to the plain exact scan)
    uint32_t dirty_cap;
    // RANGE mode (range_segs > 0): work item w = (dirty task w / range_segs, segment w % range_segs) of the task's shard, evaluated from
    // an EMPTY list into part_dist / part_row [w][k]; the workgroups of the launch take items round robin
    uint32_t range_segs;
    float* part_dist;
    uint32_t* part_row;
    // Heavy buckets are SPLIT (parts > 1: the launch has parts x tasks wavefronts, wavefront w * tasks + t = part w of task t): a bucket
    // of more than split_min candidates is cut into up to `parts` pieces, each re-scored by a wavefront of its own from the task's
    // current list; the pieces' lists go to split_dist / split_row [task][part][k] and SplitMergeKernel, the next launch, merges them.  The candidate counts are long-tailed (mean 22, maximum 1 283 in the
    // shard-sample pass of the NS workload) and a launch lasts as long as its heaviest task.
    uint32_t parts, split_min;
    float* split_dist;
    uint32_t* split_row;
    uint32_t* stat_overflowed;             // (SplitMergeKernel) += buckets of this pass that overflowed
    unsigned long long* stat_candidates;   // (SplitMergeKernel) += candidates of this pass
    // SEED mode (seed_rows > 0): the lists start EMPTY and the "candidates" of task t are the first seed_rows plane rows of its nearest
    // cell (plane_perm maps them to store rows; padding rows are skipped) — no bucket is read
    uint32_t seed_rows;
    const uint32_t* nearest;            // [tasks] nearest cell of the task, relative to its shard's first cell (indexed through task_src)
    const uint32_t* task_src;           // [tasks]
    const uint32_t* shard_cell_begin;   // [S+1]
    const uint64_t* cell_tile_begin;    // [cells+1] first 32-row plane tile of every cell
    const uint32_t* plane_perm;         // [tiles*32] plane row -> store row (0xFFFFFFFF: padding)
    uint32_t* query_tau;                // optional [nq] (GPANN_SCAN_FOR_MERGE): atomicMin of OrderedKey(k-th distance) over the query's tasks
    // SEED mode under GPANN_SCAN_FOR_MERGE with several probes per query (seed_first_slot_of != 0 = nprobe): only the task of a query's
    // FIRST slot — the shard the router ranks nearest — gets seed rows; the other tasks start with an empty list and live on the query's
    // merge bound, which that first list sets (three quarters of the seed evaluations at nprobe 4 bought lists the bound made irrelevant)
    uint32_t seed_first_slot_of;
    uint32_t seed_catch_up;             // second seed launch: the tasks of queries that still have no bound are seeded after all
    const uint32_t* task_out;           // [tasks] slot index q * nprobe + j of every task
    // SEED mode, every task seeded (seed_first_slot_of == 0): the task's first bf16 threshold U (ThresholdKernel's job) is written at the end of
    // its seed evaluation — the shard-sample pass follows without a launch in between
    float* seed_task_U;                 // [tasks] or null
    const double* task_nq;              // [tasks] |q|^2 (through task_src)
    const double* shard_rmax;           // [S]
    uint32_t seed_hi_only;
};

__device__ __forceinline__ void ListInsertF(float& ld_, uint32_t& lr_, float cd, uint32_t cr, uint32_t k, int lane) {
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

// One wavefront per task.  Candidates are taken a ROUND at a time (32 for L2, 16 for the inner product), a round in sub-rounds of 16 rows
// fetched COALESCED — half a wavefront per row: one 16-byte load per lane = one GROUP of four dimensions (a lane per candidate reading its
// own row costs the texture path 64 cache lines per load instruction); the next sub-round's rows are in flight while this one's are used.
// What the reference's sequence fixes is the ORDER OF THE ADDITIONS into `result` (src/dist.cpp:17-52); its terms — L2: s_g = ((d0 d0 +
// d1 d1) + d2 d2) + d3 d3 of a group of four dimensions, then d_i d_i of the d % 4 tail dimensions; inner product: the products q_i p_i —
// do not depend on `result`.  Phase A: the lane that loaded a group computes its term(s) from registers (the query's group sits in a
// register too) and leaves them in LDS, terms[row of the round][term index]; phase B: lane r adds row r's terms in the reference's order.
// (Until round 5 the raw rows went through LDS — 16 x 528 bytes per round written, read back by phase A, terms written, read by phase B — and
// phase B's chain of d / 4 dependent additions served 16 rows on 16 lanes; L2 terms are a quarter of the row, so a round holds twice
// the rows in half the LDS and a CU twice the tasks.)  The survivors are inserted into the task's list (one entry per lane).
constexpr int kRescoreSub = 16;                                                   // rows per sub-round (8 load instructions, two rows each)
constexpr int kRescoreRowsL2 = 32, kRescoreRowsIP = 16;                           // rows per round
template <int METRIC> struct RescoreShape {
    static constexpr int kRows = METRIC == GPANN_METRIC_L2 ? kRescoreRowsL2 : kRescoreRowsIP;
    // words between two rows' terms: L2 — d / 4 group terms + d % 4 tail terms, odd (phase B: lane r reads word r * stride + i);
    // inner product — d products, rows 16-byte aligned (phase A writes a group's four at once)
    __host__ __device__ static inline uint32_t Stride(uint32_t ld) { return METRIC == GPANN_METRIC_L2 ? (((ld >> 2) + 3u) | 1u) : ld + 4u; }
};
constexpr int kRescoreTasksPerWG = 1;   // wavefronts (= tasks) per workgroup: candidate counts are long-tailed and a workgroup lasts as long as its
                                        // slowest task, so single-wavefront workgroups pack the CUs best
template <int METRIC> __host__ __device__ inline size_t RescoreLdsFloats(uint32_t ld) {   // per wavefront: the query row + a round's terms
    return (size_t)ld + (size_t)RescoreShape<METRIC>::kRows * RescoreShape<METRIC>::Stride(ld);
}

// RANGED (a template argument: the loop over work items must not cost the one-task-per-wavefront instance its registers) = RANGE mode.
// WIDE = rows of more than 128 dimensions (further chunks of 32 groups per sub-round; the loop costs the narrow instance 30 registers).
template <int METRIC, bool RANGED = false, bool WIDE = false>
__global__ void __launch_bounds__(64 * kRescoreTasksPerWG) RescoreKernel(RescoreArgs a) {
    extern __shared__ float rsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool ranged = RANGED;
    uint32_t item = blockIdx.x * kRescoreTasksPerWG + wave;
    do {
    uint32_t t = item;
    int64_t scan_lo = 0, scan_hi = 0;
    uint32_t part = 0, n_parts = 1, c_lo = 0;
    if (!ranged && a.parts > 1) {
        part = item / a.tasks;
        t = item - part * a.tasks;
        if (part >= a.parts) return;
    }
    if (ranged) {
        const uint32_t n_dirty = min(*a.dirty_count, a.dirty_cap);
        if (item >= n_dirty * a.range_segs) return;
        t = a.dirty_list[item / a.range_segs].x;
        const uint32_t seg = item % a.range_segs, b = a.task_shard[t];
        const int64_t r0 = a.shard_offsets[b], rows = a.shard_offsets[b + 1] - r0;
        const int64_t per = (rows + a.range_segs - 1) / a.range_segs;
        scan_lo = min(r0 + rows, r0 + (int64_t)seg * per);
        scan_hi = min(r0 + rows, scan_lo + per);
    } else if (t >= a.tasks) {
        return;
    }
    const bool seed = a.seed_rows != 0;
    uint64_t seed_first = 0;   // first plane row of the task's nearest cell
    uint32_t cnt;
    // Bucket mode: what does not depend on the bucket's size — the task's list, the first 64 row ids of its bucket — is loaded BEFORE the size is
    // known (the size, the ids and the rows were three dependent round trips on every task's chain; the ids of a small bucket's slots beyond its
    // size are whatever an earlier pass left there: masked below).
    const float* q = a.queries + (size_t)a.task_query[t] * a.ld;
    const float4 q0 = (uint32_t)(lane & 31) < (a.ld >> 2) ? reinterpret_cast<const float4*>(q)[lane & 31] : make_float4(0.f, 0.f, 0.f, 0.f);   // the query's group of chunk 0
    const bool bucket = !seed && !ranged;
    uint32_t early_ids = kNoId, early_lr = kNoId;
    float early_ld = INFINITY;
    if (bucket) {
        if (part == 0 && (uint32_t)lane < a.cand_cap) early_ids = a.cand_row[(size_t)t * a.cand_cap + lane];
        if (lane < (int)a.k) {
            early_ld = a.list_dist[(size_t)t * a.k + lane];
            early_lr = a.list_row[(size_t)t * a.k + lane];
        }
    }
    if (seed && a.seed_first_slot_of != 0) {
        const bool first_slot = a.task_out[t] % a.seed_first_slot_of == 0;
        if (!a.seed_catch_up) {
            if (!first_slot) {
                if (lane < (int)a.k) {   // (an empty list: the pass's thresholds take the query's merge bound instead)
                    a.list_dist[(size_t)t * a.k + lane] = INFINITY;
                    a.list_row[(size_t)t * a.k + lane] = kNoId;
                }
                return;
            }
        } else if (first_slot || a.query_tau[a.task_query[t]] != kOrderedKeyInf) {
            return;   // (catch-up launch: only tasks of queries whose first slot gave no bound — skipped probe, shard with fewer than k rows)
        }
    }
    if (seed) {
        // plane rows [seed_first, seed_first + cnt): from the nearest cell's first row onwards, inside the shard's plane range (the
        // following cells are its neighbours along the cell chain).  (A workgroup of four wavefronts per task with merged private lists
        // was measured slower: 10k workgroups cost more to start than the long tail of the candidate counts costs to sit out.)
        const uint32_t b = a.task_shard[t];
        const uint64_t p0 = a.cell_tile_begin[a.shard_cell_begin[b]] * kGTileRows, p1 = a.cell_tile_begin[a.shard_cell_begin[b + 1]] * kGTileRows;
        seed_first = a.cell_tile_begin[a.shard_cell_begin[b] + a.nearest[a.task_src[t]]] * kGTileRows;
        if (seed_first + a.seed_rows > p1) seed_first = p1 - p0 > a.seed_rows ? p1 - a.seed_rows : p0;
        cnt = (uint32_t)min((uint64_t)a.seed_rows, p1 - seed_first);
    } else if (ranged) {
        cnt = 0;
    } else {
        cnt = a.cand_cnt[t];
        if (cnt == 0) return;
        if (cnt > a.cand_cap) {   // the bucket overflowed: the task is dirty from now on
            if (lane == 0 && atomicExch(&a.dirty_flag[t], 1u) == 0u) {
                const uint32_t slot = atomicAdd(a.dirty_count, 1u);
                if (slot < a.dirty_cap) a.dirty_list[slot] = make_uint2(t, 0u);
            }
            return;
        }
        if (a.parts > 1) {   // this wavefront's piece of the bucket
            n_parts = min(a.parts, (cnt + a.split_min - 1) / a.split_min);
            if (part >= n_parts) return;
            const uint32_t per = ((cnt + n_parts - 1) / n_parts + 15u) / 16u * 16u;
            c_lo = part * per;
            cnt = min(cnt, c_lo + per) - min(cnt, c_lo);   // (never empty: split_min >> 16 parts)
        }
    }
    const bool overflowed = ranged;   // the "candidates" are the rows [scan_lo, scan_hi) of the shard
    using Shape = RescoreShape<METRIC>;
    constexpr int kRows = Shape::kRows, kSubs = kRows / kRescoreSub, kRoundsPerBlock = 64 / kRows;
    const uint32_t ld = a.ld, ld4 = a.ld >> 2, tstride = Shape::Stride(a.ld);
    float* sq = rsm + (size_t)wave * RescoreLdsFloats<METRIC>(ld);
    float* terms = sq + ld;   // [kRows][tstride]
    if (WIDE) {   // (the narrow instance keeps the query's only chunk in registers: q0 below)
        for (uint32_t i = lane; i < ld; i += 64) sq[i] = q[i];
    }
    const uint32_t k = a.k;
    float ld_ = early_ld;      // (bucket mode: the task's list; seed / range mode: empty)
    uint32_t lr_ = early_lr;
    // (wavefront-uniform by construction; said so, or the loop's counters and bounds live in vector registers)
    uint64_t n_items = overflowed ? (uint64_t)(scan_hi - scan_lo) : cnt;
    n_items = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)n_items) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(n_items >> 32)) << 32);
    c_lo = __builtin_amdgcn_readfirstlane(c_lo);
    seed_first = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)seed_first) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(seed_first >> 32)) << 32);
    const uint32_t n_chunks = (ld4 + 31) / 32;   // 32 groups (128 dimensions) of a row per load instruction and half wavefront
    const int half = lane >> 5, c4l = lane & 31;
    const uint32_t full = a.d >> 2, tail = a.d & 3u;

    // row ids are read 64 candidates at a time (one coalesced load, a block ahead); round `sub` of a block takes its kRows from there
    auto load_block = [&](uint64_t first) -> uint32_t {
        const uint64_t idx = first + lane;
        if (idx >= n_items) return kNoId;
        if (bucket && first == 0 && c_lo == 0) return early_ids;
        if (seed) return a.plane_perm[seed_first + idx];   // (kNoId on the padding of a cell's last tile: skipped like an absent candidate)
        return overflowed ? (uint32_t)(scan_lo + (int64_t)idx) : a.cand_row[(size_t)t * a.cand_cap + c_lo + idx];
    };
    auto round_rows = [&](uint32_t block_rows, uint32_t sub) -> uint32_t {
        if (kRows == 64) return block_rows;
        const uint32_t r = __shfl(block_rows, (int)(kRows * sub + (lane & (kRows - 1))));
        return lane < kRows ? r : kNoId;
    };
    // sub-round `s` of the round whose rows are `row`, one chunk of 32 groups: instruction i serves rows 16 s + 2 i (lanes 0..31) and
    // 16 s + 2 i + 1 (lanes 32..63).  Straight-line code: an absent candidate (and a lane beyond the row's last group) loads from the store's
    // first row instead of being branched around — nothing reads what it gets (a branch per load cost this kernel its registers).
    auto fetch = [&](float4 (&v)[kRescoreSub / 2], uint32_t row, uint32_t s, uint32_t chunk) {
        const uint32_t c4 = 32u * chunk + c4l;
        const float4* lane_base = reinterpret_cast<const float4*>(a.points) + (c4 < ld4 ? c4 : 0u);
#pragma unroll
        for (int i = 0; i < kRescoreSub / 2; ++i) {
            const uint32_t r = __shfl(row, (int)(kRescoreSub * s + 2u * i) + half);
            v[i] = lane_base[(size_t)(r != kNoId ? r : 0u) * ld4];
        }
    };
    // phase A of what `fetch` loaded
    auto leave_terms = [&](const float4 (&v)[kRescoreSub / 2], uint32_t s, uint32_t chunk, const float4 qv) {
        const uint32_t g = 32u * chunk + c4l;
        float* w = terms + (size_t)(kRescoreSub * s + half) * tstride;
        if (METRIC == GPANN_METRIC_L2) {
            float sg[kRescoreSub / 2];
#pragma unroll
            for (int i = 0; i < kRescoreSub / 2; ++i) {
                const float d0 = v[i].x - qv.x, d1 = v[i].y - qv.y, d2 = v[i].z - qv.z, d3 = v[i].w - qv.w;
                float t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                sg[i] = t + d3 * d3;
            }
            if (g < full) {
#pragma unroll
                for (int i = 0; i < kRescoreSub / 2; ++i) w[(size_t)(2 * i) * tstride + g] = sg[i];
            } else if (g == full && tail != 0u) {   // (the tail dimensions' own terms, after the groups')
#pragma unroll
                for (int i = 0; i < kRescoreSub / 2; ++i) {
                    const float d0 = v[i].x - qv.x;
                    w[(size_t)(2 * i) * tstride + full] = d0 * d0;
                }
                if (tail > 1u) {
#pragma unroll
                    for (int i = 0; i < kRescoreSub / 2; ++i) {
                        const float d1 = v[i].y - qv.y;
                        w[(size_t)(2 * i) * tstride + full + 1u] = d1 * d1;
                    }
                }
                if (tail > 2u) {
#pragma unroll
                    for (int i = 0; i < kRescoreSub / 2; ++i) {
                        const float d2 = v[i].z - qv.z;
                        w[(size_t)(2 * i) * tstride + full + 2u] = d2 * d2;
                    }
                }
            }
        } else if (g < ld4) {
#pragma unroll
            for (int i = 0; i < kRescoreSub / 2; ++i)
                *reinterpret_cast<float4*>(w + (size_t)(2 * i) * tstride + 4u * g) = make_float4(qv.x * v[i].x, qv.y * v[i].y, qv.z * v[i].z, qv.w * v[i].w);
        }
    };

    float4 v[kRescoreSub / 2];
    uint32_t block_rows = load_block(0), next_block_rows = load_block(64);
    uint32_t row = round_rows(block_rows, 0), cur_row = row;   // the round being fetched / evaluated
    fetch(v, row, 0, 0);
    const uint64_t n_subs = (n_items + kRescoreSub - 1) / kRescoreSub;
    for (uint64_t sr = 0; sr < n_subs; ++sr) {
        const uint32_t s = (uint32_t)sr & (uint32_t)(kSubs - 1);
        const uint64_t base = sr / kSubs * kRows;
        const uint32_t n_here = (uint32_t)min((uint64_t)kRows, n_items - base);
        if (s == 0) {
            __builtin_amdgcn_wave_barrier();   // the previous round's LDS reads are done
            cur_row = row;
        }
        leave_terms(v, s, 0, q0);
        for (uint32_t ch = 1; WIDE && ch < n_chunks; ++ch) {   // (rows wider than 128 dimensions: the further chunks of this sub-round)
            fetch(v, cur_row, s, ch);
            const uint32_t c4 = 32u * ch + c4l;
            leave_terms(v, s, ch, c4 < ld4 ? reinterpret_cast<const float4*>(sq)[c4] : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        // the next sub-round's rows (of this round or the first of the next) are in flight while this one's terms are added up
        const bool round_done = kRescoreSub * (s + 1u) >= n_here;
        uint32_t f_s = s + 1u;
        if (round_done) {
            const uint32_t sub = (uint32_t)(base / kRows) % (uint32_t)kRoundsPerBlock;
            if (sub + 1u == (uint32_t)kRoundsPerBlock) {
                block_rows = next_block_rows;
                next_block_rows = load_block(base + kRows + 64);
                row = round_rows(block_rows, 0);
            } else {
                row = round_rows(block_rows, sub + 1);
            }
            f_s = 0;
        }
        if (!round_done || base + kRows < n_items) fetch(v, row, f_s, 0);
        if (!round_done) continue;
        __builtin_amdgcn_wave_barrier();
        const bool valid = cur_row != kNoId;
        float e = INFINITY;
        if (valid) {   // (lanes 0..kRows-1) phase B
            const float* row_terms = terms + (size_t)lane * tstride;
            float result = 0.f;
            if (METRIC == GPANN_METRIC_L2) {
                const uint32_t n_terms = full + tail;
#pragma unroll 8
                for (uint32_t g = 0; g < n_terms; ++g) result = result + row_terms[g];
                e = result;
            } else {
#pragma unroll 4
                for (uint32_t g = 0; g < full; ++g) {
                    const float4 pr = reinterpret_cast<const float4*>(row_terms)[g];
                    result = result + pr.x;
                    result = result + pr.y;
                    result = result + pr.z;
                    result = result + pr.w;
                }
                for (uint32_t i = full << 2; i < a.d; ++i) result = result + row_terms[i];
                e = 1.0f - result;
            }
        }
        float td = __shfl(ld_, k - 1);
        uint32_t tr = __shfl(lr_, k - 1);
        bool pass = valid && PairLess(e, cur_row, td, tr);
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            const float cd = __shfl(e, l);
            const uint32_t cr = __shfl(cur_row, l);
            // a row may be offered twice (cell passes overlap a stage-0 row; an overflowed bucket re-reads its range): a row that is
            // in the list stays once, one that was pushed out cannot pass the k-th entry again
            if (a.dedupe && __ballot(lane < (int)k && lr_ == cr) != 0ull) {
                pass = pass && lane != l;
                mask = __ballot(pass);
                continue;
            }
            ListInsertF(ld_, lr_, cd, cr, k, lane);
            td = __shfl(ld_, k - 1);
            tr = __shfl(lr_, k - 1);
            pass = pass && lane != l && PairLess(e, cur_row, td, tr);
            mask = __ballot(pass);
        }
    }
    if (ranged) {
        if (lane < (int)k) {
            a.part_dist[(size_t)item * k + lane] = ld_;
            a.part_row[(size_t)item * k + lane] = lr_;
        }
        __builtin_amdgcn_wave_barrier();   // (the next item re-uses this wavefront's LDS)
        item += gridDim.x * kRescoreTasksPerWG;
        continue;
    }
    if (n_parts > 1) {
        // a piece of a split bucket: its list goes to the scratch lists, SplitMergeKernel (the next launch) merges the pieces.  (Merging
        // inside this kernel — last piece to arrive, agent-scope fences — was measured 10x SLOWER than the whole unsplit launch: on this
        // multi-XCD part every agent-scope release / acquire writes back and invalidates an XCD's L2.)
        if (lane < (int)k) {
            a.split_dist[((size_t)t * a.parts + part) * k + lane] = ld_;
            a.split_row[((size_t)t * a.parts + part) * k + lane] = lr_;
        }
        return;
    }
    if (lane < (int)k) {
        a.list_dist[(size_t)t * k + lane] = ld_;
        a.list_row[(size_t)t * k + lane] = lr_;
    }
    // the query's merge bound: the smallest k-th distance among its tasks' lists (a list that is not full yet has +inf there)
    if (a.query_tau && lane == (int)k - 1 && ld_ < INFINITY) atomicMin(&a.query_tau[a.task_query[t]], OrderedKey(ld_));
    if (!RANGED) {
        if (seed && a.seed_task_U && lane == (int)k - 1)   // (the lane that holds the k-th entry)
            a.seed_task_U[t] = ThresholdOf<METRIC>((double)ld_, a.task_nq[a.task_src[t]], a.shard_rmax[a.task_shard[t]], (double)a.d, a.seed_hi_only != 0);
    }
    return;
    } while (RANGED);   // items
}

// The lists of the tasks whose bucket RescoreKernel split (same rule: cand_cnt, split_min, parts): one wavefront per task merges the pieces'
// lists — all grown from the task's list, so an entry of that list can be in several: duplicates are dropped by row — into the task's list.
// It also keeps the pass's statistics (what OverflowCountKernel did in a launch of its own): buckets that overflowed, candidates in total.
constexpr uint32_t kSplitMergeTasksPerWave = 8;   // (a wavefront per task: at C3, 40 000 wavefronts of which most return at once, the four launches were 0.30 ms of a 2.8 ms scan)
__device__ __forceinline__ void SplitMergeBody(const RescoreArgs& a, uint32_t block) {   // (all 256 threads of the workgroup)
    __shared__ unsigned long long s_total;
    __shared__ uint32_t s_overflowed;
    const int lane = threadIdx.x & 63;
    const uint32_t t0 = (block * 4 + (threadIdx.x >> 6)) * kSplitMergeTasksPerWave;   // this wavefront's tasks: t0 .. t0 + 7, lane i looks at task t0 + i
    if (threadIdx.x == 0) {
        s_total = 0ull;
        s_overflowed = 0u;
    }
    __syncthreads();
    const uint32_t t_l = t0 + (uint32_t)lane;
    const uint32_t cnt_l = lane < (int)kSplitMergeTasksPerWave && t_l < a.tasks ? a.cand_cnt[t_l] : 0u;
    if (cnt_l != 0u) {
        atomicAdd(&s_total, (unsigned long long)cnt_l);
        if (cnt_l > a.cand_cap) atomicAdd(&s_overflowed, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0 && a.stat_overflowed) {   // (one global atomic per workgroup of 32 tasks)
        if (s_total) atomicAdd(a.stat_candidates, s_total);
        if (s_overflowed) atomicAdd(a.stat_overflowed, s_overflowed);
    }
    const uint32_t parts_l = cnt_l == 0u || cnt_l > a.cand_cap ? 0u : min(a.parts, (cnt_l + a.split_min - 1) / a.split_min);
    unsigned long long todo = __ballot(parts_l > 1u);
    const uint32_t k = a.k;
    while (todo) {
        const int i = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint32_t t = t0 + (uint32_t)i, n_parts = (uint32_t)__shfl((int)parts_l, i);
        const float* pd = a.split_dist + ((size_t)t * a.parts) * k;
        const uint32_t* pr = a.split_row + ((size_t)t * a.parts) * k;
        float ld_ = INFINITY;
        uint32_t lr_ = kNoId;
        if (lane < (int)k) {
            ld_ = pd[lane];
            lr_ = pr[lane];
        }
        for (uint32_t p = 1; p < n_parts; ++p) {
            float cd_l = INFINITY;
            uint32_t cr_l = kNoId;
            if (lane < (int)k) {
                cd_l = pd[(size_t)p * k + lane];
                cr_l = pr[(size_t)p * k + lane];
            }
            for (uint32_t e = 0; e < k; ++e) {
                const float cd = __shfl(cd_l, e);
                const uint32_t cr = __shfl(cr_l, e);
                const float td = __shfl(ld_, k - 1);
                const uint32_t tr = __shfl(lr_, k - 1);
                if (!PairLess(cd, cr, td, tr)) break;   // (ascending: nothing after it fits either)
                if (__ballot(lane < (int)k && lr_ == cr) != 0ull) continue;
                ListInsertF(ld_, lr_, cd, cr, k, lane);
            }
        }
        if (lane < (int)k) {
            a.list_dist[(size_t)t * k + lane] = ld_;
            a.list_row[(size_t)t * k + lane] = lr_;
        }
        if (a.query_tau && lane == (int)k - 1 && ld_ < INFINITY) atomicMin(&a.query_tau[a.task_query[t]], OrderedKey(ld_));
    }
}
static __global__ void __launch_bounds__(256) SplitMergeKernel(RescoreArgs a) { SplitMergeBody(a, blockIdx.x); }

// The dirty tasks' lists: one wavefront per dirty task merges the range_segs partial lists RescoreKernel's RANGE mode left (ascending
// (dist, row), disjoint row ranges) into the task's list — which then is the exact top-k of the whole shard, whatever the passes did.
__device__ __forceinline__ void DirtyMergeBody(const RescoreArgs& a, uint32_t index, int lane) {   // (one wavefront)
    const uint32_t n_dirty = min(*a.dirty_count, a.dirty_cap);
    if (index >= n_dirty) return;
    const uint32_t dirty_index = index;
    const uint32_t t = a.dirty_list[dirty_index].x, k = a.k;
    float ld_ = INFINITY;
    uint32_t lr_ = kNoId;
    // 64 segments at a time: every lane looks at the FIRST (smallest) entry of one partial list; only the lists whose first entry still beats
    // the merged list's k-th are walked.  (One list after the other, each a dependent global load: 1.5 ms for the 2 170 lists of a 3.3M-row shard.)
    for (uint32_t sg0 = 0; sg0 < a.range_segs; sg0 += 64) {
        const uint32_t sg_l = sg0 + (uint32_t)lane;
        float fd = INFINITY;
        uint32_t fr = kNoId;
        if (sg_l < a.range_segs) {
            fd = a.part_dist[((size_t)dirty_index * a.range_segs + sg_l) * k];
            fr = a.part_row[((size_t)dirty_index * a.range_segs + sg_l) * k];
        }
        unsigned long long todo = __ballot(PairLess(fd, fr, __shfl(ld_, k - 1), __shfl(lr_, k - 1)));
        while (todo) {
            const uint32_t sg = sg0 + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const size_t base = ((size_t)dirty_index * a.range_segs + sg) * k;
            float cd_l = INFINITY;
            uint32_t cr_l = kNoId;
            if (lane < (int)k) {
                cd_l = a.part_dist[base + lane];
                cr_l = a.part_row[base + lane];
            }
            for (uint32_t e = 0; e < k; ++e) {
                const float cd = __shfl(cd_l, e);
                const uint32_t cr = __shfl(cr_l, e);
                const float td = __shfl(ld_, k - 1);
                const uint32_t tr = __shfl(lr_, k - 1);
                if (!PairLess(cd, cr, td, tr)) break;
                ListInsertF(ld_, lr_, cd, cr, k, lane);
            }
        }
    }
    if (lane < (int)k) {
        a.list_dist[(size_t)t * k + lane] = ld_;
        a.list_row[(size_t)t * k + lane] = lr_;
    }
    if (a.query_tau && lane == (int)k - 1 && ld_ < INFINITY) atomicMin(&a.query_tau[a.task_query[t]], OrderedKey(ld_));
}
static __global__ void __launch_bounds__(64) DirtyMergeKernel(RescoreArgs a) { DirtyMergeBody(a, blockIdx.x, (int)threadIdx.x); }
// After the last pass, in one launch: the pieces of the split buckets into their lists (blocks below merge_blocks) and the dirty tasks' partial
// lists into theirs (wavefront 0 of the blocks behind them; a dirty task has no bucket: its last passes gave it none).
static __global__ void __launch_bounds__(256) FinalMergeKernel(RescoreArgs a, uint32_t merge_blocks) {
    if (blockIdx.x < merge_blocks) SplitMergeBody(a, blockIdx.x);
    else if (threadIdx.x < 64) DirtyMergeBody(a, blockIdx.x - merge_blocks, (int)threadIdx.x);
}

// final lists -> caller arrays: out[task_out[t]][i] = (dist, permutation[row], row)
// query_tau / task_query (GPANN_SCAN_FOR_MERGE): entries beyond the query's merge bound are cut — what is left is a prefix of the
// reference's per-slot list (an entry of OUR list beyond the bound need not be one of the reference's: rows beyond the bound were never looked at)
static __global__ void ListOutputKernel(const float* list_dist, const uint32_t* list_row, const uint32_t* task_out, const uint32_t* permutation,
                                 uint32_t tasks, uint32_t k, float* out_dist, uint32_t* out_ids, uint32_t* out_rows, const uint32_t* query_tau,
                                 const uint32_t* task_query) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)tasks * k) return;
    const uint32_t t = (uint32_t)(i / k), e = (uint32_t)(i % k);
    const size_t dst = (size_t)task_out[t] * k + e;
    uint32_t row = list_row[i];
    float dist = list_dist[i];
    if (query_tau && !(dist <= OrderedKeyToFloat(query_tau[task_query[t]]))) {
        dist = INFINITY;
        row = kNoId;
    }
    if (out_dist) out_dist[dst] = dist;
    if (out_rows) out_rows[dst] = row;
    if (out_ids) out_ids[dst] = row == kNoId ? kNoId : permutation[row];
}

// row norms for the filter: -|p|^2/2 as float (from a double sum), per-shard max |p| via an ordered-bits atomic max, and a
// finiteness / magnitude check.  One wavefront per row.
static __global__ void __launch_bounds__(256) RowNormKernel(const float* points, uint64_t rows, uint32_t d, uint32_t ld, const int64_t* shard_offsets,
                                                     uint32_t S, float* neg_half_norm, unsigned long long* shard_max_bits, uint32_t* bad,
                                                     const float* mu = nullptr) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_id = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave_id; r < rows; r += n_waves) {
        const float* p = points + r * ld;
        double s = 0.0;
        for (uint32_t i = lane; i < d; i += 64) {
            const float v = mu ? p[i] - mu[i] : p[i];   // (the fp32 value the planes are split from)
            s += (double)v * (double)v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            if (!(s < 1e30)) atomicOr(bad, 1u);   // inf / NaN / absurd magnitudes: the bounds above are not meant for them
            neg_half_norm[r] = (float)(-0.5 * s);
            // shard of row r: binary search in the offsets
            uint32_t lo = 0, hi = S;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) / 2;
                if ((int64_t)r >= shard_offsets[mid]) lo = mid;
                else hi = mid;
            }
            const double nrm = sqrt(s) * (1.0 + 1e-12);
            atomicMax(&shard_max_bits[lo], (unsigned long long)__double_as_longlong(nrm));
        }
    }
}

}  // namespace gpann
