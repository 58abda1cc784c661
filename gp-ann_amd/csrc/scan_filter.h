// scan_filter.h — the MFMA pre-filter of the exact shard scan (included by scan.hip).
//
// The exact scan is bound by FP32 VALU issue: 3 unfused lane-ops per (pair, dimension).  Only the k best rows of a shard
// matter, so almost every one of those exact evaluations is spent on rows that are nowhere near the current k-th
// distance.  The filter replaces them by a matrix-core bound:
//
//   1. a first slice of every shard is scanned exactly (ScanTileKernel) and yields, per task, an exact k-th distance tau;
//   2. FilterKernel streams the remaining rows through v_mfma_f32_32x32x16_bf16 and evaluates
//          acc = q.p - |p|^2/2                       (L2)        acc = q.p          (MIPS)
//      with q and p each split into two bf16 planes (hi + lo, three products hi*hi, hi*lo, lo*hi, f32 accumulate);
//   3. a row is kept as a CANDIDATE iff acc >= U[task], where U is derived from tau and a rigorous bound on
//      |acc - true value| + |canonical float distance - true value| (ThresholdKernel) — every row whose canonical
//      distance is <= tau passes, plus a few percent of near misses;
//   4. RescoreKernel evaluates the canonical binary32 distance (the reference's exact sequence) of the candidates only and
//      inserts them into the task's exact top-k list.
// Steps 2-4 run twice (a short slice to tighten tau, then the rest of the shard).  The result lists are the exact lists:
// the filter only decides which rows get the exact evaluation, and it never drops a row that could belong to the top-k.
// If the fixed-size candidate bucket of any task overflows (adversarial inputs: masses of equal distances) the caller
// falls back to the plain exact scan for the whole call.
#pragma once

#include "shards.h"

#include <type_traits>

namespace gpann {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kFQB = 256;                 // tasks per filter unit: 4 waves x 64
constexpr int kFPT = 64;                  // rows per tile
constexpr int kFKS = 8;                   // k-steps of 16 dims held in registers (d <= 128)
constexpr int kFRowStride = 128 + 8;      // bf16 elements per LDS row: 272 B puts the 16 rows of a ds_read_b128 lane group on distinct banks
constexpr int kFPlane = kFPT * kFRowStride;   // bf16 elements per plane

struct FilterArgs {
    const float* points;          // [rows][ld]
    const float* queries;         // [nq][ld]
    const float* neg_half_norm;   // [rows]  -|p|^2/2 (L2 only)
    const FilterUnit* units;
    const uint32_t* task_query;   // [tasks]
    const float* task_U;          // [tasks] pass iff acc >= U
    uint32_t* cand_cnt;           // [tasks]
    uint32_t* cand_row;           // [tasks][cand_cap]
    uint32_t cand_cap;
    uint32_t d, ld;
    const char* frag_planes;      // optional: pre-split bf16 planes in MFMA fragment order (FilterKernelReg), see kGTileBytes
    int64_t row_limit;            // rows of the store (frag_planes and neg_half_norm are readable 64 rows beyond it)
};

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

template <int METRIC>
__global__ void __launch_bounds__(256, 2) FilterKernel(FilterArgs a) {
    extern __shared__ uint4 fsm4[];
    __bf16* sHi = reinterpret_cast<__bf16*>(fsm4);   // [64][136]
    __bf16* sLo = sHi + kFPlane;                     // [64][136]
    float* sNorm = reinterpret_cast<float*>(sLo + kFPlane);   // [64]

    const int64_t row_begin = a.units[blockIdx.x].row_begin, row_end = a.units[blockIdx.x].row_end;
    const uint32_t task_begin = a.units[blockIdx.x].task_begin, task_count = a.units[blockIdx.x].task_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t d = a.d, ld = a.ld;
    const int ksteps = (int)((d + 15) / 16);

    // ---- this wave's 64 queries as B operands, resident in registers: lane (c, h) holds dims 16 ks + 8 h .. + 7 of query c
    Frag qh[2][kFKS], ql[2][kFKS];
    float U[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t tl = 64u * wave + 32u * t + c;
        const uint32_t task = task_begin + min(tl, task_count - 1);
        U[t] = tl < task_count ? a.task_U[task] : INFINITY;   // lanes beyond the unit's tasks never pass
        const float* qrow = a.queries + (size_t)a.task_query[task] * ld;
#pragma unroll
        for (int ks = 0; ks < kFKS; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (ks < ksteps) {
                const uint32_t k0 = 16u * ks + 8u * h;
                if (k0 < ld) {
                    const float4 x = *reinterpret_cast<const float4*>(qrow + k0);
                    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
                }
                if (k0 + 4 < ld) {
                    const float4 x = *reinterpret_cast<const float4*>(qrow + k0 + 4);
                    v[4] = x.x; v[5] = x.y; v[6] = x.z; v[7] = x.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) SplitBf16(v[2 * j], v[2 * j + 1], qh[t][ks].u[j], ql[t][ks].u[j]);
        }
    }

    // staging role: float4 #(tid + 256 i) of the [64 rows][32 float4] tile
    for (int64_t row0 = row_begin; row0 < row_end; row0 += kFPT) {
        __syncthreads();   // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + 256 * i;
            const int r = f >> 5, c4 = f & 31;
            const int64_t row = min(row0 + r, row_end - 1);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((uint32_t)(4 * c4) < ld) x = *reinterpret_cast<const float4*>(a.points + (size_t)row * ld + 4 * c4);
            uint2 hi, lo;
            SplitBf16(x.x, x.y, hi.x, lo.x);
            SplitBf16(x.z, x.w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(sHi + r * kFRowStride + 4 * c4) = hi;
            *reinterpret_cast<uint2*>(sLo + r * kFRowStride + 4 * c4) = lo;
        }
        if (METRIC == GPANN_METRIC_L2 && tid < kFPT) sNorm[tid] = a.neg_half_norm[min(row0 + tid, row_end - 1)];
        __syncthreads();

        f32x16 acc[2][2];   // [point tile][query tile]
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // C rows of register r = 4g + e: 32 pt + 8 g + 4 h + e
                float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (METRIC == GPANN_METRIC_L2) nv = *reinterpret_cast<const float4*>(sNorm + 32 * pt + 8 * g + 4 * h);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    acc[pt][qt][4 * g + 0] = nv.x;
                    acc[pt][qt][4 * g + 1] = nv.y;
                    acc[pt][qt][4 * g + 2] = nv.z;
                    acc[pt][qt][4 * g + 3] = nv.w;
                }
            }
#pragma unroll
        for (int ks = 0; ks < kFKS; ++ks) {
            if (ks < ksteps) {
                Frag ah[2], al[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int off = (32 * pt + c) * kFRowStride + 16 * ks + 8 * h;
                    ah[pt].q = *reinterpret_cast<const uint4*>(sHi + off);
                    al[pt].q = *reinterpret_cast<const uint4*>(sLo + off);
                }
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        acc[pt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[pt].v, qh[qt][ks].v, acc[pt][qt], 0, 0, 0);
                        acc[pt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[pt].v, ql[qt][ks].v, acc[pt][qt], 0, 0, 0);
                        acc[pt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[pt].v, qh[qt][ks].v, acc[pt][qt], 0, 0, 0);
                    }
            }
        }

        // ---- epilogue: a row is a candidate for the lane's query iff acc >= U
        bool any = false;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) any |= acc[pt][qt][r] >= U[qt];
        if (__any(any)) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (acc[pt][qt][r] >= U[qt]) {
                            const int64_t row = row0 + 32 * pt + (r & 3) + 8 * (r >> 2) + 4 * h;
                            const uint32_t tl = 64u * wave + 32u * qt + c;
                            if (row < row_end && tl < task_count) {
                                const uint32_t task = task_begin + tl;
                                const uint32_t slot = atomicAdd(&a.cand_cnt[task], 1u);
                                if (slot < a.cand_cap) a.cand_row[(size_t)task * a.cand_cap + slot] = (uint32_t)row;
                            }
                        }
                    }
        }
    }
}

// ---- the filter with the QUERIES stationary in LDS and the points streamed through registers --------------------------------
// FilterKernel above (and an LDS-DMA fed variant of it that this file used to hold) stages POINT tiles in LDS for 256 tasks and
// keeps the query fragments in registers: every tile costs a workgroup barrier, and an LDS-DMA fill landing in LDS and the
// ds_read/MFMA stream slowed each other down (measured: filter time = fill-only time + compute-only time, no overlap).
// FilterKernelReg turns the roles around:
//   * the bf16 hi/lo fragments of the unit's <= 256 tasks are built ONCE into LDS (128 KiB, in MFMA B-operand order: a
//     ds_read_b128 with lane-linear addresses, conflict free) and stay there for the unit's whole row segment;
//   * the points are pre-split in HBM in MFMA A-operand order — per 32-row tile and k-step one contiguous KiB per plane
//     (lane (c, h) holds row 32 T + c, dims 16 ks + 8 h .. + 7; the hi plane of a tile is 8 contiguous KiB, then the lo plane) — so a wavefront loads a tile's fragments with 16 perfectly
//     coalesced global_load_dwordx4 straight into the registers the MFMA reads.  No LDS traffic for the points, no barrier in
//     the loop: the 8 wavefronts of a workgroup take tiles round robin and run independently, each prefetching its next tile
//     into a second register set while the matrix core works on the current one;
//   * every wavefront meets ALL tasks of the unit, so units with few tasks keep all 8 wavefronts busy;
//   * candidates of a tile are collected as bit masks and written after the tile's MFMA phase, when the prefetch has landed
//     (the atomic's vmcnt wait would otherwise stall on the loads issued before it).
constexpr int kGTileRows = 32;
constexpr int kGPlaneBytes = kFKS * 1024;      // one plane of a tile: [ks][lane][16 B]
constexpr int kGTileBytes = 2 * kGPlaneBytes;  // [plane hi|lo][ks][lane][16 B]: 16 KiB per 32 rows (512 B per row)
constexpr int kGTasks = 256;                   // tasks per unit: 8 query tiles of 32

template <int METRIC, int TAG, int KS>
__global__ void __launch_bounds__(512, 1) FilterKernelReg(FilterArgs a) {
    extern __shared__ uint4 gsm4[];
    char* lds = reinterpret_cast<char*>(gsm4);   // [8 query tiles][KS][2 planes][1 KiB]

    const int64_t row_begin = a.units[blockIdx.x].row_begin, row_end = a.units[blockIdx.x].row_end;
    const uint32_t task_begin = a.units[blockIdx.x].task_begin, task_count = a.units[blockIdx.x].task_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t ld = a.ld;
    const int nqt = (int)((task_count + 31u) / 32u);

    // ---- the query image: wavefront w builds query tile w (zeros for absent tasks)
    {
        const uint32_t tl = 32u * wave + c;
        const uint32_t task = task_begin + min(tl, task_count - 1);
        const bool real = tl < task_count;
        const float* qrow = a.queries + (size_t)a.task_query[task] * ld;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            const uint32_t k0 = 16u * ks + 8u * h;
            if (real && k0 < ld) {
                const float4 x = *reinterpret_cast<const float4*>(qrow + k0);
                v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            }
            if (real && k0 + 4 < ld) {
                const float4 x = *reinterpret_cast<const float4*>(qrow + k0 + 4);
                v[4] = x.x; v[5] = x.y; v[6] = x.z; v[7] = x.w;
            }
            Frag fh, fl;
#pragma unroll
            for (int j = 0; j < 4; ++j) SplitBf16(v[2 * j], v[2 * j + 1], fh.u[j], fl.u[j]);
            *reinterpret_cast<uint4*>(lds + ((wave * KS + ks) * 2 + 0) * 1024 + lane * 16) = fh.q;
            *reinterpret_cast<uint4*>(lds + ((wave * KS + ks) * 2 + 1) * 1024 + lane * 16) = fl.q;
        }
    }
    float U[8];
#pragma unroll
    for (int qt = 0; qt < 8; ++qt) {
        const uint32_t tl = 32u * qt + c;
        U[qt] = tl < task_count ? a.task_U[task_begin + tl] : INFINITY;
    }
    __syncthreads();

    const int64_t t_begin = row_begin >> 5, t_end = (row_end + 31) >> 5;
    const char* lane_planes = a.frag_planes + lane * 16;
    const char* lane_lds = lds + lane * 16;

    // the tile's 32 row norms travel as one float per lane and are spread to the accumulator layout through 128 B of
    // wave-private LDS (16 registers per buffered tile otherwise)
    float* wave_norms = reinterpret_cast<float*>(lds + 8 * KS * 2048) + wave * kGTileRows;

    auto load_tile = [&](Frag (&Ah)[KS], Frag (&Al)[KS], float& nrm, int64_t T) {
        const char* base = lane_planes + (size_t)T * kGTileBytes;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Ah[ks].q = *reinterpret_cast<const uint4*>(base + ks * 1024);
            Al[ks].q = *reinterpret_cast<const uint4*>(base + kGPlaneBytes + ks * 1024);
        }
        if (METRIC == GPANN_METRIC_L2) nrm = a.neg_half_norm[T * kGTileRows + c];
    };

    auto compute_tile = [&](const Frag (&Ah)[KS], const Frag (&Al)[KS], float nrm, int64_t T) {
        if (METRIC == GPANN_METRIC_L2) {
            if (h == 0) wave_norms[c] = nrm;
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t cmask[4] = {0u, 0u, 0u, 0u};   // bit r (+16 for the odd query tile) of cmask[qp]: accumulator register r passed
        // one pass of the k-loop for query tiles 2 qp and (PAIR) 2 qp + 1: two independent accumulator chains
        auto pass = [&](auto pair_tag, int qp) {
            constexpr bool PAIR = decltype(pair_tag)::value;
            f32x16 acc0, acc1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 n = METRIC == GPANN_METRIC_L2 ? *reinterpret_cast<const float4*>(wave_norms + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc0[4 * g + 0] = n.x; acc0[4 * g + 1] = n.y; acc0[4 * g + 2] = n.z; acc0[4 * g + 3] = n.w;
                if (PAIR) {
                    acc1[4 * g + 0] = n.x; acc1[4 * g + 1] = n.y; acc1[4 * g + 2] = n.z; acc1[4 * g + 3] = n.w;
                }
            }
            const char* q0 = lane_lds + (2 * qp * KS) * 2048;
            const char* q1 = lane_lds + ((2 * qp + 1) * KS) * 2048;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag bh0, bl0, bh1, bl1;
                bh0.q = *reinterpret_cast<const uint4*>(q0 + ks * 2048);
                bl0.q = *reinterpret_cast<const uint4*>(q0 + ks * 2048 + 1024);
                if (PAIR) {
                    bh1.q = *reinterpret_cast<const uint4*>(q1 + ks * 2048);
                    bl1.q = *reinterpret_cast<const uint4*>(q1 + ks * 2048 + 1024);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bh0.v, acc0, 0, 0, 0);
                if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bh1.v, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bl0.v, acc0, 0, 0, 0);
                if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bl1.v, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ks].v, bh0.v, acc0, 0, 0, 0);
                if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ks].v, bh1.v, acc1, 0, 0, 0);
            }
            float m0 = acc0[0], m1 = PAIR ? acc1[0] : -INFINITY;
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                m0 = fmaxf(m0, acc0[r]);
                if (PAIR) m1 = fmaxf(m1, acc1[r]);
            }
            if (__any(m0 >= U[2 * qp] || m1 >= U[2 * qp + 1])) {
                uint32_t bits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bits |= acc0[r] >= U[2 * qp] ? 1u << r : 0u;
                    if (PAIR) bits |= acc1[r] >= U[2 * qp + 1] ? 0x10000u << r : 0u;
                }
                cmask[qp] = bits;
            }
        };
#pragma unroll
        for (int qp = 0; qp < 4; ++qp) {
            if (2 * qp + 1 < nqt) pass(std::true_type{}, qp);
            else if (2 * qp < nqt) pass(std::false_type{}, qp);
        }
        if (__any((cmask[0] | cmask[1] | cmask[2] | cmask[3]) != 0u)) {
#pragma unroll
            for (int qp = 0; qp < 4; ++qp) {
                uint32_t bits = cmask[qp];
                while (bits) {
                    const int b = __builtin_ctz(bits);
                    bits &= bits - 1;
                    const int r = b & 15;
                    const int64_t row = T * kGTileRows + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const uint32_t tl = 32u * (2 * qp + (b >> 4)) + c;
                    if (row >= row_begin && row < row_end && tl < task_count) {
                        const uint32_t task = task_begin + tl;
                        const uint32_t slot = atomicAdd(&a.cand_cnt[task], 1u);
                        if (slot < a.cand_cap) a.cand_row[(size_t)task * a.cand_cap + slot] = (uint32_t)row;
                    }
                }
            }
        }
    };

    Frag A0h[KS], A0l[KS], A1h[KS], A1l[KS];
    float n0 = 0.f, n1 = 0.f;
    int64_t T = t_begin + wave;
    if (T < t_end) load_tile(A0h, A0l, n0, T);
    while (T < t_end) {
        if (T + 8 < t_end) load_tile(A1h, A1l, n1, T + 8);
        compute_tile(A0h, A0l, n0, T);
        T += 8;
        if (T >= t_end) break;
        if (T + 8 < t_end) load_tile(A0h, A0l, n0, T + 8);
        compute_tile(A1h, A1l, n1, T);
        T += 8;
    }
}

// f32 rows -> bf16 hi/lo planes in MFMA A-operand order (kGTileBytes per 32 rows); one thread per (row, 8 dimensions)
static __global__ void __launch_bounds__(256) SplitFragPlanesKernel(const float* points, uint64_t rows, uint32_t ld, char* planes) {
    const uint64_t total = rows * 16;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i >> 4;
        const uint32_t g8 = (uint32_t)(i & 15);   // dims 8 g8 .. 8 g8 + 7
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (8 * g8 < ld) {
            const float4 x = *reinterpret_cast<const float4*>(points + r * ld + 8 * g8);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        }
        if (8 * g8 + 4 < ld) {
            const float4 x = *reinterpret_cast<const float4*>(points + r * ld + 8 * g8 + 4);
            v[4] = x.x; v[5] = x.y; v[6] = x.z; v[7] = x.w;
        }
        Frag fh, fl;
#pragma unroll
        for (int j = 0; j < 4; ++j) SplitBf16(v[2 * j], v[2 * j + 1], fh.u[j], fl.u[j]);
        const uint32_t ks = g8 >> 1, h = g8 & 1, lane = (uint32_t)(r & 31) + 32 * h;
        char* base = planes + (r >> 5) * (uint64_t)kGTileBytes + ks * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(base) = fh.q;
        *reinterpret_cast<uint4*>(base + kGPlaneBytes) = fl.q;
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
};

template <int METRIC>
__global__ void ThresholdKernel(ThresholdArgs a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.tasks) return;
    const double tau = (double)a.list_dist[(size_t)t * a.k + a.k - 1];
    if (!(tau < (double)INFINITY)) {   // list not full yet (or NaN): everything is a candidate
        a.task_U[t] = -INFINITY;
        return;
    }
    const float* q = a.queries + (size_t)a.task_query[t] * a.ld;
    double nq = 0.0;
    for (uint32_t i = 0; i < a.d; ++i) nq += (double)q[i] * (double)q[i];
    const double R = a.shard_rmax[a.task_shard[t]];
    const double qn = sqrt(nq) * (1.0 + 1e-12);
    const double dd = (double)a.d;
    double U;
    if (METRIC == GPANN_METRIC_L2) {
        const double g1 = (dd + 32.0) * 0x1p-23;
        const double c2 = 2.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24 + 0x1p-20;
        // hi plane only: |(q_h + q_l).(p_l + r_p) + r_q.p| <= (2^-9 * 1.01 + 2^-17) |q||p| of the dot product is missing, twice that in D
        const double e_hi = a.hi_only ? 2.0 * 1.02 * 0x1p-9 * qn * R : 0.0;
        const double e2 = c2 * (R + qn) * (R + qn) + e_hi + 1e-30;
        U = -(tau * (1.0 + 2.0 * g1) - nq * (1.0 - 0x1p-40) + e2) * 0.5;
    } else {
        const double e1 = (dd + 4.0) * 0x1p-23 * (qn * R + 1.0) + fabs(tau) * 0x1p-22;
        const double e_hi = a.hi_only ? 1.02 * 0x1p-9 * qn * R : 0.0;
        const double e2 = (4.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24) * qn * R + e_hi + 1e-30;
        U = 1.0 - tau - e1 - e2;
    }
    float Uf = (float)U;
    if ((double)Uf > U) Uf = nextafterf(Uf, -INFINITY);
    a.task_U[t] = Uf;
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
    const uint32_t* task_shard;   // for the rare overflowed bucket: the task's shard ...
    const int64_t* shard_offsets; // ... and the pass's row range inside it, [offset + range_lo, min(end, offset + range_hi))
    int64_t range_lo, range_hi;
    uint32_t tasks, k, d, ld, cand_cap;
    uint32_t dedupe;              // 1: skip candidates already in the list (cell passes)
};

// number of tasks whose candidate bucket overflowed in this pass
static __global__ void OverflowCountKernel(const uint32_t* cand_cnt, uint32_t tasks, uint32_t cap, uint32_t* out, unsigned long long* total) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tasks) return;
    const uint32_t c = cand_cnt[t];
    if (c > cap) atomicAdd(out, 1u);
    // wave-level sum, one atomic per wavefront
    unsigned long long v = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(total, v);
}

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

// one wavefront per task: lane i evaluates candidate i (canonical distance against the query row in LDS), then the
// survivors are inserted into the task's list (one entry per lane)
template <int METRIC>
__global__ void __launch_bounds__(256) RescoreKernel(RescoreArgs a) {
    extern __shared__ float rsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * 4 + wave;
    if (t >= a.tasks) return;
    const uint32_t cnt = a.cand_cnt[t];
    if (cnt == 0) return;
    const bool overflowed = cnt > a.cand_cap;   // rare: this wave then evaluates the pass's whole row range of the shard itself
    float* sq = rsm + (size_t)wave * a.ld;
    const float* q = a.queries + (size_t)a.task_query[t] * a.ld;
    for (uint32_t i = lane; i < a.ld; i += 64) sq[i] = q[i];
    __builtin_amdgcn_wave_barrier();
    const uint32_t k = a.k;
    float ld_ = INFINITY;
    uint32_t lr_ = kNoId;
    if (lane < (int)k) {
        ld_ = a.list_dist[(size_t)t * k + lane];
        lr_ = a.list_row[(size_t)t * k + lane];
    }
    int64_t scan_lo = 0, scan_hi = 0;
    if (overflowed) {
        const uint32_t b = a.task_shard[t];
        scan_lo = min(a.shard_offsets[b + 1], a.shard_offsets[b] + a.range_lo);
        scan_hi = a.range_hi == INT64_MAX ? a.shard_offsets[b + 1] : min(a.shard_offsets[b + 1], a.shard_offsets[b] + a.range_hi);
    }
    const uint64_t n_items = overflowed ? (uint64_t)(scan_hi - scan_lo) : cnt;
    for (uint64_t base = 0; base < n_items; base += 64) {
        const bool valid = base + lane < n_items;
        uint32_t row = kNoId;
        float e = INFINITY;
        if (valid) {
            row = overflowed ? (uint32_t)(scan_lo + (int64_t)(base + lane)) : a.cand_row[(size_t)t * a.cand_cap + base + lane];
            e = ExactDistance<METRIC>(a.points + (size_t)row * a.ld, sq, a.d);
        }
        float td = __shfl(ld_, k - 1);
        uint32_t tr = __shfl(lr_, k - 1);
        bool pass = valid && PairLess(e, row, td, tr);
        unsigned long long mask = __ballot(pass);
        while (mask) {
            const int l = __builtin_ctzll(mask);
            const float cd = __shfl(e, l);
            const uint32_t cr = __shfl(row, l);
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
            pass = pass && lane != l && PairLess(e, row, td, tr);
            mask = __ballot(pass);
        }
    }
    if (lane < (int)k) {
        a.list_dist[(size_t)t * k + lane] = ld_;
        a.list_row[(size_t)t * k + lane] = lr_;
    }
}

// final lists -> caller arrays: out[task_out[t]][i] = (dist, permutation[row], row)
static __global__ void ListOutputKernel(const float* list_dist, const uint32_t* list_row, const uint32_t* task_out, const uint32_t* permutation,
                                 uint32_t tasks, uint32_t k, float* out_dist, uint32_t* out_ids, uint32_t* out_rows) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (size_t)tasks * k) return;
    const uint32_t t = (uint32_t)(i / k), e = (uint32_t)(i % k);
    const size_t dst = (size_t)task_out[t] * k + e;
    const uint32_t row = list_row[i];
    if (out_dist) out_dist[dst] = list_dist[i];
    if (out_rows) out_rows[dst] = row;
    if (out_ids) out_ids[dst] = row == kNoId ? kNoId : permutation[row];
}

// row norms for the filter: -|p|^2/2 as float (from a double sum), per-shard max |p| via an ordered-bits atomic max, and a
// finiteness / magnitude check.  One wavefront per row.
static __global__ void __launch_bounds__(256) RowNormKernel(const float* points, uint64_t rows, uint32_t d, uint32_t ld, const int64_t* shard_offsets,
                                                     uint32_t S, float* neg_half_norm, unsigned long long* shard_max_bits, uint32_t* bad) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_id = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave_id; r < rows; r += n_waves) {
        const float* p = points + r * ld;
        double s = 0.0;
        for (uint32_t i = lane; i < d; i += 64) s += (double)p[i] * (double)p[i];
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
