// scan_cells.h — the filtered exact scan: cell pruning + the MFMA kernel (see cells.hip for what a cell is and why skipping it is
// exact, scan_filter.h for the bound / candidate / re-scoring scheme).
//
// Per scan call:
//   CellDistKernel    |q - c|^2 of every task against every cell of its shard (the exact kernels' 4x4 register tile; plain fp32)
//   NearestCellKernel the nearest cell of every task; TaskReorderKernel then groups the tasks of a shard by it so that the 32-task
//                     query tiles of the MFMA kernel are spatially coherent
//   NeedKernel        which (task group, cell) blocks must be filtered, as a bit per 32-task query tile:
//                       head pass: the first tiles of each task's nearest cell (tightens the k-th distance cheaply)
//                       sample / rest passes: every cell whose bounding ball reaches the task's current k-th distance
//   FilterKernelCells the MFMA kernel.  Queries stationary, points streamed:
//     * the bf16 hi/lo fragments of a group's <= 256 tasks are built ONCE per block into LDS (128 KiB, in MFMA B-operand order: a
//       ds_read_b128 with lane-linear addresses, conflict free);
//     * the points are pre-split in HBM in MFMA A-operand order (scan_filter.h), so a wavefront loads a tile's fragments with
//       perfectly coalesced global_load_dwordx4 straight into the registers the MFMA reads.  No LDS traffic for the points, no
//       barrier in the loop: the 8 wavefronts of a workgroup take tiles round robin and run independently, each prefetching
//       its next tile (unconditionally — see the tile loop) into a second register set while the matrix core works on the current one.  (An earlier design staged
//       POINT tiles in LDS by LDS-DMA with the queries in registers: a barrier per tile, and the fill landing in LDS and the
//       ds_read/MFMA stream slowed each other down — its time was fill-only + compute-only.)
//     * only the query tiles whose bit is set run on a row tile; candidates of a tile are collected as bit masks and pushed, after
//       the tile's MFMA phase, into a per-wavefront queue in LDS as (task, plane row) pairs; the queue is emptied — plane rows mapped
//       back to store rows, a bucket slot per candidate — when it is full and at the end of the kernel (the global round trips of
//       a candidate, with their in-order vmcnt waits, stay out of the tile loop);
//     * FilterItemsKernel lists a pass's work — every (task group, cell) block with something left to run, cut into pieces of at most
//       seg_tiles row tiles, one list per task group — and one persistent workgroup per CU claims items from its group's list.
#pragma once

#include "scan_filter.h"

namespace gpann {

struct CellGroup {
    uint32_t task_begin, task_count, shard, pad;
};

struct CellItem {   // everything FilterKernelCells needs to run one piece of a (task group, cell) block: one 32-byte record
    uint32_t t_begin, t_end;   // row tiles of this piece
    uint32_t t_a, t_b;         // the cell's head segment ends at tile t_a, its sample segment at t_b
    uint32_t task_begin;
    uint32_t lo_shift;         // the lo plane of row tile T is tile T - lo_shift of the lo array (two-plane passes only; shards.h)
    uint32_t acts;             // query tiles to run on the head | sample << 8 | rest << 16 segment | (tasks of the group - 1) << 24
    uint32_t group;            // the task group (its query image)
    // 8-bit passes (MODE 2): the (task group, cell) block — row of the threshold table q8_B — and the cell's quantisation step
    uint32_t block;
    float cell_scale;
    uint32_t hi_shift;         // bf16 passes: the hi plane of row tile T is tile T - hi_shift of the hi array (shards.h: compact hi planes)
    uint32_t pad1;             // (48 bytes: three 16-byte loads)
};

struct CellTileRef {
    uint32_t shard, first, count, pad;   // <= 64 consecutive tasks of one shard
};

struct CellDistArgs {
    const float* queries;               // [nq][ld]
    const uint32_t* task_query;         // [tasks]
    const float* center;                // [cells][ld]
    const uint32_t* shard_cell_begin;   // [S+1]
    const CellTileRef* tiles;
    float* out;                         // [tasks][kmax]
    uint32_t kmax, d, ld;
    float* out_l1;                      // WITH_L1: [tasks][kmax] |q - c|_1 (plain fp32; the 8-bit passes' error bound lives in the frame of the cell, Q8Threshold)
};

// 64 tasks x 64 cells per pass of the cell loop; squared L2 distance in plain fp32 (bounds carry their own slack)
template <bool WITH_L1>
static __global__ void __launch_bounds__(256, 3) CellDistKernel(CellDistArgs a) {
    constexpr int kStride = 36;
    __shared__ float sC[64 * kStride];
    __shared__ float sQ[64 * kStride];
    const uint32_t shard = a.tiles[blockIdx.x].shard, first = a.tiles[blockIdx.x].first, count = a.tiles[blockIdx.x].count;
    const uint32_t cell0 = a.shard_cell_begin[shard], K = a.shard_cell_begin[shard + 1] - cell0;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;   // cells ty + 16 i, tasks tx + 16 j
    const uint32_t ld = a.ld;
    const uint32_t nchunks = (ld + 31) / 32;
    const int st_row0 = tid >> 3, st_row1 = (tid + 256) >> 3, st_c4 = tid & 7;
    const float* qrow0 = a.queries + (size_t)a.task_query[first + min((uint32_t)st_row0, count - 1)] * ld;
    const float* qrow1 = a.queries + (size_t)a.task_query[first + min((uint32_t)st_row1, count - 1)] * ld;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // grid.y = cell tiles of 64 (a workgroup per (task tile, cell tile): a rank of an 8-GPU run has ~20 task tiles — with the cell loop inside
    // the workgroup the launch ran on 20 CUs for 0.18 ms)
    for (uint32_t ct = 64u * blockIdx.y; ct < K; ct += 64u * gridDim.y) {
        const float* crow0 = a.center + (size_t)(cell0 + min(ct + (uint32_t)st_row0, K - 1)) * ld;
        const float* crow1 = a.center + (size_t)(cell0 + min(ct + (uint32_t)st_row1, K - 1)) * ld;
        float acc[4][4], acc1[WITH_L1 ? 4 : 1][WITH_L1 ? 4 : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = 0.f;
                if (WITH_L1) acc1[WITH_L1 ? i : 0][WITH_L1 ? j : 0] = 0.f;
            }
        for (uint32_t ch = 0; ch < nchunks; ++ch) {
            const uint32_t col = ch * 32 + 4u * st_c4;
            const bool in = col < ld;
            // (every lane loads — past the row's end its first columns again — and the VALUES are replaced by zeros: `in ? *p : zero4` is a select
            // between a global address and the address of a private constant, i.e. four flat loads and a scratch slot per wavefront)
            const uint32_t lcol = in ? col : 0u;
            float4 gc0 = *reinterpret_cast<const float4*>(crow0 + lcol), gc1 = *reinterpret_cast<const float4*>(crow1 + lcol);
            float4 gq0 = *reinterpret_cast<const float4*>(qrow0 + lcol), gq1 = *reinterpret_cast<const float4*>(qrow1 + lcol);
            if (!in) gc0 = gc1 = gq0 = gq1 = zero4;
            __syncthreads();
            *reinterpret_cast<float4*>(sC + st_row0 * kStride + 4 * st_c4) = gc0;
            *reinterpret_cast<float4*>(sC + st_row1 * kStride + 4 * st_c4) = gc1;
            *reinterpret_cast<float4*>(sQ + st_row0 * kStride + 4 * st_c4) = gq0;
            *reinterpret_cast<float4*>(sQ + st_row1 * kStride + 4 * st_c4) = gq1;
            __syncthreads();
#pragma unroll 2
            for (int g = 0; g < 8; ++g) {
                float4 cv[4], qv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[i] = *reinterpret_cast<const float4*>(sC + (ty + 16 * i) * kStride + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) qv[j] = *reinterpret_cast<const float4*>(sQ + (tx + 16 * j) * kStride + 4 * g);
                AccumulateGroup<GPANN_METRIC_L2>(acc, cv, qv);   // pad columns are zero on both sides
                if (WITH_L1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc1[WITH_L1 ? i : 0][WITH_L1 ? j : 0] += (fabsf(qv[j].x - cv[i].x) + fabsf(qv[j].y - cv[i].y)) + (fabsf(qv[j].z - cv[i].z) + fabsf(qv[j].w - cv[i].w));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t cell = ct + ty + 16 * i, tl = tx + 16 * j;
                if (cell < K && tl < count) {
                    a.out[(size_t)(first + tl) * a.kmax + cell] = acc[i][j];
                    if (WITH_L1) a.out_l1[(size_t)(first + tl) * a.kmax + cell] = acc1[WITH_L1 ? i : 0][WITH_L1 ? j : 0];
                }
            }
    }
}

// nearest cell (smallest |q - c|^2, first minimum) and |q|^2 of every task: 16 lanes per task (a thread per task walked its table
// row and its query row alone: 84 us for 10k tasks)
static __global__ void __launch_bounds__(256) NearestCellKernel(const float* cell_dist, const uint32_t* task_shard, const uint32_t* shard_cell_begin,
                                                                const float* queries, const uint32_t* task_query, uint32_t tasks, uint32_t kmax, uint32_t d,
                                                                uint32_t ld, uint32_t* nearest, double* task_nq, const float* mu, double* task_q1, float* task_tscale) {
    const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint32_t sub = threadIdx.x & 15u;
    if (t >= tasks) return;   // (whole groups of 16 lanes leave together: the shuffles below stay inside a group)
    const uint32_t b = task_shard[t];
    const uint32_t K = shard_cell_begin[b + 1] - shard_cell_begin[b];
    uint32_t best = 0xFFFFFFFFu;
    float bd = INFINITY;
    for (uint32_t j = sub; j < K; j += 16) {
        const float v = cell_dist[(size_t)t * kmax + j];
        if (v < bd) {   // (ascending j per lane: the lane's first minimum)
            bd = v;
            best = j;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float od = __shfl_xor(bd, o, 16);
        const uint32_t ob = (uint32_t)__shfl_xor((int)best, o, 16);
        if (od < bd || (od == bd && ob < best)) {
            bd = od;
            best = ob;
        }
    }
    if (sub == 0) nearest[t] = best == 0xFFFFFFFFu ? 0u : best;   // (no finite entry: cell 0, as the serial loop had it)
    if (task_nq) {
        const float* q = queries + (size_t)task_query[t] * ld;
        double nq = 0.0, q1 = 0.0;
        float qmax = 0.f;
        for (uint32_t i = sub; i < d; i += 16) {
            const float v = mu ? q[i] - mu[i] : q[i];   // the frame of the planes (BuildQueryTile)
            nq += (double)v * (double)v;
            q1 += fabs((double)v);
            qmax = fmaxf(qmax, fabsf(v));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            nq += __shfl_xor(nq, o, 16);   // (a fixed tree: the same bits in every run)
            q1 += __shfl_xor(q1, o, 16);
            qmax = fmaxf(qmax, __shfl_xor(qmax, o, 16));
        }
        if (sub == 0) {
            task_nq[t] = nq;
            // the 8-bit query planes (QueryImageQ8Kernel): |q~|_1 and the quantisation step max |q~_i| / 127 (rounded up, never zero)
            if (task_q1) task_q1[t] = q1;
            if (task_tscale) {
                const double st = fmax((double)qmax * (1.0 + 1e-6) / 127.0, 1e-30);
                float sf = (float)st;
                if ((double)sf < st) sf = nextafterf(sf, INFINITY);
                task_tscale[t] = sf;
            }
        }
    }
}

// Tasks of one shard -> grouped by nearest cell (counting sort; the order inside a cell does not matter).  One workgroup per shard.
// out_src[new position] = old position: the task x cell distance table, the nearest cells and |q|^2 stay in the OLD order and are read
// through it (a second CellDistKernel / NearestCellKernel round in the new order cost more than the indirection).
static __global__ void __launch_bounds__(256) TaskReorderKernel(const uint32_t* shard_task_begin, const uint32_t* nearest, const uint32_t* shard_cell_begin,
                                                                const uint32_t* in_query, const uint32_t* in_out, uint32_t* out_query, uint32_t* out_out,
                                                                uint32_t* out_src) {
    extern __shared__ uint32_t s_cnt[];   // [K + 1]
    const uint32_t b = blockIdx.x;
    const uint32_t t0 = shard_task_begin[b], t1 = shard_task_begin[b + 1];
    const uint32_t K = shard_cell_begin[b + 1] - shard_cell_begin[b];
    if (t1 == t0) return;
    for (uint32_t j = threadIdx.x; j <= K; j += blockDim.x) s_cnt[j] = 0;
    __syncthreads();
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) atomicAdd(&s_cnt[min(nearest[t], K - 1) + 1], 1u);
    __syncthreads();
    if (threadIdx.x == 0)
        for (uint32_t j = 0; j < K; ++j) s_cnt[j + 1] += s_cnt[j];   // s_cnt[j] = first position of cell j
    __syncthreads();
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
        const uint32_t pos = t0 + atomicAdd(&s_cnt[min(nearest[t], K - 1)], 1u);
        out_query[pos] = in_query[t];
        out_out[pos] = in_out[t];
        out_src[pos] = t;
    }
}

// The per-call counters and tables of the filtered pipeline, set in ONE launch (seven runtime memsets were seven launches): zero words,
// +inf keys of the queries' merge bounds, +inf floats of the 8-bit threshold table.
struct FilterInitArgs {
    uint32_t* zero[5];
    uint64_t zero_words[5];
    uint32_t* query_tau;     // [nq] -> OrderedKey(+inf), or null
    uint64_t nq;
    uint32_t* inf_floats;    // the (task, cell) threshold table of the 8-bit pass -> +inf, or null
    uint64_t inf_words;
    float* fill_dist;        // the call's output arrays -> (+inf, NO_ID, NO_ID): slots no task writes (skipped probes, short shards); each may be null
    uint32_t* fill_ids;
    uint32_t* fill_rows;
    uint64_t fill_n;
};
static __global__ void __launch_bounds__(256) FilterInitKernel(FilterInitArgs a) {
    const uint64_t i0 = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int z = 0; z < 5; ++z)
        for (uint64_t i = i0; i < a.zero_words[z]; i += stride) a.zero[z][i] = 0u;
    for (uint64_t i = i0; i < a.nq && a.query_tau; i += stride) a.query_tau[i] = kOrderedKeyInf;
    for (uint64_t i = i0; i < a.inf_words && a.inf_floats; i += stride) a.inf_floats[i] = 0x7F800000u;
    for (uint64_t i = i0; i < a.fill_n; i += stride) {
        if (a.fill_dist) a.fill_dist[i] = INFINITY;
        if (a.fill_ids) a.fill_ids[i] = kNoId;
        if (a.fill_rows) a.fill_rows[i] = kNoId;
    }
}

struct NeedArgs {
    const float* cell_dist;             // [tasks][kmax]
    const uint32_t* task_shard;
    const uint32_t* task_group;         // [tasks]
    const CellGroup* groups;
    const uint32_t* shard_cell_begin;
    const float* radius;                // [cells]
    const float* cell_cnorm;            // [cells] |c|^2 (MIPS)
    const double* shard_rmax;           // [S]
    const double* task_nq;              // [tasks] |q|^2 (double sum; indexed through task_src)
    const float* list_dist;             // [tasks][k]
    const uint32_t* nearest;            // [tasks]
    const uint32_t* task_src;           // [tasks] row of cell_dist / nearest / task_qnorm that belongs to task t (the order before TaskReorderKernel)
    uint32_t* need;                     // [groups][kmax] bit per query tile
    uint32_t tasks, kmax, k, d;
    uint32_t head;                      // 1: only the nearest cell of each task
    // the pass's per-task set-up rides along (thread j == 0 of every task): its threshold U, its candidate counter back to zero; thread
    // 0 of the launch zeroes the item cursor / count of the pass
    float* task_U;
    uint32_t* cand_cnt;
    uint32_t* group_zero;               // [2 n_groups]: the item counts and claim cursors of the task groups, zeroed here for the pass
    uint32_t n_groups;
    uint32_t hi_only;
    // GPANN_SCAN_FOR_MERGE: the query's merge bound (OrderedKey, [nq]) caps the tau of each of its tasks
    const uint32_t* query_tau;
    const uint32_t* task_query;
    // [tasks] != 0: the task's bucket overflowed in an earlier pass (RescoreArgs::dirty_flag) — its list comes from the exact evaluation of its
    // whole shard after the last pass, so the remaining passes leave it out: no cell, a threshold nothing passes.  (Left in with the loose
    // bound it was stuck with, ONE such task of 10 000 cost 2.4 ms of a 9 ms step at C5: cells nobody else needed, buckets that overflowed again.)
    const uint32_t* dirty_flag;
    // 8-bit pass (q8_B != nullptr): the per-(task, cell) thresholds of FilterKernelCells MODE 2, [groups][kmax][group_tasks] floats — +inf
    // where the task does not need the cell (the array is filled with +inf before the launch)
    float* q8_B;
    const float* q8_scale;              // [cells] quantisation step s_c
    const double* q8_cnorm;             // [cells] |c^|^2
    const double* task_q1;              // [tasks] |q~|_1 (through task_src)
    const float* task_tscale;           // [tasks] query quantisation step (through task_src)
    uint32_t group_tasks;
    const float* cell_l1;               // [tasks][kmax] |q - c|_1 (CellDistKernel<true>; squared L2 stores whose 8-bit plane has its own norm words), or null
    uint32_t q8_own_norm;               // the 8-bit plane has its own norm words (shards.h q8_norm): without cell_l1 the error term comes from Cauchy-Schwarz
};

// The threshold of the 8-bit pass for task t and cell j (derivation in DESIGN.md §3.1c).  With p~ = c^ + s P' + e (|e_i| <= s/2) and
// q~ = t_q (H + L / 254) + f (|f_i| <= t_q / 508), I = 254 H.P' + L.P' (exact integers):
//     q~.p~ = q~.c^ + (s t_q / 254) I + E,   |E| <= (s / 2) |q~|_1 + (t_q / 508) (sqrt(d) r' + d s / 2)  =: E_tc
// (s |P'|_1 <= |p~ - c^|_1 + d s / 2 <= sqrt(d) r' + d s / 2;  r' = r + 2^-23 (R + |c^|) bounds |p~ - c^| — r, the cell's radius, is taken in
// the original coordinates, the second term is the rounding of the centred ones).  A row whose canonical distance is within tau has
// true acc* = q~.p~ - |p~|^2/2 >= U_base (ThresholdOf without the bf16 terms), so it passes   (s t_q / 254) I + nhn_p >= B  with
//     B = U_base - q~.c^ - E_tc - (table error) - (fp32 evaluation slack),   q~.c^ = (|q~|^2 + |c^|^2 - |q~ - c^|^2) / 2  from the cell table.
template <int METRIC>
__device__ __forceinline__ float Q8Threshold(double tau, double nq, double q1, double tscale, double R, double dd, double D_table, double nc, double s_c,
                                             double radius, double l1_table = -1.0) {
    if (!(tau < (double)INFINITY)) return -INFINITY;
    const double qn = sqrt(nq) * (1.0 + 1e-12);
    double Ub;
    if (METRIC == GPANN_METRIC_L2) {
        const double g1 = (dd + 32.0) * 0x1p-23;
        // 2^-21: rounding of the centred coordinates; 2^-20: the stored norm word and slack (as ThresholdOf)
        const double e2 = (0x1p-20 + 0x1p-21) * (R + qn) * (R + qn) + 1e-30;
        Ub = -(tau * (1.0 + 2.0 * g1) - nq * (1.0 - 0x1p-40) + e2) * 0.5;
    } else {
        const double e1 = (dd + 4.0) * 0x1p-23 * (qn * R + 1.0) + fabs(tau) * 0x1p-22;
        Ub = 1.0 - tau - e1 - 1e-30;
    }
    const double qc = 0.5 * (nq + nc - D_table);
    // the table entry is a plain fp32 sum of d squared fp32 differences of the ORIGINAL coordinates: relative error (d + 2) 2^-24, plus the
    // difference between |q - c|^2 and |q~ - c^|^2 (roundings of the centred coordinates: <= 2^-22 (|q~| + |c^|) sqrt(D))
    const double cn = sqrt(nc);
    const double e_tab = 0.5 * ((dd + 8.0) * 0x1p-23 * fabs(D_table) + 0x1p-21 * (qn + cn) * sqrt(fabs(D_table)) + 0x1p-40 * (nq + nc));
    // `radius` bounds |p - c| in the ORIGINAL coordinates (CellRadiusKernel); the plane holds the residuals of the CENTRED fp32 coordinates
    // p~ = fl(p - mu), c^ = fl(c - mu):  |p~ - c^| <= |p - c| + 2^-24 (|p - mu| + |c - mu|) <= radius + 2^-23 (R + |c^|)   (R, |c^| are norms in the
    // centred frame, each within 2^-24 of the exact one).  For a cell of (nearly) equal rows far from the store's mean this term IS the radius.
    const double r_eff = radius + 0x1p-23 * (R + cn);
    // The row's quantisation error e (|e_i| <= s/2) meets the query as q~.e.  With the plain norm words (-|p~|^2/2; inner product: none) all of
    // it is unknown: |q~.e| <= (s/2) |q~|_1.  With the 8-bit plane's OWN norm words (squared L2, round 4: -|p~|^2/2 + c^.e, shards.h) the part c^.e
    // is in the row's word and only (q~ - c^).e is left: <= (s/2) |q~ - c^|_1, the query seen from the CELL — 2-3x smaller for a query inside its
    // cluster.  |q~ - c^|_1 <= l1_table (1 + (d + 8) 2^-24) + 2^-23 sqrt(d) (|q~| + |c^|): the table is a plain fp32 sum of d fp32 terms
    // |q_i - c_i| in the ORIGINAL coordinates, the second term the rounding of the centred ones.
    // (l1_table == -2: own norm words but no table of |q - c|_1 — Cauchy-Schwarz on the table of squared distances, |q - c|_1 <= sqrt(d) |q - c|_2,
    // whose fp32 error (d + 2) 2^-24 D is covered by the factor; about a quarter looser than the exact sum for Gaussian residuals)
    const double q1_eff = l1_table >= 0.0    ? l1_table * (1.0 + (dd + 8.0) * 0x1p-24) + 0x1p-23 * sqrt(dd) * (qn + cn)
                          : l1_table == -2.0 ? sqrt(dd * fabs(D_table)) * (1.0 + (dd + 8.0) * 0x1p-24) + 0x1p-23 * sqrt(dd) * (qn + cn)
                                             : q1;
    const double E = 0.5 * s_c * (1.0 + 1e-6) * q1_eff + tscale / 508.0 * (1.0 + 1e-6) * (sqrt(dd) * r_eff + 0.5 * dd * s_c);
    // fp32 evaluation of (float)I * (s t_q / 254) + nhn: four roundings of magnitudes <= |q~| (r + sqrt(d) s) and R^2 / 2 (the own norm
    // words add |c^.e| <= |c^| sqrt(d) s / 2 to the word's magnitude)
    const double e_fp = 0x1p-21 * (qn * (r_eff + sqrt(dd) * s_c) + 0.5 * R * R + cn * sqrt(dd) * s_c + 1.0);
    // the kernels for up to 128 dimensions evaluate the test on accumulators biased by 1.5 x 2^23 (FilterKernelCells, BIAS): V and the threshold
    // both carry 255 x 12582912 (x s0), three roundings at that magnitude (< 640 s0 in all) and the relative ones of the norm word (in e_fp) and
    // of the shifted threshold itself
    const double B0 = Ub - qc - E - e_tab - e_fp;
    const double B = B0 - 1024.0 * s_c * tscale / 254.0 * (1.0 + 1e-6) - 0x1p-22 * fabs(B0);
    float Bf = (float)B;
    if ((double)Bf > B) Bf = nextafterf(Bf, -INFINITY);
    return Bf;
}

// The cell test (PassSetupKernel, phase 2).  A cell is needed unless its ball provably holds no row within the task's k-th distance tau:
//   L2   canonical E <= tau  =>  D <= tau (1 + 2 g1), g1 = (d + 32) 2^-23 (ThresholdKernel);  D >= (|q - c| - r)^2.
//        |q - c|^2 comes from a plain fp32 sum of d squares: relative error <= (d + 2) 2^-24, covered by the factor (1 - s).
//   MIPS E = fl(1 - fl(q.p)) <= tau  =>  q.p >= 1 - tau - e1 (ThresholdKernel);  q.p <= q.c + |q| r, and
//        q.c = (|q|^2 + |c|^2 - |q - c|^2) / 2 from the same table, absolute error <= s (|q|^2 + |c|^2 + |q - c|^2).
// (Until round 3 NeedKernel — a thread per (task, cell) — and FilterItemsKernel were launches of their own, with SplitMergeKernel before them.)

struct CellFilterArgs {
    const float* queries;
    const uint32_t* task_query;
    const float* task_U;
    uint32_t* cand_cnt;
    uint32_t* cand_row;
    uint32_t cand_cap, ld;
    const char* frag_planes;     // all hi tiles (plane_tile_bytes each), then at +plane_lo_offset all lo tiles
    uint64_t plane_lo_offset;    // where the lo array starts
    uint32_t plane_tile_bytes;   // KiB per tile and plane = the store's k-steps (PlaneKSteps(d))
    const float* plane_norm;
    const uint32_t* plane_perm;
    const CellGroup* groups;
    const uint32_t* shard_cell_begin;
    const uint64_t* cell_tile_begin;
    const uint32_t* cell_lo_begin;   // [cells+1] first lo tile of every cell (CellItem::lo_shift)
    const uint32_t* cell_hi_begin;   // [cells+1] first hi tile of every cell (CellItem::hi_shift)
    const uint32_t* need;        // [groups][kmax] query tiles to run in this pass
    // A cell's tiles form three segments: head [0, head_tiles), sample [head_tiles, max(head_tiles, tiles / sample_div)), rest.
    // Pass p (0 head, 1 sample, 2 rest) covers segments 0..p; done0 / done1 are the query tiles that already ran on segment 0 / 1.
    const uint32_t* done0;
    const uint32_t* done0b;      // (optional) OR-ed into done0
    const uint32_t* done1;
    uint32_t kmax, head_tiles, sample_div, pass;
    unsigned long long* work;    // [0] += (query tile, row tile) products run, [1] / [2] += row tiles streamed by hi-plane / two-plane launches,
                                 // [3] = [1] before the rest pass, [4] / [5] += bf16 / int8 matrix instructions issued, [6] = [5] before the rest pass
    uint32_t n_groups;
    // the pass's work items, written by FilterItemsKernel: a (task group, cell) block's tile range cut into pieces of <= seg_tiles.
    // The items of task group g are items[group_first[g] .. + group_count[g]) (group_first: capacities, computed by the host);
    // group_cursor[g] counts the claims made on them (work stealing, FilterKernelCells).
    CellItem* items;
    const uint32_t* group_first;
    uint32_t* group_count;
    uint32_t* group_cursor;
    uint32_t seg_tiles;
    const char* image;           // [groups][QT query tiles][KS][hi | lo][1 KiB]: the groups' query fragments (QueryImageKernel / QueryImageQ8Kernel)
    // 8-bit pass (MODE 2): the int8 plane (KS KiB per tile, KS = k-steps of 32 dimensions), the tiles' largest norm words, the per-(task,
    // cell) thresholds written by NeedKernel, the tasks' query quantisation steps / 254, the cells' steps (FilterItemsKernel copies them into the items)
    const char* q8_planes;
    const float* q8_norm;        // the 8-bit passes' norm words (squared L2 stores: shards.h q8_norm; else = plane_norm)
    const float2* tile_bound;    // [tiles] (the tile's largest norm word, 127 x its largest sum of |P'| over a row — cells.hip, TileBoundKernel)
    const float* q8_B;           // [groups][kmax][32 QT]
    const float* task_qscale;    // [tasks] t_q / 254, in the CURRENT task order
    const float* q8_scale;       // [cells]
    uint32_t round_tiles;        // wavefronts x FilterUnitTiles of the kernel that runs this pass's items (FilterItemsKernel cuts pieces in whole rounds of the workgroup)
    // 0: persistent workgroups (one per CU, each claims items until the lists are empty).  n > 0: a workgroup claims at most n items and leaves —
    // the launch then has more workgroups than CUs (scan.hip sizes the grid from the item capacity) and a CU changes hands every few items: a
    // launch of ANOTHER stream gets compute units within one item's time instead of at the end of this one (GPANN_SCAN_OVERLAP: the rest pass
    // of call i beside the front of call i + 1).  Price: the claim -> record -> image -> first tile chain once per workgroup instead of once per CU.
    uint32_t wg_item_budget;
};

// wavefronts per workgroup of FilterKernelCells (one workgroup per CU): all of them stream row tiles round robin
constexpr int kFWaves = 8;
// candidates a wavefront of FilterKernelCells collects in LDS before it writes them to the tasks' buckets (what is left of the 160 KiB
// next to a 128 KiB query image and the norms)
constexpr uint32_t kFQueue = 448;
// dynamic LDS of FilterKernelCells
constexpr size_t FilterLdsBytes(size_t image_bytes, int waves = kFWaves) { return image_bytes + waves * 2 * kGTileRows * sizeof(float) + (size_t)waves * kFQueue * 8; }

// The work items of one pass (PassSetupKernel, phase 3): every (task group, cell) block with something left to run, its tile range cut into
// equal pieces of at most seg_tiles row tiles (bounded pieces keep the persistent workgroups of FilterKernelCells balanced to the end of the
// launch; blocks without work never reach them), one list per task group.  Order of the lists is irrelevant for the result.
// ---- one launch per pass (round 4): what SplitMergeKernel + NeedKernel + FilterItemsKernel did in three --------------------------------------
// One workgroup per TASK GROUP (<= 256 tasks of one shard):
//   phase 1  (a thread per task) the pieces of the task's split bucket of the PREVIOUS pass are merged into its list (SplitMergeKernel's
//            job; the pass's statistics ride along), the task's k-th distance tau is taken — capped by the query's merge bound —, its
//            bf16 threshold U is written and its candidate counter zeroed;
//   phase 2  (a thread per (task, cell) pair, tau from LDS) the cell test of NeedKernel: the group's need mask per cell in LDS, the 8-bit
//            pass's per-(task, cell) thresholds to the table;
//   phase 3  (a thread per cell) the group's work items for FilterKernelCells (FilterItemsKernel's job), counted in LDS — one workgroup
//            owns the group's list, so no global atomics — and the group's item count / claim cursor.
// Nothing here can change a result: the lists are exact at any time, and which cells / candidates a pass looks at only has to be a
// superset of what the k-th distances require (a merge bound that another group lowers while this one reads it is merely looser).
struct PassSetupArgs {
    NeedArgs need;
    CellFilterArgs filt;
    RescoreArgs merge;        // the previous pass's buckets (cand_cnt, cand_cap, parts, split_min, split lists), lists, query_tau, statistics
    uint32_t merge_splits;    // 1: the previous launch was a bucket re-scoring whose heavy buckets were split (RescoreKernel, parts > 1)
};
constexpr int kSetupThreads = 1024;   // (40 task groups at the NS workload: the launch lives on threads per group — 256 threads took 78 rounds of the cell loop)
static inline size_t PassSetupLdsBytes(uint32_t kmax) { return ((size_t)kmax + kGTasks + 8) * sizeof(uint32_t); }

// the pieces of task t's split bucket -> its list (the body of SplitMergeKernel's loop): all 64 lanes of a wavefront; returns the list's k-th distance.
// For k <= 16 every piece is fetched up front — lanes 16 q .. 16 q + k - 1 hold piece q (pieces 4 .. 7 in a second register): one memory latency
// per task instead of one per piece (a wavefront of PassSetupKernel merges its tasks one after the other).
__device__ __forceinline__ float MergeSplitPieces(const RescoreArgs& a, uint32_t t, uint32_t n_parts, int lane) {
    const uint32_t k = a.k;
    const float* pd = a.split_dist + ((size_t)t * a.parts) * k;
    const uint32_t* pr = a.split_row + ((size_t)t * a.parts) * k;
    float ld_ = INFINITY;
    uint32_t lr_ = kNoId;
    if (k <= 16 && n_parts <= 8) {
        const uint32_t q = (uint32_t)lane >> 4, e_l = (uint32_t)lane & 15u;
        float d0 = INFINITY, d1 = INFINITY;
        uint32_t r0 = kNoId, r1 = kNoId;
        if (e_l < k && q < n_parts) {
            d0 = pd[(size_t)q * k + e_l];
            r0 = pr[(size_t)q * k + e_l];
        }
        if (e_l < k && q + 4 < n_parts) {
            d1 = pd[(size_t)(q + 4) * k + e_l];
            r1 = pr[(size_t)(q + 4) * k + e_l];
        }
        // piece 0 is the starting list: entry e of it sits in lane e already
        ld_ = lane < (int)k ? d0 : INFINITY;
        lr_ = lane < (int)k ? r0 : kNoId;
        for (uint32_t p = 1; p < n_parts; ++p) {
            for (uint32_t e = 0; e < k; ++e) {
                const int src = (int)(16u * (p & 3u) + e);
                const float cd = p < 4 ? __shfl(d0, src) : __shfl(d1, src);
                const uint32_t cr = p < 4 ? __shfl(r0, src) : __shfl(r1, src);
                const float td = __shfl(ld_, k - 1);
                const uint32_t tr = __shfl(lr_, k - 1);
                if (!PairLess(cd, cr, td, tr)) break;   // (ascending: nothing after it fits either)
                if (__ballot(lane < (int)k && lr_ == cr) != 0ull) continue;
                ListInsertF(ld_, lr_, cd, cr, k, lane);
            }
        }
    } else {
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
                if (!PairLess(cd, cr, td, tr)) break;
                if (__ballot(lane < (int)k && lr_ == cr) != 0ull) continue;
                ListInsertF(ld_, lr_, cd, cr, k, lane);
            }
        }
    }
    if (lane < (int)k) {
        a.list_dist[(size_t)t * k + lane] = ld_;
        a.list_row[(size_t)t * k + lane] = lr_;
    }
    if (a.query_tau && lane == (int)k - 1 && ld_ < INFINITY) atomicMin(&a.query_tau[a.task_query[t]], OrderedKey(ld_));
    return __shfl(ld_, k - 1);
}

// Grid: (kSetupChunks, task groups).  Workgroup (p, g) owns the tasks [32 p, 32 p + 32) of group g — one query tile — for phases 1 and 2
// (with ONE workgroup per group the launch ran on 40 of the 256 CUs at the NS workload and its chains of dependent loads, 20 rounds of
// the cell loop deep, took 0.10 - 0.15 ms); the group's need masks are OR-ed together in global memory with device-scope atomics, and
// PassItemsKernel — the next launch — runs phase 3 per group.
constexpr uint32_t kSetupChunkTasks = 32;   // (one query tile per workgroup: 64 was measured no faster at 10k tasks and 1.5x slower at 1 250 — a rank of an 8-GPU run)
constexpr uint32_t kSetupChunks = kGTasks / kSetupChunkTasks;
template <int METRIC>
static __global__ void __launch_bounds__(kSetupThreads) PassSetupKernel(PassSetupArgs A) {
    extern __shared__ uint32_t psm[];
    const NeedArgs& a = A.need;
    const CellFilterArgs& f = A.filt;
    const uint32_t g = blockIdx.y, chunk = blockIdx.x;
    const uint32_t group_task_begin = a.groups[g].task_begin, group_task_count = a.groups[g].task_count, b = a.groups[g].shard;
    const uint32_t n_chunks = (group_task_count + kSetupChunkTasks - 1) / kSetupChunkTasks;
    if (chunk >= n_chunks) return;
    const uint32_t task_begin = group_task_begin + chunk * kSetupChunkTasks, task_count = min(kSetupChunkTasks, group_task_count - chunk * kSetupChunkTasks);
    const uint32_t cell0 = a.shard_cell_begin[b], K = a.shard_cell_begin[b + 1] - cell0;
    uint32_t* s_misc = psm;                                             // [0] items of the group, [1] overflowed buckets, [2..3] candidates (u64, 8-byte aligned), [4] last arrival
    float* s_tau = reinterpret_cast<float*>(psm + 8);                   // [kGTasks] tau of the chunk's tasks (capped by the merge bound); NaN = dirty task
    uint32_t* s_need = psm + 8 + kGTasks;                               // [kmax] the chunk's need mask per cell (bits of the GROUP's query tiles)
    const int tid = threadIdx.x, lane = tid & 63;
    for (uint32_t j = tid; j < K; j += kSetupThreads) s_need[j] = 0u;
    if (tid < 8) s_misc[tid] = 0u;
    __syncthreads();

    // ---- phase 1a: thread tl <-> task task_begin + tl (the first wavefront): its bucket's statistics, how many pieces the bucket was split into
    uint32_t* s_parts = reinterpret_cast<uint32_t*>(s_tau);   // (the words of s_tau hold the piece counts until the merges are done)
    if (tid < (int)kSetupChunkTasks) {
        const uint32_t tl = (uint32_t)tid, t = task_begin + tl;
        uint32_t parts_l = 0;
        if (tl < task_count && A.merge_splits) {
            const uint32_t cnt = A.merge.cand_cnt[t];
            if (cnt != 0u) {
                atomicAdd(reinterpret_cast<unsigned long long*>(s_misc + 2), (unsigned long long)cnt);
                if (cnt > A.merge.cand_cap) atomicAdd(&s_misc[1], 1u);
                else parts_l = min(A.merge.parts, (cnt + A.merge.split_min - 1) / A.merge.split_min);
            }
        }
        s_parts[tl] = parts_l;
    }
    __syncthreads();
    // ---- phase 1b: the split buckets, spread over the wavefronts of the workgroup (a wavefront per task, one after the other)
    {
        const int wave = tid >> 6;
        constexpr int n_waves = kSetupThreads / 64;
        const uint32_t p = (uint32_t)lane < task_count ? s_parts[lane] : 0u;
        unsigned long long todo = __ballot(p > 1u);
        uint32_t seen = 0;
        while (todo) {
            const int i = __builtin_ctzll(todo);
            todo &= todo - 1;
            if ((int)(seen++ % (uint32_t)n_waves) != wave) continue;
            (void)MergeSplitPieces(A.merge, task_begin + (uint32_t)i, (uint32_t)__shfl((int)p, i), lane);
        }
    }
    __syncthreads();   // (the merged lists are read back below by other threads of the workgroup: written through to L2, and nobody on this CU has read them before)
    // ---- phase 1c: thread tl <-> task: tau (capped by the merge bound), the bf16 threshold, the candidate counter back to zero
    if (tid < (int)kSetupChunkTasks) {
        const uint32_t tl = (uint32_t)tid, t = task_begin + tl;
        if (tl < task_count) {
            float tau = __builtin_nontemporal_load(&a.list_dist[(size_t)t * a.k + a.k - 1]);
            if (a.query_tau) tau = fminf(tau, OrderedKeyToFloat(a.query_tau[a.task_query[t]]));
            const bool dirty = a.dirty_flag && a.dirty_flag[t] != 0u;
            a.task_U[t] = dirty ? INFINITY : ThresholdOf<METRIC>((double)tau, a.task_nq[a.task_src[t]], a.shard_rmax[b], (double)a.d, a.hi_only != 0);
            a.cand_cnt[t] = 0;
            s_tau[tl] = dirty ? __uint_as_float(0x7FC00000u) : tau;
        }
    }
    __syncthreads();

    // ---- phase 2: the cell tests (NeedKernel), cells fastest: a task's row of the distance table is read coalesced.  The 8-bit pass's
    //      threshold table is this pass's own and arrives filled with +inf (FilterInitKernel): only the needed slots are written
    const uint32_t pairs = task_count * K;
    for (uint32_t i = (uint32_t)tid; i < pairs; i += kSetupThreads) {
        const uint32_t tl = i / K, j = i - tl * K, t = task_begin + tl;
        const uint32_t tg = chunk * kSetupChunkTasks + tl;   // the task's number in its GROUP (its query tile: tg >> 5)
        const uint32_t src = a.task_src[t];
        const float tau_f = s_tau[tl];
        const bool dirty = tau_f != tau_f;
        const double tau = (double)tau_f;
        bool need;
        if (dirty || (a.head && a.nearest[src] != j)) {
            need = false;
        } else if (!(tau < (double)INFINITY)) {
            need = true;
        } else {
            const double dd = (double)a.d;
            const double s = (dd + 64.0) * 0x1p-22;
            const double D = (double)a.cell_dist[(size_t)src * a.kmax + j];
            const double r = (double)a.radius[cell0 + j];
            if (METRIC == GPANN_METRIC_L2) {
                const double g1 = (dd + 32.0) * 0x1p-23;
                const double lb = sqrt(fmax(D, 0.0)) * (1.0 - s) - r;
                need = !(lb > 0.0 && lb * lb > tau * (1.0 + 2.0 * g1) * (1.0 + s) + 1e-30);
            } else {
                const double nq = a.task_nq[src] * (1.0 + 1e-6), nc = (double)a.cell_cnorm[cell0 + j];
                const double R = a.shard_rmax[b];
                const double qn = sqrt(nq);
                const double dot_ub = 0.5 * (nq + nc - D) + s * (nq + nc + fabs(D));
                const double e1 = (dd + 4.0) * 0x1p-23 * (qn * R + 1.0) + fabs(tau) * 0x1p-22;
                need = dot_ub + qn * r * (1.0 + s) >= 1.0 - tau - e1 - s;
            }
        }
        if (need) {
            atomicOr(&s_need[j], 1u << (tg >> 5));
            if (a.q8_B)
                a.q8_B[((size_t)g * a.kmax + j) * a.group_tasks + tg] =
                    Q8Threshold<METRIC>(tau, a.task_nq[src], a.task_q1[src], (double)a.task_tscale[src], a.shard_rmax[b], (double)a.d,
                                        (double)a.cell_dist[(size_t)src * a.kmax + j], a.q8_cnorm[cell0 + j], (double)a.q8_scale[cell0 + j], (double)a.radius[cell0 + j],
                                        a.cell_l1 ? (double)a.cell_l1[(size_t)src * a.kmax + j] : (a.q8_own_norm ? -2.0 : -1.0));
        }
    }
    __syncthreads();
    // the chunk's bits join the group's masks (device-scope atomics: the chunks of a group run on any CU of any XCD); the statistics ride along
    for (uint32_t j = (uint32_t)tid; j < K; j += kSetupThreads)
        if (s_need[j] != 0u) atomicOr(&a.need[(size_t)g * a.kmax + j], s_need[j]);
    if (tid == 0 && A.merge_splits && A.merge.stat_overflowed) {
        const unsigned long long total = *reinterpret_cast<unsigned long long*>(s_misc + 2);
        if (total) atomicAdd(A.merge.stat_candidates, total);
        if (s_misc[1]) atomicAdd(A.merge.stat_overflowed, s_misc[1]);
    }
}

// phase 3, a launch of its own (one workgroup per task group): the group's work items from the need masks the chunks have OR-ed together.
// (Run by the last chunk of a group to arrive — a ticket per group behind a device-scope fence — the set-up launch took twice as long: on this
// multi-XCD part an agent-scope release / acquire writes back and invalidates an XCD's whole L2.  The kernel boundary is the cheap fence.)
constexpr int kItemsThreads = 256;
static __global__ void __launch_bounds__(kItemsThreads) PassItemsKernel(PassSetupArgs A) {
    __shared__ uint32_t s_items;
    const NeedArgs& a = A.need;
    const CellFilterArgs& f = A.filt;
    const uint32_t g = blockIdx.x;
    const uint32_t group_task_begin = a.groups[g].task_begin, group_task_count = a.groups[g].task_count, b = a.groups[g].shard;
    const uint32_t cell0 = a.shard_cell_begin[b], K = a.shard_cell_begin[b + 1] - cell0;
    const int tid = threadIdx.x;
    if (tid == 0) s_items = 0u;
    __syncthreads();
    uint32_t* s_misc = &s_items;
    for (uint32_t j = (uint32_t)tid; j < K; j += kItemsThreads) {
        const uint32_t block = g * a.kmax + j;
        const uint32_t mask = a.need[block];   // (later passes read it as their done mask)
        if (mask == 0) continue;
        const uint32_t d0 = (f.done0 ? f.done0[block] : 0u) | (f.done0b ? f.done0b[block] : 0u);
        const uint32_t d1 = f.done1 ? f.done1[block] : 0u;
        const int64_t t_cell = (int64_t)f.cell_tile_begin[cell0 + j];
        const int64_t t_cell_end = (int64_t)f.cell_tile_begin[cell0 + j + 1];
        const int64_t t_a = min(t_cell_end, t_cell + (int64_t)f.head_tiles);
        const int64_t t_b = min(t_cell_end, max(t_a, t_cell + (t_cell_end - t_cell + f.sample_div - 1) / f.sample_div));
        const int64_t t_pass_end = f.pass == 0 ? t_a : (f.pass == 1 ? t_b : t_cell_end);
        const int64_t t_first = (mask & ~d0) ? t_cell : ((mask & ~d1) ? t_a : t_b);
        if (t_first >= t_pass_end) continue;
        const int64_t range = t_pass_end - t_first;
        const uint32_t n = (uint32_t)((range + f.seg_tiles - 1) / f.seg_tiles);
        const int64_t round = (int64_t)f.round_tiles;
        const int64_t per = ((range + n - 1) / n + round - 1) / round * round;
        const uint32_t base = f.group_first[g] + atomicAdd(&s_misc[0], n);
        CellItem it;
        it.t_a = (uint32_t)t_a;
        it.t_b = (uint32_t)t_b;
        it.task_begin = group_task_begin;
        it.lo_shift = (uint32_t)t_cell - f.cell_lo_begin[cell0 + j];
        it.acts = (mask & ~d0) | (mask & ~d1) << 8 | mask << 16 | (group_task_count - 1u) << 24;
        it.group = g;
        it.block = block;
        it.cell_scale = f.q8_scale ? f.q8_scale[cell0 + j] : 0.f;
        it.hi_shift = (uint32_t)t_cell - f.cell_hi_begin[cell0 + j];
        it.pad1 = 0u;
        for (uint32_t i = 0; i < n; ++i) {
            const int64_t bb = min(t_pass_end, t_first + (int64_t)i * per), e = min(t_pass_end, bb + per);
            it.t_begin = (uint32_t)bb;
            it.t_end = (uint32_t)e;
            f.items[base + i] = it;   // (group_first reserves tiles / seg_tiles + 1 >= n entries for every cell of the group)
        }
    }
    __syncthreads();
    if (tid == 0) {
        f.group_count[g] = s_misc[0];
        f.group_cursor[g] = 0u;
        if (g == 0 && f.pass == 2 && f.work) {
            f.work[3] = f.work[1];   // hi-plane row tiles streamed before the rest pass (its own rows = the difference)
            f.work[6] = f.work[5];   // ... and the int8 matrix instructions issued before it
        }
    }
}

// HI_ONLY: the points' lo plane is neither loaded nor multiplied (half the HBM bytes, two of the three products); the
// threshold of such a pass carries the extra error term (ThresholdArgs::hi_only).
// The bf16 hi/lo fragments of every task group's queries in MFMA B-operand order, built once per scan call: one wavefront per
// (group, query tile).  FilterKernelCells copies the tiles it needs into LDS for every work item (LDS-DMA, no registers).
template <int KS, int QT>
static __global__ void __launch_bounds__(64) QueryImageKernel(const float* queries, const uint32_t* task_query, const CellGroup* groups, uint32_t ld,
                                                              char* image, const float* mu) {
    const uint32_t g = blockIdx.x / QT, qt = blockIdx.x - g * QT;
    const int lane = threadIdx.x, c = lane & 31;
    const uint32_t task_begin = groups[g].task_begin, task_count = groups[g].task_count;
    const uint32_t tl = 32u * qt + c;
    if (32u * qt >= task_count) return;   // (tiles past the group's tasks are never read)
    const uint32_t task = task_begin + min(tl, task_count - 1);
    BuildQueryTile<KS>(image + ((size_t)g * QT + qt) * KS * 2048, queries + (size_t)task_query[task] * ld, tl < task_count, ld, lane, mu);
}

// The 8-bit query image of the rest pass: per 32-task query tile and k-step of 32 dimensions two KiB — the planes H and L of
//     q~_i = t_q (H_i + L_i / 254) + f_i,   H = round(q~_i / t_q) in [-127, 127],  L = round((q~_i / t_q - H) * 254) in [-127, 127],  |f_i| <= t_q / 508
// in the B-operand order of v_mfma_i32_32x32x32_i8 (lane (c, h): task c of the tile, dimensions 32 ks + 16 h .. + 15).  t_q = task_tscale
// (NearestCellKernel, old task order: through task_src); task_qscale[t] = t_q / 254 in the new order for the kernel.  One wavefront per (group, query tile).
template <int KS, int QT>
static __global__ void __launch_bounds__(64) QueryImageQ8Kernel(const float* queries, const uint32_t* task_query, const CellGroup* groups, uint32_t ld, char* image,
                                                                const float* mu, const float* task_tscale, const uint32_t* task_src, float* task_qscale) {
    const uint32_t g = blockIdx.x / QT, qt = blockIdx.x - g * QT;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const uint32_t task_begin = groups[g].task_begin, task_count = groups[g].task_count;
    const uint32_t tl = 32u * qt + c;
    if (32u * qt >= task_count) return;
    const bool real = tl < task_count;
    const uint32_t task = task_begin + min(tl, task_count - 1);
    const float* qrow = queries + (size_t)task_query[task] * ld;
    const float tq = task_tscale[task_src[task]];
    if (real && h == 0) task_qscale[task] = tq / 254.0f;   // (exact to one rounding; the threshold's fp32 slack covers it)
    const double inv = 1.0 / (double)tq;
    char* tile = image + ((size_t)g * QT + qt) * KS * 2048;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint32_t wh[4] = {0u, 0u, 0u, 0u}, wl[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t x = 32u * ks + 16u * h + j;
            int H = 0, L = 0;
            if (real && x < ld) {
                const float v = mu ? qrow[x] - mu[x] : qrow[x];
                const double u = (double)v * inv;
                H = max(-127, min(127, (int)rint(u)));
                L = max(-127, min(127, (int)rint((u - (double)H) * 254.0)));
            }
            wh[j >> 2] |= ((uint32_t)H & 0xFFu) << (8 * (j & 3));
            wl[j >> 2] |= ((uint32_t)L & 0xFFu) << (8 * (j & 3));
        }
        *reinterpret_cast<uint4*>(tile + (ks * 2 + 0) * 1024 + lane * 16) = make_uint4(wh[0], wh[1], wh[2], wh[3]);
        *reinterpret_cast<uint4*>(tile + (ks * 2 + 1) * 1024 + lane * 16) = make_uint4(wl[0], wl[1], wl[2], wl[3]);
    }
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS bytes [lds_addr, lds_addr + 1024).  Inline asm:
// the compiler neither sees the LDS bytes it writes nor counts it in vmcnt — the caller waits (DmaWaitAll) before the barrier that
// publishes the data, and issues it BEFORE any load the compiler tracks (a vmcnt(N) wait then only ever waits longer, never shorter).
__device__ __forceinline__ void DmaToLds16(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_addr) : "m0", "memory");
}
__device__ __forceinline__ void DmaWaitAll() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr uint32_t kNoItem = 0xFFFFFFFFu;
// query tiles (of 32 tasks) per task group: the group's image, QT * KS * 2 KiB, has to fit the 160 KiB of LDS beside the norms and the queues
constexpr int FilterQueryTiles(int ks) { return ks <= 8 ? 8 : 4; }
// consecutive row tiles a wavefront of FilterKernelCells takes at a time (its unit of prefetch)
constexpr int FilterUnitTiles(int mode, int ks) { return mode == 2 && ks <= 4 ? 2 : 1; }

// Wavefront 0 of a workgroup (all 64 lanes) looks for the task group with the most unclaimed work items and claims one there.
// Returns the item, or kNoItem once every list is exhausted; g / first / count become the group the claim was made on.
// Every failed claim coincides with a group running dry, so the loop ends after at most n_groups rounds.
__device__ __forceinline__ uint32_t ClaimElsewhere(const uint32_t* group_first, const uint32_t* group_count, uint32_t* group_cursor, uint32_t n_groups,
                                                   int lane, uint32_t& g, uint32_t& first, uint32_t& count) {
    for (;;) {
        uint32_t best = 0, best_g = kNoItem;
        for (uint32_t i = (uint32_t)lane; i < n_groups; i += 64) {
            const uint32_t cnt = group_count[i];
            const uint32_t cur = __hip_atomic_load(&group_cursor[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t rem = cur < cnt ? cnt - cur : 0u;
            if (rem > best) {
                best = rem;
                best_g = i;
            }
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const uint32_t ob = (uint32_t)__shfl_xor((int)best, off), og = (uint32_t)__shfl_xor((int)best_g, off);
            if (ob > best || (ob == best && og < best_g)) {
                best = ob;
                best_g = og;
            }
        }
        best_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)best_g);
        if (best_g == kNoItem) return kNoItem;
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(&group_cursor[best_g], 1u);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        const uint32_t cnt = group_count[best_g];
        if (idx < cnt) {
            g = best_g;
            first = group_first[best_g];
            count = cnt;
            return first + idx;
        }
    }
}

// KS k-steps of 16 dimensions; QT query tiles of 32 tasks per task group — 8 (256 tasks) up to KS = 8, 4 (128 tasks) for KS = 16: the
// query image, QT * KS * 2 KiB, has to fit the 160 KiB of LDS.  KS = 16 exists for hi-only passes only (both planes of two 16-step
// tiles do not fit the register file).
// MODE 0: both bf16 planes of the points (three products); 1: their hi plane only (two products); 2: the 8-bit plane (scan_q8 notes above
// Q8Threshold; KS then counts k-steps of 32 dimensions and the image holds the int8 planes H, L of the queries).
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
// NW: wavefronts per workgroup (one workgroup per CU).  (Measured on the 8-bit rest pass of the NS workload, profiles/r03/q8_kernel_variants.md:
// 16 wavefronts of 128 registers 2.41 ms, 8 wavefronts 2.20 ms, 8 wavefronts taking two row tiles at a time 2.24 ms.)
constexpr int FilterWaves(int mode, int ks) { return kFWaves; }
template <int METRIC, int TAG, int KS, int MODE, int QT = FilterQueryTiles(KS), int TW_ = FilterUnitTiles(MODE, KS), int NW = FilterWaves(MODE, KS)>
__global__ void __launch_bounds__(64 * NW, 1) FilterKernelCells(CellFilterArgs a) {
    TakeWholeRegisterFile();   // (common.h: no wavefront of another launch beside a matrix-core wavefront)
    constexpr bool HI_ONLY = MODE >= 1, Q8 = MODE == 2;
    static_assert(QT == 4 || QT == 8, "query tiles per group");
    static_assert(KS <= 8 || MODE == 1, "more than 8 k-steps: bf16 hi-plane passes only");
    extern __shared__ uint4 csm4[];
    char* lds = reinterpret_cast<char*>(csm4);   // [8 query tiles][KS][2 planes][1 KiB] + per-wave norms
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    // Persistent workgroups, one per CU, claim work items from the task groups' lists.  Two things keep the time between two items short:
    //  * claims run TWO ahead: while item i streams, the record of item i + 1 is already in scalar registers and the claim for i + 2
    //    is in flight, and the prefetch that would run past the end of item i fetches the first row tile of item i + 1 instead
    //    (claim -> record -> first tile were three dependent memory latencies with nothing in flight);
    //  * a workgroup stays with ONE task group while that group has unclaimed items: its query image (up to 128 KiB, which at the
    //    workgroup's share of the memory system took about as long to copy as the three latencies together) is copied when the group
    //    changes — a few times per launch instead of once per item — and consecutive items of a group need no barrier between them
    //    but the one that hands the next claim to the other wavefronts.  A workgroup whose group runs dry moves to the group with the
    //    most unclaimed items (ClaimElsewhere).
    // (Claims of two consecutive entries, streamed as one range when they are pieces of one block, measured 3 % SLOWER on the rest
    // pass: 4.21 against 4.08 ms.)
    __shared__ uint32_t s_claim[4];
    uint32_t my_products = 0, my_tiles = 0;
    uint32_t my_second = 0;   // (MODE 2) products whose second half — L.P', KS more matrix instructions — was multiplied too
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t ld = a.ld;
    (void)ld;
    const char* lane_planes = a.frag_planes + lane * 16;
    const char* lane_lds = lds + lane * 16;
    float* wave_norms = reinterpret_cast<float*>(lds + QT * KS * 2048) + wave * 2 * kGTileRows;   // (the norm words of up to two row tiles)
    // Candidates go through a per-wavefront queue in LDS: (task, plane row) pairs, mapped to store rows and written to the tasks'
    // buckets — a returning atomic per candidate for the slot — when the queue fills up and at the end of the kernel.  (Written
    // straight from the tile loop, every row tile with a candidate costs two dependent round trips — row id, slot: the head and sample
    // passes, 15 and 4 candidates per row tile, streamed at two thirds of the rest pass's rate.)
    uint2* queue = reinterpret_cast<uint2*>(lds + QT * KS * 2048 + NW * 2 * kGTileRows * sizeof(float)) + wave * kFQueue;
    uint32_t q_len = 0;
    auto flush_queue = [&]() __attribute__((always_inline)) {
        for (uint32_t i = (uint32_t)lane; i < q_len; i += 64) {
            const uint2 e = queue[i];
            const uint32_t row = a.plane_perm[e.y];
            if (row == 0xFFFFFFFFu) continue;   // (padding row of a cell's last tile)
            const uint32_t slot = atomicAdd(&a.cand_cnt[e.x], 1u);
            if (slot < a.cand_cap) a.cand_row[(size_t)e.x * a.cand_cap + slot] = row;
        }
        q_len = 0;
    };

    // A wavefront's unit of work is TW consecutive row tiles (1: two 4 KiB tiles of the 8-bit plane per unit were measured no faster).
    constexpr int TW = TW_;
    Frag Ah[2][TW][KS], Al[HI_ONLY ? 1 : 2][KS];   // (the lo set is never touched when HI_ONLY)
    float nrm[2][TW];
    float2 nmax[2][TW];   // (MODE 2: the tile's largest norm word and its bound of |L.P'|)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int tw = 0; tw < TW; ++tw) {
            nrm[b][tw] = 0.f;
            nmax[b][tw] = make_float2(0.f, 0.f);
        }
    const char* lane_q8 = Q8 ? a.q8_planes + lane * 16 : nullptr;
    // (tiles T .. T + TW - 1: the tile after a range is the next cell's or the zero padding behind the planes — loaded, never multiplied)
    auto load_tile = [&](Frag (&Ah)[TW][KS], Frag (&Al)[KS], float (&nrm)[TW], float2 (&nmax)[TW], uint32_t T, uint32_t T_hi, uint32_t T_lo) __attribute__((always_inline)) {
        if (Q8) {
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) {
                const char* base = lane_q8 + (size_t)(T + tw) * (KS * 1024);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Ah[tw][ks].v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base + ks * 1024));
                if (METRIC == GPANN_METRIC_L2) nrm[tw] = a.q8_norm[(size_t)(T + tw) * kGTileRows + c];
                nmax[tw] = a.tile_bound[T + tw];   // (inner product: .x is 0)
            }
            return;
        }
        const char* base = lane_planes + (size_t)T_hi * a.plane_tile_bytes;
        const char* base_lo = lane_planes + a.plane_lo_offset + (size_t)T_lo * a.plane_tile_bytes;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // (read-once stream: non-temporal, leaves L2 to the query rows / tables)
            Ah[0][ks].v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base + ks * 1024));
            if (!HI_ONLY) Al[ks].v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(base_lo + ks * 1024));
        }
        if (METRIC == GPANN_METRIC_L2) nrm[0] = a.plane_norm[(size_t)T * kGTileRows + c];
    };
    // an item's record as wave-uniform scalars (tile indices are 32-bit all the way: 64-bit copies of these wave-uniform values ended
    // up in vector registers and were spilled — and a scratch reload inside the tile loop has to wait for the tile prefetch issued
    // before it)
    auto uniform = [](uint32_t v) __attribute__((always_inline)) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };

    // claim state: meaningful in wavefront 0 only (wave-uniform there)
    uint32_t c_g = 0, c_first = 0, c_count = 0;
    bool drained = false;
    // wavefront 0: the item behind cursor value idx of group c_g, or a claim on another group
    auto claim_for = [&](uint32_t idx) __attribute__((always_inline)) {
        if (idx < c_count) return c_first + idx;
        if (drained) return kNoItem;
        const uint32_t got = ClaimElsewhere(a.group_first, a.group_count, a.group_cursor, a.n_groups, lane, c_g, c_first, c_count);
        drained = got == kNoItem;
        return got;
    };
    if (wave == 0) {
        c_g = (uint32_t)((uint64_t)blockIdx.x * a.n_groups / gridDim.x);
        c_first = a.group_first[c_g];
        c_count = a.group_count[c_g];
        for (int j = 0; j < 2; ++j) {
            uint32_t idx = 0;
            if (lane == 0 && !drained) idx = atomicAdd(&a.group_cursor[c_g], 1u);
            const uint32_t got = claim_for(uniform(idx));
            if (lane == 0) s_claim[j] = got;
        }
    }
    __syncthreads();
    uint32_t item = uniform(s_claim[0]), nxt = uniform(s_claim[1]);
    if (item == kNoItem) return;
    uint32_t loaded_group = kNoItem, loaded = 0;   // the query image in LDS: whose, and which of its query tiles
    // thresholds of this lane's task in each query tile: U (bf16 modes; filled with the query image) — MODE 2: the task's query step / 254
    // (with the image) and, per ITEM, the (task, cell) threshold Bq and the cell's step
    float U[QT];
    float Bq[Q8 ? QT : 1];
    float cell_scale = 0.f;
    // MODE 2 up to 128 dimensions: the integer accumulators start at 0x4B400000 — the bit pattern of 1.5 x 2^23 — so that an accumulator
    // 0x4B400000 + h (|h| <= 127 x 127 x 128 < 2^22) READ AS A FLOAT is 12582912 + h exactly: the row tests need no int -> float
    // conversion (two of their six vector instructions per accumulator).  The constant sits in 16 registers of its own (the matrix
    // instruction's C operand) and is opaque to the optimiser, which would otherwise rebuild it with 16 moves per product.
    constexpr bool BIAS = Q8 && KS <= 4;
    constexpr int kBiasBits = 0x4B400000;
    constexpr float kBias = 12582912.0f, kBias255 = 3208642560.0f;   // (255 x 12582912 = 765 x 2^22: a float)
    i32x16 acc_init;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_init[r] = BIAS ? kBiasBits : 0;
    if constexpr (BIAS) asm volatile("" : "+v"(acc_init));
    uint32_t q8_hits = 0;   // (MODE 2) products of this wavefront that had a candidate (of my_products: counted per unit of work, so it runs up to a unit ahead)
    uint32_t iter = 0;
    uint32_t r_t_begin, r_t_end, r_t_a, r_t_b, r_task_begin, r_lo_shift, r_acts, r_group, r_block = 0, r_scale_bits = 0, r_hi_shift = 0;
    {
        const uint4 lo = reinterpret_cast<const uint4*>(a.items + item)[0], hi = reinterpret_cast<const uint4*>(a.items + item)[1];
        r_t_begin = uniform(lo.x); r_t_end = uniform(lo.y); r_t_a = uniform(lo.z); r_t_b = uniform(lo.w);
        r_task_begin = uniform(hi.x); r_lo_shift = uniform(hi.y); r_acts = uniform(hi.z); r_group = uniform(hi.w);
        const uint4 x = reinterpret_cast<const uint4*>(a.items + item)[2];
        if (Q8) {
            r_block = uniform(x.x); r_scale_bits = uniform(x.y);
        } else {
            r_hi_shift = uniform(x.z);
        }
    }
    {
        const uint32_t T0 = min(r_t_begin + wave * TW, r_t_end - 1);
        load_tile(Ah[0], Al[0], nrm[0], nmax[0], T0, T0 - r_hi_shift, T0 - r_lo_shift);
    }
    for (;;) {
    // here: r_* is this item's record, its first row tile is on its way to buffer 0, nxt is the item after it (or kNoItem)
    const uint32_t t_begin = r_t_begin, t_end = r_t_end, t_a = r_t_a, t_b = r_t_b;
    const uint32_t acts = r_acts;
    const uint32_t task_begin = r_task_begin, task_count = (acts >> 24) + 1u;
    const uint32_t act_head = acts & 0xFFu, act_sample = acts >> 8 & 0xFFu, mask = acts >> 16 & 0xFFu;
    uint32_t next_idx = 0;
    // (wg_item_budget >= 2: this workgroup leaves after that many items.  `iter` items are done, two more are claimed: the budget is met when
    // iter + 2 reaches it — decided BEFORE the cursor is advanced: an index taken from it is an item that has to be run.  Uniform over the
    // workgroup, and once true it stays true.)
    // (Compiled into the hi-plane and 8-bit kernels only — the ones a rest pass runs on; the two-plane kernel, already at its register limit, keeps its code.)
    const bool budget_met = MODE >= 1 && a.wg_item_budget != 0u && iter + 2u >= a.wg_item_budget;
    if (tid == 0 && !drained && !budget_met) next_idx = atomicAdd(&a.group_cursor[c_g], 1u);
    const CellItem* rn = a.items + (nxt != kNoItem ? nxt : item);
    const uint4 n_lo = reinterpret_cast<const uint4*>(rn)[0], n_hi = reinterpret_cast<const uint4*>(rn)[1];
    const uint4 n_x = reinterpret_cast<const uint4*>(rn)[2];


    // query tiles to run on row tile T.  (Written with masks on purpose: `T < t_a ? act_head : ...`, or the same with ifs, is a
    // conditional between captured variables, which the optimiser turns into a select of ADDRESSES before the closure is
    // dissolved — the closure, and the kernel arguments it refers to, then stay in scratch memory and every global access becomes
    // a flat one.)
    auto act_of = [&](uint32_t T) __attribute__((always_inline)) {
        const uint32_t in_head = T < t_a ? 0xFFu : 0u, in_sample = T < t_b ? 0xFFu & ~in_head : 0u;
        return (act_head & in_head) | (act_sample & in_sample) | (mask & ~(in_head | in_sample));
    };
    // the candidate bits of a row tile (bit r, + 16 for the odd query tile of pair qp: accumulator register r passed) -> the LDS queue
    auto push_candidates = [&](const uint32_t (&cmask)[QT / 2], uint32_t T) __attribute__((always_inline)) {
        uint32_t any_bits = 0;
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) any_bits |= cmask[qp];
        if (!__any(any_bits != 0u)) return;
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) {
            uint32_t bits = cmask[qp];
            while (__any(bits != 0u)) {   // (wave-uniform: every lane takes its lowest set bit, if it has one)
                const bool has = bits != 0u;
                const int b = has ? __builtin_ctz(bits) : 0;
                bits &= bits - 1u;
                const int r = b & 15;
                const uint32_t tl = 32u * (2 * qp + (b >> 4)) + c;
                const uint32_t prow = T * kGTileRows + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h);   // (plane rows fit 32 bits: checked by the host)
                const bool ok = has && tl < task_count;
                const unsigned long long votes = __ballot(ok);
                if (votes != 0ull) {
                    if (q_len + 64u > kFQueue) flush_queue();
                    const uint32_t pos = q_len + __builtin_amdgcn_mbcnt_hi((uint32_t)(votes >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)votes, 0u));
                    if (ok) queue[pos] = make_uint2(task_begin + tl, prow);
                    q_len += (uint32_t)__builtin_popcountll(votes);
                }
            }
        }
    };
    // MODE 2: the test of one product's accumulators h (H.P') and l (L.P') — V = 254 h + l is the exact integer dot product of the quantised
    // vectors, a row passes iff  fl(V) * s0 + nhn_row >= Bq  (Q8Threshold); `norms`: the tile's 32 norm words in LDS.
    //   x = fl(fl(fl(254 h + l)) * s0 + nhn_row) - b0 >= 0,   s0 = the cell's step x the task's step / 254,
    // with 254 h + l formed in fp32 (|h|, |l| < 2^24 are exact, the sum is rounded once: Q8Threshold's fp32 slack).  The sign bits of the
    // 16 x are shifted into one word (v_alignbit), a set bit = "no candidate".  (Formed with integer multiplies, compares and selects the
    // same test took 165 vector instructions per product against 8 matrix instructions.)
    auto full8 = [&](const i32x16& h0, const i32x16& l0, const float* norms, int q0, uint32_t (&cmask)[QT / 2]) __attribute__((always_inline)) {
        // (BIAS: the accumulators read as floats are 12582912 + h, 12582912 + l; fv = fl(255 x 12582912 + V) and the threshold moves by
        // 255 x 12582912 x s0 — Q8Threshold's e_bias covers the roundings at that magnitude)
        const float s0 = cell_scale * U[q0], b0 = BIAS ? __builtin_fmaf(kBias255, s0, Bq[Q8 ? q0 : 0]) : Bq[Q8 ? q0 : 0];
        uint32_t neg = 0;
        // written on pairs of accumulators (the matrix instruction's 16 result registers are consecutive); built without packed fp32 (Makefile) every
        // element is one v_fma_f32 / v_sub_f32 — the same IEEE operation as the packed form computed per lane: same bits
        const f32x2 k254 = {254.0f, 254.0f}, s02 = {s0, s0}, b02 = {b0, b0};
#pragma unroll
        for (int gg = 3; gg >= 0; --gg) {   // (accumulator 15 first: the last sign shifted in, accumulator 0's, is bit 0)
            const float4 n = METRIC == GPANN_METRIC_L2 ? *reinterpret_cast<const float4*>(norms + 8 * gg + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 1; p >= 0; --p) {
                const int r = 4 * gg + 2 * p;
                f32x2 hv, lv;
                if (BIAS) {
                    hv = f32x2{__int_as_float(h0[r]), __int_as_float(h0[r + 1])};
                    lv = f32x2{__int_as_float(l0[r]), __int_as_float(l0[r + 1])};
                } else {
                    hv = f32x2{(float)h0[r], (float)h0[r + 1]};
                    lv = f32x2{(float)l0[r], (float)l0[r + 1]};
                }
                const f32x2 nv = p ? f32x2{n.z, n.w} : f32x2{n.x, n.y};
                const f32x2 fv = __builtin_elementwise_fma(hv, k254, lv);
                const f32x2 x = __builtin_elementwise_fma(fv, s02, nv) - b02;
                neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(x.y), 31);   // neg = neg << 1 | sign(x): accumulator r + 1, then r
                neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(x.x), 31);
            }
        }
        // (a NaN — a padding row's -inf norm against a -inf threshold — may read as a candidate: padding rows are dropped at the flush)
        const uint32_t bits = ~neg & 0xFFFFu;
        q8_hits += __any(bits != 0u) ? 1u : 0u;
        cmask[q0 >> 1] |= bits << (16 * (q0 & 1));
    };
    // ... and what decides whether the product's second half is run at all, while fewer than a quarter of the wavefront's products had a
    // candidate (data without structure — a uniform cloud: one product in nine; the NS rest pass, with a candidate in most products,
    // runs both halves unconditionally): H.P' alone bounds the test value of every row of the tile from above,
    //   x <= fl(fl(fl(254 max h + lb)) * s0 + nmax) - b0,    lb = 127 max_row sum |P'| >= L.P' (|L| <= 127),  nmax = the tile's largest norm word
    // — each operand replaced by one that is no smaller, every operation monotone.  (The rows behind a cell's front are ordered by norm
    // word, so nmax is as good as a row's own; lb costs about 0.13 standard deviations of V.)  No lane >= 0: no row of the tile can pass,
    // the 4 KS matrix instructions of L.P' and the 16 row tests are skipped.
    auto quick8 = [&](const i32x16& h0, float2 bound, int q0) __attribute__((always_inline)) {
        if (!(my_products >= 16u && q8_hits * 4u < my_products)) return true;
        int hm = h0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) hm = max(hm, h0[r]);
        // (BIAS: the largest accumulator is the largest float too; lb moves by 12582912 like every L.P')
        const float s0 = cell_scale * U[q0], b0 = BIAS ? __builtin_fmaf(kBias255, s0, Bq[Q8 ? q0 : 0]) : Bq[Q8 ? q0 : 0];
        const float xub = __builtin_fmaf(__builtin_fmaf(BIAS ? __int_as_float(hm) : (float)hm, 254.0f, BIAS ? bound.y + kBias : bound.y), s0, bound.x) - b0;
        return __any(!(xub < 0.f)) != 0;
    };
    // MODE 2, two row tiles that run the same query tiles: every B fragment read from LDS feeds both tiles' products (an MFMA per KiB of LDS
    // read is what the LDS can just deliver: with one tile per fragment the uniform-cloud rest pass ran the matrix cores at 38 %)
    auto compute_pair8 = [&](const Frag (&A0)[KS], const Frag (&A1)[KS], float nrm0, float nrm1, float2 bound0, float2 bound1, uint32_t T0, uint32_t act)
                             __attribute__((always_inline)) {
        if (METRIC == GPANN_METRIC_L2) {
            if (h == 0) {
                wave_norms[c] = nrm0;
                wave_norms[kGTileRows + c] = nrm1;
            }
            __builtin_amdgcn_wave_barrier();
        }
        my_products += 2u * (uint32_t)__builtin_popcount(act);
        my_tiles += 2;
        uint32_t cm0[QT / 2], cm1[QT / 2];
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) cm0[qp] = cm1[qp] = 0u;
#pragma unroll
        for (int q0 = 0; q0 < QT; ++q0) {
            if (!(act >> q0 & 1u)) continue;
            i32x16 h0 = acc_init, l0 = acc_init, h1 = acc_init, l1 = acc_init;
            const char* p0 = lane_lds + (q0 * KS) * 2048;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i32 a0 = __builtin_bit_cast(v4i32, A0[ks].q), a1 = __builtin_bit_cast(v4i32, A1[ks].q);
                const v4i32 bh = *reinterpret_cast<const v4i32*>(p0 + ks * 2048);
                h0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bh, h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bh, h1, 0, 0, 0);
            }
            const bool run0 = quick8(h0, bound0, q0), run1 = quick8(h1, bound1, q0);
            if (!(run0 || run1)) continue;
            my_second += 2u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i32 a0 = __builtin_bit_cast(v4i32, A0[ks].q), a1 = __builtin_bit_cast(v4i32, A1[ks].q);
                const v4i32 bl = *reinterpret_cast<const v4i32*>(p0 + ks * 2048 + 1024);
                l0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bl, l0, 0, 0, 0);
                l1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bl, l1, 0, 0, 0);
            }
            if (run0) full8(h0, l0, wave_norms, q0, cm0);
            if (run1) full8(h1, l1, wave_norms + kGTileRows, q0, cm1);
        }
        push_candidates(cm0, T0);
        push_candidates(cm1, T0 + 1);
    };
    auto compute_tile = [&](const Frag (&Ah)[KS], const Frag (&Al)[KS], float nrm, float2 nmax, uint32_t T) __attribute__((always_inline)) {
        const uint32_t act = act_of(T);
        if (act == 0) return;
        if (METRIC == GPANN_METRIC_L2) {
            if (h == 0) wave_norms[c] = nrm;
            __builtin_amdgcn_wave_barrier();
        }
        my_products += (uint32_t)__builtin_popcount(act);
        my_tiles += 1;
        uint32_t cmask[QT / 2];   // bit r (+16 for the odd query tile) of cmask[qp]: accumulator register r passed
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) cmask[qp] = 0u;
        // one pass of the k-loop: query tiles q0 and (PAIR) q0 + 1, two independent accumulator chains
        auto pass = [&](auto pair_tag, int q0, int qp, int shift) __attribute__((always_inline)) {
            constexpr bool PAIR = decltype(pair_tag)::value;
            f32x16 acc0, acc1;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float4 n = METRIC == GPANN_METRIC_L2 ? *reinterpret_cast<const float4*>(wave_norms + 8 * gg + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc0[4 * gg + 0] = n.x; acc0[4 * gg + 1] = n.y; acc0[4 * gg + 2] = n.z; acc0[4 * gg + 3] = n.w;
                if (PAIR) {
                    acc1[4 * gg + 0] = n.x; acc1[4 * gg + 1] = n.y; acc1[4 * gg + 2] = n.z; acc1[4 * gg + 3] = n.w;
                }
            }
            const char* p0 = lane_lds + (q0 * KS) * 2048;
            const char* p1 = lane_lds + ((q0 + 1) * KS) * 2048;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag bh0, bl0, bh1, bl1;
                bh0.q = *reinterpret_cast<const uint4*>(p0 + ks * 2048);
                bl0.q = *reinterpret_cast<const uint4*>(p0 + ks * 2048 + 1024);
                if (PAIR) {
                    bh1.q = *reinterpret_cast<const uint4*>(p1 + ks * 2048);
                    bl1.q = *reinterpret_cast<const uint4*>(p1 + ks * 2048 + 1024);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bh0.v, acc0, 0, 0, 0);
                if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bh1.v, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bl0.v, acc0, 0, 0, 0);
                if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bl1.v, acc1, 0, 0, 0);
                if (!HI_ONLY) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ks].v, bh0.v, acc0, 0, 0, 0);
                    if (PAIR) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ks].v, bh1.v, acc1, 0, 0, 0);
                }
            }
            const float u0 = (q0 & 1) ? U[2 * qp + 1] : U[2 * qp];
            const float u1 = U[2 * qp + 1];
            float m0 = acc0[0], m1 = PAIR ? acc1[0] : -INFINITY;
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                m0 = fmaxf(m0, acc0[r]);
                if (PAIR) m1 = fmaxf(m1, acc1[r]);
            }
            if (__any(m0 >= u0 || (PAIR && m1 >= u1))) {
                uint32_t bits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    bits |= acc0[r] >= u0 ? 1u << (r + shift) : 0u;
                    if (PAIR) bits |= acc1[r] >= u1 ? 0x10000u << r : 0u;
                }
                cmask[qp] = bits;
            }
        };
        // MODE 2: the same for the 8-bit planes: per query tile two integer accumulator chains, H.P' and L.P' (test8)
        auto pass8 = [&](int q0) __attribute__((always_inline)) {   // (one query tile: its two chains already keep the matrix core busy)
            i32x16 h0 = acc_init, l0 = acc_init;
            const char* p0 = lane_lds + (q0 * KS) * 2048;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i32 av = __builtin_bit_cast(v4i32, Ah[ks].q);
                const v4i32 bh0 = *reinterpret_cast<const v4i32*>(p0 + ks * 2048);
                h0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bh0, h0, 0, 0, 0);
            }
            if (!quick8(h0, nmax, q0)) return;
            my_second += 1u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i32 av = __builtin_bit_cast(v4i32, Ah[ks].q);
                const v4i32 bl0 = *reinterpret_cast<const v4i32*>(p0 + ks * 2048 + 1024);
                l0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bl0, l0, 0, 0, 0);
            }
            full8(h0, l0, wave_norms, q0, cmask);
        };
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) {
            const uint32_t two = act >> (2 * qp) & 3u;
            if constexpr (Q8) {
                if (two & 1u) pass8(2 * qp);
                if (two & 2u) pass8(2 * qp + 1);
            } else {
                if (two == 3u) pass(std::true_type{}, 2 * qp, qp, 0);
                else if (two == 1u) pass(std::false_type{}, 2 * qp, qp, 0);
                else if (two == 2u) pass(std::false_type{}, 2 * qp + 1, qp, 16);
            }
        }
        push_candidates(cmask, T);
    };

    // Row tiles stream through registers, one tile ahead of the products (with one 512-thread workgroup per CU this, not occupancy,
    // is what covers the HBM latency).  The prefetch is UNCONDITIONAL — past the end of the range it fetches the first tile of the
    // next item (or, after the last item, re-reads the last tile) — because a load under a branch leaves the compiler unable to count
    // the loads in flight at the join, and it then drains them all (s_waitcnt vmcnt(0)) before the first use of the current tile: no
    // overlap at all (measured on the rest pass of the NS workload: 5.33 ms with the branch, 4.86 ms without).  (A third buffer for
    // the hi-only passes was measured slower: spills.)
    // ---- the query image: wavefront w copies query tile w of the group's prebuilt image if this item uses it and it is not there yet
    //      (a new group's image replaces the old one: the barrier at the end of the previous item made sure nobody reads that any more;
    //      further tiles of the SAME group go to slots nobody reads yet)
    const bool same_group = r_group == loaded_group;
    const uint32_t fetch = same_group ? mask & ~loaded : mask;
    if (fetch >> wave & 1u) {
        const char* src = a.image + ((size_t)r_group * QT + wave) * KS * 2048 + lane * 16;
        const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + wave * KS * 2048));
#pragma unroll
        for (int i = 0; i < 2 * KS; ++i) DmaToLds16(src + i * 1024, dst + i * 1024);
    }
    if (!same_group) {   // (a query tile outside the item's mask is never multiplied: its thresholds need no masking)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const uint32_t tl = 32u * qt + c;
            if (Q8) U[qt] = tl < task_count ? a.task_qscale[task_begin + tl] : 0.f;
            else U[qt] = tl < task_count ? a.task_U[task_begin + tl] : INFINITY;
        }
    }
    if (Q8) {   // this item's cell: the (task, cell) thresholds (+inf: the task does not need the cell / no such task) and step products
        cell_scale = __uint_as_float(r_scale_bits);
        const float* brow = a.q8_B + (size_t)r_block * (32u * QT) + c;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) Bq[Q8 ? qt : 0] = brow[32 * qt];
    }
    loaded = same_group ? loaded | mask : mask;
    loaded_group = r_group;
    DmaWaitAll();      // this wavefront's part of the image is in LDS (so are its first tile, the next record and the claim)
    if (wave == 0) {
        const uint32_t got = budget_met ? kNoItem : claim_for(uniform(next_idx));
        if (lane == 0) s_claim[2 + (iter & 1u)] = got;
    }
    __syncthreads();
    const uint32_t after = uniform(s_claim[2 + (iter & 1u)]);
    ++iter;
    const uint32_t n_t_begin = uniform(n_lo.x), n_t_end = uniform(n_lo.y);
    const uint32_t lo_shift = r_lo_shift, n_lo_shift = HI_ONLY ? 0u : uniform(n_hi.y);   // (n_hi is this item's record after the last item)
    const uint32_t hi_shift = r_hi_shift, n_hi_shift = Q8 ? 0u : uniform(n_x.z);
    const uint32_t t_last = t_end - 1;
    // this wavefront's first tile of the next item
    const uint32_t n_first = nxt != kNoItem ? min(n_t_begin + wave * TW, n_t_end - 1) : t_last;
    constexpr uint32_t kStride = NW * TW;   // tiles between two units of a wavefront
    // the tiles of the unit that begins at tile T0, from buffer b
    auto compute_unit = [&](int b, uint32_t T0) __attribute__((always_inline)) {
        if constexpr (Q8 && TW == 2) {
            if (T0 + 1 < t_end) {
                const uint32_t a0 = act_of(T0), a1 = act_of(T0 + 1);
                if (a0 == a1) {   // (the rule: both tiles in one segment of the cell)
                    if (a0 != 0u) compute_pair8(Ah[b][0], Ah[b][TW - 1], nrm[b][0], nrm[b][TW - 1], nmax[b][0], nmax[b][TW - 1], T0, a0);
                    return;
                }
            }
        }
#pragma unroll
        for (int tw = 0; tw < TW; ++tw)
            if (tw == 0 || T0 + tw < t_end) compute_tile(Ah[b][tw], Al[HI_ONLY ? 0 : b], nrm[b][tw], nmax[b][tw], T0 + tw);
    };
    uint32_t T = t_begin + wave * TW;
    if (T < t_end) {
        const uint32_t steps = (t_end - T + kStride - 1) / kStride;
        for (;;) {   // (written out: every exit is a plain break, so the compiler keeps an exact count of the loads in flight)
            uint32_t Tn = T + kStride;
            uint32_t own = Tn <= t_last ? 0xFFFFFFFFu : 0u;
            load_tile(Ah[1], Al[HI_ONLY ? 0 : 1], nrm[1], nmax[1], (Tn & own) | (n_first & ~own), ((Tn - hi_shift) & own) | ((n_first - n_hi_shift) & ~own),
                      ((Tn - lo_shift) & own) | ((n_first - n_lo_shift) & ~own));
            compute_unit(0, T);
            T = Tn;
            if (T >= t_end) break;
            Tn = T + kStride;
            own = Tn <= t_last ? 0xFFFFFFFFu : 0u;
            load_tile(Ah[0], Al[0], nrm[0], nmax[0], (Tn & own) | (n_first & ~own), ((Tn - hi_shift) & own) | ((n_first - n_hi_shift) & ~own),
                      ((Tn - lo_shift) & own) | ((n_first - n_lo_shift) & ~own));
            compute_unit(1, T);
            T = Tn;
            if (T >= t_end) break;
        }
        if (steps & 1u) {   // the next item's first unit went to buffer 1
#pragma unroll
            for (int tw = 0; tw < TW; ++tw) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Ah[0][tw][ks].v = Ah[1][tw][ks].v;
                nrm[0][tw] = nrm[1][tw];
                nmax[0][tw] = nmax[1][tw];
            }
            if (!HI_ONLY) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) Al[0][ks].v = Al[HI_ONLY ? 0 : 1][ks].v;
            }
        }
    } else {
        load_tile(Ah[0], Al[0], nrm[0], nmax[0], n_first, n_first - n_hi_shift, n_first - n_lo_shift);   // (buffer 0 holds a tile of this item that this wavefront had no use for)
    }
    if (nxt == kNoItem) break;
    const uint32_t n_group = uniform(n_hi.w);
    if (n_group != r_group) __syncthreads();   // the next item replaces the query image: everyone has to be done with this one
    r_t_begin = n_t_begin; r_t_end = n_t_end; r_t_a = uniform(n_lo.z); r_t_b = uniform(n_lo.w);
    r_task_begin = uniform(n_hi.x); r_lo_shift = uniform(n_hi.y); r_acts = uniform(n_hi.z); r_group = n_group;
    if (Q8) {
        r_block = uniform(n_x.x);
        r_scale_bits = uniform(n_x.y);
    } else {
        r_hi_shift = n_hi_shift;
    }
    item = nxt;
    nxt = after;
    }   // next item
    // (no wavefront of the workgroup leaves — and frees its registers for another launch's wavefront — while a sibling on the same SIMD may still
    // issue matrix instructions: TakeWholeRegisterFile, common.h.  What follows the barrier is the queue's flush and the counters: no matrix work.)
    __syncthreads();
    flush_queue();
    if (a.work && (threadIdx.x & 63) == 0 && my_tiles) {
        atomicAdd(&a.work[0], (unsigned long long)my_products);
        atomicAdd(&a.work[HI_ONLY ? 1 : 2], (unsigned long long)my_tiles);   // (row tiles of hi-plane launches / of two-plane launches)
        // matrix instructions this wavefront ISSUED: v_mfma_f32_32x32x16_bf16 — KS per plane product, two or three products — into [4],
        // v_mfma_i32_32x32x32_i8 — KS for H.P' of every product, KS more where L.P' was multiplied — into [5]
        if (Q8) atomicAdd(&a.work[5], (unsigned long long)(my_products + my_second) * (unsigned long long)KS);
        else atomicAdd(&a.work[4], (unsigned long long)my_products * (unsigned long long)(KS * (HI_ONLY ? 2 : 3)));
    }
}

}  // namespace gpann
