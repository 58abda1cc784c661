// router_bound.h — KMeansTreeRouter::Query when the budget covers the whole tree, through the matrix cores (included by router.hip).
//
// Every node is then visited (kmeans_tree_router.cpp:112-123), so min_dist[shard] is the minimum of the canonical distance over
// ALL centroids of the shard.  Evaluating nq x C canonical distances on the vector ALUs is what DenseDistKernel<.., MIN_ONLY>
// does (1.8 ms for 10k queries x 19k centroids).  Here the matrix cores first BOUND every (query, centroid) pair and only the
// pairs that can be the minimum get the canonical evaluation:
//   pass 1   acc = q.c - |c|^2/2 (MIPS: q.c) with the bf16 hi/lo split of scan_filter.h (3 products), max per (query, shard)
//   margin   thr[q][shard] = max - (what |acc - true value| and the canonical distance's own rounding can hide)
//   pass 2   the same products again; pairs with acc >= thr become candidates (typically one or two per (query, shard))
//   exact    canonical distance of the candidates, atomicMin on order-preserving bits -> min_dist
// The true minimiser always passes: if c* minimises the canonical distance E and c~ maximises acc, then
//   L2:   D(c*) <= D(c~) (1 + g1) / (1 - g1) <= D(c~) (1 + 3 g1)   (E = D (1 +- g1), g1 = (d + 32) 2^-23), and with |Dapprox - D| <= e2:
//         acc(c*) >= acc(c~) - (3 g1 Dapprox(c~) + e2 (2 + 3 g1)) / 2
//   MIPS: q.c* >= q.c~ - 2 e1 (rounding of 1 - q.c), acc within e2' of q.c:   acc(c*) >= acc(c~) - 2 e1 - 2 e2'
// with e1, e2, e2' as in ThresholdKernel (R = largest centroid norm).  Kernel layout: FilterKernelCells' (scan_cells.h) — the
// queries' fragments stationary in LDS, the centroids' pre-split planes streamed through registers.
#pragma once

#include "scan_filter.h"

namespace gpann {

struct BoundArgs {
    const float* queries;         // [nq][ld]
    const char* cent_planes;      // centroids in MFMA A-operand order, kGTileBytes per 32 centroids
    const float* cent_nhn;        // [C + 64] -|c|^2/2 (L2)
    const uint32_t* cent_shard;   // [C]
    uint32_t* max_bits;           // [nq][S] order-preserving bits of the largest acc (pass 1)
    const float* thr;             // [nq][S] (pass 2)
    uint32_t* cand_count;         // [1]
    uint2* cand;                  // [cand_cap] (query, centroid)
    uint32_t cand_cap;
    uint32_t nq, C, S, ld, n_seg;
    uint32_t plane_ks;            // k-steps (16 dimensions each) per plane of a centroid tile: 8 (d <= 128) or 16 (d <= 256); a tile = 2 planes of plane_ks KiB
    const float* mu;              // squared L2: the centre the centroid planes were built around ([ld]); nullptr = not centred
    float* table;                 // PASS 3: [nq][table_ld] approximate distances  base[q] - (L2: 2, MIPS: 1) * acc  (router.hip, bounded replay)
    uint64_t table_ld;            //         row stride in floats: a multiple of kGTileRows, >= C
    const float* table_base;      //         [nq]  L2: |q - mu|^2 (float), MIPS: 1
};

__device__ __forceinline__ uint32_t OrderedBitsB(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float FromOrderedBitsB(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// grid = (n_seg, query groups of 32 QT); one workgroup = 8 wavefronts.  QT query tiles of 32 queries: 8 up to KS = 8, 4 for KS = 16 (129 - 200
// dimensions: the query image, QT * KS * 2 KiB, has to fit LDS).
template <int METRIC, int KS, int PASS, int QT = (KS <= 8 ? 8 : 4)>
__global__ void __launch_bounds__(512, 1) RouterBoundKernel(BoundArgs a) {
    TakeWholeRegisterFile();   // (common.h: no wavefront of another launch beside a matrix-core wavefront)
    extern __shared__ uint4 bsm4[];
    char* lds = reinterpret_cast<char*>(bsm4);   // [QT query tiles][KS][2 planes][1 KiB] + per-wave norms
    const uint32_t q_base = blockIdx.y * (32u * QT);
    const uint32_t q_count = min(32u * QT, a.nq - q_base);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const uint32_t ld = a.ld;
    const int nqt = (int)((q_count + 31u) / 32u);

    if (wave < QT) {   // the query image: wavefront w builds query tile w (zeros for absent queries)
        const uint32_t ql = 32u * wave + c;
        BuildQueryTile<KS>(lds + (size_t)wave * KS * 2048, a.queries + (size_t)(q_base + min(ql, q_count - 1)) * ld, ql < q_count, ld, lane, a.mu);
    }
    __syncthreads();

    const int64_t n_tiles = ((int64_t)a.C + kGTileRows - 1) / kGTileRows;
    const int64_t per = (n_tiles + a.n_seg - 1) / a.n_seg;
    const int64_t b_begin = (int64_t)blockIdx.x * per, b_end = min(n_tiles, b_begin + per);
    // every wavefront takes a CONTIGUOUS run of the workgroup's tiles: consecutive tiles mostly belong to one shard, so the running
    // maximum (pass 1) / the threshold (pass 2) of a (query, shard) pair stays in a register and global memory is touched only
    // when the shard changes
    const int64_t wper = (b_end - b_begin + 7) / 8;
    const int64_t t_begin = b_begin + (int64_t)wave * wper, t_end = min(b_end, t_begin + wper);
    const char* lane_planes = a.cent_planes + lane * 16;
    const char* lane_lds = lds + lane * 16;
    float* wave_norms = reinterpret_cast<float*>(lds + QT * KS * 2048) + wave * kGTileRows;
    const size_t plane_bytes = (size_t)a.plane_ks * 1024, tile_bytes = 2 * plane_bytes;

    // candidates (pass 2) go to the global list in chunks of 256 slots per wavefront: one atomic on the shared counter per chunk
    // instead of one per candidate (400k same-address atomics were most of this kernel's time); unused slots stay (~0, ~0)
    uint32_t chunk_base = 0, chunk_used = 256;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    auto append = [&](bool pass, uint32_t q, uint32_t cidx) {
        const unsigned long long m = __ballot(pass);
        if (m == 0) return;
        const uint32_t need = (uint32_t)__popcll(m);
        if (chunk_used + need > 256u) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(a.cand_count, 256u);
            chunk_base = (uint32_t)__shfl((int)b, 0);
            chunk_used = 0;
        }
        if (pass) {
            const uint32_t slot = chunk_base + chunk_used + (uint32_t)__popcll(m & lanes_below);
            if (slot < a.cand_cap) a.cand[slot] = make_uint2(q, cidx);
        }
        chunk_used += need;
    };
    uint32_t cur_shard = 0xFFFFFFFFu;
    float run[QT];   // pass 1: running max of the current shard; pass 2: its threshold; pass 3: the query's table base
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) run[qt] = PASS == 1 ? -INFINITY : INFINITY;
    if (PASS == 3) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) run[qt] = a.table_base[q_base + min(32u * qt + c, q_count - 1)];
    }
    auto flush = [&]() {
        if (PASS != 1 || cur_shard == 0xFFFFFFFFu) return;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const uint32_t ql = 32u * qt + c;
            if (qt < nqt && h == 0 && ql < q_count && run[qt] > -INFINITY)
                atomicMax(&a.max_bits[(size_t)(q_base + ql) * a.S + cur_shard], OrderedBitsB(run[qt]));
            run[qt] = -INFINITY;
        }
    };

    auto load_tile = [&](Frag (&Ah)[KS], Frag (&Al)[KS], float& nrm, int64_t T) {
        const char* base = lane_planes + (size_t)T * tile_bytes;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            Ah[ks].q = *reinterpret_cast<const uint4*>(base + ks * 1024);
            Al[ks].q = *reinterpret_cast<const uint4*>(base + plane_bytes + ks * 1024);
        }
        if (METRIC == GPANN_METRIC_L2) nrm = a.cent_nhn[T * kGTileRows + c];
    };

    auto compute_tile = [&](const Frag (&Ah)[KS], const Frag (&Al)[KS], float nrm, int64_t T) {
        const uint32_t c0 = (uint32_t)T * kGTileRows;
        if (METRIC == GPANN_METRIC_L2) {
            if (h == 0) wave_norms[c] = nrm;
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t shard_first = 0, shard_last = 0;
        if (PASS != 3) {
            shard_first = a.cent_shard[min(c0, a.C - 1)];
            shard_last = a.cent_shard[min(c0 + 31u, a.C - 1)];
        }
        const bool one_shard = shard_first == shard_last && c0 + 31u < a.C;
        if (PASS != 3 && one_shard && shard_first != cur_shard) {
            flush();
            cur_shard = shard_first;
            if (PASS == 2) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const uint32_t ql = 32u * qt + c;
                    run[qt] = (qt < nqt && ql < q_count) ? a.thr[(size_t)(q_base + ql) * a.S + cur_shard] : INFINITY;
                }
            }
        }
        // two query tiles per pass of the k-loop: two independent accumulator chains keep the matrix core busy
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) {
            if (2 * qp >= nqt) break;
            f32x16 acc[2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 n = METRIC == GPANN_METRIC_L2 ? *reinterpret_cast<const float4*>(wave_norms + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][4 * g + 0] = n.x; acc[e][4 * g + 1] = n.y; acc[e][4 * g + 2] = n.z; acc[e][4 * g + 3] = n.w;
                }
            }
            const char* p0 = lane_lds + (2 * qp * KS) * 2048;
            const char* p1 = lane_lds + ((2 * qp + 1) * KS) * 2048;   // (an absent odd tile is all zeros with no valid query)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag bh0, bl0, bh1, bl1;
                bh0.q = *reinterpret_cast<const uint4*>(p0 + ks * 2048);
                bl0.q = *reinterpret_cast<const uint4*>(p0 + ks * 2048 + 1024);
                bh1.q = *reinterpret_cast<const uint4*>(p1 + ks * 2048);
                bl1.q = *reinterpret_cast<const uint4*>(p1 + ks * 2048 + 1024);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bh0.v, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bh1.v, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bl0.v, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[ks].v, bl1.v, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ks].v, bh0.v, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[ks].v, bh1.v, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int qt = 2 * qp + e;
                const uint32_t ql = 32u * qt + c;
                const bool q_ok = ql < q_count;
                if (PASS == 3) {   // the table: 32 consecutive distances of a query's row per tile, 16 bytes per lane and store
                    if (q_ok) {
                        float* row = a.table + (size_t)(q_base + ql) * a.table_ld + c0 + 4 * h;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float4 v;
                            if (METRIC == GPANN_METRIC_L2) {
                                v.x = run[qt] - 2.f * acc[e][4 * g + 0]; v.y = run[qt] - 2.f * acc[e][4 * g + 1];
                                v.z = run[qt] - 2.f * acc[e][4 * g + 2]; v.w = run[qt] - 2.f * acc[e][4 * g + 3];
                            } else {
                                v.x = run[qt] - acc[e][4 * g + 0]; v.y = run[qt] - acc[e][4 * g + 1];
                                v.z = run[qt] - acc[e][4 * g + 2]; v.w = run[qt] - acc[e][4 * g + 3];
                            }
                            *reinterpret_cast<float4*>(row + 8 * g) = v;
                        }
                    }
                    continue;
                }
                const size_t qrow = (size_t)(q_base + min(ql, q_count - 1)) * a.S;
                if (one_shard) {
                    float m = acc[e][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[e][r]);
                    if (PASS == 1) {
                        m = fmaxf(m, __shfl_xor(m, 32));
                        run[qt] = fmaxf(run[qt], m);
                    } else if (__any(m >= run[qt] && q_ok)) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) append(acc[e][r] >= run[qt] && q_ok, q_base + ql, c0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                    }
                } else {   // a tile across shard boundaries (or the last, partly filled one): shard by shard (centroids are numbered
                           // shard by shard, so the tile holds the shards shard_first .. shard_last)
                    uint32_t rs[16];   // shard of every accumulator row (0xFFFFFFFF beyond the last centroid)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t cidx = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        rs[r] = cidx < a.C ? a.cent_shard[cidx] : 0xFFFFFFFFu;
                    }
                    for (uint32_t b = shard_first; b <= shard_last; ++b) {
                        float m = -INFINITY;
#pragma unroll
                        for (int r = 0; r < 16; ++r) m = rs[r] == b ? fmaxf(m, acc[e][r]) : m;
                        if (PASS == 1) {
                            m = fmaxf(m, __shfl_xor(m, 32));
                            if (h == 0 && q_ok && m > -INFINITY) atomicMax(&a.max_bits[qrow + b], OrderedBitsB(m));
                        } else {
                            const float t = a.thr[qrow + b];
                            if (__any(m >= t && q_ok)) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) append(rs[r] == b && acc[e][r] >= t && q_ok, q_base + ql, c0 + (r & 3) + 8 * (r >> 2) + 4 * h);
                            }
                        }
                    }
                }
            }
        }
    };

    // (one register set: the workgroup's other wavefronts cover the load latency; a second set spills)
    Frag Ah[KS], Al[KS];
    float nrm = 0.f;
    for (int64_t T = t_begin; T < t_end; ++T) {
        load_tile(Ah, Al, nrm, T);
        compute_tile(Ah, Al, nrm, T);
    }
    flush();
}

// thr[q][shard] from the pass-1 maxima; |q|^2 in double per query.  One workgroup (64 threads) per query.
// Always-on check of pass 1 (no extra launch): every shard that has centroids must have received a maximum (max_bits != 0: the array
// is zeroed before pass 1 and OrderedBits of any float is non-zero) and its threshold must be finite — otherwise pass 2 would let
// nothing (NaN) or everything through.  Violations are counted into *bad_pairs, the same counter the unset-minima check after pass 2
// feeds: the caller then evaluates every pair canonically (or fails under GPANN_ROUTER_STRICT).
template <int METRIC>
__global__ void __launch_bounds__(64) BoundMarginKernel(const float* queries, const uint32_t* max_bits, float* thr, uint32_t S, uint32_t d, uint32_t ld,
                                                        double cent_rmax, const float* mu, const uint32_t* shard_has_cent, uint32_t* bad_pairs) {
    const uint32_t q = blockIdx.x;
    const float* row = queries + (size_t)q * ld;
    double nq = 0.0;
    for (uint32_t i = threadIdx.x; i < d; i += 64) {
        const float v = mu ? row[i] - mu[i] : row[i];   // the frame of the planes (BuildQueryTile)
        nq += (double)v * (double)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nq += __shfl_xor(nq, o);
    const double qn = sqrt(nq) * (1.0 + 1e-12), R = cent_rmax, dd = (double)d;
    for (uint32_t b = threadIdx.x; b < S; b += 64) {
        const uint32_t mb = max_bits[(size_t)q * S + b];
        const double m = (double)FromOrderedBitsB(mb);
        double t;
        if (METRIC == GPANN_METRIC_L2) {
            const double g1 = (dd + 32.0) * 0x1p-23;
            const double c2 = 2.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24 + 0x1p-20 + 0x1p-21;   // (+ 2^-21: rounding of the centred coordinates)
            const double e2 = c2 * (R + qn) * (R + qn) + 1e-30;
            const double Dap = fmax(nq - 2.0 * m, 0.0) + e2;
            t = m - 0.5 * (3.0 * g1 * Dap + e2 * (2.0 + 3.0 * g1)) - 1e-30;
        } else {
            const double e1 = (dd + 4.0) * 0x1p-23 * (qn * R + 1.0) + (1.0 + qn * R) * 0x1p-22;
            const double e2 = (4.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24) * qn * R + 1e-30;
            t = m - 2.0 * e1 - 2.0 * e2;
        }
        float tf = (float)t;
        if ((double)tf > t) tf = nextafterf(tf, -INFINITY);
        thr[(size_t)q * S + b] = tf;
        if (shard_has_cent[b] != 0u && (mb == 0u || !(fabsf(tf) < INFINITY))) atomicAdd(bad_pairs, 1u);
    }
}

// The bounded replay's frame of a query (router.hip): the base of its table row (PASS 3 writes  base - 2 acc  /  base - acc) and eps[q], a
// bound on |table value - canonical distance| for EVERY centroid:
//   L2:   |(nq - 2 acc) - D| <= e2 = c2 (R + qn)^2 as above; nq as a float and the rounding of the subtraction add <= 2^-22 (R + qn)^2;
//         the canonical float sequence is within g1 D <= g1 (R + qn)^2 of D
//   MIPS: |acc - q.c| <= e2', the canonical 1 - q.c within e1 (as above), the rounding of 1 - acc <= 2^-23 (1 + qn R)
// times (1 + 2^-6): the consumers add and subtract multiples of eps in float.  h_lo / h_scale: the interval that holds every table value
// of the query, cut into 2^20 bins.
template <int METRIC>
__global__ void __launch_bounds__(64) BoundFrameKernel(const float* queries, uint32_t d, uint32_t ld, double cent_rmax, const float* mu, float* base, float* eps,
                                                       float* h_lo, float* h_scale, double eps_scale) {
    const uint32_t q = blockIdx.x;
    const float* row = queries + (size_t)q * ld;
    double nq = 0.0;
    for (uint32_t i = threadIdx.x; i < d; i += 64) {
        const float v = mu ? row[i] - mu[i] : row[i];
        nq += (double)v * (double)v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nq += __shfl_xor(nq, o);
    if (threadIdx.x != 0) return;
    const double qn = sqrt(nq) * (1.0 + 1e-12), R = cent_rmax, dd = (double)d;
    double e, lo, span;   // every table value of the query lies in [lo + e, lo + span - e]
    if (METRIC == GPANN_METRIC_L2) {
        const double g1 = (dd + 32.0) * 0x1p-23;
        const double c2 = 2.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24 + 0x1p-20 + 0x1p-21 + 0x1p-22;
        e = (c2 + g1) * (R + qn) * (R + qn);
        base[q] = (float)nq;
    } else {
        const double e1 = (dd + 4.0) * 0x1p-23 * (qn * R + 1.0) + (1.0 + qn * R) * 0x1p-22;
        const double e2 = (4.0 * 0x1p-16 + 8.0 * (dd + 16.0) * 0x1p-24) * qn * R;
        e = e1 + e2 + 0x1p-23 * (1.0 + qn * R);
        base[q] = 1.0f;
    }
    e = e * (1.0 + 0x1p-6) * eps_scale + 1e-30;   // (eps_scale >= 1: a test knob — a wider band must give the same bits)
    float ef = (float)e;
    if ((double)ef < e) ef = nextafterf(ef, INFINITY);
    eps[q] = ef;
    if (METRIC == GPANN_METRIC_L2) {
        lo = -2.0 * e;
        span = (R + qn) * (R + qn) + 4.0 * e;
    } else {
        lo = 1.0 - qn * R - 2.0 * e;
        span = 2.0 * qn * R + 4.0 * e;
    }
    // the bins of the bounded replay (router.hip, BinOf): 2^20 over the span
    h_lo[q] = (float)lo;
    h_scale[q] = (float)(1048574.0 / span);
}

// canonical distance of every candidate pair -> min per (query, shard)
template <int METRIC>
__global__ void ExactCandidatesKernel(const float* centroids, const float* queries, const uint2* cand, const uint32_t* cand_count, uint32_t cand_cap,
                                      const uint32_t* cent_shard, uint32_t* out_min_bits, uint32_t S, uint32_t d, uint32_t ld) {
    const uint32_t n = min(*cand_count, cand_cap);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint2 qc = cand[i];
        if (qc.x == 0xFFFFFFFFu) continue;   // unused slot of a wavefront's chunk
        const float e = ExactDistanceVec4<METRIC>(centroids + (size_t)qc.y * ld, queries + (size_t)qc.x * ld, d);
        atomicMin(&out_min_bits[(size_t)qc.x * S + cent_shard[qc.y]], OrderedBitsB(e));
    }
}

// centroids -> bf16 hi/lo planes in MFMA A-operand order (as SplitFragPlanes of the shard store), and -|c|^2/2
static __global__ void __launch_bounds__(256) CentroidPlanesKernel(const float* centroids, uint64_t C, uint32_t d, uint32_t ld, char* planes, float* nhn,
                                                                   const float* mu, uint32_t plane_ks) {
    const uint32_t g_per_row = 2 * plane_ks;   // groups of 8 dimensions per row
    const uint64_t total = C * g_per_row;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / g_per_row;
        const uint32_t g8 = (uint32_t)(i - r * g_per_row);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (8 * g8 < ld) {
            const float4 x = *reinterpret_cast<const float4*>(centroids + r * ld + 8 * g8);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            if (mu) {
                const float4 m = *reinterpret_cast<const float4*>(mu + 8 * g8);
                v[0] -= m.x; v[1] -= m.y; v[2] -= m.z; v[3] -= m.w;
            }
        }
        if (8 * g8 + 4 < ld) {
            const float4 x = *reinterpret_cast<const float4*>(centroids + r * ld + 8 * g8 + 4);
            v[4] = x.x; v[5] = x.y; v[6] = x.z; v[7] = x.w;
            if (mu) {
                const float4 m = *reinterpret_cast<const float4*>(mu + 8 * g8 + 4);
                v[4] -= m.x; v[5] -= m.y; v[6] -= m.z; v[7] -= m.w;
            }
        }
        Frag fh, fl;
#pragma unroll
        for (int j = 0; j < 4; ++j) SplitBf16(v[2 * j], v[2 * j + 1], fh.u[j], fl.u[j]);
        const uint32_t ks = g8 >> 1, h = g8 & 1, lane = (uint32_t)(r & 31) + 32 * h;
        char* base = planes + (r >> 5) * ((uint64_t)plane_ks * 2048) + ks * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(base) = fh.q;
        *reinterpret_cast<uint4*>(base + (uint64_t)plane_ks * 1024) = fl.q;
        if (g8 == 0) {
            double s = 0.0;
            for (uint32_t x = 0; x < d; ++x) {
                const float v = mu ? centroids[r * ld + x] - mu[x] : centroids[r * ld + x];
                s += (double)v * (double)v;
            }
            nhn[r] = (float)(-0.5 * s);
        }
    }
}

}  // namespace gpann
