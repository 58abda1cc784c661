// cells.hip — sub-clusters ("cells") of every shard for the MFMA pre-filter of the exact scan (gfx950).
//
// The reference's QueryBucket streams EVERY row of a probed shard (src/inverted_index.h:64-68).  The result only depends on
// the rows within the k-th distance, so whole groups of rows can be skipped when a bound proves them farther than that.
// At store build every shard is cut into cells by a few Lloyd rounds of the batched k-means engine (kmeans.hip, L2 geometry
// whatever the metric) and the filter's bf16 planes are written cell by cell; per cell the store keeps centre c and radius
// r = max |p - c|.  For a task (query q, current k-th distance tau) every row p of the cell has
//     L2:   |q - p| >= |q - c| - r                      MIPS:   q.p <= q.c + |q| r
// so the cell is skipped when that bound already exceeds tau (scan_cells.h makes the comparison safe against the rounding of
// the canonical binary32 distance).  Exactness is untouched: skipped rows cannot enter the top-k, all other rows go through
// the same filter + exact re-scoring as before.
#include "kmeans.h"
#include "scan_filter.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

namespace gpann {

constexpr uint32_t kNoRow = 0xFFFFFFFFu;

// One piece of a cell's member list — at most kSortChunk rows — ordered by norm word, descending (ties: by row): a bitonic network over
// (key, row) pairs in LDS, one workgroup per piece.  piece = (first position in the member array, rows).
constexpr uint32_t kSortChunk = 16384;   // 8 bytes per pair: 128 KiB of the 160 KiB of LDS
__global__ void __launch_bounds__(1024) SortPieceByNormKernel(const uint2* pieces, const float* neg_half_norm, uint32_t* members) {
    extern __shared__ uint2 pairs[];   // (key bits, row)
    const uint2 piece = pieces[blockIdx.x];
    const uint32_t first = piece.x, count = piece.y;
    uint32_t n = 64;
    while (n < count) n <<= 1;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t row = i < count ? members[first + i] : kNoRow;
        // (norm words are finite and <= 0: -inf pads to the end; floats are compared as floats below)
        pairs[i] = make_uint2(__float_as_uint(i < count ? neg_half_norm[row] : -INFINITY), row);
    }
    __syncthreads();
    // `before(a, b)`: a belongs in front of b
    auto before = [](uint2 a, uint2 b) {
        const float ka = __uint_as_float(a.x), kb = __uint_as_float(b.x);
        return ka > kb || (ka == kb && a.y < b.y);
    };
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint2 a = pairs[i], b = pairs[l];
                    const bool up = (i & k) == 0;   // this pair's run is sorted front-to-back
                    if (before(b, a) == up) {
                        pairs[i] = b;
                        pairs[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) members[first + i] = pairs[i].y;
}

// The cells' member lists with everything behind a cell's front (FilterFrontTiles) ordered by the rows' norm words, descending.  The head and
// sample passes of the filter read the front of a cell and want a random sample of it (slab order is one); the rest pass streams what is
// behind, tile by tile, and its 8-bit kernel first tests a (row tile, query tile) product against the LARGEST norm word of the tile
// (scan_cells.h, quick8): with 32 neighbours of the sorted order in a tile that bound is as good as the row's own word, where a random
// tile's largest word sits about two standard deviations above its typical one (uniform 128-d cloud: 58 % of the products passed that
// test, 12 % held a candidate).  A tail of more than kSortChunk rows is ordered piece by piece — what matters is that the rows of a TILE
// are alike.  Results do not depend on the order.  Returns nullptr (and leaves the lists alone) when it cannot.
static uint32_t* RowsByNorm(const uint32_t* members, uint64_t rows, const std::vector<uint64_t>& member_begin, const std::vector<uint32_t>& cell_size,
                            const std::vector<uint64_t>& cell_tile_begin, const float* neg_half_norm) {
    const size_t cells = cell_size.size();
    if (rows == 0 || rows >= (1ull << 32)) return nullptr;
    std::vector<uint2> pieces;
    for (size_t c = 0; c < cells; ++c) {
        const uint64_t tiles = cell_tile_begin[c + 1] - cell_tile_begin[c];
        const uint64_t keep = std::min<uint64_t>(cell_size[c], FilterFrontTiles(tiles) * kGTileRows);
        for (uint64_t at = keep; at + kGTileRows < cell_size[c]; at += kSortChunk)   // (a last piece of one tile or less: nothing to order)
            pieces.push_back(make_uint2((uint32_t)(member_begin[c] + at), (uint32_t)std::min<uint64_t>(kSortChunk, cell_size[c] - at)));
    }
    if (pieces.empty() || pieces.size() >= (1ull << 31)) return nullptr;
    uint32_t* out = nullptr;
    uint2* d_pieces = nullptr;
    const size_t lds = (size_t)kSortChunk * sizeof(uint2);
    bool ok = hipMalloc(&out, rows * 4) == hipSuccess && hipMalloc(&d_pieces, pieces.size() * sizeof(uint2)) == hipSuccess &&
              hipMemcpy(d_pieces, pieces.data(), pieces.size() * sizeof(uint2), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(out, members, rows * 4, hipMemcpyDeviceToDevice) == hipSuccess &&   // (rows outside the pieces stay where they are)
              hipFuncSetAttribute(reinterpret_cast<const void*>(SortPieceByNormKernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(SortPieceByNormKernel, dim3((uint32_t)pieces.size()), dim3(1024), lds, 0, d_pieces, neg_half_norm, out);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (d_pieces) (void)hipFree(d_pieces);
    if (!ok) {
        (void)hipGetLastError();
        if (out) (void)hipFree(out);
        return nullptr;
    }
    return out;
}

// plane_perm / plane_norm of one cell: members in list order, then padding up to the tile boundary.  One workgroup per cell.
__global__ void __launch_bounds__(256) CellLayoutKernel(const uint32_t* members, const uint64_t* cell_member_begin, const uint32_t* cell_size,
                                                        const uint64_t* cell_tile_begin, const float* neg_half_norm, uint32_t* plane_perm,
                                                        float* plane_norm) {
    const uint32_t cell = blockIdx.x;
    const uint64_t src = cell_member_begin[cell], dst = cell_tile_begin[cell] * kGTileRows;
    const uint32_t size = cell_size[cell];
    const uint32_t padded = (uint32_t)((cell_tile_begin[cell + 1] - cell_tile_begin[cell]) * kGTileRows);
    for (uint32_t i = threadIdx.x; i < padded; i += blockDim.x) {
        const uint32_t row = i < size ? members[src + i] : kNoRow;
        plane_perm[dst + i] = row;
        plane_norm[dst + i] = row != kNoRow ? neg_half_norm[row] : -INFINITY;
    }
}

// bf16 hi/lo planes in MFMA A-operand order, rows taken through plane_perm (padding rows are zero)
__global__ void __launch_bounds__(256) SplitFragPlanesPermKernel(const float* points, const uint32_t* plane_perm, uint64_t plane_rows, uint32_t ld,
                                                                 char* planes, uint64_t lo_offset, uint32_t plane_ks, const float* mu,
                                                                 const uint32_t* tile_hi_index, const uint32_t* tile_lo_index) {
    const uint32_t g_per_row = 2 * plane_ks;   // groups of 8 dims per row: 16 or 32
    const uint64_t total = plane_rows * g_per_row;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / g_per_row;
        const uint32_t g8 = (uint32_t)(i - r * g_per_row);
        const uint32_t row = plane_perm[r];
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (row != kNoRow && 8 * g8 < ld) {
            const float4 x = *reinterpret_cast<const float4*>(points + (size_t)row * ld + 8 * g8);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            if (mu) {
                const float4 m = *reinterpret_cast<const float4*>(mu + 8 * g8);
                v[0] -= m.x; v[1] -= m.y; v[2] -= m.z; v[3] -= m.w;
            }
        }
        if (row != kNoRow && 8 * g8 + 4 < ld) {
            const float4 x = *reinterpret_cast<const float4*>(points + (size_t)row * ld + 8 * g8 + 4);
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
        const uint64_t tile_bytes = (uint64_t)plane_ks * 1024;
        // hi tile: at the row's own tile index (tile_hi_index == nullptr), or at tile_hi_index[tile] of the hi array if the tile has one
        const uint64_t hi = tile_hi_index ? tile_hi_index[r >> 5] : (r >> 5);
        char* base = planes + hi * tile_bytes + ks * 1024 + lane * 16;   // all hi tiles first, then the lo tiles
        if (!tile_hi_index || tile_hi_index[r >> 5] != kNoRow) *reinterpret_cast<uint4*>(base) = fh.q;
        // lo tile: at the same index as the hi tile (tile_lo_index == nullptr), or at tile_lo_index[tile] of the lo array if the tile has one
        if (!tile_lo_index) {
            *reinterpret_cast<uint4*>(base + lo_offset) = fl.q;
        } else {
            const uint32_t li = tile_lo_index[r >> 5];
            if (li != kNoRow) *reinterpret_cast<uint4*>(planes + lo_offset + li * tile_bytes + ks * 1024 + lane * 16) = fl.q;
        }
    }
}

// column sums of the rows (double atomics): the store's mean row.  Thread t of a block sums column t % ld over the block's row slab.
__global__ void __launch_bounds__(256) ColumnSumKernel(const float* points, uint64_t rows, uint32_t ld, uint64_t rows_per_block, double* sums) {
    const uint32_t col = threadIdx.x % ld, lane_rows = 256 / ld;   // (ld <= 256)
    const uint32_t r_off = threadIdx.x / ld;
    if (r_off >= lane_rows) return;
    const uint64_t r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    double acc = 0.0;
    for (uint64_t r = r0 + r_off; r < r1; r += lane_rows) acc += (double)points[r * ld + col];
    atomicAdd(&sums[col], acc);
}

// norms of the shard-sample rows (padding: -inf)
__global__ void SampleNormKernel(const uint32_t* sample_perm, uint64_t n, const float* neg_half_norm, float* sample_norm) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) sample_norm[i] = sample_perm[i] != kNoRow ? neg_half_norm[sample_perm[i]] : -INFINITY;
}

// radius of every cell: max over its rows of |p - c| (double accumulation), rounded up.  One workgroup per cell.
__global__ void __launch_bounds__(256) CellRadiusKernel(const float* points, const uint32_t* plane_perm, const uint64_t* cell_tile_begin,
                                                        const float* center, uint32_t d, uint32_t ld, float* radius) {
    __shared__ double s_max[256];
    const uint32_t cell = blockIdx.x;
    const uint64_t begin = cell_tile_begin[cell] * kGTileRows, end = cell_tile_begin[cell + 1] * kGTileRows;
    const float* c = center + (size_t)cell * ld;
    double mx = 0.0;
    for (uint64_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t row = plane_perm[i];
        if (row == kNoRow) continue;
        const float* p = points + (size_t)row * ld;
        double acc = 0.0;
        for (uint32_t x = 0; x < d; ++x) {
            const double df = (double)p[x] - (double)c[x];
            acc += df * df;
        }
        mx = fmax(mx, acc);
    }
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double r = sqrt(s_max[0]) * (1.0 + 1e-6);
        float rf = (float)r;
        if ((double)rf < r) rf = nextafterf(rf, INFINITY);
        radius[cell] = rf;
    }
}

// ---- the 8-bit plane of the rest pass (scan_q8.h) ----------------------------------------------------------------------------------
// per cell: max over its rows and dimensions of |p~_i - c^_i| (p~ = fl(p - mu), c^ = fl(c - mu); mu absent: the values themselves) -> the
// cell's quantisation step, and |c^|^2.  One workgroup per cell.
__global__ void __launch_bounds__(256) CellQ8ScaleKernel(const float* points, const uint32_t* plane_perm, const uint64_t* cell_tile_begin, const float* center,
                                                         const float* mu, uint32_t d, uint32_t ld, float* scale, double* cnorm) {
    __shared__ float s_max[256];
    const uint32_t cell = blockIdx.x;
    const uint64_t begin = cell_tile_begin[cell] * kGTileRows, end = cell_tile_begin[cell + 1] * kGTileRows;
    const float* c = center + (size_t)cell * ld;
    float mx = 0.f;
    // thread t handles dimension t % ld4 * 4 .. + 3 of rows t / ld4, + 256 / ld4, ...: coalesced over a row
    const uint32_t ld4 = ld / 4, rows_per_iter = max(1u, 256u / ld4);
    const uint32_t my_c4 = threadIdx.x % ld4, my_r = threadIdx.x / ld4;
    if (my_r < rows_per_iter) {
        float4 cc = *reinterpret_cast<const float4*>(c + 4 * my_c4);
        float4 mm = mu ? *reinterpret_cast<const float4*>(mu + 4 * my_c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (mu) cc = make_float4(cc.x - mm.x, cc.y - mm.y, cc.z - mm.z, cc.w - mm.w);
        for (uint64_t i = begin + my_r; i < end; i += rows_per_iter) {
            const uint32_t row = plane_perm[i];
            if (row == kNoRow) continue;
            float4 p = *reinterpret_cast<const float4*>(points + (size_t)row * ld + 4 * my_c4);
            if (mu) p = make_float4(p.x - mm.x, p.y - mm.y, p.z - mm.z, p.w - mm.w);
            // (float differences of floats: within one rounding of the exact residual — the factor below covers it)
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(p.x - cc.x), fabsf(p.y - cc.y)), fmaxf(fabsf(p.z - cc.z), fabsf(p.w - cc.w))));
        }
    }
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double st = fmax((double)s_max[0] * (1.0 + 1e-6) / 127.0, 1e-30);
        float sf = (float)st;
        if ((double)sf < st) sf = nextafterf(sf, INFINITY);
        scale[cell] = sf;
        double acc = 0.0;
        for (uint32_t x = 0; x < d; ++x) {
            const float v = mu ? c[x] - mu[x] : c[x];
            acc += (double)v * (double)v;
        }
        cnorm[cell] = acc;
    }
}

// the int8 plane of one cell: P' = round((p~ - c^) / scale) in [-127, 127], v_mfma_i32_32x32x32_i8 A-operand order.  One workgroup per cell;
// a thread converts 16 dimensions of a row (one 16-byte store).
__global__ void __launch_bounds__(256) CellQ8PlaneKernel(const float* points, const uint32_t* plane_perm, const uint64_t* cell_tile_begin, const float* center,
                                                         const float* mu, const float* scale, uint32_t ld, uint32_t q8_ks, char* planes) {
    const uint32_t cell = blockIdx.x;
    const uint64_t begin = cell_tile_begin[cell] * kGTileRows, end = cell_tile_begin[cell + 1] * kGTileRows;
    const float* c = center + (size_t)cell * ld;
    const double inv = 1.0 / (double)scale[cell];
    const uint32_t g_per_row = 2 * q8_ks;   // groups of 16 dimensions
    const uint64_t total = (end - begin) * g_per_row;
    for (uint64_t i = threadIdx.x; i < total; i += blockDim.x) {
        const uint64_t r = begin + i / g_per_row;
        const uint32_t g16 = (uint32_t)(i % g_per_row);
        const uint32_t row = plane_perm[r];
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (row != kNoRow) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t x = 16 * g16 + j;
                int v = 0;
                if (x < ld) {   // (pad columns of a row are zero on both sides: residual 0)
                    const float pv = mu ? points[(size_t)row * ld + x] - mu[x] : points[(size_t)row * ld + x];
                    const float cv = mu ? c[x] - mu[x] : c[x];
                    v = (int)rint(((double)pv - (double)cv) * inv);
                    v = max(-127, min(127, v));
                }
                w[j >> 2] |= ((uint32_t)v & 0xFFu) << (8 * (j & 3));
            }
        }
        const uint32_t ks = g16 >> 1, h = g16 & 1, lane = (uint32_t)(r & 31) + 32 * h;
        *reinterpret_cast<uint4*>(planes + (r >> 5) * ((uint64_t)q8_ks * 1024) + ks * 1024 + lane * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// The 8-bit passes' norm words (squared L2; shards.h q8_norm): per plane row  -|p~|^2/2 + c^.e,  e = (p~ - c^) - s P' with P' READ BACK from the
// plane (so e is exactly the error of what the kernels multiply), all in double, rounded once to a float; -inf on padding rows.  One wavefront
// per plane row.
__global__ void __launch_bounds__(256) CellQ8NormKernel(const float* points, const uint32_t* plane_perm, const uint64_t* cell_tile_begin, const uint32_t* row_cell,
                                                        const float* center, const float* mu, const float* scale, uint32_t d, uint32_t ld, uint32_t q8_ks,
                                                        const char* planes, uint64_t plane_rows, float* q8_norm) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_id = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave_id; r < plane_rows; r += n_waves) {
        const uint32_t row = plane_perm[r];
        if (row == kNoRow) {
            if (lane == 0) q8_norm[r] = -INFINITY;
            continue;
        }
        const uint32_t cell = row_cell[r >> 5];   // (a tile belongs to one cell)
        const float* c = center + (size_t)cell * ld;
        const double sc = (double)scale[cell];
        const char* tile = planes + (r >> 5) * ((uint64_t)q8_ks * 1024);
        double acc = 0.0;
        for (uint32_t x = lane; x < d; x += 64) {
            const float pv = mu ? points[(size_t)row * ld + x] - mu[x] : points[(size_t)row * ld + x];
            const float cv = mu ? c[x] - mu[x] : c[x];
            const int P = (int)(int8_t)tile[(x >> 5) * 1024 + ((uint32_t)(r & 31) + 32u * ((x >> 4) & 1u)) * 16 + (x & 15u)];
            const double e = ((double)pv - (double)cv) - sc * (double)P;
            acc += -0.5 * (double)pv * (double)pv + (double)cv * e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) q8_norm[r] = (float)acc;
    }
}
// tile -> cell of the cell tiles (one thread per cell writes its tiles' entries)
__global__ void TileCellKernel(const uint64_t* cell_tile_begin, uint32_t n_cells, uint32_t* row_cell) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    for (uint64_t t = cell_tile_begin[cell]; t < cell_tile_begin[cell + 1]; ++t) row_cell[t] = cell;
}

// per 32-row tile of the 8-bit plane, what its products' first half is tested against (scan_cells.h, quick8): .x the largest norm word of
// its rows (-inf rows: padding; 0 for the inner product, whose test has no norm word), .y = 127 max_row sum_i |P'_i| >= |L.P'| for every
// int8 query plane L.  One wavefront per tile: lane (c, h) holds 16 dimensions of row c per k-step.
__global__ void __launch_bounds__(64) TileBoundKernel(const float* plane_norm, const char* q8_planes, uint32_t q8_ks, uint64_t tiles, int l2, float2* tile_bound) {
    const uint64_t t = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    uint32_t l1 = 0;
    for (uint32_t ks = 0; ks < q8_ks; ++ks) {
        const uint4 w = *reinterpret_cast<const uint4*>(q8_planes + t * ((uint64_t)q8_ks * 1024) + ks * 1024 + lane * 16);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        for (int j = 0; j < 4; ++j)
            for (int b = 0; b < 4; ++b) {
                const int v = (int)(int8_t)(ww[j] >> (8 * b) & 0xFFu);
                l1 += (uint32_t)(v < 0 ? -v : v);
            }
    }
    l1 += __shfl_xor(l1, 32);   // the row's two halves
    float m = l2 && lane < 32 ? plane_norm[t * kGTileRows + lane] : (l2 ? -INFINITY : 0.f);
    for (int o = 16; o > 0; o >>= 1) {
        l1 = max(l1, (uint32_t)__shfl_xor(l1, o));
        m = fmaxf(m, __shfl_xor(m, o));
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lane == 0) tile_bound[t] = make_float2(m, 127.0f * (float)l1);   // (127 x 127 x 256 < 2^24: exact)
}

void ReleaseCells(gpann_shards* s) {
    auto& c = s->cells;
    for (void* p : {(void*)c.d_shard_cell_begin, (void*)c.d_cell_tile_begin, (void*)c.d_center, (void*)c.d_radius, (void*)c.d_cnorm, (void*)c.plane_perm,
                    (void*)c.plane_norm, (void*)c.d_mu, (void*)c.d_cell_lo_begin, (void*)c.d_cell_hi_begin, (void*)c.q8_planes, (void*)c.d_q8_scale, (void*)c.d_q8_cnorm, (void*)c.d_tile_bound, (void*)c.q8_norm})
        if (p) (void)hipFree(p);
    c = gpann_shards::Cells{};
}

// Builds s->cells and s->frag_planes (cell order).  On any failure the store is left without cells and without planes
// (the caller then builds the plain row-order planes); returns GPANN_OK unless the device itself failed.
gpann_status BuildCells(gpann_shards* s, uint32_t target_rows) {
    const uint32_t S = s->S, ld = s->ld;
    auto give_up = [&](const char* why) {
        // not silent: a store without filter data answers every scan with the (40x slower) exact kernel
        const hipError_t last = hipGetLastError();
        fprintf(stderr, "[gpann] filter support data (cells, bf16 planes) could not be built — step '%s'%s%s; scans on this store run in exact mode\n", why,
                last != hipSuccess ? ", last HIP error: " : "", last != hipSuccess ? hipGetErrorString(last) : "");
        SetLastError("filter support data could not be built (step '%s'%s%s)", why, last != hipSuccess ? ": " : "", last != hipSuccess ? hipGetErrorString(last) : "");
        // the cause, for callers that branch on the status (gpann_shards_build_filter): retry with less memory vs report a defect
        s->filter_fail_status = last == hipErrorOutOfMemory ? GPANN_ERR_OOM : (last != hipSuccess ? GPANN_ERR_HIP : (std::string(why) == "no cells" ? GPANN_ERR_INTERNAL : GPANN_ERR_OOM));
        ReleaseCells(s);
        if (s->frag_planes) (void)hipFree(s->frag_planes);
        s->frag_planes = nullptr;
        return GPANN_OK;
    };
    LloydSession session;
    if (session.Begin(s, GPANN_METRIC_L2) != GPANN_OK) return give_up("session");

    // one problem per non-empty shard: K_b cells, initial centres at evenly spaced rows
    std::vector<LloydTask> level;
    std::vector<uint32_t> pick, shard_of;
    for (uint32_t b = 0; b < S; ++b) {
        const uint64_t n = (uint64_t)(s->offsets[b + 1] - s->offsets[b]);
        if (n == 0) continue;
        const uint32_t K = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(1, (n + target_rows / 2) / target_rows));
        level.push_back({(uint64_t)s->offsets[b], (uint32_t)n, K});
        for (uint32_t x = 0; x < K; ++x) pick.push_back((uint32_t)((uint64_t)x * n / K));
        shard_of.push_back(b);
    }
    LloydResult res;
    const int rounds = 4;
    if (session.RunLevel(level, pick, rounds, res) != GPANN_OK) return give_up("lloyd");

    // surviving cells, shard by shard, in slot order; tiles
    auto& C = s->cells;
    C.shard_cell_begin.assign(S + 1, 0);
    C.cell_tile_begin.assign(1, 0);
    std::vector<uint64_t> member_begin;
    std::vector<uint32_t> cell_size;
    std::vector<float> centers;
    {
        size_t p = 0;
        for (uint32_t b = 0; b < S; ++b) {
            C.shard_cell_begin[b] = (uint32_t)cell_size.size();
            if (p < shard_of.size() && shard_of[p] == b) {
                const uint32_t cb = res.cent_begin[p];
                // surviving cells in slot order, then renumbered along a greedy nearest-neighbour chain over their centres: cells
                // with neighbouring numbers are neighbours in space, so tasks sorted by nearest cell form spatially coherent
                // 32-task query tiles (the unit the filter can skip by)
                struct Surv {
                    uint32_t slot, index;
                };
                std::vector<Surv> surv;
                for (uint32_t j = 0; j < level[p].k0; ++j)
                    if (res.csize[cb + j] != 0) surv.push_back({j, (uint32_t)surv.size()});
                const uint32_t ns = (uint32_t)surv.size();
                std::vector<uint32_t> chain;
                std::vector<uint8_t> used(ns, 0);
                uint32_t cur = 0;
                for (uint32_t step = 0; step < ns; ++step) {
                    used[cur] = 1;
                    chain.push_back(cur);
                    const float* cc = &res.cent[(size_t)(cb + surv[cur].index) * ld];
                    double best = INFINITY;
                    uint32_t nxt = cur;
                    for (uint32_t o = 0; o < ns; ++o) {
                        if (used[o]) continue;
                        const float* oc = &res.cent[(size_t)(cb + surv[o].index) * ld];
                        double acc = 0.0;
                        for (uint32_t x = 0; x < s->d; ++x) acc += ((double)cc[x] - oc[x]) * ((double)cc[x] - oc[x]);
                        if (acc < best) {
                            best = acc;
                            nxt = o;
                        }
                    }
                    cur = nxt;
                }
                uint32_t survivor = 0;
                for (uint32_t ci : chain) {
                    const uint32_t j = surv[ci].slot;
                    const uint32_t sz = res.csize[cb + j];
                    member_begin.push_back(level[p].row_begin + res.coff[cb + j]);
                    cell_size.push_back(sz);
                    C.cell_tile_begin.push_back(C.cell_tile_begin.back() + (sz + kGTileRows - 1) / kGTileRows);
                    const float* c = &res.cent[(size_t)(cb + surv[ci].index) * ld];
                    centers.insert(centers.end(), c, c + ld);
                    ++survivor;
                }
                C.kmax = std::max(C.kmax, survivor);
                ++p;
            }
        }
        C.shard_cell_begin[S] = (uint32_t)cell_size.size();
    }
    C.n_cells = (uint32_t)cell_size.size();
    C.n_tiles = C.cell_tile_begin.back();
    if (C.n_cells == 0) return give_up("no cells");
    const uint64_t plane_rows = C.n_tiles * kGTileRows;
    C.plane_ks = PlaneKSteps(s->d);
    const size_t tile_bytes = (size_t)C.plane_ks * 1024;   // one plane of one tile
    // the shard sample: the first rows of every shard (store order = the order of Clusters[b] = a random sample), appended as tiles of their own
    std::vector<uint32_t> sample_perm_host;
    {
        const char* sr = s->opt.Get("GPANN_FILTER_EXACT_ROWS");   // (historic name: the rows of the first, shard-wide sample)
        const uint64_t want = sr ? (uint64_t)std::max(64L, std::atol(sr)) : 4096;
        C.sample_tile_begin.assign(S + 1, 0);
        for (uint32_t b = 0; b < S; ++b) {
            const uint64_t n = (uint64_t)(s->offsets[b + 1] - s->offsets[b]), m = std::min(n, want);
            const uint64_t tiles = (m + kGTileRows - 1) / kGTileRows;
            C.sample_tile_begin[b + 1] = C.sample_tile_begin[b] + tiles;
            for (uint64_t i = 0; i < tiles * kGTileRows; ++i) sample_perm_host.push_back(i < m ? (uint32_t)(s->offsets[b] + i) : kNoRow);
        }
        C.n_sample_tiles = C.sample_tile_begin[S];
        C.sample_tile_base = C.n_tiles + 2;
    }
    const uint64_t all_tiles = C.sample_tile_base + C.n_sample_tiles;   // tile numbering of plane_perm / plane_norm / the 8-bit plane: cell tiles | 2 zero tiles | sample tiles
    const uint64_t all_rows = all_tiles * kGTileRows;
    // The 8-bit plane comes first: when the store has one, the rest pass never reads the bf16 planes, and only the tiles the earlier
    // passes can reach keep a bf16 hi tile (below).
    {
        const char* q8_env = s->opt.Get("GPANN_FILTER_Q8");
        if (!(q8_env && q8_env[0] == '0')) {
            C.q8_ks = (s->d + 31) / 32;
            const bool qok = hipMalloc(&C.q8_planes, (C.n_tiles + 2) * (size_t)C.q8_ks * 1024) == hipSuccess && hipMalloc(&C.d_q8_scale, (size_t)C.n_cells * 4) == hipSuccess &&
                             hipMalloc(&C.d_q8_cnorm, (size_t)C.n_cells * 8) == hipSuccess && hipMalloc(&C.d_tile_bound, (C.n_tiles + 2) * 8) == hipSuccess &&
                             hipMemset(C.q8_planes + C.n_tiles * (size_t)C.q8_ks * 1024, 0, 2 * (size_t)C.q8_ks * 1024) == hipSuccess &&
                             // (GPANN_FILTER_Q8_CELL_FRAME=0 at store build: the plain norm words and the looser (s/2) |q~|_1 error term of round 3)
                             (s->metric != GPANN_METRIC_L2 || (s->opt.Get("GPANN_FILTER_Q8_CELL_FRAME") && s->opt.Get("GPANN_FILTER_Q8_CELL_FRAME")[0] == '0') ||
                              hipMalloc(&C.q8_norm, ((C.n_tiles + 2) * kGTileRows + 64) * 4) == hipSuccess);
            if (!qok) {   // the store works without it
                (void)hipGetLastError();
                for (void* p : {(void*)C.q8_planes, (void*)C.d_q8_scale, (void*)C.d_q8_cnorm, (void*)C.d_tile_bound, (void*)C.q8_norm})
                    if (p) (void)hipFree(p);
                C.q8_norm = nullptr;
                C.q8_planes = nullptr;
                C.d_q8_scale = nullptr;
                C.d_q8_cnorm = nullptr;
                C.d_tile_bound = nullptr;
                C.q8_ks = 0;
                fprintf(stderr, "[gpann] the 8-bit plane of the filter's rest pass could not be allocated; the rest pass streams the bf16 plane\n");
            }
        }
    }
    // Which tiles keep a bf16 hi / lo tile (shards.h).  lo: every one (GPANN_FILTER_LO_ROWS=all), or the sample tiles and the first rows of every
    // cell — the only ones the two-plane passes read (beyond 128 dimensions every pass is a hi-plane pass: no lo tiles at all).  hi: every one, or —
    // when the rest pass has the 8-bit plane — the sample tiles and the first max(head, 1 / kHiKeepDiv) of every cell: the head and sample passes
    // never reach further (the sample pass covers at most 1 / 8 of a cell: smaller shares were measured no better, profiles/r03/sweep_sample_div.md).  GPANN_FILTER_HI_ROWS=all keeps all (then GPANN_FILTER_Q8=0 works at scan time).
    constexpr uint32_t kHiKeepDiv = kFilterFrontDiv;
    std::vector<uint32_t> cell_lo_begin(C.n_cells + 1, 0), cell_hi_begin(C.n_cells + 1, 0), tile_lo_index, tile_hi_index;
    {
        const char* lr = s->opt.Get("GPANN_FILTER_LO_ROWS");
        const char* hr = s->opt.Get("GPANN_FILTER_HI_ROWS");
        C.lo_full = lr && std::string(lr) == "all";
        C.hi_compact = C.q8_planes != nullptr && !C.lo_full && !(hr && std::string(hr) == "all");
        const uint64_t lo_rows = C.plane_ks > 8 ? 0 : (lr && !C.lo_full ? (uint64_t)std::max(0L, std::atol(lr)) : 512);
        C.lo_head_tiles = C.lo_full ? 0xFFFFFFFFu : (uint32_t)((lo_rows + kGTileRows - 1) / kGTileRows);
        C.hi_keep_div = C.hi_compact ? kHiKeepDiv : 1u;
        uint64_t at_lo = 0, at_hi = 0;
        if (!C.lo_full) tile_lo_index.assign(C.n_tiles, kNoRow);
        if (C.hi_compact) tile_hi_index.assign(C.n_tiles, kNoRow);
        for (uint32_t c = 0; c < C.n_cells; ++c) {
            const uint64_t t0 = C.cell_tile_begin[c], tiles = C.cell_tile_begin[c + 1] - t0;
            cell_lo_begin[c] = (uint32_t)(C.lo_full ? t0 : at_lo);
            cell_hi_begin[c] = (uint32_t)(C.hi_compact ? at_hi : t0);
            if (!C.lo_full) {
                const uint64_t keep = std::min<uint64_t>(C.lo_head_tiles, tiles);
                for (uint64_t i = 0; i < keep; ++i) tile_lo_index[t0 + i] = (uint32_t)(at_lo + i);
                at_lo += keep;
            }
            if (C.hi_compact) {
                // (head: 512 rows, the scan's default and upper limit here; + 1: the sample segment's end is a rounded-up quotient)
                const uint64_t keep = std::min<uint64_t>(tiles, FilterFrontTiles(tiles));
                for (uint64_t i = 0; i < keep; ++i) tile_hi_index[t0 + i] = (uint32_t)(at_hi + i);
                at_hi += keep;
            }
        }
        cell_lo_begin[C.n_cells] = (uint32_t)(C.lo_full ? C.n_tiles : at_lo);
        cell_hi_begin[C.n_cells] = (uint32_t)(C.hi_compact ? at_hi : C.n_tiles);
        // hi array: the cells' tiles | 2 zero tiles | the sample's tiles | 2 zero tiles;  lo array behind it, laid out the same way
        C.n_hi_cell_tiles = cell_hi_begin[C.n_cells];
        C.hi_sample_begin = C.n_hi_cell_tiles + 2;
        C.lo_offset_tiles = C.hi_sample_begin + C.n_sample_tiles + 2;
        C.lo_sample_begin = (uint64_t)cell_lo_begin[C.n_cells] + 2;
        C.n_lo_tiles = C.lo_sample_begin + (C.plane_ks > 8 ? 0 : C.n_sample_tiles);
    }

    uint64_t* d_member_begin = nullptr;
    uint32_t* d_cell_size = nullptr;
    bool ok = hipMalloc(&C.d_shard_cell_begin, (S + 1) * 4) == hipSuccess && hipMalloc(&C.d_cell_tile_begin, (C.n_cells + 1) * 8) == hipSuccess &&
              hipMalloc(&C.d_center, (size_t)C.n_cells * ld * 4) == hipSuccess && hipMalloc(&C.d_radius, (size_t)C.n_cells * 4) == hipSuccess &&
              hipMalloc(&C.d_cnorm, (size_t)C.n_cells * 4) == hipSuccess &&
              hipMalloc(&C.plane_perm, all_rows * 4) == hipSuccess && hipMalloc(&C.plane_norm, (all_rows + 64) * 4) == hipSuccess &&
              hipMalloc(&s->frag_planes, (C.lo_offset_tiles + C.n_lo_tiles + 2) * tile_bytes) == hipSuccess && hipMalloc(&d_member_begin, C.n_cells * 8) == hipSuccess &&
              hipMalloc(&d_cell_size, C.n_cells * 4) == hipSuccess && hipMalloc(&C.d_cell_lo_begin, (size_t)(C.n_cells + 1) * 4) == hipSuccess &&
              hipMalloc(&C.d_cell_hi_begin, (size_t)(C.n_cells + 1) * 4) == hipSuccess;
    uint32_t *d_tile_lo_index = nullptr, *d_tile_hi_index = nullptr;
    if (ok && !tile_lo_index.empty())
        ok = hipMalloc(&d_tile_lo_index, tile_lo_index.size() * 4) == hipSuccess &&
             hipMemcpy(d_tile_lo_index, tile_lo_index.data(), tile_lo_index.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && !tile_hi_index.empty())
        ok = hipMalloc(&d_tile_hi_index, tile_hi_index.size() * 4) == hipSuccess &&
             hipMemcpy(d_tile_hi_index, tile_hi_index.data(), tile_hi_index.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    std::vector<float> cnorm(C.n_cells);
    for (uint32_t cidx = 0; cidx < C.n_cells; ++cidx) {
        double acc = 0.0;
        for (uint32_t x = 0; x < s->d; ++x) acc += (double)centers[(size_t)cidx * ld + x] * (double)centers[(size_t)cidx * ld + x];
        cnorm[cidx] = (float)(acc * (1.0 + 1e-6));
    }
    if (ok) {
        ok = hipMemcpy(C.d_cnorm, cnorm.data(), (size_t)C.n_cells * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(C.d_shard_cell_begin, C.shard_cell_begin.data(), (S + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(C.d_cell_tile_begin, C.cell_tile_begin.data(), (C.n_cells + 1) * 8, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(C.d_center, centers.data(), (size_t)C.n_cells * ld * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_member_begin, member_begin.data(), C.n_cells * 8, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(d_cell_size, cell_size.data(), C.n_cells * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemset(C.plane_norm + plane_rows, 0, (all_rows - plane_rows + 64) * 4) == hipSuccess &&
             hipMemset(C.plane_perm + plane_rows, 0xFF, (all_rows - plane_rows) * 4) == hipSuccess &&   // (kNoRow: the gap tiles and, for now, the sample)
             hipMemcpy(C.d_cell_lo_begin, cell_lo_begin.data(), (size_t)(C.n_cells + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(C.d_cell_hi_begin, cell_hi_begin.data(), (size_t)(C.n_cells + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
             // everything of both arrays that is not a cell's tile: the gaps, the sample's tiles until they are written
             hipMemset(s->frag_planes + C.n_hi_cell_tiles * tile_bytes, 0, (C.lo_offset_tiles - C.n_hi_cell_tiles) * tile_bytes) == hipSuccess &&
             hipMemset(s->frag_planes + (C.lo_offset_tiles + cell_lo_begin[C.n_cells]) * tile_bytes, 0, (C.n_lo_tiles + 2 - cell_lo_begin[C.n_cells]) * tile_bytes) == hipSuccess;
    }
    // squared L2: everything the matrix-core bound touches is taken around the store's mean row (translation leaves the distances alone
    // and makes |p|, |q| — which the bound's error terms scale with — as small as the data allows; a cloud that sits at +100 in every
    // coordinate would otherwise have error terms larger than its k-th distances).  GPANN_FILTER_CENTER=0 switches it off.
    const char* centre_env = s->opt.Get("GPANN_FILTER_CENTER");
    if (ok && s->metric == GPANN_METRIC_L2 && !(centre_env && centre_env[0] == '0')) {
        double* d_sums = nullptr;
        unsigned long long* d_max_bits = reinterpret_cast<unsigned long long*>(s->d_shard_rmax);
        uint32_t* d_bad = nullptr;
        ok = hipMalloc(&C.d_mu, (size_t)ld * 4) == hipSuccess && hipMalloc(&d_sums, (size_t)ld * 8) == hipSuccess && hipMalloc(&d_bad, 4) == hipSuccess &&
             hipMemset(d_sums, 0, (size_t)ld * 8) == hipSuccess && hipMemset(d_bad, 0, 4) == hipSuccess;
        if (ok) {
            const uint64_t rows_per_block = std::max<uint64_t>(256, (s->rows + 8191) / 8192);
            hipLaunchKernelGGL(ColumnSumKernel, dim3((uint32_t)((s->rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, 0, s->points, s->rows, ld, rows_per_block,
                               d_sums);
            std::vector<double> sums(ld);
            ok = hipMemcpy(sums.data(), d_sums, (size_t)ld * 8, hipMemcpyDeviceToHost) == hipSuccess;
            std::vector<float> mu(ld, 0.f);
            for (uint32_t x = 0; x < s->d; ++x) mu[x] = (float)(sums[x] / (double)s->rows);
            ok = ok && hipMemcpy(C.d_mu, mu.data(), (size_t)ld * 4, hipMemcpyHostToDevice) == hipSuccess;
            // the rows' norms and the shards' max norms once more, in the centred frame
            ok = ok && hipMemset(d_max_bits, 0, (size_t)S * 8) == hipSuccess;
            if (ok) {
                const int grid = (int)std::min<uint64_t>((s->rows + 3) / 4, 256 * 32);
                hipLaunchKernelGGL(RowNormKernel, dim3(grid), dim3(256), 0, 0, s->points, s->rows, s->d, ld, s->d_offsets, S, s->neg_half_norm, d_max_bits, d_bad, C.d_mu);
                uint32_t bad = 0;
                ok = hipGetLastError() == hipSuccess && hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost) == hipSuccess && bad == 0;
            }
        }
        if (d_sums) (void)hipFree(d_sums);
        if (d_bad) (void)hipFree(d_bad);
    }
    if (ok) {
        // (GPANN_FILTER_SORT_ROWS=0: the cells' rows stay in slab order throughout — the layout of the earlier rounds, for comparison)
        const char* sort_env = s->opt.Get("GPANN_FILTER_SORT_ROWS");
        uint32_t* by_norm = sort_env && sort_env[0] == '0' ? nullptr : RowsByNorm(session.rows(), s->rows, member_begin, cell_size, C.cell_tile_begin, s->neg_half_norm);
        hipLaunchKernelGGL(CellLayoutKernel, dim3(C.n_cells), dim3(256), 0, 0, by_norm ? by_norm : session.rows(), d_member_begin, d_cell_size, C.d_cell_tile_begin,
                           s->neg_half_norm, C.plane_perm, C.plane_norm);
        if (by_norm) {
            (void)hipDeviceSynchronize();
            (void)hipFree(by_norm);
        }
        const int g2 = (int)std::min<uint64_t>((plane_rows * 2 * C.plane_ks + 255) / 256, 256 * 64);
        hipLaunchKernelGGL(SplitFragPlanesPermKernel, dim3(g2), dim3(256), 0, 0, s->points, C.plane_perm, plane_rows, ld, s->frag_planes,
                           (uint64_t)C.lo_offset_tiles * tile_bytes, C.plane_ks, C.d_mu, d_tile_hi_index, d_tile_lo_index);
        hipLaunchKernelGGL(CellRadiusKernel, dim3(C.n_cells), dim3(256), 0, 0, s->points, C.plane_perm, C.d_cell_tile_begin, C.d_center, s->d, ld, C.d_radius);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (d_member_begin) (void)hipFree(d_member_begin);
    if (d_cell_size) (void)hipFree(d_cell_size);
    if (d_tile_lo_index) (void)hipFree(d_tile_lo_index);
    if (d_tile_hi_index) (void)hipFree(d_tile_hi_index);
    if (!ok) return give_up("device");

    // the 8-bit plane of the rest pass (allocated above)
    if (C.q8_planes) {
        hipLaunchKernelGGL(CellQ8ScaleKernel, dim3(C.n_cells), dim3(256), 0, 0, s->points, C.plane_perm, C.d_cell_tile_begin, C.d_center, C.d_mu, s->d, ld, C.d_q8_scale,
                           C.d_q8_cnorm);
        hipLaunchKernelGGL(CellQ8PlaneKernel, dim3(C.n_cells), dim3(256), 0, 0, s->points, C.plane_perm, C.d_cell_tile_begin, C.d_center, C.d_mu, C.d_q8_scale, ld, C.q8_ks,
                           C.q8_planes);
        if (C.q8_norm) {   // squared L2: the 8-bit passes' own norm words (shards.h)
            uint32_t* d_tile_cell = nullptr;
            bool nok = hipMalloc(&d_tile_cell, (C.n_tiles + 2) * 4) == hipSuccess && hipMemset(d_tile_cell, 0, (C.n_tiles + 2) * 4) == hipSuccess &&
                       hipMemset(C.q8_norm + (C.n_tiles + 2) * kGTileRows, 0, 64 * 4) == hipSuccess;
            if (nok) {
                hipLaunchKernelGGL(TileCellKernel, dim3((C.n_cells + 255) / 256), dim3(256), 0, 0, C.d_cell_tile_begin, C.n_cells, d_tile_cell);
                // (the two zero tiles behind the cell tiles: perm = kNoRow there -> -inf, like plane_norm)
                hipLaunchKernelGGL(CellQ8NormKernel, dim3(4096), dim3(256), 0, 0, s->points, C.plane_perm, C.d_cell_tile_begin, d_tile_cell, C.d_center, C.d_mu, C.d_q8_scale,
                                   s->d, ld, C.q8_ks, C.q8_planes, (C.n_tiles + 2) * kGTileRows, C.q8_norm);
                nok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            }
            if (d_tile_cell) (void)hipFree(d_tile_cell);
            if (!nok) return give_up("8-bit norm words");
        }
        hipLaunchKernelGGL(TileBoundKernel, dim3((uint32_t)(C.n_tiles + 2)), dim3(64), 0, 0, C.q8_norm ? C.q8_norm : C.plane_norm, C.q8_planes, C.q8_ks, C.n_tiles + 2,
                           s->metric == GPANN_METRIC_L2 ? 1 : 0, C.d_tile_bound);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return give_up("8-bit plane");
    }

    // the shard sample's rows, norms and planes behind the cell tiles
    {
        const uint64_t srows = C.n_sample_tiles * kGTileRows;
        bool sok = true;
        if (srows) {
            uint32_t* sperm = C.plane_perm + C.sample_tile_base * kGTileRows;
            float* snorm = C.plane_norm + C.sample_tile_base * kGTileRows;
            sok = hipMemcpy(sperm, sample_perm_host.data(), srows * 4, hipMemcpyHostToDevice) == hipSuccess;
            if (sok) {
                hipLaunchKernelGGL(SampleNormKernel, dim3((uint32_t)((srows + 255) / 256)), dim3(256), 0, 0, sperm, srows, s->neg_half_norm, snorm);
                const int g3 = (int)std::min<uint64_t>((srows * 2 * C.plane_ks + 255) / 256, 256 * 64);
                // (the sample's lo tiles: lo_sample_begin + j of the lo array; beyond 128 dimensions there are none — they land in the two spare tiles' place,
                // which the allocation covers only for j < 2: write them nowhere instead)
                if (C.plane_ks > 8) {
                    // hi planes only: a lo index table that says "none" for every sample tile
                    uint32_t* d_none = nullptr;
                    sok = hipMalloc(&d_none, C.n_sample_tiles * 4) == hipSuccess && hipMemset(d_none, 0xFF, C.n_sample_tiles * 4) == hipSuccess;
                    if (sok)
                        hipLaunchKernelGGL(SplitFragPlanesPermKernel, dim3(g3), dim3(256), 0, 0, s->points, sperm, srows, ld, s->frag_planes + C.hi_sample_begin * tile_bytes,
                                           (uint64_t)0, C.plane_ks, C.d_mu, (const uint32_t*)nullptr, d_none);
                    sok = sok && hipDeviceSynchronize() == hipSuccess;
                    if (d_none) (void)hipFree(d_none);
                } else {
                    hipLaunchKernelGGL(SplitFragPlanesPermKernel, dim3(g3), dim3(256), 0, 0, s->points, sperm, srows, ld, s->frag_planes + C.hi_sample_begin * tile_bytes,
                                       (uint64_t)(C.lo_offset_tiles + C.lo_sample_begin - C.hi_sample_begin) * tile_bytes, C.plane_ks, C.d_mu, (const uint32_t*)nullptr,
                                       (const uint32_t*)nullptr);
                }
                sok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
            }
        }
        if (!sok) return give_up("sample planes");
    }
    return GPANN_OK;
}

}  // namespace gpann
