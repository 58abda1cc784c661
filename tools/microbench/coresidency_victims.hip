// coresidency_victims.hip — WHICH instructions of a wavefront go wrong beside another kernel's v_mfma_i32_32x32x32_i8 wavefronts?
//
// Follow-up of coresidency_repro.hip (profiles/r05/overlap_race.md): there the victim was the product's CellDistKernel (LDS reads + fp32 vector
// arithmetic) and the finding "lanes 48 - 63 of single accumulator registers differ, every launch, only beside the int8 matrix instruction".  Here the
// victims are minimal kernels that each exercise ONE class of instruction on values that depend on nothing but the thread index, so that the class that
// is disturbed names itself:
//   0  scalar fp32 vector arithmetic (v_mul_f32 / v_add_f32 chains in registers)          1  packed fp32 (v_pk_mul_f32 / v_pk_add_f32)
//   2  integer vector arithmetic (v_mul_lo_u32 / v_add_u32)                                3  LDS reads (ds_read_b128) folded with integer xor
//   4  cross-lane (ds_bpermute / DPP via __shfl_xor) folded with integer adds              5  fp64 (v_fma_f64 / v_mul_f64 / v_add_f64)
//   6  global loads (coalesced dwordx4) folded with integer xor                            7  fused fp32 (v_fma_f32)
// Every victim is launched over and over on three streams (256-thread workgroups, <= 64 registers: they fit beside a 128-register neighbour) and each
// launch's output is compared with a launch made on an idle chip.  Neighbours: none / v_mfma_i32_32x32x32_i8 / v_mfma_f32_32x32x16_bf16 /
// v_mfma_i32_16x16x64_i8 (the other int8 shape new on gfx950) / the int8 one with s_nop between issues.
// Output per (neighbour, victim): launches checked, launches that differ, and the histogram of differing words by 16-lane quarter of the wavefront.
// Victim 8 is the positive control: the product's CellDistKernel<true> on coresidency_repro.hip's inputs — if IT is not disturbed on a box, the table says nothing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o coresidency_victims coresidency_victims.hip && timeout 300 ./coresidency_victims [seconds per cell]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../gp-ann_amd/csrc/scan_filter.h"
#include "../../gp-ann_amd/csrc/scan_cells.h"

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// neighbour: persistent 512-thread workgroups, one per CU, 128 registers per wavefront (four fit a SIMD: room for victims), one instruction kind
template <int KIND>
__global__ void __launch_bounds__(512, 1) Neighbour(const int* stop, int* sink, int burst, unsigned long long tick_limit) {
    extern __shared__ char nlds[];
    asm volatile("" ::: "v127");
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<int*>(nlds)[i] = (i * 2654435761u) & 0x3C003C00u;
    __syncthreads();
    v4i a = {lane, lane * 3, lane * 5, lane * 7};
    v16i acc0 = {0}, acc1 = {0};
    v4i c0 = {0}, c1 = {0};
    v16f f0 = {0}, f1 = {0};
    // (burst is a run-time value, as in coresidency_repro.hip: with a constant the compiler pipelines the loop into a gapless stream of matrix
    // instructions beside which fp32 victims got 5 launches per second and, beside the bf16 one, none at all — the first attempt never ended.
    // The neighbour also ends by itself after tick_limit ticks of the 100 MHz wall clock: no launch of this program can outlive its cell.)
    const unsigned long long t_begin = wall_clock64();
    for (;;) {
        for (int b = 0; b < burst; ++b) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const v4i bv = *reinterpret_cast<const v4i*>(nlds + ((b * 8 + ks) & 31) * 1024 + lane * 16);
                if (KIND == 0) {
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, a, acc1, 0, 0, 0);
                } else if (KIND == 1) {
                    f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, bv), f0, 0, 0, 0);
                    f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, bv), f1, 0, 0, 0);
                } else if (KIND == 2) {
                    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bv, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(bv, a, c1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bv, acc0, 0, 0, 0);
                    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, a, acc1, 0, 0, 0);
                    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
                }
            }
        }
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t_begin > tick_limit) break;
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= acc0[i] ^ acc1[i] ^ __float_as_int(f0[i]) ^ __float_as_int(f1[i]);
    s ^= c0[0] ^ c0[1] ^ c0[2] ^ c0[3] ^ c1[0] ^ c1[1] ^ c1[2] ^ c1[3];
    if (s == 0x12345678) sink[0] = s;
}

constexpr int kVictimKinds = 41;   // (8 = the positive control: the product's CellDistKernel<true>, the victim of coresidency_repro.hip; 9 = CellDistKernel<false>;
                                    //  10 - 13: Victim2 below)
constexpr int kOutPerThread = 4;
// victims: out[thread * 4 .. +4) = a function of the thread index only
template <int KIND>
__global__ void __launch_bounds__(256) Victim(uint32_t* out, const uint32_t* table, int iters) {
    __shared__ uint32_t lds[4096];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t r[kOutPerThread] = {gid * 2654435761u + 1u, gid * 40503u + 7u, gid ^ 0x9E3779B9u, gid + 12345u};
    if (KIND == 0 || KIND == 7) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __uint_as_float(0x3F800000u | (((gid + 17u * i) & 1023u) << 13));   // 1 + j / 1024, without a float instruction
        const float m = 1.0000003f, c = 0.25f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // (inline assembly: left to itself the compiler pairs these chains into v_pk_mul_f32 / v_pk_add_f32 — victim 1's instructions)
                if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
                else asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(x[i]) : "v"(m), "v"(c));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i & 3] ^= __float_as_uint(x[i]) * (2u * i + 1u);
    } else if (KIND == 1) {
        f32x2 x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = f32x2{1.0f + (float)((gid + 17u * i) & 1023u) * (1.0f / 1024.0f), 1.5f + (float)((gid + 29u * i) & 511u) * (1.0f / 512.0f)};
        const f32x2 m = {1.0000003f, 0.9999997f}, c = {0.25f, 0.125f}, lim = {63.f, 63.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] = x[i] * m + c;   // v_pk_mul_f32 + v_pk_add_f32
                if (x[i].x > 64.f) x[i] = x[i] - lim;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i & 3] ^= (__float_as_uint(x[i].x) * (2u * i + 1u)) ^ (__float_as_uint(x[i].y) * (2u * i + 3u));
    } else if (KIND == 2) {
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = gid * (2u * i + 3u) + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] * 1664525u + 1013904223u + (x[(i + 1) & 7] >> 7);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i & 3] ^= x[i];
    } else if (KIND == 3) {
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (i * 2246822519u) ^ (blockIdx.x * 374761393u);
        __syncthreads();
        uint32_t pos = (threadIdx.x * 4u) & 4095u;
        for (int it = 0; it < iters; ++it) {
            const uint4 v = *reinterpret_cast<const uint4*>(&lds[pos & 4092u]);
            r[0] ^= v.x + it;
            r[1] += v.y;
            r[2] ^= v.z;
            r[3] += v.w ^ (uint32_t)it;
            pos = (pos + 4u * 67u) & 4095u;
        }
    } else if (KIND == 4) {
        uint32_t x = r[0], y = r[1];
        for (int it = 0; it < iters; ++it) {
            x += (uint32_t)__shfl_xor((int)y, 1 + (it & 31));
            y ^= (uint32_t)__shfl((int)x, (lane * 5 + it) & 63) + 0x9E3779B9u;
        }
        r[2] ^= x;
        r[3] += y;
    } else if (KIND == 5) {
        double x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = 1.0 + (double)((gid + 13u * i) & 4095u) / 4096.0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i] = x[i] * 1.000000001 + 0.25;
                if (x[i] > 64.0) x[i] -= 63.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(x[i]);
            r[i] ^= (uint32_t)b ^ (uint32_t)(b >> 32);
        }
    } else if (KIND == 6) {
        uint32_t pos = gid * 4u;
        for (int it = 0; it < iters; ++it) {
            const uint4 v = *reinterpret_cast<const uint4*>(&table[pos & ((1u << 20) - 4u)]);
            r[0] ^= v.x + it;
            r[1] += v.y;
            r[2] ^= v.z;
            r[3] += v.w ^ (uint32_t)it;
            pos += 4u * 65537u;
        }
    }
#pragma unroll
    for (int i = 0; i < kOutPerThread; ++i) out[(size_t)gid * kOutPerThread + i] = r[i];
}

// second family (round 6): the distance sequence of the product's tile kernels — d = p - q (q broadcast to both halves: op_sel), acc += d * d — which
// compiles to v_pk_add_f32 with op_sel / neg_lo / neg_hi modifiers and v_pk_mul_f32; the plain packed victim (1) has no modifiers.
//   MODE 0: operands in registers          1: operands from LDS (ds_read_b128), each thread its own slots, no barrier
//   MODE 2: registers, NACC = 32 accumulator pairs (more than 100 live registers)      3: LDS rewritten every round between two s_barriers
template <int MODE, int NACC>
__global__ void __launch_bounds__(256) Victim2(uint32_t* out, int iters) {
    __shared__ float lds2[256 * 8 + 64];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    f32x2 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x2{0.f, 0.f};
    float4 pv = {__uint_as_float(0x3F800000u | ((gid & 1023u) << 13)), __uint_as_float(0x3F800000u | (((gid >> 3) & 1023u) << 13)),
                 __uint_as_float(0x40000000u | (((gid * 7u) & 1023u) << 12)), __uint_as_float(0x3F000000u | (((gid * 13u) & 1023u) << 13))};
    float4 qv = {__uint_as_float(0x3F800000u | (((gid * 3u) & 1023u) << 13)), __uint_as_float(0x3FC00000u | (((gid * 5u) & 511u) << 13)),
                 __uint_as_float(0x3F000000u | (((gid * 11u) & 1023u) << 13)), __uint_as_float(0x40000000u | (((gid * 17u) & 1023u) << 12))};
    if (MODE == 1 || MODE == 3) {
        reinterpret_cast<float4*>(lds2)[threadIdx.x * 2] = pv;
        reinterpret_cast<float4*>(lds2)[threadIdx.x * 2 + 1] = qv;
        __syncthreads();
    }
    for (int it = 0; it < iters; ++it) {
        float4 p = pv, q = qv;
        if (MODE == 1 || MODE == 3) {
            const int slot = MODE == 3 ? ((threadIdx.x + it) & 255) : threadIdx.x;
            p = reinterpret_cast<const float4*>(lds2)[slot * 2];
            q = reinterpret_cast<const float4*>(lds2)[slot * 2 + 1];
        }
        asm volatile("" : "+v"(p.x), "+v"(p.y), "+v"(p.z), "+v"(p.w), "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));   // (opaque: nothing of the round is hoisted out of the loop)
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const float qs = (i & 1) ? q.y : q.x, qt = (i & 1) ? q.w : q.z;
            const f32x2 px = {p.x, p.y}, py = {p.z, p.w};
            const f32x2 qx = {qs, qs}, qy = {qt, qt};     // broadcast: op_sel on the instruction
            const f32x2 d0 = px - qx, d1 = py - qy;       // neg_lo / neg_hi
            f32x2 t = d0 * d0 + d1 * d1;
            acc[i] = acc[i] + t;
        }
        if (MODE == 3) {   // every thread rewrites its slot (same values) between two barriers: the staging pattern of the tile kernels
            __syncthreads();
            reinterpret_cast<float4*>(lds2)[threadIdx.x * 2] = pv;
            reinterpret_cast<float4*>(lds2)[threadIdx.x * 2 + 1] = qv;
            __syncthreads();
        }
    }
    uint32_t r[kOutPerThread] = {gid, gid * 3u, gid * 5u, gid * 7u};
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        r[i & 3] ^= __float_as_uint(acc[i].x) * (2u * i + 1u);
        r[(i + 1) & 3] += __float_as_uint(acc[i].y);
    }
#pragma unroll
    for (int i = 0; i < kOutPerThread; ++i) out[(size_t)gid * kOutPerThread + i] = r[i];
}

// third family (round 6): the INNER LOOP of the product's tile kernels (CellDistKernel / ScanTileKernel / AssignKernel): per group of four dimensions
// eight ds_read_b128 (four rows of each operand tile) feed a 4 x 4 register tile through AccumulateGroup — the compiler waits for the reads one at a time
// (s_waitcnt lgkmcnt(7) ... lgkmcnt(0)) and starts multiplying while the later ones are still in flight.
//   MODE 0: as the compiler schedules it          1: s_waitcnt lgkmcnt(0) between the eight reads and the first use
//   MODE 2: the full wait followed by s_nop       3: full wait + every operand register copied (v_mov_b32) before use
//   MODE 4: MODE 0 with the tiles re-staged from global memory through registers every chunk, two barriers per chunk (CellDistKernel's shape)
//   MODE 5: full wait + ~160 cycles of s_nop, nothing copied        6 / 7: full wait, only the c / only the q operands copied
//   MODE 10 - 12: the operands' ROLES swapped — the rows read with one address per 16-lane quarter (index ty) become the instruction's plain operand, the
//                 rows read with one address per lane (index tx) its op_sel-broadcast operand: 10 no copy, 11 the quarter-uniform read copied, 12 the per-lane read
//   MODE 13 / 14: both operands straight from global memory (global_load_dwordx4), no LDS: 13 no copy, 14 the op_sel operand (c) copied
//   MODE 8: the arithmetic on scalar fp32 instructions (inline assembly: the compiler would pair them)        9: only the .x word of every operand copied
template <int MODE>
__global__ void __launch_bounds__(256) Victim3(uint32_t* out, const uint32_t* table, int iters) {
    constexpr int kStride = 36;
    __shared__ float sC[64 * kStride];
    __shared__ float sQ[64 * kStride];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const uint32_t gid = blockIdx.x * 256u + tid;
    const int st_row0 = tid >> 3, st_row1 = (tid + 256) >> 3, st_c4 = tid & 7;
    auto value = [&](uint32_t row, uint32_t col, uint32_t salt) {   // a finite float in [1, 2) from integers only
        return __uint_as_float(0x3F800000u | ((((row * 131u + col * 7u + salt) * 2654435761u) >> 9) & 0x007FFFFFu));
    };
    auto stage = [&](uint32_t salt) {
        float4 c0, c1, q0, q1;
        if (MODE == 4) {   // through global memory, as the product stages its tiles
            const uint4 a = *reinterpret_cast<const uint4*>(&table[((blockIdx.x * 64u + st_row0) * 32u + 4u * st_c4 + salt * 4u) & ((1u << 20) - 4u)]);
            const uint4 b = *reinterpret_cast<const uint4*>(&table[((blockIdx.x * 64u + st_row1) * 32u + 4u * st_c4 + salt * 4u + 2048u) & ((1u << 20) - 4u)]);
            auto f = [](uint32_t w) { return __uint_as_float(0x3F800000u | (w & 0x007FFFFFu)); };
            c0 = make_float4(f(a.x), f(a.y), f(a.z), f(a.w));
            c1 = make_float4(f(b.x), f(b.y), f(b.z), f(b.w));
            q0 = make_float4(f(a.y ^ b.x), f(a.z ^ b.y), f(a.w ^ b.z), f(a.x ^ b.w));
            q1 = make_float4(f(b.y ^ a.x), f(b.z ^ a.y), f(b.w ^ a.z), f(b.x ^ a.w));
        } else {
            c0 = make_float4(value(st_row0, 4 * st_c4, salt), value(st_row0, 4 * st_c4 + 1, salt), value(st_row0, 4 * st_c4 + 2, salt), value(st_row0, 4 * st_c4 + 3, salt));
            c1 = make_float4(value(st_row1, 4 * st_c4, salt), value(st_row1, 4 * st_c4 + 1, salt), value(st_row1, 4 * st_c4 + 2, salt), value(st_row1, 4 * st_c4 + 3, salt));
            q0 = make_float4(value(st_row0 + 64, 4 * st_c4, salt), value(st_row0 + 64, 4 * st_c4 + 1, salt), value(st_row0 + 64, 4 * st_c4 + 2, salt), value(st_row0 + 64, 4 * st_c4 + 3, salt));
            q1 = make_float4(value(st_row1 + 64, 4 * st_c4, salt), value(st_row1 + 64, 4 * st_c4 + 1, salt), value(st_row1 + 64, 4 * st_c4 + 2, salt), value(st_row1 + 64, 4 * st_c4 + 3, salt));
        }
        *reinterpret_cast<float4*>(sC + st_row0 * kStride + 4 * st_c4) = c0;
        *reinterpret_cast<float4*>(sC + st_row1 * kStride + 4 * st_c4) = c1;
        *reinterpret_cast<float4*>(sQ + st_row0 * kStride + 4 * st_c4) = q0;
        *reinterpret_cast<float4*>(sQ + st_row1 * kStride + 4 * st_c4) = q1;
    };
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    if (MODE != 4) {
        stage(gid >> 8);
        __syncthreads();
    }
    const int chunks = MODE == 4 ? iters / 8 : iters / 8;
    for (int ch = 0; ch < chunks; ++ch) {
        if (MODE == 4) {
            __syncthreads();
            stage((uint32_t)ch);
            __syncthreads();
        }
#pragma unroll 2
        for (int g = 0; g < 8; ++g) {
            float4 cv[4], qv[4];
            if (MODE == 13 || MODE == 14) {
                auto f = [](uint32_t w) { return __uint_as_float(0x3F800000u | (w & 0x007FFFFFu)); };
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint4 a = *reinterpret_cast<const uint4*>(&table[(((uint32_t)ch * 8u + g) * 4096u + (ty + 16 * i) * 4u) & ((1u << 20) - 4u)]);
                    const uint4 b = *reinterpret_cast<const uint4*>(&table[(((uint32_t)ch * 8u + g) * 4096u + 2048u + (tx + 16 * i) * 4u) & ((1u << 20) - 4u)]);
                    // (the words are used as they arrive: only the exponent bits are forced, by integer instructions on a COPY would defeat the purpose —
                    // so the table itself holds finite floats: see main)
                    cv[i] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
                    qv[i] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
                }
                (void)f;
                if (MODE == 14) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(cv[i].x), "+v"(cv[i].y), "+v"(cv[i].z), "+v"(cv[i].w));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[i] = *reinterpret_cast<const float4*>(sC + (ty + 16 * i) * kStride + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) qv[j] = *reinterpret_cast<const float4*>(sQ + (tx + 16 * j) * kStride + 4 * g);
            }
            if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 9 || MODE == 11 || MODE == 12) {
                // every result register is an operand of the statement: it sits behind all eight reads and nothing moves above it
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv[0].x), "+v"(cv[1].x), "+v"(cv[2].x), "+v"(cv[3].x), "+v"(qv[0].x), "+v"(qv[1].x), "+v"(qv[2].x), "+v"(qv[3].x));
                asm volatile("" : "+v"(cv[0].w), "+v"(cv[1].w), "+v"(cv[2].w), "+v"(cv[3].w), "+v"(qv[0].w), "+v"(qv[1].w), "+v"(qv[2].w), "+v"(qv[3].w));
                if (MODE == 2) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                if (MODE == 5) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
                if (MODE == 6 || MODE == 7 || MODE == 11 || MODE == 12) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (MODE == 6 || MODE == 11) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(cv[i].x), "+v"(cv[i].y), "+v"(cv[i].z), "+v"(cv[i].w));
                        else asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(qv[i].x), "+v"(qv[i].y), "+v"(qv[i].z), "+v"(qv[i].w));
                    }
                }
                if (MODE == 9) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(cv[i].x), "+v"(qv[i].x));
                }
                if (MODE == 3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(cv[i].x), "+v"(cv[i].y), "+v"(cv[i].z), "+v"(cv[i].w));
                        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(qv[i].x), "+v"(qv[i].y), "+v"(qv[i].z), "+v"(qv[i].w));
                    }
                }
            }
            if (MODE == 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float d0, d1, d2, d3;
                        asm volatile("v_sub_f32 %0, %4, %5\n\tv_sub_f32 %1, %6, %7\n\tv_sub_f32 %2, %8, %9\n\tv_sub_f32 %3, %10, %11"
                                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                                     : "v"(qv[j].x), "v"(cv[i].x), "v"(qv[j].y), "v"(cv[i].y), "v"(qv[j].z), "v"(cv[i].z), "v"(qv[j].w), "v"(cv[i].w));
                        asm volatile("v_mul_f32 %0, %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\t"
                                     "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %4, %4, %0"
                                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(acc[i][j]));
                    }
            } else if (MODE >= 10) {
                gpann::AccumulateGroup<GPANN_METRIC_L2>(acc, qv, cv);
            } else {
                gpann::AccumulateGroup<GPANN_METRIC_L2>(acc, cv, qv);
            }
        }
    }
    uint32_t r[kOutPerThread] = {gid, gid * 3u, gid * 5u, gid * 7u};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] ^= __float_as_uint(acc[i][j]) * (2u * i + 1u);
#pragma unroll
    for (int i = 0; i < kOutPerThread; ++i) out[(size_t)gid * kOutPerThread + i] = r[i];
}

// fourth family (round 6): ONE packed instruction form at a time, in inline assembly, so that the operand that comes straight from a load and the
// operand the op_sel modifier acts on are known:   t = A (+) B [modifier];  t = t * t;  acc += t       (A = freshly loaded pair, B = a pair written by the ALU)
//   SRC   0 ds_read_b64 (one pair per load)   1 ds_read_b128 (two pairs per load; the first is used)   2 global_load_dwordx2
//   FORM  0 plain   1 op_sel_hi:[0,1] on A (A's low half broadcast: the loaded source)   2 the same after A was copied by two v_mov_b32
//         3 op_sel_hi:[1,0] on B (the ALU-written source's low half broadcast; A plain)
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int SRC, int FORM>
__global__ void __launch_bounds__(256) Victim4(uint32_t* out, const uint32_t* table, int iters) {
    __shared__ float lds4[256 * 4 * 8];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    for (int i = threadIdx.x; i < 256 * 4 * 8; i += 256) lds4[i] = __uint_as_float(0x3F800000u | ((((uint32_t)i * 2654435761u) >> 9) & 0x007FFFFFu));
    __syncthreads();
    f32x2 acc[4], bsrc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c] = f32x2{0.f, 0.f};
        bsrc[c] = f32x2{__uint_as_float(0x3F800000u | (((gid * (2u * c + 3u)) & 1023u) << 13)), __uint_as_float(0x3FC00000u | (((gid * (2u * c + 5u)) & 511u) << 13))};
        asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(bsrc[c].x), "+v"(bsrc[c].y));   // (written by the ALU, whatever the compiler made of the initialisation)
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x2 a, t;
            const uint32_t slot = ((threadIdx.x + 37u * (uint32_t)it + 64u * c) & 255u) * 4u + ((uint32_t)it & 3u) * 1024u;   // (floats; 16-byte aligned)
            if (SRC == 0) {
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(slot * 4u) : "memory");
            } else if (SRC == 1) {
                f32x4v a4;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a4) : "v"(slot * 4u) : "memory");
                a = f32x2{a4.x, a4.y};   // (the low register pair of the four: no instruction)
            } else {
                const uint32_t* gp = table + ((gid * 4u + 4099u * (uint32_t)it + 256u * c) & ((1u << 20) - 4u));
                asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(gp) : "memory");
            }
            if (FORM == 2) asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(a.x), "+v"(a.y));
            if (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(bsrc[c]));
            else if (FORM == 1 || FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(bsrc[c]));
            else asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(bsrc[c]));
            asm volatile("v_pk_mul_f32 %0, %0, %0\n\ts_nop 1\n\tv_pk_add_f32 %1, %1, %0" : "+v"(t), "+v"(acc[c]));
        }
    }
    uint32_t r[kOutPerThread] = {gid, gid * 3u, gid * 5u, gid * 7u};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r[c] ^= __float_as_uint(acc[c].x) * (2u * c + 1u);
        r[(c + 1) & 3] += __float_as_uint(acc[c].y);
    }
#pragma unroll
    for (int i = 0; i < kOutPerThread; ++i) out[(size_t)gid * kOutPerThread + i] = r[i];
}
template <int SRC>
static void LaunchVictim4(int form, uint32_t* out, const uint32_t* table, int blocks, int iters, hipStream_t st) {
    switch (form) {
        case 0: hipLaunchKernelGGL((Victim4<SRC, 0>), dim3(blocks), dim3(256), 0, st, out, table, iters); break;
        case 1: hipLaunchKernelGGL((Victim4<SRC, 1>), dim3(blocks), dim3(256), 0, st, out, table, iters); break;
        case 2: hipLaunchKernelGGL((Victim4<SRC, 2>), dim3(blocks), dim3(256), 0, st, out, table, iters); break;
        default: hipLaunchKernelGGL((Victim4<SRC, 3>), dim3(blocks), dim3(256), 0, st, out, table, iters); break;
    }
}

static const char* kVictimName[kVictimKinds] = {"fp32 scalar (v_mul/v_add)", "fp32 packed (v_pk_mul/v_pk_add)", "integer (v_mul_lo/v_add)", "LDS reads (ds_read_b128)",
                                                "cross-lane (shfl / bpermute)", "fp64 (v_mul/v_add_f64)", "global loads (dwordx4)", "fp32 fused (v_fma_f32)",
                                                "control: CellDistKernel<true> (LDS + packed fp32)", "CellDistKernel<false> (no |.|_1 table: no v_and)",
                                                "packed fp32 with op_sel / neg modifiers, registers", "the same fed by ds_read_b128 (no barrier)",
                                                "the same, 32 accumulator pairs (> 100 registers)", "the same fed by LDS with s_barrier per round",
                                                "tile loop: 8 ds_read_b128 -> 4x4 tile (partial lgkmcnt waits)", "tile loop, s_waitcnt lgkmcnt(0) before the arithmetic",
                                                "tile loop, s_nop 7 x2 behind the full wait", "tile loop, operands copied (v_mov) before use",
                                                "tile loop with the global -> LDS staging and two barriers per chunk",
                                                "tile loop, full wait + s_nop 15 x10 (~160 cycles, no copy)", "tile loop, only the c operands copied (v_mov)",
                                                "tile loop, only the q operands copied (v_mov)", "tile loop on SCALAR fp32 (v_sub/v_mul/v_add, no packed ops)",
                                                "tile loop, operands copied by ONE v_mov each (.x only)",
                                                "tile loop, roles swapped (op_sel operand = the per-lane read), no copy", "roles swapped, quarter-uniform read copied",
                                                "roles swapped, per-lane read copied",
                                                "tile loop fed by GLOBAL loads (no LDS), no copy", "tile loop fed by global loads, op_sel operand copied",
                                                "asm: ds_read_b64 pair, plain sources", "asm: ds_read_b64 pair, op_sel_hi:[1,0] on the loaded source",
                                                "asm: ds_read_b64 pair, op_sel on the loaded source, copied first", "asm: ds_read_b64 pair, op_sel on the OTHER (ALU-written) source",
                                                "asm: ds_read_b128, plain sources", "asm: ds_read_b128, op_sel_hi:[1,0] on the loaded source",
                                                "asm: ds_read_b128, op_sel on the loaded source, copied first", "asm: ds_read_b128, op_sel on the OTHER (ALU-written) source",
                                                "asm: global_load_dwordx2, plain sources", "asm: global_load_dwordx2, op_sel_hi:[1,0] on the loaded source",
                                                "asm: global_load_dwordx2, op_sel on the loaded source, copied first", "asm: global_load_dwordx2, op_sel on the OTHER source"};

template <int KIND>
static void LaunchVictim(uint32_t* out, const uint32_t* table, int blocks, int iters, hipStream_t st) {
    hipLaunchKernelGGL(Victim<KIND>, dim3(blocks), dim3(256), 0, st, out, table, iters);
}
static void LaunchVictimKind(int kind, uint32_t* out, const uint32_t* table, int blocks, int iters, hipStream_t st) {
    switch (kind) {
        case 0: LaunchVictim<0>(out, table, blocks, iters, st); break;
        case 1: LaunchVictim<1>(out, table, blocks, iters, st); break;
        case 2: LaunchVictim<2>(out, table, blocks, iters, st); break;
        case 3: LaunchVictim<3>(out, table, blocks, iters, st); break;
        case 4: LaunchVictim<4>(out, table, blocks, iters, st); break;
        case 5: LaunchVictim<5>(out, table, blocks, iters, st); break;
        case 6: LaunchVictim<6>(out, table, blocks, iters, st); break;
        default: LaunchVictim<7>(out, table, blocks, iters, st); break;
    }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int first_kind = argc > 2 ? atoi(argv[2]) : 0;          // victims first_kind .. kVictimKinds - 1
    const unsigned neighbour_mask = argc > 3 ? (unsigned)strtoul(argv[3], nullptr, 0) : 0x1Fu;   // bit i: neighbour configuration i
    // (small launches, as coresidency_repro.hip's victim: 64 workgroups of 256 threads fit into the room the neighbours leave; launches of thousands of
    // workgroups beside persistent low-priority neighbours ran at three launches per second on the first attempt and one configuration never ended)
    const int blocks = 64, iters = 400;
    const size_t words = (size_t)blocks * 256 * kOutPerThread;
    int n_cus = 256;
    CHECK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t* d_table;
    CHECK(hipMalloc(&d_table, (size_t)4 << 20));
    {
        std::vector<uint32_t> h((size_t)1 << 20);
        // (every word is also a finite float in [1, 2) with a pseudo-random mantissa: the global-fed tile victims use the words as they arrive)
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3F800000u | ((((uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 3)) >> 9) & 0x007FFFFFu);
        CHECK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    // the control's inputs (coresidency_repro.hip): 64 task tiles x 36 cells, 64 dimensions
    const uint32_t cd_d = 64, cd_K = 36, cd_tiles = 64, cd_tasks = cd_tiles * 64;
    const size_t cd_words = (size_t)cd_tasks * cd_K * 2;   // both tables
    gpann::CellDistArgs cd{};
    {
        std::mt19937 rng(7);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> h_q((size_t)cd_tasks * cd_d), h_c((size_t)cd_K * cd_d);
        for (auto& v : h_q) v = 3.f * nd(rng);
        for (auto& v : h_c) v = 3.f * nd(rng);
        std::vector<uint32_t> h_tq(cd_tasks), h_scb = {0, cd_K};
        for (uint32_t t = 0; t < cd_tasks; ++t) h_tq[t] = (t * 2654435761u) % cd_tasks;
        std::vector<gpann::CellTileRef> h_tiles(cd_tiles);
        for (uint32_t i = 0; i < cd_tiles; ++i) h_tiles[i] = {0, i * 64, 64, 0};
        float *d_q, *d_c;
        uint32_t *d_tq, *d_scb;
        gpann::CellTileRef* d_tiles;
        CHECK(hipMalloc(&d_q, h_q.size() * 4));
        CHECK(hipMalloc(&d_c, h_c.size() * 4));
        CHECK(hipMalloc(&d_tq, cd_tasks * 4));
        CHECK(hipMalloc(&d_scb, 8));
        CHECK(hipMalloc(&d_tiles, cd_tiles * sizeof(gpann::CellTileRef)));
        CHECK(hipMemcpy(d_q, h_q.data(), h_q.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_c, h_c.data(), h_c.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_tq, h_tq.data(), cd_tasks * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_scb, h_scb.data(), 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_tiles, h_tiles.data(), cd_tiles * sizeof(gpann::CellTileRef), hipMemcpyHostToDevice));
        cd.queries = d_q; cd.task_query = d_tq; cd.center = d_c; cd.shard_cell_begin = d_scb; cd.tiles = d_tiles; cd.kmax = cd_K; cd.d = cd_d; cd.ld = cd_d;
    }
    const size_t buf_words = std::max(words, cd_words);
    constexpr int kStreams = 3, kSlots = 4;
    uint32_t *d_ref, *d_out[kStreams][kSlots];
    CHECK(hipMalloc(&d_ref, buf_words * 4));
    for (auto& per_stream : d_out)
        for (auto& p : per_stream) CHECK(hipMalloc(&p, buf_words * 4));
    auto launch = [&](int kind, uint32_t* out, hipStream_t st) {
        if (kind < 8) {
            LaunchVictimKind(kind, out, d_table, blocks, iters, st);
        } else if (kind == 8 || kind == 9) {
            gpann::CellDistArgs a = cd;
            a.out = reinterpret_cast<float*>(out);
            a.out_l1 = kind == 8 ? reinterpret_cast<float*>(out) + (size_t)cd_tasks * cd_K : nullptr;
            if (kind == 8) hipLaunchKernelGGL(gpann::CellDistKernel<true>, dim3(cd_tiles, 1), dim3(256), 0, st, a);
            else hipLaunchKernelGGL(gpann::CellDistKernel<false>, dim3(cd_tiles, 1), dim3(256), 0, st, a);
        } else if (kind == 10) {
            hipLaunchKernelGGL((Victim2<0, 8>), dim3(blocks), dim3(256), 0, st, out, iters);
        } else if (kind == 11) {
            hipLaunchKernelGGL((Victim2<1, 8>), dim3(blocks), dim3(256), 0, st, out, iters);
        } else if (kind == 12) {
            hipLaunchKernelGGL((Victim2<2, 32>), dim3(blocks), dim3(256), 0, st, out, iters);
        } else if (kind == 13) {
            hipLaunchKernelGGL((Victim2<3, 8>), dim3(blocks), dim3(256), 0, st, out, iters);
        } else if (kind == 14) {
            hipLaunchKernelGGL(Victim3<0>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 15) {
            hipLaunchKernelGGL(Victim3<1>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 16) {
            hipLaunchKernelGGL(Victim3<2>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 17) {
            hipLaunchKernelGGL(Victim3<3>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 18) {
            hipLaunchKernelGGL(Victim3<4>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 19) {
            hipLaunchKernelGGL(Victim3<5>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 20) {
            hipLaunchKernelGGL(Victim3<6>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 21) {
            hipLaunchKernelGGL(Victim3<7>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 22) {
            hipLaunchKernelGGL(Victim3<8>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 23) {
            hipLaunchKernelGGL(Victim3<9>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 24) {
            hipLaunchKernelGGL(Victim3<10>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 25) {
            hipLaunchKernelGGL(Victim3<11>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 26) {
            hipLaunchKernelGGL(Victim3<12>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 27) {
            hipLaunchKernelGGL(Victim3<13>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else if (kind == 28) {
            hipLaunchKernelGGL(Victim3<14>, dim3(blocks), dim3(256), 0, st, out, d_table, iters);
        } else {
            const int v = kind - 29, src = v / 4, form = v % 4;
            if (src == 0) LaunchVictim4<0>(form, out, d_table, blocks, iters, st);
            else if (src == 1) LaunchVictim4<1>(form, out, d_table, blocks, iters, st);
            else LaunchVictim4<2>(form, out, d_table, blocks, iters, st);
        }
    };
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sv[kStreams], sn;
    for (auto& s : sv) CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    CHECK(hipStreamCreateWithPriority(&sn, hipStreamNonBlocking, least));
    int *d_stop, *d_sink;
    CHECK(hipMalloc(&d_stop, 4));
    CHECK(hipMalloc(&d_sink, 4));
    struct NeighbourConfig { const void* fn; const char* what; };
    const NeighbourConfig neighbours[] = {
        {nullptr, "no neighbour"},
        {(const void*)Neighbour<0>, "v_mfma_i32_32x32x32_i8"},
        {(const void*)Neighbour<1>, "v_mfma_f32_32x32x16_bf16"},
        {(const void*)Neighbour<2>, "v_mfma_i32_16x16x64_i8"},
        {(const void*)Neighbour<3>, "v_mfma_i32_32x32x32_i8 + s_nop 15 x2 between issues"},
    };
    for (const NeighbourConfig& c : neighbours)
        if (c.fn) CHECK(hipFuncSetAttribute(c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    std::vector<uint32_t> ref(buf_words), got(buf_words);
    printf("%-52s | %-46s | launches | differ | words by lane quarter (0-15, 16-31, 32-47, 48-63)\n", "neighbour on every CU", "victim");
    for (int nbi = 0; nbi < (int)(sizeof(neighbours) / sizeof(neighbours[0])); ++nbi) {
        const NeighbourConfig& nb = neighbours[nbi];
        if (!((neighbour_mask >> nbi) & 1u)) continue;
        for (int kind = first_kind; kind < kVictimKinds; ++kind) {
            const bool table_kind = kind == 8 || kind == 9;
            const size_t w_kind = !table_kind ? words : kind == 8 ? cd_words : cd_words / 2;
            // reference: the victim alone on an idle chip
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(d_ref, 0xFF, buf_words * 4));
            launch(kind, d_ref, nullptr);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ref.data(), d_ref, w_kind * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemset(d_stop, 0, 4));
            CHECK(hipDeviceSynchronize());
            if (nb.fn) {
                int burst = 64;
                unsigned long long tick_limit = (unsigned long long)((seconds + 1.0) * 1e8);
                const int* stop_p = d_stop;
                int* sink_p = d_sink;
                void* kargs[] = {&stop_p, &sink_p, &burst, &tick_limit};
                CHECK(hipLaunchKernel(nb.fn, dim3(n_cus), dim3(512), kargs, 65536, sn));
            }
            unsigned long long launches = 0, bad = 0, quarter[4] = {0, 0, 0, 0};
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                for (int slot = 0; slot < kSlots; ++slot)
                    for (int s = 0; s < kStreams; ++s) launch(kind, d_out[s][slot], sv[s]);
                for (auto& s : sv) CHECK(hipStreamSynchronize(s));
                for (int slot = 0; slot < kSlots; ++slot)
                    for (int s = 0; s < kStreams; ++s) {
                        CHECK(hipMemcpy(got.data(), d_out[s][slot], w_kind * 4, hipMemcpyDeviceToHost));
                        ++launches;
                        if (std::memcmp(got.data(), ref.data(), w_kind * 4) == 0) continue;
                        ++bad;
                        for (size_t w = 0; w < w_kind; ++w) {
                            if (got[w] == ref[w]) continue;
                            // the lane that wrote word w: victims 0-7 thread = w / 4; the control: row = task, column = cell; cell = ct + ty + 16 i with
                            // ty = tid >> 4: the lane quarter is (cell & 3) (CellDistKernel's thread layout: lanes 16 q .. 16 q + 15 hold ty = 4 wave + q)
                            const unsigned q = !table_kind ? (unsigned)(((w / kOutPerThread) & 63) >> 4) : (unsigned)((w % cd_K) & 3);
                            quarter[q]++;
                        }
                    }
            }
            int one = 1;
            CHECK(hipMemcpy(d_stop, &one, 4, hipMemcpyHostToDevice));
            CHECK(hipDeviceSynchronize());
            printf("%-52s | %-46s | %8llu | %6llu | %llu %llu %llu %llu\n", nb.what, kVictimName[kind], launches, bad, quarter[0], quarter[1], quarter[2], quarter[3]);
            fflush(stdout);
        }
    }
    return 0;
}
