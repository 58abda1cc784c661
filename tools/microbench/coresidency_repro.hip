// coresidency_repro.hip — does a wavefront's result change when it shares a SIMD with matrix-core wavefronts of ANOTHER launch?
//
// Background: profiles/r05/overlap_race.md.  With GPANN_SCAN_OVERLAP and 64-dimensional rows, CellDistKernel wavefronts that ran beside
// FilterKernelCells wavefronts (140 - 230 registers: room on the SIMD) now and then wrote table entries that differ from the same launch alone —
// one accumulator register, lanes 48 - 63.  This program takes the product out of the picture as far as it can:
//   * the VICTIM is the product's own CellDistKernel (csrc/scan_cells.h) on fixed synthetic inputs, launched over and over on three streams; every
//     launch's table is compared bit for bit with the table of a launch that had the chip to itself;
//   * the NEIGHBOUR is a kernel of this file: persistent 512-thread workgroups, one per CU, that do nothing but v_mfma_i32_32x32x32_i8 on
//     registers and LDS (64 KB of LDS, REGS architectural registers: 184 leaves room on the SIMD for the victim's 116, 256 does not), on a fourth, low-priority stream;
//   * FILLERS: single-wavefront workgroups that load, stage through LDS and shuffle (what RescoreKernel looks like to the hardware).
// Output: launches checked, launches whose table differed, and for the first few the (row, column) blocks that differ.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o coresidency_repro coresidency_repro.hip
//   ./coresidency_repro [seconds per configuration]
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

// the neighbour: persistent 512-thread workgroups that keep one kind of instruction busy on registers and LDS until *stop != 0 (or max_rounds)
//   KIND 0: v_mfma_i32_32x32x32_i8 (new on gfx950)   1: no matrix instruction at all (integer multiply-adds on the same operands)
//   KIND 2: v_mfma_f32_32x32x16_bf16 (new on gfx950)  3: v_mfma_f32_32x32x8_f16 (the instruction of earlier CDNA parts)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int REGS, int KIND>
__global__ void __launch_bounds__(512, 1) MatrixNeighbour(const int* stop, int* sink, int burst, int max_rounds) {
    extern __shared__ char nlds[];
    if (REGS >= 256) asm volatile("" ::: "v255");
    else if (REGS >= 184) asm volatile("" ::: "v183");   // (184 registers = the product's 8-bit sample kernel at 64 dimensions: two wavefronts leave 144 per SIMD lane)
    else asm volatile("" ::: "v127");
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<int*>(nlds)[i] = (i * 2654435761u) & 0x3C003C00u;   // (small finite halves / bf16s)
    __syncthreads();
    v4i a = {lane, lane * 3, lane * 5, lane * 7};
    v16i acc0 = {0}, acc1 = {0};
    v16f f0 = {0}, f1 = {0};
    for (int r = 0; r < max_rounds; ++r) {
        for (int b = 0; b < burst; ++b) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const v4i bv = *reinterpret_cast<const v4i*>(nlds + ((b * 8 + ks) & 31) * 1024 + lane * 16);
                if (KIND == 0) {
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, a, acc1, 0, 0, 0);
                } else if (KIND == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        acc0[i] = acc0[i] * bv[i & 3] + a[i & 3];
                        acc1[i] = acc1[i] * a[i & 3] + bv[(i + 1) & 3];
                    }
                } else if (KIND == 2) {
                    f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, bv), f0, 0, 0, 0);
                    f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, bv), f1, 0, 0, 0);
                } else {
                    const f16x4_t h = {(_Float16)1, (_Float16)0.5, (_Float16)0.25, (_Float16)2};
                    f0 = __builtin_amdgcn_mfma_f32_32x32x8f16(h, h, f0, 0, 0, 0);
                    f1 = __builtin_amdgcn_mfma_f32_32x32x8f16(h, h, f1, 0, 0, 0);
                    acc0[0] += bv[0];
                }
            }
        }
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= acc0[i] ^ acc1[i] ^ __float_as_int(f0[i]) ^ __float_as_int(f1[i]);
    if (s == 0x12345678) sink[0] = s;
}

// Do SIBLING wavefronts disturb each other?  512-thread workgroups of 256 registers (two wavefronts per SIMD, as the product's filter kernels run): every
// wavefront interleaves v_mfma_i32_32x32x32_i8 with vector arithmetic on 16 registers of its own, all from fixed inputs; check[block * 8 + wave] = a checksum
// of both — the same in every workgroup unless something changed a register.
__global__ void __launch_bounds__(512, 1) SiblingCheck(int rounds, uint32_t* check) {
    asm volatile("" ::: "v255");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i a = {lane * 3 + 1, lane * 5 + 2, lane * 7 + 3, lane * 11 + 4}, b = {lane + wave, lane * 2 + 1, lane * 4 + 1, lane * 8 + 1};
    v16i acc = {0};
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = (float)(lane + i);
    for (int r = 0; r < rounds; ++r) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = w[i] * 1.0000001f + 0.5f;
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc, 0, 0, 0);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = s * 31u + (uint32_t)acc[i] + __float_as_uint(w[i]);
#pragma unroll
    for (int off = 32; off; off >>= 1) s = s * 17u + (uint32_t)__shfl_xor((int)s, off);
    if (lane == 0) check[blockIdx.x * 8 + wave] = s;
}

// fillers: one wavefront per workgroup, 4.5 KB of LDS, coalesced loads of rows, LDS round trip, shuffles
__global__ void __launch_bounds__(64) Filler(const float* rows, uint32_t n_rows, uint32_t ld, float* out) {
    extern __shared__ float flds[];
    const int lane = threadIdx.x;
    float acc = 0.f;
    uint32_t r = (blockIdx.x * 2654435761u) % n_rows;
    for (int it = 0; it < 8; ++it) {
        const float4 v = reinterpret_cast<const float4*>(rows + (size_t)r * ld)[lane & 15];
        reinterpret_cast<float4*>(flds)[lane] = v;
        __builtin_amdgcn_wave_barrier();
        const float4 w = reinterpret_cast<const float4*>(flds)[(lane * 7) & 63];
        acc += (w.x - v.x) * (w.y - v.y) + __shfl(w.z, (lane + 5) & 63);
        r = (r * 1664525u + 1013904223u) % n_rows;
    }
    out[(size_t)blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 8.0;
    const uint32_t d = 64, ld = 64, K = 36, kmax = 36, tiles = 64, tasks = tiles * 64, nq = tasks;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> h_q((size_t)nq * ld), h_c((size_t)K * ld);
    for (auto& v : h_q) v = 3.f * nd(rng);
    for (auto& v : h_c) v = 3.f * nd(rng);
    std::vector<uint32_t> h_tq(tasks), h_scb = {0, K};
    for (uint32_t t = 0; t < tasks; ++t) h_tq[t] = (t * 2654435761u) % nq;
    std::vector<gpann::CellTileRef> h_tiles(tiles);
    for (uint32_t i = 0; i < tiles; ++i) h_tiles[i] = {0, i * 64, 64, 0};
    float *d_q, *d_c, *d_ref, *d_ref1;
    uint32_t *d_tq, *d_scb;
    gpann::CellTileRef* d_tiles;
    CHECK(hipMalloc(&d_q, h_q.size() * 4));
    CHECK(hipMalloc(&d_c, h_c.size() * 4));
    CHECK(hipMalloc(&d_tq, tasks * 4));
    CHECK(hipMalloc(&d_scb, 8));
    CHECK(hipMalloc(&d_tiles, tiles * sizeof(gpann::CellTileRef)));
    CHECK(hipMemcpy(d_q, h_q.data(), h_q.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_c, h_c.data(), h_c.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tq, h_tq.data(), tasks * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_scb, h_scb.data(), 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tiles, h_tiles.data(), tiles * sizeof(gpann::CellTileRef), hipMemcpyHostToDevice));
    const size_t table = (size_t)tasks * kmax;
    CHECK(hipMalloc(&d_ref, table * 4));
    CHECK(hipMalloc(&d_ref1, table * 4));
    constexpr int kStreams = 3, kSlots = 4;
    float* d_out[kStreams][kSlots][2];
    for (auto& per_stream : d_out)
        for (auto& slot : per_stream)
            for (auto& p : slot) CHECK(hipMalloc(&p, table * 4));
    gpann::CellDistArgs cd{};
    cd.queries = d_q; cd.task_query = d_tq; cd.center = d_c; cd.shard_cell_begin = d_scb; cd.tiles = d_tiles; cd.kmax = kmax; cd.d = d; cd.ld = ld;
    auto launch_victim = [&](float* out, float* out1, hipStream_t st) {
        cd.out = out; cd.out_l1 = out1;
        hipLaunchKernelGGL(gpann::CellDistKernel<true>, dim3(tiles, 1), dim3(256), 0, st, cd);
    };
    CHECK(hipMemset(d_ref, 0xFF, table * 4));
    CHECK(hipMemset(d_ref1, 0xFF, table * 4));
    launch_victim(d_ref, d_ref1, nullptr);
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref(table), ref1(table), got(table);
    CHECK(hipMemcpy(ref.data(), d_ref, table * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ref1.data(), d_ref1, table * 4, hipMemcpyDeviceToHost));

    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sv[kStreams], sn, sf;
    for (auto& s : sv) CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
    CHECK(hipStreamCreateWithPriority(&sn, hipStreamNonBlocking, least));
    CHECK(hipStreamCreateWithPriority(&sf, hipStreamNonBlocking, greatest));
    int *d_stop, *d_sink;
    float* d_fill;
    CHECK(hipMalloc(&d_stop, 4));
    CHECK(hipMalloc(&d_sink, 4));
    CHECK(hipMalloc(&d_fill, (size_t)65536 * 64 * 4));
    int n_cus = 256;
    CHECK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    struct Config { const void* fn; int grid; const char* what; };
    const Config configs[] = {
        {nullptr, 0, "victims and fillers only"},
        {(const void*)MatrixNeighbour<184, 0>, n_cus, "beside v_mfma_i32_32x32x32_i8 wavefronts of 184 registers, every CU"},
        {(const void*)MatrixNeighbour<256, 0>, n_cus / 2, "v_mfma_i32_32x32x32_i8 wavefronts of 256 registers on half of the CUs (nothing fits beside them)"},
        {(const void*)MatrixNeighbour<184, 1>, n_cus, "beside wavefronts of 184 registers WITHOUT matrix instructions, every CU"},
        {(const void*)MatrixNeighbour<184, 2>, n_cus, "beside v_mfma_f32_32x32x16_bf16 wavefronts of 184 registers, every CU"},
        {(const void*)MatrixNeighbour<184, 3>, n_cus, "beside v_mfma_f32_32x32x8_f16 wavefronts of 184 registers, every CU"},
        {(const void*)MatrixNeighbour<128, 0>, n_cus, "beside v_mfma_i32_32x32x32_i8 wavefronts of 128 registers, every CU"},
    };
    for (const Config& c : configs)
        if (c.fn) CHECK(hipFuncSetAttribute(c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));

    for (int config = 0; config < (int)(sizeof(configs) / sizeof(configs[0])); ++config) {
        CHECK(hipMemset(d_stop, 0, 4));
        CHECK(hipDeviceSynchronize());
        if (configs[config].fn) {
            int burst = 64, max_rounds = 1 << 30;
            const int* stop_p = d_stop;
            int* sink_p = d_sink;
            void* kargs[] = {&stop_p, &sink_p, &burst, &max_rounds};
            CHECK(hipLaunchKernel(configs[config].fn, dim3(configs[config].grid), dim3(512), kargs, 65536, sn));
        }
        CHECK(hipGetLastError());
        unsigned long long launches = 0, bad_launches = 0, bad_words = 0;
        int shown = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int slot = 0; slot < kSlots; ++slot)
                for (int s = 0; s < kStreams; ++s) {
                    launch_victim(d_out[s][slot][0], d_out[s][slot][1], sv[s]);
                    hipLaunchKernelGGL(Filler, dim3(16384), dim3(64), 4608, sf, d_q, nq, ld, d_fill);
                }
            for (auto& s : sv) CHECK(hipStreamSynchronize(s));
            CHECK(hipStreamSynchronize(sf));
            for (int slot = 0; slot < kSlots; ++slot)
                for (int s = 0; s < kStreams; ++s)
                    for (int which = 0; which < 2; ++which) {
                        CHECK(hipMemcpy(got.data(), d_out[s][slot][which], table * 4, hipMemcpyDeviceToHost));
                        const std::vector<float>& want = which ? ref1 : ref;
                        ++launches;
                        if (std::memcmp(got.data(), want.data(), table * 4) == 0) continue;
                        ++bad_launches;
                        for (size_t w = 0; w < table; ++w) {
                            if (std::memcmp(&got[w], &want[w], 4) == 0) continue;
                            ++bad_words;
                            if (shown < 6) {
                                printf("  config %d: table %d row %zu (row %% 16 = %zu) column %zu: %.9g instead of %.9g\n", config, which, w / kmax, (w / kmax) % 16, w % kmax, got[w], want[w]);
                                ++shown;
                            }
                        }
                    }
        }
        int one = 1;
        CHECK(hipMemcpy(d_stop, &one, 4, hipMemcpyHostToDevice));
        CHECK(hipDeviceSynchronize());
        printf("config %d (%s): %llu tables checked, %llu differ from the launch made alone (%llu words)\n", config, configs[config].what, launches, bad_launches, bad_words);
        fflush(stdout);
    }
    {   // sibling wavefronts of one launch
        uint32_t* d_check;
        const int blocks = n_cus * 8;
        CHECK(hipMalloc(&d_check, (size_t)blocks * 8 * 4));
        unsigned long long launches = 0, bad = 0;
        std::vector<uint32_t> h(blocks * 8);
        for (int rep = 0; rep < 40; ++rep) {
            hipLaunchKernelGGL(SiblingCheck, dim3(blocks), dim3(512), 0, nullptr, 20000, d_check);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), d_check, h.size() * 4, hipMemcpyDeviceToHost));
            for (int blk = 0; blk < blocks; ++blk)
                for (int wv = 0; wv < 8; ++wv) {
                    ++launches;
                    if (h[blk * 8 + wv] != h[wv]) ++bad;
                }
        }
        printf("sibling wavefronts (v_mfma_i32_32x32x32_i8 + vector arithmetic, 256 registers, two per SIMD): %llu wavefronts checked, %llu checksums differ from workgroup 0's\n", launches, bad);
    }
    return 0;
}
