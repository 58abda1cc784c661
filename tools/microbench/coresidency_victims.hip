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
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o coresidency_victims coresidency_victims.hip && ./coresidency_victims [seconds per cell]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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
__global__ void __launch_bounds__(512, 1) Neighbour(const int* stop, int* sink, int max_rounds) {
    extern __shared__ char nlds[];
    asm volatile("" ::: "v127");
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<int*>(nlds)[i] = (i * 2654435761u) & 0x3C003C00u;
    __syncthreads();
    v4i a = {lane, lane * 3, lane * 5, lane * 7};
    v16i acc0 = {0}, acc1 = {0};
    v4i c0 = {0}, c1 = {0};
    v16f f0 = {0}, f1 = {0};
    for (int r = 0; r < max_rounds; ++r) {
        for (int b = 0; b < 64; ++b) {
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
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= acc0[i] ^ acc1[i] ^ __float_as_int(f0[i]) ^ __float_as_int(f1[i]);
    s ^= c0[0] ^ c0[1] ^ c0[2] ^ c0[3] ^ c1[0] ^ c1[1] ^ c1[2] ^ c1[3];
    if (s == 0x12345678) sink[0] = s;
}

constexpr int kVictimKinds = 8;
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

static const char* kVictimName[kVictimKinds] = {"fp32 scalar (v_mul/v_add)", "fp32 packed (v_pk_mul/v_pk_add)", "integer (v_mul_lo/v_add)", "LDS reads (ds_read_b128)",
                                                "cross-lane (shfl / bpermute)", "fp64 (v_mul/v_add_f64)", "global loads (dwordx4)", "fp32 fused (v_fma_f32)"};

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
    const int blocks = 2048, iters = 400;
    const size_t words = (size_t)blocks * 256 * kOutPerThread;
    int n_cus = 256;
    CHECK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    uint32_t* d_table;
    CHECK(hipMalloc(&d_table, (size_t)4 << 20));
    {
        std::vector<uint32_t> h((size_t)1 << 20);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 3);
        CHECK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    constexpr int kStreams = 3;
    uint32_t *d_ref, *d_out[kStreams];
    CHECK(hipMalloc(&d_ref, words * 4));
    for (auto& p : d_out) CHECK(hipMalloc(&p, words * 4));
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
    std::vector<uint32_t> ref(words), got(words);
    printf("%-52s | %-34s | launches | differ | words by lane quarter (0-15, 16-31, 32-47, 48-63)\n", "neighbour on every CU (128 registers per wavefront)", "victim");
    for (const NeighbourConfig& nb : neighbours) {
        for (int kind = 0; kind < kVictimKinds; ++kind) {
            // reference: the victim alone on an idle chip
            CHECK(hipDeviceSynchronize());
            LaunchVictimKind(kind, d_ref, d_table, blocks, iters, nullptr);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ref.data(), d_ref, words * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemset(d_stop, 0, 4));
            CHECK(hipDeviceSynchronize());
            if (nb.fn) {
                int max_rounds = 1 << 30;
                const int* stop_p = d_stop;
                int* sink_p = d_sink;
                void* kargs[] = {&stop_p, &sink_p, &max_rounds};
                CHECK(hipLaunchKernel(nb.fn, dim3(n_cus), dim3(512), kargs, 65536, sn));
            }
            unsigned long long launches = 0, bad = 0, quarter[4] = {0, 0, 0, 0};
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                for (int rep = 0; rep < 4; ++rep)
                    for (int s = 0; s < kStreams; ++s) LaunchVictimKind(kind, d_out[s], d_table, blocks, iters, sv[s]);
                for (auto& s : sv) CHECK(hipStreamSynchronize(s));
                for (int s = 0; s < kStreams; ++s) {   // (the last of the four launches per stream is the one compared)
                    CHECK(hipMemcpy(got.data(), d_out[s], words * 4, hipMemcpyDeviceToHost));
                    ++launches;
                    if (std::memcmp(got.data(), ref.data(), words * 4) == 0) continue;
                    ++bad;
                    for (size_t w = 0; w < words; ++w)
                        if (got[w] != ref[w]) quarter[((w / kOutPerThread) & 63) >> 4]++;
                }
            }
            int one = 1;
            CHECK(hipMemcpy(d_stop, &one, 4, hipMemcpyHostToDevice));
            CHECK(hipDeviceSynchronize());
            printf("%-52s | %-34s | %8llu | %6llu | %llu %llu %llu %llu\n", nb.what, kVictimName[kind], launches, bad, quarter[0], quarter[1], quarter[2], quarter[3]);
            fflush(stdout);
        }
    }
    return 0;
}
