// matrix_neighbour.hip — a process that does nothing but keep matrix-core wavefronts resident on every compute unit for a while: the "other work on
// the GPU" of the co-residency experiments (profiles/r05/overlap_race.md, profiles/r06/router_event.md), as a program of its own so that the product
// under test runs in ANOTHER process, unchanged.
//   ./matrix_neighbour SECONDS [KIND] [THREADS] [WORKGROUPS]
//     KIND        0 v_mfma_i32_32x32x32_i8 (default)   1 v_mfma_f32_32x32x16_bf16   2 integer multiply-adds, no matrix instruction
//     THREADS     256 (default: ONE wavefront per SIMD, 128 registers — three quarters of every SIMD's register file stay free for others) or 512
//     WORKGROUPS  default: the number of compute units
// The kernel leaves by itself after SECONDS of the 100 MHz wall clock (no host signal needed; a killed host cannot leave it spinning for long).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o matrix_neighbour matrix_neighbour.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

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
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(512) Neighbour(int* sink, int burst, unsigned long long tick_limit) {
    __shared__ int nlds[4096];
    asm volatile("" ::: "v127");
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) nlds[i] = (i * 2654435761u) & 0x3C003C00u;
    __syncthreads();
    v4i a = {lane, lane * 3, lane * 5, lane * 7};
    v16i acc0 = {0}, acc1 = {0};
    v16f f0 = {0}, f1 = {0};
    const unsigned long long t_begin = wall_clock64();
    for (;;) {
        for (int b = 0; b < burst; ++b) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const v4i bv = *reinterpret_cast<const v4i*>(reinterpret_cast<const char*>(nlds) + ((b * 8 + ks) & 15) * 1024 + lane * 16);
                if (KIND == 0) {
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bv, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bv, a, acc1, 0, 0, 0);
                } else if (KIND == 1) {
                    f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, bv), f0, 0, 0, 0);
                    f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bv), __builtin_bit_cast(bf16x8_t, bv), f1, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        acc0[i] = acc0[i] * bv[i & 3] + a[i & 3];
                        acc1[i] = acc1[i] * a[i & 3] + bv[(i + 1) & 3];
                    }
                }
            }
        }
        if (wall_clock64() - t_begin > tick_limit) break;
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= acc0[i] ^ acc1[i] ^ __float_as_int(f0[i]) ^ __float_as_int(f1[i]);
    if (s == 0x12345678) sink[0] = s;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const int threads = argc > 3 ? atoi(argv[3]) : 256;
    int n_cus = 256;
    CHECK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int wgs = argc > 4 ? atoi(argv[4]) : n_cus;
    int* d_sink;
    CHECK(hipMalloc(&d_sink, 4));
    int least = 0, greatest = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st;
    CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least));
    // one launch per ~second: between two launches the compute units are handed out anew (the product's workgroups get their turn everywhere)
    const double slice = seconds < 1.0 ? seconds : 1.0;
    const unsigned long long ticks = (unsigned long long)(slice * 1e8);
    int launches = 0;
    for (double done = 0; done < seconds; done += slice, ++launches) {
        const int burst = 64;
        if (kind == 0) hipLaunchKernelGGL(Neighbour<0>, dim3(wgs), dim3(threads), 0, st, d_sink, burst, ticks);
        else if (kind == 1) hipLaunchKernelGGL(Neighbour<1>, dim3(wgs), dim3(threads), 0, st, d_sink, burst, ticks);
        else hipLaunchKernelGGL(Neighbour<2>, dim3(wgs), dim3(threads), 0, st, d_sink, burst, ticks);
        CHECK(hipGetLastError());
        CHECK(hipStreamSynchronize(st));
    }
    printf("matrix_neighbour: kind %d, %d workgroups of %d threads, %d launches of %.2f s\n", kind, wgs, threads, launches, slice);
    return 0;
}
