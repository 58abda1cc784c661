// clock_under_load.hip — what shader clock does the chip sustain under a given kind of load?  Every wavefront reads the shader-clock counter
// (s_memtime) and the 100 MHz wall clock (s_memrealtime) before and after its loop: cycles / wall time = the clock it ran at.  The nominal peaks of
// MI355X_MICROARCH.md are quoted at 2.4 GHz; a kernel that keeps every vector ALU busy is power-limited below that, and a roofline fraction measured
// against the nominal peak cannot exceed (sustained clock / 2.4 GHz) however good the kernel is.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o clock_under_load clock_under_load.hip && ./clock_under_load [milliseconds per load]
//   loads: 0 scalar fp32 (v_mul / v_add chains, every SIMD full: 8 wavefronts)   1 the same with one ds_read_b128 per 24 vector instructions
//          2 v_mfma_f32_32x32x16_bf16 back to back   3 v_mfma_i32_32x32x32_i8   4 streaming global loads (HBM-bound)   5 idle (s_sleep)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct Sample {
    unsigned long long cycles, ticks, insts;
};

template <int LOAD>
__global__ void __launch_bounds__(512) Load(Sample* out, const float4* stream, size_t stream_n, unsigned long long tick_limit, float* sink) {
    __shared__ float4 lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = make_float4(1.0f + i * 1e-6f, 1.0f, 1.0f, 1.0f);
    __syncthreads();
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 1.0f + (float)(lane + i) * 1e-3f;
    v16f facc = {0};
    v16i iacc = {0};
    v4i a = {lane, lane * 3, lane * 5, lane * 7};
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long insts = 0;
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long t0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (;;) {
        for (int rep = 0; rep < 64; ++rep) {
            if (LOAD == 0 || LOAD == 1) {
                if (LOAD == 1) {
                    const float4 v = lds[(lane * 7 + rep) & 1023];
                    x[0] += v.x * 1e-9f;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(x[i]) : "v"(1.0000001f), "v"(1e-7f));
                insts += LOAD == 1 ? 24 : 32;   // (LOAD 1: counted per the same 32, one read in between — see the table's note)
            } else if (LOAD == 2) {
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, a);
#pragma unroll
                for (int i = 0; i < 8; ++i) facc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, facc, 0, 0, 0);
                insts += 8;
            } else if (LOAD == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) iacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, iacc, 0, 0, 0);
                insts += 8;
            } else if (LOAD == 4) {
                const float4 v = stream[pos % stream_n];
                pos += (size_t)gridDim.x * blockDim.x;
                keep.x += v.x; keep.y += v.y; keep.z += v.z; keep.w += v.w;
                insts += 1;
            } else {
                __builtin_amdgcn_s_sleep(64);
                insts += 1;
            }
        }
        if (wall_clock64() - t0 > tick_limit) break;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = wall_clock64();
    float s = keep.x + keep.y + keep.z + keep.w + facc[0] + (float)iacc[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[(size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = {c1 - c0, t1 - t0, insts};
}

int main(int argc, char** argv) {
    const double ms = argc > 1 ? atof(argv[1]) : 200.0;
    int n_cus = 256;
    CHECK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, 0));
    const size_t stream_n = (size_t)1 << 28;   // 4 GiB of float4
    float4* d_stream;
    CHECK(hipMalloc(&d_stream, stream_n * sizeof(float4)));
    CHECK(hipMemset(d_stream, 0, stream_n * sizeof(float4)));
    float* d_sink;
    CHECK(hipMalloc(&d_sink, 4));
    const int wgs = n_cus * 4, waves = wgs * 8;   // 512-thread workgroups, four per CU: eight wavefronts per SIMD
    Sample* d_out;
    CHECK(hipMalloc(&d_out, waves * sizeof(Sample)));
    const char* names[] = {"scalar fp32 v_mul/v_add, 8 wavefronts per SIMD", "the same + one ds_read_b128 per 32 vector instructions", "v_mfma_f32_32x32x16_bf16 back to back",
                           "v_mfma_i32_32x32x32_i8 back to back", "streaming global loads (HBM-bound)", "idle (s_sleep)"};
    printf("%-58s | shader clock MHz (median of wavefronts) | min | max | vector or matrix instructions per cycle and SIMD\n", "load");
    for (int load = 0; load < 6; ++load) {
        const unsigned long long ticks = (unsigned long long)(ms * 1e5);
        switch (load) {
            case 0: hipLaunchKernelGGL(Load<0>, dim3(wgs), dim3(512), 0, 0, d_out, d_stream, stream_n, ticks, d_sink); break;
            case 1: hipLaunchKernelGGL(Load<1>, dim3(wgs), dim3(512), 0, 0, d_out, d_stream, stream_n, ticks, d_sink); break;
            case 2: hipLaunchKernelGGL(Load<2>, dim3(wgs), dim3(512), 0, 0, d_out, d_stream, stream_n, ticks, d_sink); break;
            case 3: hipLaunchKernelGGL(Load<3>, dim3(wgs), dim3(512), 0, 0, d_out, d_stream, stream_n, ticks, d_sink); break;
            case 4: hipLaunchKernelGGL(Load<4>, dim3(wgs), dim3(512), 0, 0, d_out, d_stream, stream_n, ticks, d_sink); break;
            default: hipLaunchKernelGGL(Load<5>, dim3(wgs), dim3(512), 0, 0, d_out, d_stream, stream_n, ticks, d_sink); break;
        }
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        std::vector<Sample> h(waves);
        CHECK(hipMemcpy(h.data(), d_out, waves * sizeof(Sample), hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        double inst_rate = 0;
        for (const Sample& s : h) {
            if (s.ticks == 0) continue;
            mhz.push_back((double)s.cycles / ((double)s.ticks / 100.0));   // cycles per microsecond
            inst_rate += (double)s.insts / (double)s.cycles;                 // this wavefront's instructions per shader cycle
        }
        std::sort(mhz.begin(), mhz.end());
        // eight wavefronts share a SIMD: their rates add up to the SIMD's
        printf("%-58s | %8.0f | %6.0f | %6.0f | %.3f\n", names[load], mhz[mhz.size() / 2], mhz.front(), mhz.back(), inst_rate / (n_cus * 4.0));
        fflush(stdout);
    }
    return 0;
}
