// Microbenchmark: VALU issue rates on gfx950 that decide the shape of the exact (non-FMA) scan kernel.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
// Measures wave64 instruction throughput per SIMD for plain / packed f32 ops, with SGPR operands,
// with LDS broadcast reads interleaved, and the scalar-load (SMEM) stream rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

// 16 independent accumulators, 3 ops each per inner step => 48 VALU per step.
template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, const float* in, int iters) {
    float a[16];
    float b = in[threadIdx.x & 63];
    float s = in[0];   // uniform
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = in[(threadIdx.x + i) & 255];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {   // plain VGPR ops: sub, mul, add
                asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %0\n v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            } else if (MODE == 1) {  // SGPR operand
                asm volatile("v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %0, %0\n v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// packed: 8 float2 accumulators, 3 pk ops each => 24 pk instrs = 48 lane-ops per step
__global__ void __launch_bounds__(256) k_pk(float* out, const float* in, int iters) {
    float2v a[8];
    float2v b = {in[threadIdx.x & 63], in[(threadIdx.x + 1) & 63]};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = float2v{in[(threadIdx.x + i) & 255], in[(threadIdx.x + i + 7) & 255]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %0, %0, %0\n v_pk_add_f32 %0, %0, %1"
                         : "+v"(a[i]) : "v"(b));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// VALU + LDS broadcast: per 12 VALU one ds_read_b128 from a wave-uniform address.
template <int READS_PER_48>
__global__ void __launch_bounds__(256) k_lds(float* out, const float* in, int iters) {
    __shared__ float4v tile[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = float4v{in[i & 255], in[(i + 1) & 255], in[(i + 2) & 255], in[(i + 3) & 255]};
    __syncthreads();
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = in[(threadIdx.x + i) & 255];
    int w = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        float4v p[READS_PER_48 > 0 ? READS_PER_48 : 1];
#pragma unroll
        for (int r = 0; r < READS_PER_48; ++r) p[r] = tile[(it * READS_PER_48 + r + w * 16) & 1023];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float b = READS_PER_48 > 0 ? p[i % (READS_PER_48 > 0 ? READS_PER_48 : 1)][i & 3] : 1.0f;
            asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %0\n v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// VALU + SMEM stream: per 48 VALU one s_load_dwordx16 (64 B) from a per-wave advancing pointer.
__global__ void __launch_bounds__(256) k_smem(float* out, const float* in, const float* stream, size_t stream_floats, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = in[(threadIdx.x + i) & 255];
    // all waves of a block read the same stream segment (like all query-waves reading one point block)
    const float* base = stream + (size_t)blockIdx.x * 16 * (size_t)iters % stream_floats;
    typedef float f16v __attribute__((ext_vector_type(16)));
    for (int it = 0; it < iters; ++it) {
        const f16v* p = (const f16v*)(base + (size_t)it * 16);
        f16v s;
        asm volatile("s_load_dwordx16 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(p));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %0, %0\n v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s[i]));
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename F>
static double time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        f();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    float *in, *out, *stream;
    size_t stream_floats = (size_t)64 << 20;   // 256 MB
    CHECK(hipMalloc(&in, 4096)); CHECK(hipMalloc(&out, (size_t)cus * 64 * 256 * 4)); CHECK(hipMalloc(&stream, stream_floats * 4));
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97) + 0.5f;
    CHECK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    CHECK(hipMemset(stream, 0, stream_floats * 4));
    const int iters = 4096;
    // waves per SIMD = blocks_per_cu (each block = 4 waves = 1 per SIMD)
    for (int bpc : {1, 2, 4, 8}) {
        int grid = cus * bpc;
        double lane_ops = (double)grid * 256 * iters * 48.0;
        double wave_instr_per_simd = (double)bpc * iters * 48.0;   // per SIMD
        auto report = [&](const char* name, double ms, double ops_per_instr) {
            double cyc = ms * 1e-3 * 2.4e9;
            printf("%-28s waves/SIMD %d: %8.3f ms  %7.2f Tlane-op/s  %.2f cyc/wave-instr/SIMD (at 2.4GHz)\n", name, bpc, ms,
                   lane_ops / (ms * 1e-3) / 1e12, cyc / (wave_instr_per_simd / ops_per_instr));
        };
        report("plain v_sub/mul/add", time_ms([&] { hipLaunchKernelGGL(k_valu<0>, grid, 256, 0, 0, out, in, iters); }), 1);
        report("sgpr-operand", time_ms([&] { hipLaunchKernelGGL(k_valu<1>, grid, 256, 0, 0, out, in, iters); }), 1);
        report("v_pk_add/mul (2 ops/instr)", time_ms([&] { hipLaunchKernelGGL(k_pk, grid, 256, 0, 0, out, in, iters); }), 2);
        report("valu + 0 lds", time_ms([&] { hipLaunchKernelGGL(k_lds<0>, grid, 256, 0, 0, out, in, iters); }), 1);
        report("valu + 1 b128 bcast /48", time_ms([&] { hipLaunchKernelGGL(k_lds<1>, grid, 256, 0, 0, out, in, iters); }), 1);
        report("valu + 2 b128 bcast /48", time_ms([&] { hipLaunchKernelGGL(k_lds<2>, grid, 256, 0, 0, out, in, iters); }), 1);
        report("valu + 4 b128 bcast /48", time_ms([&] { hipLaunchKernelGGL(k_lds<4>, grid, 256, 0, 0, out, in, iters); }), 1);
        report("valu + smem x16 /48", time_ms([&] { hipLaunchKernelGGL(k_smem, grid, 256, 0, 0, out, in, stream, stream_floats, iters); }), 1);
    }
    return 0;
}
