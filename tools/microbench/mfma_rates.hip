// Issue rate of the matrix instructions the filter kernels use, register operands only (no memory): 4 independent accumulator chains per wavefront,
// 8 wavefronts per workgroup, one workgroup per CU x 4.   hipcc --offload-arch=gfx950 -O3 -o mfma_rates mfma_rates.hip && ./mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ void __launch_bounds__(512) Rate(int iters, float* out) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)blockIdx.x};
    if (KIND == 0) {
        v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = (float)(c0[0] + c1[1] + c2[2] + c3[3]);
    } else {
        bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
        f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
}

int main() {
    float* out;
    hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, grid = 1024;
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(Rate<0>, dim3(grid), dim3(512), 0, 0, iters, out);
            else hipLaunchKernelGGL(Rate<1>, dim3(grid), dim3(512), 0, 0, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = (double)grid * 8 * 4 * iters;   // per launch
            const double ops = mfmas * 32 * 32 * (kind == 0 ? 32 : 16) * 2;
            if (rep) printf("%s: %.3f ms, %.1f T%s/s, %.1f cycles per instruction and SIMD at 2.4 GHz (1024 SIMDs)\n", kind == 0 ? "v_mfma_i32_32x32x32_i8" : "v_mfma_f32_32x32x16_bf16", ms,
                            ops / ms / 1e9, kind == 0 ? "OP" : "FLOP", ms * 1e-3 * 2.4e9 * 1024 / mfmas);
        }
    }
    return 0;
}
