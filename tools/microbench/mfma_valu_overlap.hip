// Do vector instructions run under the matrix core's instructions on gfx950?  One "product" = 8 v_mfma_i32_32x32x32_i8 (two accumulator chains of 4, the
// 8-bit rest pass at 128 dimensions) + a test of the 2 x 16 accumulators (2 v_cvt, 2 v_fma, v_sub, v_alignbit each = 96 vector instructions), 8 wavefronts per
// workgroup (2 per SIMD), register operands only.
//   0  matrix instructions only            1  vector instructions only
//   2  product, then its test (dependent; what FilterKernelCells MODE 2 does)
//   3  software-pipelined in one wavefront: the test of product i is written after the MFMAs of product i+1 (two accumulator sets)
//   4  as 3, with the instruction streams interleaved by __builtin_amdgcn_sched_group_barrier (1 MFMA : 12 VALU)
// hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void Product(const v4i& a, const v4i& b, v16i& h, v16i& l) {
    const v16i z = {0};
    h = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
    l = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, z, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks) {
        h = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, h, 0, 0, 0);
        l = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, l, 0, 0, 0);
    }
}

__device__ __forceinline__ uint32_t Test(const v16i& h, const v16i& l, float s, float n, float b) {
    uint32_t neg = 0;
#pragma unroll
    for (int r = 15; r >= 0; --r) {
        const float fv = __builtin_fmaf((float)h[r], 254.0f, (float)l[r]);
        const float x = __builtin_fmaf(fv, s, n) - b;
        neg = __builtin_amdgcn_alignbit(neg, __float_as_uint(x), 31);
    }
    return neg;
}

template <int KIND>
__global__ void __launch_bounds__(512) Overlap(int iters, float s, float n, float b, uint32_t* out) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, bb = {3, 2, 1, (int)blockIdx.x};
    uint32_t acc = 0;
    if (KIND == 0) {
        v16i h, l;
        for (int i = 0; i < iters; ++i) {
            a[0] += i;
            Product(a, bb, h, l);
            acc += h[0] + l[5];
        }
    } else if (KIND == 1) {
        v16i h = {(int)threadIdx.x, 5, 6, 7, 1, 2, 3, 4, 8, 9, 10, 11, 12, 13, 14, 15}, l = h;
        for (int i = 0; i < iters; ++i) {
            h[i & 15] += i;
            l[(i + 3) & 15] -= i;
            acc += Test(h, l, s, n, b);
        }
    } else if (KIND == 2) {
        v16i h, l;
        for (int i = 0; i < iters; ++i) {
            a[0] += i;
            Product(a, bb, h, l);
            acc += Test(h, l, s, n, b);
        }
    } else {
        v16i h0, l0, h1, l1;
        Product(a, bb, h0, l0);
        for (int i = 0; i < iters; i += 2) {
            a[0] += i;
            Product(a, bb, h1, l1);
            acc += Test(h0, l0, s, n, b);
            if (KIND == 4) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);   // 12 VALU
                }
            }
            a[1] += i;
            Product(a, bb, h0, l0);
            acc += Test(h1, l1, s, n, b);
            if (KIND == 4) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                }
            }
        }
        acc += h0[3] + l0[4];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 1024 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000, grid = 256;
    const char* names[5] = {"MFMAs only", "tests only", "product then its test", "pipelined (source order)", "pipelined + sched_group_barrier"};
    for (int kind = 0; kind < 5; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (kind) {
                case 0: hipLaunchKernelGGL(Overlap<0>, dim3(grid), dim3(512), 0, 0, iters, 1e-3f, 5.f, 7.f, out); break;
                case 1: hipLaunchKernelGGL(Overlap<1>, dim3(grid), dim3(512), 0, 0, iters, 1e-3f, 5.f, 7.f, out); break;
                case 2: hipLaunchKernelGGL(Overlap<2>, dim3(grid), dim3(512), 0, 0, iters, 1e-3f, 5.f, 7.f, out); break;
                case 3: hipLaunchKernelGGL(Overlap<3>, dim3(grid), dim3(512), 0, 0, iters, 1e-3f, 5.f, 7.f, out); break;
                default: hipLaunchKernelGGL(Overlap<4>, dim3(grid), dim3(512), 0, 0, iters, 1e-3f, 5.f, 7.f, out); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            // cycles per product and SIMD: 2 wavefronts per SIMD each run `iters` products
            if (rep) printf("%-34s %.3f ms = %.0f cycles per product and SIMD at 2.4 GHz\n", names[kind], ms, ms * 1e-3 * 2.4e9 / (2.0 * iters));
        }
    }
    return 0;
}
