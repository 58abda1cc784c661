// hbm_read_peak.hip — what a plain streaming read reaches on this MI355X: every lane loads 16-byte vectors from a large buffer
// (grid-stride, non-temporal), sums them and writes one word per block.  The figure the filter kernel's big pass is compared with in
// DESIGN.md §5 (the data-sheet peak is 8 TB/s).   hipcc --offload-arch=gfx950 -O3 tools/microbench/hbm_read_peak.hip -o hbm_read_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(256) ReadKernel(const u32x4* __restrict__ in, size_t n_vec, unsigned* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n_vec; i += stride) {
        u32x4 v = __builtin_nontemporal_load(in + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;   // keeps the loads alive, practically never taken
}

int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)atol(argv[1]) : 24;
    const size_t bytes = gib << 30, n_vec = bytes / 16;
    u32x4* buf = nullptr;
    unsigned* out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks_per_cu : {4, 8, 16, 32}) {
        const int grid = 256 * blocks_per_cu;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(ReadKernel<4>, dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("read %zu GiB, %2d blocks/CU x 256 threads, 4 x 16 B in flight per lane: %.3f ms = %.2f TB/s\n", gib, blocks_per_cu, best,
               bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
