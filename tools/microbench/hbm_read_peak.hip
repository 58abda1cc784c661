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

// the same bytes as `streams` independent sequential streams: workgroup b (512 threads, 8 waves taking 8 KiB tiles round robin with
// one tile of prefetch — the filter kernel's access pattern) reads its own contiguous slice of the buffer
__global__ void __launch_bounds__(512) StreamsKernel(const u32x4* __restrict__ in, size_t vec_per_stream, unsigned* __restrict__ out) {
    const u32x4* base = in + (size_t)blockIdx.x * vec_per_stream;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t tiles = vec_per_stream / 512;   // 8 KiB = 512 vectors
    unsigned acc = 0;
    u32x4 a[8], b[8];
    size_t T = wave;
    if (T < tiles) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_nontemporal_load(base + T * 512 + k * 64 + lane);
        for (;;) {
            size_t Tn = T + 8 < tiles ? T + 8 : tiles - 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) b[k] = __builtin_nontemporal_load(base + Tn * 512 + k * 64 + lane);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += a[k].x ^ a[k].y ^ a[k].z ^ a[k].w;
            T += 8;
            if (T >= tiles) break;
            Tn = T + 8 < tiles ? T + 8 : tiles - 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = __builtin_nontemporal_load(base + Tn * 512 + k * 64 + lane);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += b[k].x ^ b[k].y ^ b[k].z ^ b[k].w;
            T += 8;
            if (T >= tiles) break;
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
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
    for (int streams : {256, 512, 1024, 4096}) {
        const size_t vec_per_stream = n_vec / streams / 512 * 512;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(StreamsKernel, dim3(streams), dim3(512), 0, 0, buf, vec_per_stream, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("read %zu GiB as %4d independent streams (512-thread workgroups, 8 KiB tiles, 1 tile of prefetch per wave): %.3f ms = %.2f TB/s\n", gib,
               streams, best, (double)vec_per_stream * streams * 16 / (best * 1e-3) / 1e12);
    }
    return 0;
}
