// synth.hip — bench support: synthetic Gaussian-mixture rows generated straight into HBM (include/gpann_synth.h) and
// nearest-centroid assignment (Top1Neighbor, reference src/kmeans.cpp:12-23) used to cut the synthetic cloud into shards.
// Neither is on the route+scan hot path; they exist because the 100M-point benchmark inputs (51 GB) are produced on the
// device rather than shipped through PCIe.
#define GPANN_SYNTH_HD __host__ __device__
#include "../../include/gpann_synth.h"

#include "common.h"

#include <cfloat>

namespace gpann {

__global__ void __launch_bounds__(256) SynthRowsKernel(uint64_t seed, uint64_t center_seed, uint32_t n_centers, float sigma, uint64_t row0,
                                                       uint64_t nrows, uint32_t d, float* __restrict__ out) {
    const uint64_t total = nrows * d;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / d;
        const uint32_t c = (uint32_t)(i % d);
        const uint64_t row = row0 + r;
        const uint32_t ctr = gpann_synth_center_of(seed, row, n_centers);
        const float centre = gpann_synth_center_coord(center_seed, ctr, c);
        const float noise = gpann_synth_gauss(seed, row, c);
        out[i] = __fadd_rn(centre, __fmul_rn(sigma, noise));
    }
}

// One thread per point; centroids are few (number of shards) and sit in LDS.
template <int METRIC>
__global__ void __launch_bounds__(256) NearestCentroidKernel(const float* __restrict__ points, uint64_t n, uint32_t d,
                                                             const float* __restrict__ centroids, uint32_t c, int32_t* __restrict__ out) {
    extern __shared__ float s_cent[];
    for (uint32_t i = threadIdx.x; i < c * d; i += blockDim.x) s_cent[i] = centroids[i];
    __syncthreads();
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < n; p += (uint64_t)gridDim.x * blockDim.x) {
        const float* P = points + p * d;
        int best = -1;
        float best_dist = FLT_MAX;
        for (uint32_t j = 0; j < c; ++j) {
            // distance(centroid, point): kmeans.cpp:16
            float nd = ExactDistance<METRIC>(s_cent + (size_t)j * d, P, d);
            if (nd < best_dist) {
                best_dist = nd;
                best = (int)j;
            }
        }
        out[p] = best;
    }
}

}  // namespace gpann

using namespace gpann;

extern "C" {

gpann_status gpann_synth_rows_device(uint64_t seed, uint64_t center_seed, uint32_t n_centers, float sigma, uint64_t row0, uint64_t nrows,
                                     uint32_t d, float* out, void* stream) {
    GPANN_REQUIRE(out != nullptr && d > 0 && n_centers > 0, "bad synth arguments");
    int dev = 0;
    GPANN_HIP_TRY(hipGetDevice(&dev));
    if (nrows == 0) return GPANN_OK;
    const uint64_t total = nrows * d;
    const int grid = (int)std::min<uint64_t>((total + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(SynthRowsKernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), seed, center_seed, n_centers, sigma, row0,
                       nrows, d, out);
    GPANN_HIP_TRY(hipGetLastError());
    return GPANN_OK;
}

gpann_status gpann_nearest_centroid_device(const float* points, uint64_t n, uint32_t d, int metric, const float* centroids, uint32_t c,
                                           int32_t* out_assign, void* stream) {
    GPANN_REQUIRE(points && centroids && out_assign && d > 0 && c > 0, "bad arguments");
    const size_t lds = (size_t)c * d * sizeof(float);
    if (lds > 64 * 1024) {
        SetLastError("nearest-centroid: %u centroids x %u dims exceed the 64 KiB LDS table", c, d);
        return GPANN_ERR_UNSUPPORTED;
    }
    if (n == 0) return GPANN_OK;
    const int grid = (int)std::min<uint64_t>((n + 255) / 256, 256 * 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (metric == GPANN_METRIC_L2)
        hipLaunchKernelGGL(NearestCentroidKernel<GPANN_METRIC_L2>, dim3(grid), dim3(256), lds, st, points, n, d, centroids, c, out_assign);
    else
        hipLaunchKernelGGL(NearestCentroidKernel<GPANN_METRIC_MIPS>, dim3(grid), dim3(256), lds, st, points, n, d, centroids, c, out_assign);
    GPANN_HIP_TRY(hipGetLastError());
    return GPANN_OK;
}

}  // extern "C"
