/* gpann_synth.h — the synthetic-input generator shared by bench.py, the tests and the CPU oracle.
 *
 * There are no datasets in the build environment, so every configuration of BASELINE.json is run on a
 * Gaussian-mixture point cloud of the same shape: `n_centers` centres ~ N(0, I_d),
 * point = centre[c(i)] + sigma * g(i, j).  The generator is counter-based (a pure function of (seed, row, column)),
 * uses only integer hashing, one exact int->float conversion, one float multiply and one float add, so the CPU
 * (gcc, strict flags) and the GPU (hipcc, contraction off) produce bit-identical floats.  It deliberately avoids
 * std::normal_distribution / logf / cosf, which differ between libstdc++/libm and the device libraries.
 *
 * "Gaussian" here is an Irwin-Hall sum of eight 16-bit uniforms (variance-normalised); good enough for a
 * clustered benchmark cloud and exactly reproducible.
 *
 * Plain C99 / C++ / HIP.  All functions are `static inline`; define GPANN_SYNTH_HD before including to decorate
 * them with `__host__ __device__` in a HIP translation unit.
 */
#ifndef GPANN_SYNTH_H
#define GPANN_SYNTH_H

#include <stdint.h>

#ifndef GPANN_SYNTH_HD
#define GPANN_SYNTH_HD
#endif

#define GPANN_SYNTH_CENTER_STREAM 0x43454E5445525321ULL /* mixed into the seed for the centre table */
#define GPANN_SYNTH_ASSIGN_COLUMN 0xFFFFFFFFULL         /* column index used to draw a row's centre id */

GPANN_SYNTH_HD static inline uint64_t gpann_synth_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

GPANN_SYNTH_HD static inline uint64_t gpann_synth_key(uint64_t seed, uint64_t row, uint64_t col) {
    uint64_t a = gpann_synth_mix64(seed + row * 0x9E3779B97F4A7C15ULL);
    return gpann_synth_mix64(a ^ (col * 0xC2B2AE3D27D4EB4FULL + 0x165667B19E3779F9ULL));
}

/* unit-variance, zero-mean pseudo-Gaussian; exactly representable steps, |g| <= 4.9 */
GPANN_SYNTH_HD static inline float gpann_synth_gauss(uint64_t seed, uint64_t row, uint64_t col) {
    uint64_t r0 = gpann_synth_key(seed, row, col);
    uint64_t r1 = gpann_synth_mix64(r0 ^ 0xD6E8FEB86659FD93ULL);
    int32_t s = 0;
    for (int i = 0; i < 4; ++i) {
        s += (int32_t)((r0 >> (16 * i)) & 0xFFFFu);
        s += (int32_t)((r1 >> (16 * i)) & 0xFFFFu);
    }
    s -= 4 * 65535;                      /* centre: mean of eight U{0..65535} */
    return (float)s * 1.8688000e-05f;    /* 1 / sqrt(8 * (65536^2 - 1) / 12) ~= 1.8688e-5; one rounding */
}

GPANN_SYNTH_HD static inline uint32_t gpann_synth_center_of(uint64_t seed, uint64_t row, uint32_t n_centers) {
    return (uint32_t)(gpann_synth_key(seed, row, GPANN_SYNTH_ASSIGN_COLUMN) % (uint64_t)n_centers);
}

GPANN_SYNTH_HD static inline float gpann_synth_center_coord(uint64_t center_seed, uint32_t c, uint32_t col) {
    return gpann_synth_gauss(center_seed ^ GPANN_SYNTH_CENTER_STREAM, c, col);
}

/* coordinate `col` of row `row`: centre + sigma * noise, as two separately rounded float ops (callers compile
 * with contraction off; the HIP side uses __fmul_rn/__fadd_rn).  `center_seed` is shared between the point set
 * and the query set so that both are drawn from the same mixture; `seed` differs (555 points / 556 queries). */
GPANN_SYNTH_HD static inline float gpann_synth_coord(uint64_t seed, uint64_t center_seed, uint32_t n_centers, float sigma,
                                                     uint64_t row, uint32_t col) {
    uint32_t c = gpann_synth_center_of(seed, row, n_centers);
    float centre = gpann_synth_center_coord(center_seed, c, col);
    float noise = gpann_synth_gauss(seed, row, col);
    float scaled = sigma * noise;
    return centre + scaled;
}

#endif /* GPANN_SYNTH_H */
