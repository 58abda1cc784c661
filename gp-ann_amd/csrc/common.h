// common.h — shared host/device helpers of libgpann_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <string>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>

#include "../../include/gpann.h"

// All distance arithmetic in this library must be the reference's strict binary32 sequence: no FMA contraction.
// The translation units are compiled with -ffp-contract=off; this pragma makes it explicit per file as well.
#pragma clang fp contract(off)

namespace gpann {

constexpr int kWave = 64;
constexpr uint32_t kNoId = GPANN_NO_ID;

// ---- error plumbing ---------------------------------------------------------------------------------------------
void SetLastError(const char* fmt, ...);

#define GPANN_HIP_TRY(expr)                                                                                   \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) {                                                                               \
            ::gpann::SetLastError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* the runtime keeps a failed launch as the thread's "last error": consumed here, or the NEXT call's check reports it */ \
            return _e == hipErrorOutOfMemory ? GPANN_ERR_OOM : GPANN_ERR_HIP;                                 \
        }                                                                                                     \
    } while (0)

#define GPANN_REQUIRE(cond, msg)                                                        \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ::gpann::SetLastError("invalid argument: %s (%s:%d)", msg, __FILE__, __LINE__); \
            return GPANN_ERR_INVALID;                                                   \
        }                                                                               \
    } while (0)

#define GPANN_TRY(expr)                      \
    do {                                     \
        gpann_status _s = (expr);            \
        if (_s != GPANN_OK) return _s;       \
    } while (0)

gpann_status EnsureDevice(int device);

// The knobs of a handle (include/gpann.h: gpann_shards_set_option / gpann_router_set_option).  A handle takes a SNAPSHOT of the process
// environment's GPANN_* variables when it is created — the defaults of a deployment still come from the environment — and never looks at the
// environment again: what a call does afterwards depends on the handle alone (set_option), is visible to a C-ABI caller (get_option) and cannot
// race with another thread's setenv.  Only debug output (GPANN_CELLS_DEBUG, GPANN_ROUTER_DEBUG, the path GPANN_VERIFY_REPORT) is still read
// from the environment where it is used.  Not synchronised: set an option while no call is in flight on the handle.
struct Options {
    std::map<std::string, std::string> kv;
    void SnapshotEnvironment();                       // (common.h users link shards.hip, which defines these)
    static std::string Canonical(const char* name);   // "FILTER_HEAD_ROWS" and "GPANN_FILTER_HEAD_ROWS" name the same knob
    static bool Known(const std::string& canonical);
    const char* Get(const char* name) const {         // the interface of getenv: nullptr = not set
        auto it = kv.find(name);
        return it == kv.end() ? nullptr : it->second.c_str();
    }
    bool IsSet(const char* name) const { return Get(name) != nullptr; }
    bool IsZero(const char* name) const {             // set and begins with '0'
        const char* v = Get(name);
        return v && v[0] == '0';
    }
    uint32_t U32(const char* name, uint32_t dflt) const {
        const char* v = Get(name);
        if (!v || !*v) return dflt;
        char* end = nullptr;
        long x = std::strtol(v, &end, 10);
        return (end != v && x >= 0) ? (uint32_t)x : dflt;
    }
};
gpann_status SetOption(Options& o, const char* name, const char* value);
gpann_status GetOption(const Options& o, const char* name, char* out, size_t cap);

// Growable device buffer owned by a handle.  (Growth goes through hipFree, which synchronises the whole DEVICE: a buffer never shrinks, every
// scan call slot and every router block owns its own buffers, and nothing that is still read by an enqueued kernel is ever re-grown — the callers
// wait for the slot's previous call first (scan.hip ScanToDevice, stage_in_flight).  Do not switch this to hipFreeAsync / a pool without
// revisiting those waits.)
struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    gpann_status Reserve(size_t need) {
        if (need <= bytes) return GPANN_OK;
        if (ptr) {
            hipError_t e = hipFree(ptr);
            ptr = nullptr;
            bytes = 0;
            if (e != hipSuccess) {
                SetLastError("hipFree failed: %s", hipGetErrorString(e));
                return GPANN_ERR_HIP;
            }
        }
        size_t grow = need + need / 8;
        hipError_t e = hipMalloc(&ptr, grow);
        if (e != hipSuccess) {
            ptr = nullptr;
            SetLastError("hipMalloc(%zu bytes) failed: %s", grow, hipGetErrorString(e));
            return GPANN_ERR_OOM;
        }
        bytes = grow;
        return GPANN_OK;
    }
    void Release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(ptr); }
};

inline uint32_t RoundUp4(uint32_t d) { return (d + 3u) & ~3u; }

// ---- device-side exact distance -----------------------------------------------------------------------------------
// The reference's scalar sequences (src/dist.cpp:17-37 L2, :42-52 MIPS), one (point,query) pair, rows in global memory.
// `p` and `q` need only 4-byte alignment here; used by the router walk, rescoring and the small helper kernels.
template <int METRIC>
__device__ __forceinline__ float ExactDistance(const float* __restrict__ p, const float* __restrict__ q, uint32_t d) {
    float result = 0.f;
    if (METRIC == GPANN_METRIC_L2) {
        uint32_t i = 0;
        for (; i + 4 <= d; i += 4) {
            float d0 = p[i] - q[i];
            float d1 = p[i + 1] - q[i + 1];
            float d2 = p[i + 2] - q[i + 2];
            float d3 = p[i + 3] - q[i + 3];
            float s = d0 * d0 + d1 * d1;
            s = s + d2 * d2;
            s = s + d3 * d3;
            result = result + s;
        }
        for (; i < d; ++i) {
            float d0 = p[i] - q[i];
            result = result + d0 * d0;
        }
        return result;
    } else {
        for (uint32_t i = 0; i < d; ++i) result = result + q[i] * p[i];
        return 1.0f - result;
    }
}

// The same sequences with 16-byte loads: `p` and `q` 16-byte aligned, rows padded with zeros to a multiple of 4 floats (the pad
// never enters the arithmetic: the tail loop stops at d).  Used where one lane evaluates one whole pair (re-scoring).
template <int METRIC>
__device__ __forceinline__ float ExactDistanceVec4(const float* __restrict__ p, const float* __restrict__ q, uint32_t d) {
    float result = 0.f;
    const uint32_t full = d >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* q4 = reinterpret_cast<const float4*>(q);
    if (METRIC == GPANN_METRIC_L2) {
#pragma unroll 4
        for (uint32_t g = 0; g < full; ++g) {
            const float4 a = p4[g], b = q4[g];
            const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
            float s = d0 * d0 + d1 * d1;
            s = s + d2 * d2;
            s = s + d3 * d3;
            result = result + s;
        }
        for (uint32_t i = full << 2; i < d; ++i) {
            const float d0 = p[i] - q[i];
            result = result + d0 * d0;
        }
        return result;
    } else {
#pragma unroll 4
        for (uint32_t g = 0; g < full; ++g) {
            const float4 a = p4[g], b = q4[g];
            result = result + b.x * a.x;
            result = result + b.y * a.y;
            result = result + b.z * a.z;
            result = result + b.w * a.w;
        }
        for (uint32_t i = full << 2; i < d; ++i) result = result + q[i] * p[i];
        return 1.0f - result;
    }
}

// ---- 4x4 (query x point) register tile, the reference's exact op sequence per group of 4 dimensions ---------------
template <int METRIC>
__device__ __forceinline__ void AccumulateGroup(float (&acc)[4][4], const float4 (&qv)[4], const float4 (&pv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (METRIC == GPANN_METRIC_L2) {
                // src/dist.cpp:23-31: diffN = a[N] - b[N] (a = point, b = query); result += d0*d0 + d1*d1 + d2*d2 + d3*d3
                float d0 = pv[j].x - qv[i].x;
                float d1 = pv[j].y - qv[i].y;
                float d2 = pv[j].z - qv[i].z;
                float d3 = pv[j].w - qv[i].w;
                float s = d0 * d0 + d1 * d1;
                s = s + d2 * d2;
                s = s + d3 * d3;
                acc[i][j] = acc[i][j] + s;
            } else {
                // src/dist.cpp:44-46: result += q[i] * p[i], one dimension after the other
                float r = acc[i][j];
                r = r + qv[i].x * pv[j].x;
                r = r + qv[i].y * pv[j].y;
                r = r + qv[i].z * pv[j].z;
                r = r + qv[i].w * pv[j].w;
                acc[i][j] = r;
            }
        }
    }
}

// the 0..3 trailing dimensions (src/dist.cpp:33-36): one at a time
template <int METRIC>
__device__ __forceinline__ float TailStep(float r, float pe, float qe) {
    if (METRIC == GPANN_METRIC_L2) {
        float d0 = pe - qe;
        return r + d0 * d0;
    }
    return r + qe * pe;
}

template <int METRIC>
__device__ __forceinline__ void AccumulateTail(float (&acc)[4][4], const float4 (&qv)[4], const float4 (&pv)[4], int tail) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r = acc[i][j];
            if (tail > 0) r = TailStep<METRIC>(r, pv[j].x, qv[i].x);
            if (tail > 1) r = TailStep<METRIC>(r, pv[j].y, qv[i].y);
            if (tail > 2) r = TailStep<METRIC>(r, pv[j].z, qv[i].z);
            acc[i][j] = r;
        }
    }
}

// lexicographic (dist, id) < (dist2, id2): the order of std::pair<float,uint32_t> that TopN uses (src/topn.h:15): PairLess, below
// A kernel that calls this is allocated all 256 architectural vector registers per wavefront: two of its wavefronts fill a SIMD's register
// file, so no wavefront of ANOTHER launch can be resident on that SIMD beside them.  The matrix-core kernels (FilterKernelCells, RouterBoundKernel:
// 512-thread workgroups, one per CU — nothing of their own is lost) call it.  History: with GPANN_SCAN_OVERLAP and 64-dimensional rows (filter kernels
// of 140 - 230 registers: room beside them for the short kernels of another call's front) about one call in 500 lost a task's candidates — traced to
// single accumulator registers of a co-resident CellDistKernel wavefront, lanes 48 - 63 (round 5, profiles/r05/overlap_race.md); gone once the int8
// kernels took their SIMDs whole and left them together (a barrier after the last work item): 0 events in 1.8 million overlapped calls
// (profiles/r06/NOTES.md §1).  Round 6 named the victim's side: the RESULTS OF PACKED fp32 INSTRUCTIONS (v_pk_add/mul_f32 with an op_sel-broadcast
// source that came from a load) go wrong in lanes 48 - 63 while ANOTHER kernel — or another PROCESS — issues gfx950's int8 matrix instructions
// (v_mfma_i32_32x32x32_i8 / 16x16x64_i8) on the same SIMD; scalar fp32 instructions are not disturbed.  The library is therefore also built WITHOUT
// packed fp32 (Makefile: -target-feature -packed-fp32-ops; same IEEE operations, same bits, no measurable cost) — beside an int8 neighbour process 186 of
// 285 GPU tests failed before, 302 of 302 pass now (profiles/r06/NOTES.md §2; tools/microbench/coresidency_victims.hip, matrix_neighbour.hip).  Both
// properties are checked on the built library: tools/kernel_resources.py --check-matrix-kernels, tests/test_capi_exports.py.
#ifdef GPANN_EXPERIMENT_SHARED_SIMDS
// (measurement builds only — `make -C gp-ann_amd experiment-shared-simds` -> lib_experiment/libgpann_hip.so, never the shipped library: the matrix-core
// kernels keep their natural register counts, i.e. leave room on their SIMDs, so that tools/router_event_experiment.sh can put another process's
// int8 matrix wavefronts beside RouterBoundKernel and see what its self-check then reports.  tests/test_capi_exports.py checks the SHIPPED code objects.)
__device__ __forceinline__ void TakeWholeRegisterFile() {}
#else
__device__ __forceinline__ void TakeWholeRegisterFile() { asm volatile("" ::: "v255"); }
#endif

__device__ __forceinline__ bool PairLess(float d1, uint32_t i1, float d2, uint32_t i2) { return d1 < d2 || (d1 == d2 && i1 < i2); }
// float <-> uint32 whose unsigned order is the float order (negative inner-product "distances" included): atomicMin on the keys
__device__ __forceinline__ uint32_t OrderedKey(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float OrderedKeyToFloat(uint32_t key) { return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key); }
constexpr uint32_t kOrderedKeyInf = 0xFF800000u;   // OrderedKey(+inf)

// ---- the same sequence on PAIRS of point rows ---------------------------------------------------------------------------------------
// The rows of a tile lie in LDS as pairs [pair][dim][member] (one ds_read_b128 = two dimensions of both rows) and the arithmetic is written on
// two-element vectors.  Until round 6 the compiler turned that into packed instructions (v_pk_add/mul_f32, the query broadcast by op_sel); the
// library is now built WITHOUT packed fp32 (Makefile, -packed-fp32-ops: their results go wrong beside another kernel's int8 matrix wavefronts —
// TakeWholeRegisterFile, above), so every element is a scalar v_sub / v_mul / v_add_f32 — the same IEEE operation, the same bits, and on this part
// the same rate: a SIMD issues a scalar fp32 instruction for 64 lanes in 2 cycles, a packed one in 4 (ScanTileKernel: 255.6 -> 259.5 ms per NS step).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// one group of 4 dimensions of the reference's sequence (src/dist.cpp:23-31 / :44-46) for a thread's 4 queries x 2 point pairs
template <int METRIC>
__device__ __forceinline__ void AccumulatePairs(f32x2_t (&acc)[4][2], const float4 (&qv)[4], const f32x4_t (&p0)[2], const f32x4_t (&p1)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const f32x2_t px = p0[pr].xy, py = p0[pr].zw, pz = p1[pr].xy, pw = p1[pr].zw;
            const f32x2_t qx = {qv[i].x, qv[i].x}, qy = {qv[i].y, qv[i].y}, qz = {qv[i].z, qv[i].z}, qw = {qv[i].w, qv[i].w};
            if (METRIC == GPANN_METRIC_L2) {
                const f32x2_t d0 = px - qx, d1 = py - qy, d2 = pz - qz, d3 = pw - qw;
                f32x2_t t = d0 * d0 + d1 * d1;
                t = t + d2 * d2;
                t = t + d3 * d3;
                acc[i][pr] = acc[i][pr] + t;
            } else {
                f32x2_t r = acc[i][pr];
                r = r + qx * px;
                r = r + qy * py;
                r = r + qz * pz;
                r = r + qw * pw;
                acc[i][pr] = r;
            }
        }
    }
}

// the 1..3 trailing dimensions (src/dist.cpp:33-36), one at a time
template <int METRIC>
__device__ __forceinline__ void AccumulatePairsTail(f32x2_t (&acc)[4][2], const float4 (&qv)[4], const f32x4_t (&p0)[2], const f32x4_t (&p1)[2], int tail) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            f32x2_t r = acc[i][pr];
            const f32x2_t pe[3] = {p0[pr].xy, p0[pr].zw, p1[pr].xy};
            const float qe[3] = {qv[i].x, qv[i].y, qv[i].z};
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                if (x < tail) {
                    const f32x2_t qb = {qe[x], qe[x]};
                    if (METRIC == GPANN_METRIC_L2) {
                        const f32x2_t d0 = pe[x] - qb;
                        r = r + d0 * d0;
                    } else {
                        r = r + qb * pe[x];
                    }
                }
            }
            acc[i][pr] = r;
        }
    }
}
}  // namespace gpann
