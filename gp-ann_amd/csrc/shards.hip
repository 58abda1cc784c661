// shards.hip — building and querying metadata of the device-resident shard store.
//
// Replaces InvertedIndex::InvertedIndex (reference src/inverted_index.h:12-41): the points of every shard are gathered
// into one shard-contiguous row-major array so that a shard is one coalesced HBM stream; offsets[S+1] and
// permutation[row] -> global id are kept exactly as the reference keeps them.
#include "shards.h"
#include "scan_filter.h"

#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

namespace gpann {

thread_local std::string g_last_error;

void SetLastError(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

gpann_status EnsureDevice(int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        SetLastError("no usable HIP device (hipGetDeviceCount: %s, count %d) — libgpann_hip has no CPU fallback",
                     hipGetErrorString(e), count);
        return GPANN_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        SetLastError("device %d out of range (count %d)", device, count);
        return GPANN_ERR_INVALID;
    }
    GPANN_HIP_TRY(hipSetDevice(device));
    return GPANN_OK;
}

// One thread per float4 of the destination: dst row r <- src row ids[r].  Source rows have stride d (caller layout),
// destination rows stride ld (multiple of 4, zero padded).
__global__ void __launch_bounds__(256) GatherRowsKernel(const float* __restrict__ src, const uint32_t* __restrict__ ids, uint64_t rows,
                                                        uint32_t d, uint32_t ld, float* __restrict__ dst) {
    const uint32_t vec_per_row = ld / 4;
    const uint64_t total = rows * vec_per_row;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t r = i / vec_per_row;
        uint32_t c = (uint32_t)(i % vec_per_row) * 4;
        const float* s = src + (uint64_t)ids[r] * d;
        float4 v;
        v.x = c + 0 < d ? s[c + 0] : 0.f;
        v.y = c + 1 < d ? s[c + 1] : 0.f;
        v.z = c + 2 < d ? s[c + 2] : 0.f;
        v.w = c + 3 < d ? s[c + 3] : 0.f;
        *reinterpret_cast<float4*>(dst + r * ld + c) = v;
    }
}

// ---- the handles' knobs (common.h Options) -------------------------------------------------------------------------------
// Every name a handle understands.  `when`: B = read while the store / router is BUILT (gpann_shards_create, gpann_shards_build_filter,
// gpann_router_create / _train): set it in the environment before the handle is created, or — stores built in two steps — between
// gpann_shards_create(GPANN_STORE_PLAIN) and gpann_shards_build_filter; C = read by every call.
struct Knob {
    const char* name;
    char when;
    const char* what;
};
static const Knob kKnobs[] = {
    // store build
    {"GPANN_CELL_ROWS", 'B', "target rows per cell (default: rows per shard / 128, at most 32768; shards below 262144 rows: rows per shard / 256, at least 512)"},
    {"GPANN_FILTER_PLANES", 'B', "0: no planes / cells — the store offers the plain exact scan only"},
    {"GPANN_FILTER_LO_ROWS", 'B', "rows per cell that keep a bf16 lo tile (default 512; all: every row — allows GPANN_FILTER_HI_ONLY=0)"},
    {"GPANN_FILTER_HI_ROWS", 'B', "all: every tile keeps its bf16 hi tile although the store has the 8-bit plane"},
    {"GPANN_FILTER_Q8", 'B', "0: no 8-bit plane (store build); per call: 0 = the sample / rest passes on the bf16 hi plane where the store has one"},
    {"GPANN_FILTER_Q8_CELL_FRAME", 'B', "0: plain norm words for the 8-bit passes (round 3's error term)"},
    {"GPANN_FILTER_SORT_ROWS", 'B', "0: a cell's rows stay in slab order behind its front"},
    {"GPANN_FILTER_CENTER", 'B', "0: planes / router images in the original frame, not around the mean row"},
    {"GPANN_FILTER_EXACT_ROWS", 'B', "rows of the shard sample (default 4096)"},
    // scan calls
    {"GPANN_SCAN_MODE", 'C', "exact | filter: overrides gpann_shards_set_scan_mode"},
    {"GPANN_FILTER_CAND_CAP", 'C', "candidate bucket size per task (default: 1024 .. 16384 from the 6 GiB budget)"},
    {"GPANN_FILTER_HEAD_ROWS", 'C', "rows of the head pass (default 512 — small shards, GPANN_FILTER_SMALL_SHARD_ROWS: 0; 0: no head pass)"},
    {"GPANN_FILTER_SAMPLE_DIV", 'C', "the sample pass covers 1 / n of a needed cell (default: 8 .. 64 from the bucket size)"},
    {"GPANN_FILTER_Q8_FROM_PASS", 'C', "first pass on the 8-bit plane: 1 sample (default), 2 rest"},
    {"GPANN_FILTER_Q8_L1", 'C', "0: no |q - c|_1 table — Cauchy-Schwarz on the squared distances"},
    {"GPANN_FILTER_SEG_TILES", 'C', "row tiles per work item (default: 16 .. 256 from the store size)"},
    {"GPANN_FILTER_SEED_ROWS", 'C', "rows of the nearest cell evaluated exactly for the first lists (default 128; 0: exact scan of the shard sample)"},
    {"GPANN_FILTER_SHARD_SAMPLE", 'C', "0: no shard-sample pass, 1: with it (default 1 — small shards, GPANN_FILTER_SMALL_SHARD_ROWS: 0)"},
    {"GPANN_FILTER_SMALL_SHARD_ROWS", 'C', "stores whose shards average fewer rows run without the shard-sample and head passes unless those knobs are set (default 262144; 0: never) — while the cells prune: GPANN_FILTER_SMALL_SHARD_MAX_SHARE"},
    {"GPANN_FILTER_SMALL_SHARD_MAX_SHARE", 'C', "the small-shard rule applies while the store's previous filtered call sent at most this percentage of the (task, row) pairs it was asked for to the matrix cores (default 50; a first call has no history and takes the rule)"},
    {"GPANN_FILTER_SEED_ALL", 'C', "1: seed every task under the merge bound, not only a query's first slot"},
    {"GPANN_FILTER_SAMPLE_UNIT", 'C', "2: the sample pass on the rest pass's kernel (two row tiles per unit)"},
    {"GPANN_FILTER_HI_ONLY", 'C', "0: all three bf16 products in every pass (needs GPANN_FILTER_LO_ROWS=all)"},
    {"GPANN_FILTER_HI_FROM_PASS", 'C', "first hi-plane-only pass (default 1)"},
    {"GPANN_RESCORE_PARTS", 'C', "wavefronts a heavy candidate bucket is split over (default: 2 .. 8 from the task count)"},
    {"GPANN_RESCORE_SPLIT_MIN", 'C', "candidates per piece of a split bucket (default 128)"},
    {"GPANN_SCAN_TARGET_UNITS", 'C', "exact mode: workgroups a launch is cut into (default 8192)"},
    {"GPANN_SCAN_MIN_SEG_ROWS", 'C', "exact mode: smallest row segment of a unit"},
    {"GPANN_OVERLAP_FRONT_CUS", 'C', "GPANN_SCAN_OVERLAP: workgroups of a front's filter launches (default 96; stores whose planes are below 4 GB: all CUs)"},
    {"GPANN_OVERLAP_REST_ITEMS", 'C', "GPANN_SCAN_OVERLAP: work items after which a rest-pass workgroup leaves (default 0: persistent workgroups, GPANN_OVERLAP_REST_CUS of them)"},
    {"GPANN_OVERLAP_REST_CUS", 'C', "GPANN_SCAN_OVERLAP with persistent rest-pass workgroups: how many (default 160; stores whose planes are below 4 GB: all CUs)"},
    {"GPANN_OVERLAP_SPLIT_MAX_PRODUCTS", 'C', "GPANN_SCAN_OVERLAP: the CU split applies while the previous call ran at most this many (query tile, row tile) products per rest-pass row tile (default 3: an HBM-bound rest pass)"},
    {"GPANN_OVERLAP_PACE", 'C', "GPANN_SCAN_OVERLAP: 0 = a call's front may start at once (default 1: not before the rest pass of the call two before it has started)"},
    {"GPANN_OVERLAP_FRONT_STREAMS", 'C', "GPANN_SCAN_OVERLAP: 1 = the fronts of the calls in flight one after the other, 3 (default) = side by side, a stream per call slot"},
    {"GPANN_OVERLAP_ASSUME_SHARED_SIMDS", 'C', "test hook: 1 = GPANN_SCAN_OVERLAP behaves as if a matrix-core kernel of this build had fewer than 256 vector registers (the flag is ignored, calls run one after the other)"},
    {"GPANN_DEBUG_Q8_NO_CAND", 'C', "measurement aid, WRONG RESULTS: the 8-bit passes find no candidate"},
    // router
    {"GPANN_ROUTER_MODE", 'C', "walk | dense: force the tree walk / the distance table + replay (default: by tree size and budget)"},
    {"GPANN_ROUTER_DENSE_FACTOR", 'C', "table + replay up to trees this many times the budget (default 32)"},
    {"GPANN_ROUTER_REPLAY", 'C', "heap | threshold: force a replay kernel"},
    {"GPANN_ROUTER_BOUND", 'C', "0: canonical evaluation of every (query, centroid) pair instead of the matrix-core bound"},
    {"GPANN_ROUTER_BOUND_TABLE", 'C', "0: the canonical distance table instead of the matrix-core table + bounded replay"},
    {"GPANN_ROUTER_BAND_CAP", 'C', "band list capacity of the bounded replay (96 .. 4096, default 512)"},
    {"GPANN_ROUTER_EPS_SCALE", 'C', "test knob: widens the bounded replay's band (same bits)"},
    {"GPANN_ROUTER_HOST_REDO", 'C', "0: marked queries are replayed by the device heap kernel, not on host threads"},
    {"GPANN_ROUTER_FREQ_SMALL", 'C', "FrequencyQuery: 0 never records only the smallest pairs; n: that many records per query"},
    {"GPANN_ROUTER_VERIFY", 'C', "set: every block of the bound path is compared with the canonical evaluation"},
    {"GPANN_ROUTER_STRICT", 'C', "set: a fired self-check is GPANN_ERR_INTERNAL instead of a canonical re-evaluation"},
    {"GPANN_ROUTER_PROBE", 'C', "set: every bound block runs twice and every bit is compared"},
    {"GPANN_ROUTER_FORCE_RECHECK", 'C', "test hook: take the canonical re-evaluation"},
    {"GPANN_ROUTER_FORCE_EVIDENCE", 'C', "test hook: write the evidence report for a healthy block"},
    {"GPANN_ROUTER_LEGACY_MEMSET", 'C', "debug: set the bound block up with three runtime memsets + a fill launch (the pre-66a007d path) instead of BoundInitKernel"},
};

std::string Options::Canonical(const char* name) {
    std::string n = name ? name : "";
    if (n.rfind("GPANN_", 0) != 0) n = "GPANN_" + n;
    return n;
}
bool Options::Known(const std::string& canonical) {
    for (const Knob& k : kKnobs)
        if (canonical == k.name) return true;
    return false;
}
void Options::SnapshotEnvironment() {
    for (const Knob& k : kKnobs)
        if (const char* v = std::getenv(k.name)) kv[k.name] = v;
}
gpann_status SetOption(Options& o, const char* name, const char* value) {
    GPANN_REQUIRE(name != nullptr && *name, "option name is null");
    const std::string n = Options::Canonical(name);
    if (!Options::Known(n)) {
        SetLastError("unknown option %s (gpann_option_name lists them)", n.c_str());
        return GPANN_ERR_INVALID;
    }
    if (value) o.kv[n] = value;
    else o.kv.erase(n);
    return GPANN_OK;
}
gpann_status GetOption(const Options& o, const char* name, char* out, size_t cap) {
    GPANN_REQUIRE(name != nullptr && out != nullptr && cap > 0, "null argument");
    const std::string n = Options::Canonical(name);
    if (!Options::Known(n)) {
        SetLastError("unknown option %s", n.c_str());
        return GPANN_ERR_INVALID;
    }
    const char* v = o.Get(n.c_str());
    std::snprintf(out, cap, "%s", v ? v : "");
    return GPANN_OK;
}

}  // namespace gpann

using namespace gpann;

// Target rows per cell (GPANN_CELL_ROWS overrides): about 128 cells per shard, at most 32768 rows; shards below 262144 rows: about 256 cells, at least
// 512 rows.  Measured on the
// BASELINE.json shapes: 100M x 128 in 40 shards is flat between 16k and 32k rows (bigger cells stream rows for task groups that do not need
// them, smaller ones add work items); 10M x 96 in 16 shards is flat between 2k and 5k rows; 1M x 128 in 8 shards (mixture components of 977
// rows: larger cells mix components and their balls prune nothing) 2048 rows 1.365 ms per step, 1024 1.242, 512 1.188, 384 1.188, 256 1.196
// (profiles/r06/sweeps/small_store_passes.txt; until round 6 the floor was 2048).
static uint32_t DefaultCellRows(const gpann_shards* s) {
    if (const char* cr = s->opt.Get("GPANN_CELL_ROWS")) return (uint32_t)std::max(64L, std::atol(cr));
    const uint64_t per_shard = s->rows / std::max<uint32_t>(1, s->S);
    if (per_shard / 128 >= 2048) return (uint32_t)std::min<uint64_t>(32768, per_shard / 128);
    return (uint32_t)std::max<uint64_t>(512, per_shard / 256);   // small shards (below 262144 rows): about 256 cells, not below 512 rows
}

extern "C" {

const char* gpann_last_error(void) { return g_last_error.c_str(); }

const char* gpann_version(void) { return "gpann-hip 0.6 (gfx950)"; }
uint32_t gpann_abi_version(void) { return GPANN_ABI_VERSION; }

uint32_t gpann_option_count(void) { return (uint32_t)(sizeof(kKnobs) / sizeof(kKnobs[0])); }
const char* gpann_option_name(uint32_t i) { return i < gpann_option_count() ? kKnobs[i].name : nullptr; }
const char* gpann_option_help(uint32_t i) { return i < gpann_option_count() ? kKnobs[i].what : nullptr; }
int gpann_option_read_at_build(uint32_t i) { return i < gpann_option_count() && kKnobs[i].when == 'B' ? 1 : 0; }

gpann_status gpann_shards_set_option(gpann_shards* s, const char* name, const char* value) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    std::lock_guard<std::recursive_mutex> turn(s->call_mutex);
    return SetOption(s->opt, name, value);
}
gpann_status gpann_shards_get_option(const gpann_shards* s, const char* name, char* out, size_t cap) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    return GetOption(s->opt, name, out, cap);
}

gpann_status gpann_device_count(int* count) {
    GPANN_REQUIRE(count != nullptr, "count is null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) c = 0;
    *count = c;
    return GPANN_OK;
}

gpann_status gpann_shards_create(const float* points, uint64_t n, uint32_t d, int metric, const uint32_t* ids_concat,
                                 const uint64_t* shard_offsets, uint32_t S, int device, int flags, gpann_shards** out) {
    GPANN_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GPANN_REQUIRE(points != nullptr && shard_offsets != nullptr, "points/shard_offsets null");
    GPANN_REQUIRE(d > 0 && S > 0, "d and S must be positive");
    GPANN_REQUIRE(metric == GPANN_METRIC_L2 || metric == GPANN_METRIC_MIPS, "unknown metric");
    GPANN_REQUIRE(shard_offsets[0] == 0, "shard_offsets[0] must be 0");
    const uint64_t rows = shard_offsets[S];
    GPANN_REQUIRE(rows < 0xFFFFFFFFull, "total rows must fit the reference's 32-bit row index (inverted_index.h:9)");
    GPANN_REQUIRE(rows == 0 || ids_concat != nullptr, "ids_concat null");
    for (uint32_t b = 0; b < S; ++b) GPANN_REQUIRE(shard_offsets[b] <= shard_offsets[b + 1], "shard_offsets not monotone");
    bool ids_in_order = true;
    for (uint64_t i = 0; i < rows; ++i) {
        GPANN_REQUIRE(ids_concat[i] < n, "point id out of range");
        ids_in_order = ids_in_order && ids_concat[i] == i;
    }
    GPANN_TRY(EnsureDevice(device));

    auto* s = new gpann_shards();
    s->opt.SnapshotEnvironment();   // (the one look at the environment: gpann_shards_set_option from here on)
    s->device = device;
    s->metric = metric;
    s->d = d;
    s->ld = RoundUp4(d);
    s->S = S;
    s->rows = rows;
    s->ids_in_order = ids_in_order;
    s->offsets.assign(shard_offsets, shard_offsets + S + 1);
    auto fail = [&](gpann_status st) {
        gpann_shards_destroy(s);
        return st;
    };
#define TRY_OR_FAIL(expr)                                                                                  \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            SetLastError("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);       \
            return fail(_e == hipErrorOutOfMemory ? GPANN_ERR_OOM : GPANN_ERR_HIP);                        \
        }                                                                                                  \
    } while (0)

    const size_t row_bytes = (size_t)s->ld * sizeof(float);
    TRY_OR_FAIL(hipMalloc(&s->points, std::max<size_t>(rows * row_bytes, 256)));
    TRY_OR_FAIL(hipMalloc(&s->permutation, std::max<size_t>(rows * sizeof(uint32_t), 256)));
    TRY_OR_FAIL(hipMalloc(&s->d_offsets, (S + 1) * sizeof(int64_t)));
    TRY_OR_FAIL(hipMemcpy(s->d_offsets, s->offsets.data(), (S + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (rows) TRY_OR_FAIL(hipMemcpy(s->permutation, ids_concat, rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (auto& c : s->cs) {
        for (auto& e : c.ev) TRY_OR_FAIL(hipEventCreate(&e));
        for (auto& e : c.ev_filter) TRY_OR_FAIL(hipEventCreate(&e));
        for (auto& e : c.ev_x) TRY_OR_FAIL(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        TRY_OR_FAIL(hipHostMalloc(reinterpret_cast<void**>(&c.pin_words), 96, hipHostMallocDefault));
    }

    if (rows) {
        if (flags & GPANN_POINTS_ON_DEVICE) {
            uint64_t total_vec = rows * (s->ld / 4);
            int grid = (int)std::min<uint64_t>((total_vec + 255) / 256, 256 * 32);
            hipLaunchKernelGGL(GatherRowsKernel, dim3(grid), dim3(256), 0, 0, points, s->permutation, rows, d, s->ld, s->points);
            TRY_OR_FAIL(hipGetLastError());
            TRY_OR_FAIL(hipDeviceSynchronize());
        } else {
            // Host points: gather on the host into pinned staging chunks (what the reference's constructor does with
            // parlay, inverted_index.h:27-40) and stream them up; needs no second device copy of the whole point set.
            const size_t chunk_rows = std::max<size_t>(1, (64u << 20) / row_bytes);
            float* staging[2] = {nullptr, nullptr};
            hipEvent_t done[2];
            for (int i = 0; i < 2; ++i) {
                TRY_OR_FAIL(hipHostMalloc((void**)&staging[i], chunk_rows * row_bytes, hipHostMallocDefault));
                TRY_OR_FAIL(hipEventCreate(&done[i]));
            }
            unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
            int buf = 0;
            for (uint64_t r0 = 0; r0 < rows; r0 += chunk_rows, buf ^= 1) {
                uint64_t r1 = std::min<uint64_t>(rows, r0 + chunk_rows);
                TRY_OR_FAIL(hipEventSynchronize(done[buf]));
                float* st = staging[buf];
                auto work = [&](uint64_t a, uint64_t b) {
                    for (uint64_t r = a; r < b; ++r) {
                        const float* src = points + (uint64_t)ids_concat[r] * d;
                        float* dst = st + (r - r0) * s->ld;
                        std::memcpy(dst, src, d * sizeof(float));
                        for (uint32_t c = d; c < s->ld; ++c) dst[c] = 0.f;
                    }
                };
                uint64_t cnt = r1 - r0;
                unsigned T = (unsigned)std::min<uint64_t>(hw, std::max<uint64_t>(1, cnt / 4096));
                if (T <= 1) {
                    work(r0, r1);
                } else {
                    std::vector<std::thread> th;
                    for (unsigned t = 0; t < T; ++t) th.emplace_back(work, r0 + cnt * t / T, r0 + cnt * (t + 1) / T);
                    for (auto& x : th) x.join();
                }
                TRY_OR_FAIL(hipMemcpyAsync(s->points + r0 * s->ld, st, cnt * row_bytes, hipMemcpyHostToDevice, 0));
                TRY_OR_FAIL(hipEventRecord(done[buf], 0));
            }
            TRY_OR_FAIL(hipDeviceSynchronize());
            for (int i = 0; i < 2; ++i) {
                (void)hipHostFree(staging[i]);
                (void)hipEventDestroy(done[i]);
            }
        }
    }
    // support data of the MFMA pre-filter: row norms, per-shard max norm, sanity of the values
    if (rows) {
        unsigned long long* d_max_bits = nullptr;
        uint32_t* d_bad = nullptr;
        TRY_OR_FAIL(hipMalloc(&s->neg_half_norm, (rows + 64) * sizeof(float)));
        TRY_OR_FAIL(hipMemset(s->neg_half_norm + rows, 0, 64 * sizeof(float)));
        TRY_OR_FAIL(hipMalloc(&s->d_shard_rmax, S * sizeof(double)));
        TRY_OR_FAIL(hipMalloc(&d_bad, 4));
        d_max_bits = reinterpret_cast<unsigned long long*>(s->d_shard_rmax);   // non-negative doubles order like their bit patterns
        TRY_OR_FAIL(hipMemset(d_max_bits, 0, S * sizeof(double)));
        TRY_OR_FAIL(hipMemset(d_bad, 0, 4));
        const int grid = (int)std::min<uint64_t>((rows + 3) / 4, 256 * 32);
        hipLaunchKernelGGL(RowNormKernel, dim3(grid), dim3(256), 0, 0, s->points, rows, d, s->ld, s->d_offsets, S, s->neg_half_norm, d_max_bits, d_bad);
        TRY_OR_FAIL(hipGetLastError());
        uint32_t bad = 0;
        TRY_OR_FAIL(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
        (void)hipFree(d_bad);
        s->filter_ok = bad == 0;
        // filter support (scan mode 1): the rows once more as bf16 hi/lo planes grouped by cell (cells.hip).  Optional: without it
        // (allocation failure, GPANN_FILTER_PLANES=0, d > 256, non-finite data) the store only offers the plain exact scan.
        // GPANN_CELL_ROWS sets the target cell size.
        const char* want = s->opt.Get("GPANN_FILTER_PLANES");
        const uint32_t cell_rows = DefaultCellRows(s);
        if (s->filter_ok && d <= 256 && !(want && want[0] == '0') && !(flags & GPANN_STORE_PLAIN)) {
            gpann_status cst = BuildCells(s, cell_rows);
            if (cst != GPANN_OK) return fail(cst);
        }
    }
#undef TRY_OR_FAIL
    *out = s;
    return GPANN_OK;
}

gpann_status gpann_shards_build_filter(gpann_shards* s) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    std::lock_guard<std::recursive_mutex> turn(s->call_mutex);
    GPANN_TRY(EnsureDevice(s->device));
    if (s->cells.n_cells != 0 && s->frag_planes != nullptr) return GPANN_OK;   // already there
    if (s->rows == 0) return GPANN_OK;
    if (!s->filter_ok || s->d > 256) {
        SetLastError("the store's data is not fit for the filter (non-finite values or d > 256)");
        return GPANN_ERR_UNSUPPORTED;
    }
    s->filter_fail_status = GPANN_OK;
    GPANN_TRY(BuildCells(s, DefaultCellRows(s)));
    // asked for explicitly: a store that still has no filter data is an error here (gpann_last_error() names the step that failed; the status
    // says what kind: GPANN_ERR_OOM an allocation, GPANN_ERR_HIP a device / kernel error, GPANN_ERR_INTERNAL no cells)
    if (s->cells.n_cells == 0 || s->frag_planes == nullptr) return s->filter_fail_status != GPANN_OK ? s->filter_fail_status : GPANN_ERR_OOM;
    return GPANN_OK;
}

gpann_status gpann_shards_set_scan_mode(gpann_shards* s, int mode) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    GPANN_REQUIRE(mode == 0 || mode == 1, "scan mode must be 0 (exact) or 1 (filter)");
    s->scan_mode = mode;
    return GPANN_OK;
}

gpann_status gpann_shards_destroy(gpann_shards* s) {
    if (!s) return GPANN_OK;
    (void)hipSetDevice(s->device);
    if (s->points) (void)hipFree(s->points);
    if (s->permutation) (void)hipFree(s->permutation);
    if (s->d_offsets) (void)hipFree(s->d_offsets);
    if (s->neg_half_norm) (void)hipFree(s->neg_half_norm);
    if (s->d_shard_rmax) (void)hipFree(s->d_shard_rmax);
    if (s->frag_planes) (void)hipFree(s->frag_planes);
    ReleaseCells(s);
    for (auto& c : s->cs) {
        c.ReleaseWorkspaces();
        c.stage.Release();
        if (c.pin_words) (void)hipHostFree(c.pin_words);
        for (auto& e : c.ev)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : c.ev_filter)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : c.ev_x)
            if (e) (void)hipEventDestroy(e);
    }
    for (hipStream_t st : s->st_front)
        if (st) (void)hipStreamDestroy(st);
    if (s->st_tail) (void)hipStreamDestroy(s->st_tail);
    if (s->st_rest) (void)hipStreamDestroy(s->st_rest);

    delete s;
    return GPANN_OK;
}

gpann_status gpann_shards_info(const gpann_shards* s, uint64_t* rows, uint32_t* d, uint32_t* S, int* metric, int* device) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    if (rows) *rows = s->rows;
    if (d) *d = s->d;
    if (S) *S = s->S;
    if (metric) *metric = s->metric;
    if (device) *device = s->device;
    return GPANN_OK;
}

gpann_status gpann_shards_export(const gpann_shards* s, int64_t* offsets, uint32_t* permutation, float* clustered) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    GPANN_TRY(EnsureDevice(s->device));
    if (offsets) std::copy(s->offsets.begin(), s->offsets.end(), offsets);
    if (permutation && s->rows) GPANN_HIP_TRY(hipMemcpy(permutation, s->permutation, s->rows * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (clustered && s->rows) {
        GPANN_HIP_TRY(hipMemcpy2D(clustered, (size_t)s->d * sizeof(float), s->points, (size_t)s->ld * sizeof(float),
                                  (size_t)s->d * sizeof(float), s->rows, hipMemcpyDeviceToHost));
    }
    return GPANN_OK;
}

gpann_status gpann_shards_export_range(const gpann_shards* s, uint64_t row_begin, uint64_t row_count, uint32_t* permutation,
                                       float* clustered) {
    GPANN_REQUIRE(s != nullptr, "shards handle is null");
    GPANN_REQUIRE(row_begin <= s->rows && row_count <= s->rows - row_begin, "row range out of bounds");
    GPANN_TRY(EnsureDevice(s->device));
    if (row_count == 0) return GPANN_OK;
    if (permutation) GPANN_HIP_TRY(hipMemcpy(permutation, s->permutation + row_begin, row_count * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (clustered) {
        GPANN_HIP_TRY(hipMemcpy2D(clustered, (size_t)s->d * sizeof(float), s->points + row_begin * s->ld, (size_t)s->ld * sizeof(float),
                                  (size_t)s->d * sizeof(float), row_count, hipMemcpyDeviceToHost));
    }
    return GPANN_OK;
}

}  // extern "C"
