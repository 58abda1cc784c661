// shards.h — the device-resident shard store (InvertedIndex layout, reference src/inverted_index.h:8-41).
#pragma once

#include <mutex>
#include <vector>
#include <cstdint>

#include "common.h"

namespace gpann {

// One exact-scan workgroup's job: <= 64 tasks of one shard x one contiguous row segment (see scan.hip).
struct ScanUnit {
    int64_t row_begin;
    int64_t row_end;
    uint32_t task_begin;   // index into the shard-sorted task arrays
    uint32_t task_count;   // 1..64
    uint32_t seg;          // segment number within the shard
    uint32_t pad;
};

// One filter workgroup's job: <= 256 tasks of one shard x one row segment (see scan_filter.h).
// Host-side launch plan of one scan call; kept alive in the handle until the next call because its arrays are the
// sources of asynchronous uploads.
struct ScanPlan {
    std::vector<uint32_t> task_query, task_out, task_part, task_nseg, task_shard, shard_task_begin;
    std::vector<ScanUnit> units;
    uint64_t n_lists = 0, pairs = 0, padded_pairs = 0, filter_pairs = 0, filter_candidates = 0, filter_rows = 0, filter_rest_rows = 0, filter_two_plane_rows = 0;
    uint64_t filter_mfma_bf16 = 0, filter_mfma_i8 = 0, filter_rest_mfma_i8 = 0;
    bool filter_rest_hi_only = false;
    uint32_t filter_launches = 0, filter_dirty_tasks = 0, filter_rest_row_bytes = 0;
    void Clear() {
        task_query.clear();
        task_out.clear();
        task_part.clear();
        task_nseg.clear();
        task_shard.clear();
        shard_task_begin.clear();
        units.clear();
        n_lists = pairs = padded_pairs = filter_pairs = filter_candidates = filter_rows = filter_rest_rows = filter_two_plane_rows = 0;
        filter_mfma_bf16 = filter_mfma_i8 = filter_rest_mfma_i8 = 0;
        filter_launches = filter_dirty_tasks = filter_rest_row_bytes = 0;
    }
};

// Pinned host staging area of a handle: the small per-call arrays (task lists, group tables, counters) travel through it, so their
// hipMemcpyAsync calls are real asynchronous copies — from pageable memory every one of them is a synchronous staging copy of the runtime
// (12 us of idle GPU each, a dozen per scan call).  Reset at the start of a call, after waiting for the previous call's last upload
// (gpann_shards::stage_in_flight): an untimed call with device output returns without synchronising.
struct PinnedStage {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    void Reserve(size_t bytes) {   // only between calls
        used = 0;
        if (bytes <= cap) return;
        if (base) (void)hipHostFree(base);
        base = nullptr;
        cap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&base), bytes, hipHostMallocDefault) == hipSuccess) cap = bytes;
        else (void)hipGetLastError();
    }
    void* Take(size_t bytes) {
        const size_t at = (used + 63) & ~(size_t)63;
        if (!base || at + bytes > cap) return nullptr;
        used = at + bytes;
        return base + at;
    }
    void Release() {
        if (base) (void)hipHostFree(base);
        base = nullptr;
        cap = used = 0;
    }
};

}  // namespace gpann

struct gpann_shards {
    // The reference's callers invoke Query / QueryBucket concurrently, one query per call (shard_searches.cpp:48); a handle owns ONE set of
    // device workspaces and one pinned staging area, so concurrent calls on it take turns (recursive: entry points call each other).
    std::recursive_mutex call_mutex;
    gpann::Options opt;  // the handle's knobs: the environment's GPANN_* at creation, then gpann_shards_set_option (common.h)
    int device = 0;
    int metric = 0;
    uint32_t d = 0;      // true dimension
    uint32_t ld = 0;     // row stride in floats (d rounded up to 4; pad floats are zero and never enter the arithmetic)
    uint32_t S = 0;
    uint64_t rows = 0;   // sum of shard sizes
    bool ids_in_order = false;   // row i holds point i for every row (what gpann_ground_truth asks of a store it is handed: ties are then broken by id)
    float* points = nullptr;         // [rows][ld]   HBM, shard-contiguous
    uint32_t* permutation = nullptr; // [rows]       row -> global id
    int64_t* d_offsets = nullptr;    // [S+1]        device copy
    std::vector<int64_t> offsets;    // [S+1]        host copy

    // MFMA pre-filter support (scan_filter.h): -|p|^2/2 per row, max |p| per shard, and whether the data is fit for it
    float* neg_half_norm = nullptr;  // [rows]
    double* d_shard_rmax = nullptr;  // [S]
    char* frag_planes = nullptr;     // bf16 hi/lo planes in MFMA fragment order, 16 KiB per 32 rows, grouped by cell (with `cells`)

    // Cells (cells.hip): with them the planes hold the rows of every shard GROUPED BY CELL (a k-means sub-cluster of the shard,
    // padded to whole 32-row tiles); a cell's bounding ball lets the filter skip it for all tasks it cannot serve.
    struct Cells {
        uint32_t n_cells = 0, kmax = 0;            // all cells of the store (numbered shard by shard); most cells in one shard
        uint64_t n_tiles = 0;                      // 32-row tiles of the planes
        std::vector<uint32_t> shard_cell_begin;    // [S+1] host
        uint32_t plane_ks = 8;                     // k-steps (16 dims each) per plane tile = PlaneKSteps(d): ceil(d / 16) rounded up to one the kernels exist for; a tile of a plane is plane_ks KiB
        std::vector<uint64_t> cell_tile_begin;     // [n_cells+1] host
        uint32_t* d_shard_cell_begin = nullptr;    // [S+1]
        uint64_t* d_cell_tile_begin = nullptr;     // [n_cells+1]
        float* d_center = nullptr;                 // [n_cells][ld]
        float* d_radius = nullptr;                 // [n_cells] max |p - center| over the cell's rows, rounded up
        float* d_cnorm = nullptr;                  // [n_cells] |center|^2
        uint32_t* plane_perm = nullptr;            // [n_tiles*32] plane row -> store row, 0xFFFFFFFF for padding
        float* plane_norm = nullptr;               // [n_tiles*32 + 64] -|p|^2/2 in plane order (-inf for padding)
        // the shard SAMPLE: the first rows of every shard in store order (a random sample of the shard) once more as planes, so that
        // the filter's first threshold comes from a matrix-core pass over them instead of an exact VALU scan (scan.hip, sample pass)
        // They are appended to the cell tiles — tile sample_tile_base + sample_tile_begin[b] .. of the same plane / perm / norm arrays —
        // so one filter launch can run items of both kinds.
        std::vector<uint64_t> sample_tile_begin;   // [S+1] host: first sample tile of every shard, relative to sample_tile_base
        uint64_t n_sample_tiles = 0;
        uint64_t sample_tile_base = 0;             // = n_tiles + 2 (two zero tiles follow the cell tiles: the kernels' prefetch may run past a range)
        // The lo planes follow the hi tiles (+ 2 zero tiles) as an array of their own, lo_offset_tiles tiles into `frag_planes`.  Only the
        // passes that multiply both planes read them — the shard-sample pass and the head pass — so by default only the sample tiles and
        // the first lo_head_tiles tiles of every cell HAVE a lo tile (NS: 0.5 GB instead of 51 GB): lo tile cell_lo_begin[c] + i belongs
        // to tile i of cell c, lo tile lo_sample_begin + j to sample tile j.  lo_full (GPANN_FILTER_LO_ROWS=all at store build): every
        // tile has one, at its own index (the layout of rounds 1-2; allows GPANN_FILTER_HI_ONLY=0 / GPANN_FILTER_HI_FROM_PASS=2).
        uint64_t lo_offset_tiles = 0;
        uint32_t lo_head_tiles = 0;                // lo tiles kept per cell (its first tiles); 0xFFFFFFFF when lo_full
        bool lo_full = false;
        uint64_t n_lo_tiles = 0;                   // tiles of the lo array (cells' heads | 2 zero tiles | sample tiles)
        uint64_t lo_sample_begin = 0;              // first lo tile of the shard sample
        uint32_t* d_cell_lo_begin = nullptr;       // [n_cells+1] first lo tile of every cell
        // The hi array likewise holds only the tiles a bf16 pass can reach when the store has the 8-bit plane for the rest pass (hi_compact):
        // the sample tiles and the first max(16, tiles / hi_keep_div + 1) tiles of every cell — hi tile cell_hi_begin[c] + i belongs to tile i of
        // cell c, hi tile hi_sample_begin + j to sample tile j (NS: 3.2 GB instead of 25.6 GB).  Not compact: every tile at its own index.
        bool hi_compact = false;
        uint32_t hi_keep_div = 1;                  // the sample pass must not cover more than 1 / hi_keep_div of a cell
        uint64_t n_hi_cell_tiles = 0, hi_sample_begin = 0;
        uint32_t* d_cell_hi_begin = nullptr;       // [n_cells+1] first hi tile of every cell
        // The 8-bit plane of the rest pass (scan_q8.h; absent when GPANN_FILTER_Q8=0 or the allocation failed): per 32-row tile and k-step of 32
        // dimensions one KiB in the A-operand order of v_mfma_i32_32x32x32_i8 — lane (c, h) holds row c, dimensions 32 ks + 16 h .. + 15 as int8
        // P' = round((p~ - c^) / q8_scale[cell]) (p~, c^: the row and its cell's centre in the frame of the bf16 planes).  q8_ks KiB per tile.
        char* q8_planes = nullptr;
        uint32_t q8_ks = 0;                        // k-steps of 32 dimensions = ceil(d / 32)
        float* d_q8_scale = nullptr;               // [n_cells] quantisation step of the cell: max |p~_i - c^_i| / 127, rounded up
        double* d_q8_cnorm = nullptr;              // [n_cells] |c^|^2
        // Squared L2 (round 4): the 8-bit passes' own norm words, -|p~|^2/2 + c^.e per plane row (e = p~ - c^ - s P': the row's quantisation error,
        // known at store build), so that the only unknown part of q~.e is (q~ - c^).e, bounded by (s/2) |q~ - c^|_1 — the query in the frame of the
        // CELL, 2-3x smaller than (s/2) |q~|_1 — without a per-cell query image (scan_cells.h, Q8Threshold).  [(cell tiles + 2) * 32], -inf on padding.
        float* q8_norm = nullptr;
        float2* d_tile_bound = nullptr;            // [cell tiles + 2] (max over the tile's rows of plane_norm = -min |p~|^2 / 2, 127 max_row sum |P'|): cells.hip, TileBoundKernel
        // squared L2 only: the planes, their norms, the shards' max norms and the query images are taken around the store's mean row
        // `mu` ([ld] device, pad zero; nullptr = not centred: inner-product stores)
        float* d_mu = nullptr;
    } cells;
    bool filter_ok = false;
    gpann_status filter_fail_status = GPANN_OK;   // why BuildCells gave up (cells.hip): GPANN_ERR_OOM (an allocation), GPANN_ERR_HIP (a device / kernel error), GPANN_ERR_INTERNAL (no cells)
    int scan_mode = 0;               // 0 = exact scan, 1 = MFMA-filtered exact scan (gpann_shards_set_scan_mode)

    // Per-call host state, TWO sets (round 4): a call issued with GPANN_SCAN_DEFER returns once its work is enqueued, and the NEXT call may be
    // planned, staged and enqueued behind it before gpann_scan_finish completes the first — the host-side planning of batch i + 1 then runs under
    // the kernels of batch i.  `cur` is the slot of the most recent call.
    static constexpr int kCallSlots = 3;   // a call in progress + two deferred ones behind it (GPANN_SCAN_DEFER, include/gpann.h)
    struct CallState {
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [0..2] begin / scan end / merge end of the call, [3] the end of its enqueued device work
        bool stage_in_flight = false;    // ev[3] is recorded and not yet waited for: the pinned stage / plan arrays may still be read by the device
        hipEvent_t ev_filter[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // begin/end of the head pass, of the sample + rest passes,
                                                                                              // and of the rest pass's filter launch alone
        hipEvent_t ev_x[4] = {nullptr, nullptr, nullptr, nullptr};   // GPANN_SCAN_OVERLAP: [0] what the caller's stream held when the call was issued (the front
                                                            // stream waits for it), [1] the end of the call's front (the rest stream waits for it),
                                                            // [2] the end of its rest-pass launch (the tail stream waits for it),
                                                            // [3] the START of its rest-pass launch (the front of the call two after it waits for it)
        bool rest_begin_recorded = false;                   // ev_x[3] was recorded for the call this slot holds
        int filter_passes_timed = 0;
        gpann::ScanPlan plan;
        gpann::PinnedStage stage;
        uint32_t* pin_words = nullptr;   // pinned [24]: the filtered pipeline's counters land here (overflow / dirty words [0..3], work counters [4..17])
        // what the completion of the call (CompleteCall: at once, or in gpann_scan_finish) needs to know
        bool pending = false;            // enqueued, not completed yet
        bool filtered = false;           // the filtered pipeline ran (its counters are on their way to pin_words)
        bool overlapped = false;         // GPANN_SCAN_OVERLAP took effect: the call's work went to the store's own streams
        uint32_t dirty_cap = 0;
        const float* d_queries = nullptr;   // the device queries the kernels read (for the exact redo of an overflowed call)
        uint32_t nq = 0, k = 0, nprobe = 0, T = 0;
        int flags = 0;
        void* stream = nullptr;
        float* out_dist = nullptr;       // the call's DEVICE result arrays
        uint32_t* out_ids = nullptr;
        uint32_t* out_rows = nullptr;
        float* host_dist = nullptr;      // host destinations of a call without GPANN_OUTPUT_ON_DEVICE (null: none)
        uint32_t* host_ids = nullptr;
        uint32_t* host_rows = nullptr;
        // The call's DEVICE workspaces, grown on demand and reused by the next call on this slot.  One set per slot (round 5; rounds 1-4: one set
        // per handle, shared by both slots and kept apart only by stream order): two calls of a handle can be on the device AT ONCE — the passes
        // before the rest pass of call i + 1 run beside the rest pass of call i on streams of their own (GPANN_SCAN_OVERLAP) — and a workspace
        // that grows (hipFree synchronises the whole device) can no longer stall or clobber the other call.  ws_stage_dev: the device mirror of
        // this slot's pinned `stage` (the call's small arrays arrive in it with one copy: scan.hip UploadBatch).
        gpann::DeviceBuffer ws_queries, ws_tasks, ws_units, ws_partials, ws_out, ws_misc, ws_lists, ws_cand, ws_celld, ws_need, ws_image, ws_q8, ws_split, ws_stage_dev;
        void ReleaseWorkspaces() {
            for (gpann::DeviceBuffer* b : {&ws_queries, &ws_tasks, &ws_units, &ws_partials, &ws_out, &ws_misc, &ws_lists, &ws_cand, &ws_celld, &ws_need, &ws_image, &ws_q8,
                                           &ws_split, &ws_stage_dev})
                b->Release();
        }
    } cs[kCallSlots];
    int cur = 0;
    double filter_pair_share = -1.0;   // (task, row) pairs the last completed filtered call sent to the matrix cores / pairs its caller asked for (< 0: none yet): how well the cells prune — the small-shard rule of RunCellFilteredScan goes by it
    double rest_products_per_tile = 0.0;   // (query tile, row tile) products per row tile of the last completed filtered call (0: none yet) — all passes' products over the rest pass's tiles
    gpann_scan_stats stats{};
    // GPANN_SCAN_OVERLAP: the handle's own streams (created at the first such call): a front stream per call slot (high priority: short,
    // latency-bound launches; GPANN_OVERLAP_FRONT_STREAMS = 1: the fronts one after the other, 3: the fronts of all calls in flight side by side —
    // chains of dependent launches fill each other's gaps), the rest-pass launches of all calls back to back on ONE stream (low priority: it takes
    // what the others leave), what follows a rest pass — the re-scoring of its candidates, the final merge, the lists' way out — on ONE tail stream (the tails are
    // short and ordered like the rest passes anyway; every stream more is one more candidate for sharing a hardware queue with the rest pass:
    // the runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, and a front that lands in the rest
    // pass's queue waits for it — INTEGRATION.md)
    hipStream_t st_front[kCallSlots] = {nullptr, nullptr, nullptr}, st_rest = nullptr, st_tail = nullptr;
    int overlap_kernels_state = 0;   // 0: not asked yet, 1: this store's matrix-core kernels take whole SIMDs (overlap allowed), 2: they do not (scan.hip OverlapKernelsOwnTheirSimds)
    int overlap_kernels_regs = 0;    // the fewest vector registers the runtime reports for one of them (0: the question failed)
    bool overlap_refusal_said = false;
};

namespace gpann {
// cells.hip
gpann_status BuildCells(gpann_shards* s, uint32_t target_rows);
void ReleaseCells(gpann_shards* s);
}  // namespace gpann
