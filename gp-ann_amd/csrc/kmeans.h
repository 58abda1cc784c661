// kmeans.h — the batched Lloyd engine of kmeans.hip as a host-side session: every call runs one LEVEL (a batch of independent
// k-means problems over rows of a shard store) and leaves the rows grouped by cluster on the device for the next level.
// Users: gpann_router_train (KMeansTreeRouter::Train) and the shard store's cell builder (cells.hip).
#pragma once

#include "shards.h"

#include <vector>

namespace gpann {

struct LloydTask {
    uint64_t row_begin;   // first entry of the problem in the session's current row array
    uint32_t n;           // rows
    uint32_t k0;          // initial number of centroids
};

struct LloydResult {
    std::vector<uint32_t> cent_begin;   // [problems] first centroid slot
    std::vector<uint32_t> cnum;         // [problems] surviving centroids (rows cent_begin .. cent_begin + cnum of `cent`)
    std::vector<uint32_t> csize;        // [slots] members of every cluster of the LAST round, empty ones included (slot order)
    std::vector<uint32_t> coff;         // [slots] start of the cluster's member slab inside its problem
    std::vector<float> cent;            // [slots][ld] compacted per problem
    uint32_t n_slots = 0, ld = 0;
};

class LloydSession {
public:
    ~LloydSession();
    gpann_status Begin(gpann_shards* store, int metric);   // row array = identity over the store's rows
    // `pick[slot]` = position (inside its problem) of the row that initialises centroid `slot`; slots are numbered problem by problem
    gpann_status RunLevel(const std::vector<LloydTask>& level, const std::vector<uint32_t>& pick, int rounds, LloydResult& out);
    const uint32_t* rows() const;   // device: the current row array (after RunLevel: store rows grouped by cluster, ascending inside)
    void End();

private:
    gpann_shards* store_ = nullptr;
    int metric_ = 0, cur_ = 0;
    DeviceBuffer b_rows[2], b_owner, b_mpos, b_sqrt, b_probs, b_tiles, b_chunks, b_ccnt, b_cbase, b_cent, b_csum, b_cnorm, b_cnum, b_csize, b_coff,
        b_slotprob, b_pick;
};

}  // namespace gpann
