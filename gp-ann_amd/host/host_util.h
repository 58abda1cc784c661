// host_util.h — internal helpers of the host mirror (not part of the mirrored interface).
#pragma once

#include <atomic>
#include <cstddef>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpann.h"

namespace gpann_host {

int NumThreads();

// Fork-join loop in place of parlay::parallel_for: dynamic chunks claimed from an atomic cursor.  Bodies used with it are
// independent per index, so results never depend on the schedule.
template <typename Body>
void ParallelFor(size_t begin, size_t end, int threads, Body body) {
    if (end <= begin) return;
    const size_t count = end - begin;
    size_t workers = threads <= 0 ? (size_t)NumThreads() : (size_t)threads;
    if (workers > count) workers = count;
    if (workers <= 1) {
        for (size_t i = begin; i < end; ++i) body(i);
        return;
    }
    const size_t chunk = count / (workers * 8) + 1;
    std::atomic<size_t> cursor{begin};
    auto run = [&]() {
        for (;;) {
            const size_t lo = cursor.fetch_add(chunk);
            if (lo >= end) return;
            const size_t hi = lo + chunk < end ? lo + chunk : end;
            for (size_t i = lo; i < hi; ++i) body(i);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(workers - 1);
    for (size_t w = 1; w < workers; ++w) pool.emplace_back(run);
    run();
    for (auto& t : pool) t.join();
}

// throws std::runtime_error carrying gpann_last_error() unless status == GPANN_OK (the reference reports internal
// failures with exceptions, SURVEY.md §8b "Errors")
void Check(gpann_status status, const char* what);

}   // namespace gpann_host
