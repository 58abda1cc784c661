// host_util.h — internal helpers of the host mirror (not part of the mirrored interface).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <exception>
#include <mutex>
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

// Group commit for the reference's one-query-per-call interfaces (KMeansTreeRouter::Query, InvertedIndex::Query / QueryBucket): the
// reference's callers issue them from up to 32 threads at once (routes.cpp:194-200, small_scale_queries.cpp:114-116), and on the GPU a
// batch of one costs a whole launch sequence.  Concurrent callers Submit() their request; whoever finds no batch in flight becomes the
// leader of the next one, takes EVERYTHING that has queued up — its own request included —, runs it as one device batch and wakes the
// owners; callers that arrive meanwhile form the following batch.  A single caller pays one mutex round trip and never waits; with several
// callers a leader gives the callers of the previous batches up to 50 us to come back with their next query (else T callers settle into
// two alternating half-size batches).
// Req needs the members `bool done` and `std::exception_ptr error`.
template <typename Req>
class Coalescer {
public:
    template <typename RunBatch>
    void Submit(Req& req, RunBatch run) {
        std::unique_lock<std::mutex> lk(m_);
        req.done = false;
        req.error = nullptr;
        queue_.push_back(&req);
        if (gathering_) cv_.notify_all();   // (a leader is giving its batch a moment to fill up)
        for (;;) {
            if (req.done) break;
            if (!leader_) {   // (no batch in flight and this request not served: it is in the queue — this thread leads the next batch)
                leader_ = true;
                // Callers that were served by the batch before are on their way back with their next query: in the steady state of T callers
                // the batches would alternate between two halves of them.  When the last two batches together were larger than what has queued
                // up, the leader gives the others up to 50 us (a fraction of a launch sequence) to arrive.
                const size_t expect = std::min<size_t>(last_batch_ + before_last_, 4096);
                if ((last_batch_ > 1 || before_last_ > 1) && expect > queue_.size()) {   // (a lone caller — batches of one — never waits)
                    gathering_ = true;
                    cv_.wait_for(lk, std::chrono::microseconds(50), [&] { return queue_.size() >= expect; });
                    gathering_ = false;
                }
                std::vector<Req*> batch;
                batch.swap(queue_);
                lk.unlock();
                std::exception_ptr err;
                try {
                    run(batch);
                } catch (...) {
                    err = std::current_exception();
                }
                lk.lock();
                for (Req* r : batch) {
                    if (err && !r->error) r->error = err;
                    r->done = true;
                }
                leader_ = false;
                ++batches_;
                requests_ += batch.size();
                before_last_ = last_batch_;
                last_batch_ = batch.size();
                cv_.notify_all();
                continue;
            }
            cv_.wait(lk);
        }
        lk.unlock();
        if (req.error) std::rethrow_exception(req.error);
    }
    // batches run / requests served so far (a test hook: requests / batches = the mean batch size)
    void Counters(uint64_t* batches, uint64_t* requests) {
        std::lock_guard<std::mutex> lk(m_);
        *batches = batches_;
        *requests = requests_;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Req*> queue_;
    bool leader_ = false, gathering_ = false;
    size_t last_batch_ = 0, before_last_ = 0;   // sizes of the two most recent batches
    uint64_t batches_ = 0, requests_ = 0;
};

// throws std::runtime_error carrying gpann_last_error() unless status == GPANN_OK (the reference reports internal
// failures with exceptions, SURVEY.md §8b "Errors")
void Check(gpann_status status, const char* what);

}   // namespace gpann_host
