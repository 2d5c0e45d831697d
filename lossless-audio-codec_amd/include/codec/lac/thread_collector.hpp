// LAC::ThreadCollector with the reference's interface (src/codec/lac/thread_collector.hpp:8-23: record(id),
// snapshot()), kept so that callers handing one to LAC::Encoder::encode compile unchanged.  The MI355X path has
// no per-block CPU workers: the encoder records the calling thread.
#pragma once
#include <algorithm>
#include <mutex>
#include <set>
#include <thread>
#include <vector>

namespace LAC {

class ThreadCollector {
public:
    // Remembers a thread id (idempotent).
    void record(std::thread::id who) {
        const std::scoped_lock guard(lock_);
        if (std::find(seen_.begin(), seen_.end(), who) == seen_.end()) seen_.push_back(who);
    }

    // The distinct ids recorded so far.
    std::set<std::thread::id> snapshot() const {
        const std::scoped_lock guard(lock_);
        return std::set<std::thread::id>(seen_.begin(), seen_.end());
    }

private:
    mutable std::mutex lock_;
    std::vector<std::thread::id> seen_;
};

}  // namespace LAC
