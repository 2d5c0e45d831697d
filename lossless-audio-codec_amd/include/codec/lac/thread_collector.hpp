// Mirror of the reference's src/codec/lac/thread_collector.hpp:8-23: kept so that callers passing a
// ThreadCollector* compile unchanged.  The GPU path has no per-block CPU workers; the collector records
// the calling thread only.
#pragma once
#include <mutex>
#include <set>
#include <thread>

class ThreadCollector {
public:
    void record(std::thread::id id) {
        std::lock_guard<std::mutex> lock(mutex_);
        ids_.insert(id);
    }
    std::size_t count() const {
        std::lock_guard<std::mutex> lock(mutex_);
        return ids_.size();
    }

private:
    mutable std::mutex mutex_;
    std::set<std::thread::id> ids_;
};
