// popsift/common/sync_queue.h -- blocking FIFO between caller threads and the dispatcher
// (same contract as the reference's SyncQueue, common/sync_queue.h:14-55).
#pragma once

#include <condition_variable>
#include <deque>
#include <mutex>

namespace popsift {

template <typename T>
class SyncQueue
{
public:
    SyncQueue() = default;

    void push(const T& value)
    {
        {
            std::lock_guard<std::mutex> g(_m);
            _q.push_back(value);
        }
        _cv.notify_one();
    }

    bool empty()
    {
        std::lock_guard<std::mutex> g(_m);
        return _q.empty();
    }

    /// blocks until an item is available
    T pull()
    {
        std::unique_lock<std::mutex> lk(_m);
        _cv.wait(lk, [this] { return !_q.empty(); });
        T v = _q.front();
        _q.pop_front();
        return v;
    }

    /// non-blocking variant used by the dispatcher to keep several frames in flight
    bool try_pull(T& out)
    {
        std::lock_guard<std::mutex> g(_m);
        if (_q.empty()) return false;
        out = _q.front();
        _q.pop_front();
        return true;
    }

private:
    std::mutex              _m;
    std::deque<T>           _q;
    std::condition_variable _cv;
};

} // namespace popsift
