// Host worker pool of a context (header-only; used by akz_api.cpp, exercised on its own under ThreadSanitizer by
// tools/fuzz/pool_tsan.cpp).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace akz {

// Host worker threads of a context, started on first use and kept for the context's lifetime: the finish half of an
// extraction runs several short parallel phases per batch (bucketing the candidates, one selection job per image, the
// libm calls per keypoint), and creating ~60 threads per batch cost more than some of those phases.
// run(count, fn) calls fn(i) for every i in [0, count) on the workers and the calling thread and returns when all
// calls have finished.  One run at a time (a context is used by one host thread).
class WorkerPool {
public:
    explicit WorkerPool(unsigned workers) {
        for (unsigned t = 0; t < workers; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& th : threads_) th.join();
    }
    unsigned size() const { return (unsigned)threads_.size() + 1; }
    template <typename F>
    void run(size_t count, F&& fn) {
        if (count <= 1 || threads_.empty()) {
            for (size_t i = 0; i < count; ++i) fn(i);
            return;
        }
        auto r = std::make_shared<Run>();
        r->job = [&fn](size_t i) { fn(i); };
        r->count = count;
        r->pending = count;
        {
            std::lock_guard<std::mutex> lk(m_);
            current_ = r;
            ++generation_;
        }
        // wake as many workers as there is work for besides the caller: with 15 sleeping workers and two pieces (the two
        // images of a 4K pair) notify_all woke all of them to find nothing -- one after the other through the mutex, ~0.1 ms
        // per run() and several runs per finish (2 x 4K call: 4.06 ms with 16 threads against 3.29 with 2)
        if (count - 1 >= threads_.size()) {
            cv_.notify_all();
        } else {
            for (size_t i = 0; i + 1 < count; ++i) cv_.notify_one();
        }
        drain(*r);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return r->pending == 0; });
        current_.reset();
    }

private:
    struct Run {  // one run() call; workers that wake up late hold their own reference and find nothing left to do
        std::function<void(size_t)> job;
        size_t count = 0;
        std::atomic<size_t> next{0};
        size_t pending = 0;  // guarded by m_
    };
    void drain(Run& r) {
        size_t finished = 0;
        for (size_t i = r.next.fetch_add(1); i < r.count; i = r.next.fetch_add(1)) {
            r.job(i);
            ++finished;
        }
        if (finished) {
            std::lock_guard<std::mutex> lk(m_);
            r.pending -= finished;
            if (r.pending == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Run> r;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                r = current_;
            }
            if (r) drain(*r);
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Run> current_;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

}  // namespace akz
