// Host worker pool: packs caller rows into pinned staging buffers while the GPU works (hg_capi.cpp run_host_rows).
// Plain C++ (no HIP) so that tests/tsan_pool_driver.cpp can build it with -fsanitize=thread.
//
// One parallel region at a time (callers from several threads queue up); workers are created on first use.  A region is ONE
// immutable Job object (function, task count, its own ticket counter, its own completion counter) that workers pick up as a
// shared_ptr under the lock: a worker that is still leaving the previous region holds the PREVIOUS job, whose ticket counter
// is exhausted, and can neither take a task of the new region nor count against its completions (with one shared
// counter a late worker ran a new region's task a second time and parallel_for returned while a row was still being packed).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace hg {

class HostPool {
public:
    static HostPool& get() {
        static HostPool p(0);
        return p;
    }
    // n_threads 0: hardware concurrency (at most 16), HIGSFA_HOST_THREADS overrides
    explicit HostPool(int n_threads) {
        int n = n_threads;
        if (n <= 0) {
            n = (int)std::thread::hardware_concurrency();
            if (const char* e = getenv("HIGSFA_HOST_THREADS")) n = atoi(e);
            n = std::max(1, std::min(n, 16));
        }
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;

    int size() const { return (int)workers_.size() + 1; }

    // fn(task) for task in [0, n_tasks); the calling thread takes part.  Returns when every task has finished; the first
    // exception a task threw is rethrown here.
    void parallel_for(int n_tasks, const std::function<void(int)>& fn) {
        if (n_tasks <= 1 || workers_.empty()) {
            for (int t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        std::lock_guard<std::mutex> region(region_);
        auto job = std::make_shared<Job>(&fn, n_tasks);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = job;
            ++gen_;
        }
        cv_.notify_all();
        run(*job);
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return job->left.load(std::memory_order_acquire) == 0; });
            job_.reset();
        }
        if (job->error) std::rethrow_exception(job->error);
    }

private:
    struct Job {
        Job(const std::function<void(int)>* f, int n_) : fn(f), n(n_), left(n_) {}
        const std::function<void(int)>* fn;      // alive until left == 0: parallel_for does not return before
        const int n;
        std::atomic<int> next{0};
        std::atomic<int> left;
        std::mutex err_m;
        std::exception_ptr error;
    };
    void run(Job& j) {
        for (;;) {
            const int t = j.next.fetch_add(1, std::memory_order_relaxed);
            if (t >= j.n) return;
            try {
                (*j.fn)(t);
            } catch (...) {
                std::lock_guard<std::mutex> lk(j.err_m);
                if (!j.error) j.error = std::current_exception();
            }
            if (j.left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);      // pairs with the wait in parallel_for: no lost wake-up
                done_.notify_all();
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                j = job_;          // snapshot of THIS region (null if it is already over)
            }
            if (j) run(*j);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, region_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Job> job_;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace hg
