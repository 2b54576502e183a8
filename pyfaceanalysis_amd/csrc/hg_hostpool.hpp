// Host worker pool: packs caller rows into pinned staging buffers while the GPU works (hg_hostpipe.hpp, hg_capi.cpp).
// Plain C++ (no HIP) so that tests/tsan_pool_driver.cpp can build it with -fsanitize=thread.
//
// One parallel region at a time (callers from several threads queue up); workers are created with the pool.  A region is ONE
// immutable Job object (function, task count, its own ticket counter, its own completion counter) that workers pick up as a
// shared_ptr under the lock: a worker that is still leaving the previous region holds the PREVIOUS job, whose ticket counter
// is exhausted, and can neither take a task of the new region nor count against its completions (with one shared
// counter a late worker ran a new region's task a second time and parallel_for returned while a row was still being packed).
//
// Two forms of a region:
//   parallel_for(n, fn)      the calling thread takes part and returns when every task has finished;
//   begin(n, fn) ... end()   only the workers run tasks (in ticket order: task t is started before task t + 1); the caller goes
//                            on with other work — the host pipeline enqueues copies and kernels meanwhile — and end() waits.
//
// NUMA (round 4): the rows a call packs sit on one memory node of a two-socket box, and a worker on the other socket reads them
// at a third of the rate (tools/ubench/host_pack_bw.cpp: 104 against 288 GB/s with 16 threads).  bind_to_node(k) makes every
// worker restrict itself to the CPUs of node k before its next task; -1 lifts the restriction.
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hg {

// CPUs of a memory node from /sys (empty when the file is not there: no NUMA information, nothing is pinned)
inline std::vector<int> cpus_of_node(int node) {
    std::vector<int> out;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return out;
    char buf[4096];
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    for (char* p = buf; *p;) {
        char* e = nullptr;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        p = e;
        if (*p == '-') {
            b = strtol(p + 1, &e, 10);
            p = e;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) out.push_back((int)c);
        while (*p == ',' || *p == '\n' || *p == ' ') ++p;
    }
    return out;
}

// The CPUs of `cpus` grouped by the last-level cache they share (cache/index3/shared_cpu_list: one group per core complex on
// an EPYC, whose complexes each have their own link to memory); one group holding everything when the files are not there.
inline std::vector<std::vector<int>> group_by_llc(const std::vector<int>& cpus) {
    std::vector<std::vector<int>> groups;
    std::vector<std::string> keys;
    for (int c : cpus) {
        char path[128], key[512];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", c);
        FILE* f = fopen(path, "r");
        if (!f || !fgets(key, sizeof key, f)) {
            if (f) fclose(f);
            return {cpus};
        }
        fclose(f);
        size_t g = 0;
        while (g < keys.size() && keys[g] != key) ++g;
        if (g == keys.size()) {
            keys.push_back(key);
            groups.emplace_back();
        }
        groups[g].push_back(c);
    }
    return groups;
}

// Threads this process may keep busy: hardware threads, cut to a cgroup-v2 CPU quota when there is one (a GPU box of this pool
// shows 256 hardware threads and "1600000 100000" in cpu.max: 16 CPUs' worth of time — more busy threads are throttled).
inline int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
        fclose(f);
    }
    return n;
}

class HostPool {
public:
    static HostPool& get() {
        static HostPool p(0);
        return p;
    }
    // n_threads 0: usable_cpus() less two for the driving thread and the runtime's own threads — under a CPU quota, busy threads
    // beyond it get the whole process throttled for the rest of the period, which shows as calls of 5 ms among calls of 2 ms —
    // (at most 16: with its threads on the data's memory node the packing rate does not grow beyond that),
    // HIGSFA_HOST_THREADS overrides
    explicit HostPool(int n_threads) {
        int n = n_threads;
        if (n <= 0) {
            n = std::min(std::max(1, usable_cpus() - 2), 16);
            if (const char* e = getenv("HIGSFA_HOST_THREADS")) n = atoi(e);
            n = std::max(1, std::min(n, 256));
        }
        const char* pin = getenv("HIGSFA_HOST_PIN");      // ccd (default) | node | spread | off
        pin_mode_ = !pin ? 2 : std::string(pin) == "off" ? 0 : std::string(pin) == "node" ? 1 : std::string(pin) == "spread" ? 3 : 2;
        CPU_ZERO(&all_cpus_);
        (void)sched_getaffinity(0, sizeof all_cpus_, &all_cpus_);      // what the creating thread may use: the "no restriction" mask
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { loop(i); });
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

    int size() const { return (int)workers_.size(); }

    // Workers move to the CPUs of memory node `node` before their next task (-1: any CPU).  Cheap when nothing changes.
    void bind_to_node(int node) {
        if (pin_mode_ == 0) return;
        want_node_.store(node, std::memory_order_release);
    }

    // fn(task) for task in [0, n_tasks); the calling thread takes part.  Returns when every task has finished; the first
    // exception a task threw is rethrown here.
    void parallel_for(int n_tasks, const std::function<void(int)>& fn) {
        if (n_tasks <= 1) {
            for (int t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        begin(n_tasks, fn);
        run(*cur_);
        end();
    }

    // Asynchronous region: the workers run fn(0 .. n_tasks-1), tickets handed out in order; the caller must call end() (which
    // waits and rethrows a task's exception) before `fn` goes out of scope.  The region lock is held from begin() to end().
    // max_workers: only workers 0 .. max_workers-1 take tickets (rows that are only copied into device memory are bound by the
    // link, which four to six writers fill; more of them only add contention)
    void begin(int n_tasks, const std::function<void(int)>& fn, int max_workers = 1 << 30) {
        auto job = std::make_shared<Job>(&fn, n_tasks);      // allocated BEFORE the region lock: a bad_alloc must not leave it held
        job->cap = std::max(1, max_workers);
        region_.lock();
        cur_ = job;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = job;
            ++gen_;
        }
        cv_.notify_all();
    }
    void end() {
        std::shared_ptr<Job> job = std::move(cur_);
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return job->left.load(std::memory_order_acquire) <= 0; });
            job_.reset();
        }
        region_.unlock();
        if (job->error) std::rethrow_exception(job->error);
    }

private:
    struct Job {
        Job(const std::function<void(int)>* f, int n_) : fn(f), n(n_), left(n_) {}
        const std::function<void(int)>* fn;      // alive until left == 0: end() does not return before
        const int n;
        int cap = 1 << 30;
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
                std::lock_guard<std::mutex> lk(m_);      // pairs with the wait in end(): no lost wake-up
                done_.notify_all();
            }
        }
    }
    // pin modes — off: workers run wherever the scheduler puts them; node: anywhere on the rows' memory node; ccd (default):
    // worker i on core complex i mod (complexes of the node) — woken together by one thread, workers otherwise gather in the
    // waker's complex and share its one memory link (153 GB/s against 250 with the node-wide mask on the same box); spread: one
    // fixed CPU each, stepping through the node
    void rebind(int index, int node) {
        cpu_set_t set;
        CPU_ZERO(&set);
        std::vector<int> cpus = node >= 0 ? cpus_of_node(node) : std::vector<int>();
        if (cpus.empty()) {      // no restriction / no information: every CPU this process started with
            set = all_cpus_;
        } else if (pin_mode_ == 3) {
            const size_t step = std::max<size_t>(1, cpus.size() / 2 / std::max<size_t>(1, workers_.size()));
            CPU_SET(cpus[(size_t)index * step % std::max<size_t>(1, cpus.size() / 2)], &set);
        } else if (pin_mode_ == 2) {
            std::vector<std::vector<int>> groups;
            {
                std::lock_guard<std::mutex> lk(topo_m_);      // read once per node, by whichever worker gets here first
                if (topo_node_ != node) {
                    topo_ = group_by_llc(cpus);
                    topo_node_ = node;
                }
                groups = topo_;
            }
            for (int c : groups[(size_t)index % groups.size()]) CPU_SET(c, &set);
        } else {
            for (int c : cpus) CPU_SET(c, &set);
        }
        (void)sched_setaffinity(0, sizeof set, &set);      // refused (cpuset without these CPUs): stay where we are
    }
    void loop(int index) {
        uint64_t seen = 0;
        int node = -1;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                j = job_;          // snapshot of THIS region (null if it is already over)
            }
            const int want = want_node_.load(std::memory_order_acquire);
            if (want != node) {
                rebind(index, want);
                node = want;
            }
            if (j && index < j->cap) run(*j);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, region_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Job> job_, cur_;      // cur_: the open region, touched only by the thread that holds region_
    uint64_t gen_ = 0;
    bool stop_ = false;
    int pin_mode_ = 2;
    std::mutex topo_m_;
    std::vector<std::vector<int>> topo_;
    int topo_node_ = -2;
    std::atomic<int> want_node_{-1};
    cpu_set_t all_cpus_;
};

}  // namespace hg
