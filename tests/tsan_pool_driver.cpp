// ThreadSanitizer driver for the host worker pool (pyfaceanalysis_amd/csrc/hg_hostpool.hpp, host only): regions of
// alternating small and large task counts back to back, from two calling threads, every task writing its own cell —
// a task run twice, a task of another region, or a return before the last task finished shows up as a wrong cell
// count or as a data race report.
#include <cstdio>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>
#include "hg_hostpool.hpp"

static int hammer(hg::HostPool& pool, int rounds, unsigned seed) {
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        seed = seed * 1664525u + 1013904223u;
        const int n = (r & 1) ? 4 : 2 + (int)((seed >> 16) % 127);        // a 4-task region right after a large one, and back
        std::vector<int> cells((size_t)n, 0);                            // plain ints: TSAN sees any unsynchronised second writer
        pool.parallel_for(n, [&](int t) { cells[(size_t)t] += 1 + t; });
        for (int t = 0; t < n; ++t) bad += cells[(size_t)t] != 1 + t;    // read right after the return: every task must be over
    }
    return bad;
}

int main() {
    hg::HostPool pool(8);
    int bad[2] = {0, 0};
    std::thread other([&] { bad[1] = hammer(pool, 20000, 7u); });
    bad[0] = hammer(pool, 20000, 1u);
    other.join();
    // an exception in a task reaches the caller, and the pool keeps working
    bool thrown = false;
    try {
        pool.parallel_for(64, [&](int t) { if (t == 13) throw std::runtime_error("task 13"); });
    } catch (const std::runtime_error&) { thrown = true; }
    std::vector<int> after(100, 0);
    pool.parallel_for(100, [&](int t) { after[(size_t)t] = t; });
    int sum = std::accumulate(after.begin(), after.end(), 0);
    printf("bad %d %d thrown %d sum %d\n", bad[0], bad[1], (int)thrown, sum);
    return (bad[0] || bad[1] || !thrown || sum != 4950) ? 1 : 0;
}
