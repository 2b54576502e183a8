// Host side of the ndarray-in call in isolation: how fast can T threads narrow float64 rows of integer pixel values to
// uint8 (hg_hostpack.cpp), how fast does pinned / pageable memory cross PCIe.  Sizes the packing pool and the chunk
// schedule of run_host_rows (hg_capi.cpp).   build: see tools/ubench/host_pack_bw.sh
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <fstream>
#include <sstream>
#include <string>

static int node_of(const void* p) {      // NUMA node that holds the page of p (get_mempolicy with MPOL_F_NODE | MPOL_F_ADDR)
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, (unsigned long)p, 3ul) != 0) return -1;
    return node;
}
static std::vector<int> cpus_of_node(int node) {
    std::vector<int> out;
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s;
    std::getline(f, s);
    std::stringstream ss(s);
    std::string part;
    while (std::getline(ss, part, ',')) {
        int a, b;
        if (sscanf(part.c_str(), "%d-%d", &a, &b) == 2) for (int c = a; c <= b; ++c) out.push_back(c);
        else if (sscanf(part.c_str(), "%d", &a) == 1) out.push_back(a);
    }
    return out;
}

namespace hg {
bool narrow_row_f64(const double* src, uint8_t* dst, int64_t n);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 4096, cols = 16384;
    std::vector<double> x((size_t)rows * cols);
    for (size_t i = 0; i < x.size(); ++i) x[i] = (double)(i * 2654435761u >> 24 & 255);      // first touch: this thread
    uint8_t *pin = nullptr, *dev = nullptr;
    if (hipHostMalloc((void**)&pin, (size_t)rows * cols, hipHostMallocDefault) != hipSuccess) return 1;
    if (hipMalloc((void**)&dev, (size_t)rows * cols) != hipSuccess) return 1;
    printf("hardware_concurrency %u  data on node %d  pinned slot on node %d  this thread on cpu %d\n", std::thread::hardware_concurrency(),
           node_of(x.data()), node_of(pin), sched_getcpu());
    const int data_node = node_of(x.data());
    for (int mode = 0; mode < 3; ++mode)      // 0: unpinned, 1: pinned to the data's node, stride 4 over its first-thread cpus, 2: other node
    for (int T : {8, 16, 24, 32}) {
        if (T > (int)std::thread::hardware_concurrency()) break;
        std::vector<int> cpus = cpus_of_node(mode == 2 ? !data_node : (data_node < 0 ? 0 : data_node));
        if (mode && cpus.size() < 64) continue;
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            std::atomic<int64_t> next{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    if (mode) {      // first hardware threads of the node only (the first half of the list), spread: t -> cpu (t * 64 / T)
                        cpu_set_t set;
                        CPU_ZERO(&set);
                        CPU_SET(cpus[(size_t)t * 64 / T % 64], &set);
                        sched_setaffinity(0, sizeof(set), &set);
                    }
                    for (;;) {
                        const int64_t r = next.fetch_add(8);
                        if (r >= rows) return;
                        for (int64_t q = r; q < r + 8 && q < rows; ++q) hg::narrow_row_f64(x.data() + q * cols, pin + q * cols, cols);
                    }
                });
            for (auto& t : th) t.join();
            best = std::min(best, now() - t0);
        }
        printf("narrow f64 -> u8  mode %d  threads %3d  %.3f ms  %.1f GB/s read (incl. thread start)\n", mode, T, best * 1e3, rows * cols * 8 / best / 1e9);
    }
    hipStream_t s;
    (void)hipStreamCreate(&s);
    for (size_t mb : {1, 4, 16, 64}) {
        const size_t bytes = std::min<size_t>(mb << 20, (size_t)rows * cols);
        double best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            const double t0 = now();
            (void)hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            best = std::min(best, now() - t0);
        }
        printf("H2D pinned   %3zu MiB  %.3f ms  %.1f GB/s\n", mb, best * 1e3, bytes / best / 1e9);
        std::vector<uint8_t> pg(bytes, 1);
        best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            const double t0 = now();
            (void)hipMemcpyAsync(dev, pg.data(), bytes, hipMemcpyHostToDevice, s);
            (void)hipStreamSynchronize(s);
            best = std::min(best, now() - t0);
        }
        printf("H2D pageable %3zu MiB  %.3f ms  %.1f GB/s\n", mb, best * 1e3, bytes / best / 1e9);
    }
    return 0;
}
