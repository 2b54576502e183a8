// Micro-benchmark: what does ordering a side stream behind the kernels of a main stream cost the MAIN stream?
// Per iteration: a ~100 us kernel on s1, then "side work may start" is signalled to s2 (a tiny kernel there), then the next
// iteration's kernel on s1.  Variants of the signal:
//   0  nothing on s2 (baseline)
//   1  hipEventRecord(default event, s1) + hipStreamWaitEvent(s2)
//   2  the same with a hipEventDisableTiming | hipEventDisableSystemFence event
//   3  the event is the STOP event of the kernel's own dispatch (hipExtLaunchKernelGGL), no marker packet on s1
//   4  as ShardedFlow: 2 + s1 first waits for the side work of two iterations ago (event recorded on s2), two buffers
//   5  the same with the stop event of variant 3 for the s1 -> s2 direction
//   6  as 4, and the side stream also records a default event (the one handed to the caller)
// Build: hipcc -O2 --offload-arch=gfx950 tools/ubench/event_gap.hip -o tools/ubench/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
__global__ void tiny(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int* d;
    CK(hipMalloc(&d, 64));
    CK(hipMemset(d, 0, 64));
    hipEvent_t ev_def, ev_light, ev_ext;
    CK(hipEventCreate(&ev_def));
    CK(hipEventCreateWithFlags(&ev_light, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&ev_ext, hipEventDisableTiming | hipEventDisableSystemFence));
    const long long cyc = 10000;      // wall_clock64 runs at 100 MHz: 100 us
    const int iters = 300;
    hipEvent_t back[2], user[2];
    for (int b = 0; b < 2; ++b) {
        CK(hipEventCreateWithFlags(&back[b], hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipEventCreateWithFlags(&user[b], hipEventDisableTiming));
    }
    for (int variant = 0; variant < 7; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; ++i) {
                const int b = i & 1;
                if (variant >= 4 && i >= 2) CK(hipStreamWaitEvent(s1, back[b], 0));
                if (variant == 3 || variant == 5) {
                    hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, nullptr, ev_ext, 0, cyc, (int*)nullptr);
                    CK(hipStreamWaitEvent(s2, ev_ext, 0));
                } else {
                    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, cyc, (int*)nullptr);
                    if (variant == 1) { CK(hipEventRecord(ev_def, s1)); CK(hipStreamWaitEvent(s2, ev_def, 0)); }
                    if (variant == 2 || variant == 4 || variant == 6) { CK(hipEventRecord(ev_light, s1)); CK(hipStreamWaitEvent(s2, ev_light, 0)); }
                }
                if (variant) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s2, d);
                if (variant >= 4) CK(hipEventRecord(back[b], s2));
                if (variant == 6) CK(hipEventRecord(user[b], s2));
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            if (rep) printf("variant %d: %.2f us per iteration\n", variant, us);
        }
    }
    return 0;
}
