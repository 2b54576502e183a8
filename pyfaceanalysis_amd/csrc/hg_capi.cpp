// C ABI (include/higsfa.h): flow handle, host/device execute, profiling.
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include <dlfcn.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <hsa/hsa_ext_amd.h>      // types only: hsa_amd_pointer_info is looked up in the runtime the process already has

#include "hg_common.hpp"
#include "hg_hostpipe.hpp"

namespace {

thread_local std::string g_last_error;

template <typename F>
int guarded(F&& fn) {
    try {
        fn();
        return HG_OK;
    } catch (const hg::Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of host memory";
        return HG_ERR_NOMEM;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return HG_ERR_STATE;
    }
}

}  // namespace

namespace hg {
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace hg

namespace {

using hg::HostPool;      // hg_hostpool.hpp

constexpr int NB = 6;        // passes in flight per replica: input / feature buffers rotate over NB slots
constexpr int NC = 2;        // copy queues: pieces alternate between them, so that the fixed cost of one copy (~13 us on this
                             // runtime: queue packet, completion signal) runs under the transfer of the other
constexpr int MK = 128;      // events that mark points in the copy queue (ring reuse of calls longer than the pinned ring)

// NUMA node that holds the page of p (get_mempolicy(MPOL_F_NODE | MPOL_F_ADDR)); -1 when the kernel does not say
int node_of_address(const void* p) {
    int node = -1;
    if (syscall(SYS_get_mempolicy, &node, nullptr, 0ul, (unsigned long)p, 3ul) != 0) return -1;
    return node;
}

// NUMA node the device hangs off (/sys/bus/pci/devices/<bus id>/numa_node); -1 when unknown
int node_of_device(int device) {
    char bdf[64] = {0}, path[160];
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) return -1;
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = fopen(path, "r")) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    return node;
}

// May the HOST store into this device allocation?  hipDeviceAttributeIsLargeBar says what the device can do, not what THIS
// allocation is: the documented answer is hsa_amd_pointer_info — `hostBaseAddress`, "base address at which the host agent may access
// the allocation" (hsa_ext_amd.h) — which must be non-null and, because the packers store through the device pointer's own value,
// equal to the address the device uses; the range written must lie inside the allocation.  The function is taken from the HSA
// runtime that is ALREADY in the process (the one HIP runs on; RTLD_NOLOAD: never a second copy); without it, or with any answer
// but a clear yes, the call falls back to the pinned ring and the copy queues — never discovered by faulting.
// HIGSFA_HOST_PROBE_DENY=1 (read per probe; tests) forces the "not mappable" answer.
bool host_can_store(const void* p, size_t bytes) {
    typedef hsa_status_t (*info_fn)(const void*, hsa_amd_pointer_info_t*, void* (*)(size_t), uint32_t*, hsa_agent_t**);
    static const info_fn fn = [] {
        info_fn f = nullptr;
        for (const char* name : {"libhsa-runtime64.so.1", "libhsa-runtime64.so"})
            if (void* h = dlopen(name, RTLD_NOLOAD | RTLD_NOW)) {
                f = (info_fn)dlsym(h, "hsa_amd_pointer_info");
                if (f) break;
            }
        return f;
    }();
    if (!fn || !p || getenv("HIGSFA_HOST_PROBE_DENY")) return false;
    hsa_amd_pointer_info_t info;
    memset(&info, 0, sizeof info);
    info.size = sizeof info;
    if (fn(p, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA || !info.hostBaseAddress || info.hostBaseAddress != info.agentBaseAddress) return false;
    const char *base = (const char*)info.agentBaseAddress, *q = (const char*)p;
    return q >= base && q + bytes <= base + info.sizeInBytes;
}

// One execution context: the executor on one device with its streams and staging buffers.
struct Replica {
    int device = -1;
    std::unique_ptr<hg::Executor> exec;
    hipStream_t compute = nullptr, copy[NC] = {};
    void* ring = nullptr;                  // pinned host staging the packers write and the copy queue reads (hg_hostpipe.hpp)
    size_t ring_bytes = 0;
    void* hy[NB] = {};      // pinned: features of a pass on their way back
    size_t hy_bytes = 0;
    hg::DevBuf dx[NB], dy[NB];
    hipEvent_t ev_h2d[NB][NC] = {}, ev_out[NB] = {};
    hipEvent_t mk_ev[MK] = {};             // created on first use
    std::unique_ptr<HostPool> pool;        // shard replicas pack with a few threads of their own; null: the process-wide pool
    bool large_bar = false;                // the host can store into this device's memory (hipDeviceAttributeIsLargeBar)
    const void* dx_probed[NB] = {};        // host_can_store() asked about dx[b] at this address / size ...
    size_t dx_probed_bytes[NB] = {};
    bool dx_host_ok[NB] = {};              // ... and answered this
    int last_transport = -1;               // of the last host-rows call: 0 pinned ring + copy queues, 1 direct stores (hg_flow_host_transport)
    // every input buffer of the rotation confirmed host-writable (asked once per allocation)
    bool inputs_host_writable() {
        bool ok = true;
        for (int b = 0; b < NB; ++b) {
            if (dx_probed[b] != dx[b].p || dx_probed_bytes[b] != dx[b].bytes) {
                dx_host_ok[b] = host_can_store(dx[b].p, dx[b].bytes);
                dx_probed[b] = dx[b].p;
                dx_probed_bytes[b] = dx[b].bytes;
            }
            ok = ok && dx_host_ok[b];
        }
        return ok;
    }

    void create(int dev) {
        HG_HIP(hipSetDevice(dev));
        exec->to_device();
        need_streams();
        HG_HIP(hipDeviceSynchronize());
        int lb = 0;
        large_bar = hipDeviceGetAttribute(&lb, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && lb != 0;
        device = dev;
    }
    // Streams and events of the host path.  Created with the replica, not on first use: which hardware queue a later stream of
    // the process lands on depends on how many exist already, and with these three in place the side stream of the sharded step
    // (sharded.py) shares its queue with the kernels' stream — a hand-off inside one queue costs 12-17 us per step, across two
    // queues 35 (profiles/r04_rccl_world1.txt: measured both ways).
    void need_streams() {
        if (compute) return;
        HG_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
        for (int c = 0; c < NC; ++c) HG_HIP(hipStreamCreateWithFlags(&copy[c], hipStreamNonBlocking));
        for (int b = 0; b < NB; ++b) {
            for (int c = 0; c < NC; ++c) HG_HIP(hipEventCreateWithFlags(&ev_h2d[b][c], hipEventDisableTiming));
            HG_HIP(hipEventCreateWithFlags(&ev_out[b], hipEventDisableTiming));
        }
    }
    void need_pinned(size_t ring_b, size_t yb) {
        if (ring_b > ring_bytes) {
            if (ring) (void)hipHostFree(ring);
            ring = nullptr;
            ring_bytes = 0;
            HG_HIP(hipHostMalloc(&ring, ring_b, hipHostMallocDefault));
            ring_bytes = ring_b;
        }
        if (yb > hy_bytes) {
            for (int b = 0; b < NB; ++b) {
                if (hy[b]) (void)hipHostFree(hy[b]);
                hy[b] = nullptr;
            }
            hy_bytes = 0;
            for (int b = 0; b < NB; ++b) HG_HIP(hipHostMalloc(&hy[b], yb, hipHostMallocDefault));
            hy_bytes = yb;
        }
    }
    void destroy() {
        if (device < 0 || hipSetDevice(device) != hipSuccess) return;
        if (ring) (void)hipHostFree(ring);
        ring = nullptr;
        ring_bytes = 0;
        for (int b = 0; b < NB; ++b) {
            if (hy[b]) (void)hipHostFree(hy[b]);
            for (int c = 0; c < NC; ++c) {
                if (ev_h2d[b][c]) (void)hipEventDestroy(ev_h2d[b][c]);
                ev_h2d[b][c] = nullptr;
            }
            if (ev_out[b]) (void)hipEventDestroy(ev_out[b]);
            dx[b].free();
            dy[b].free();
            hy[b] = nullptr;
            ev_out[b] = nullptr;
        }
        hy_bytes = 0;
        for (auto& e : mk_ev) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        if (compute) (void)hipStreamDestroy(compute);
        for (int c = 0; c < NC; ++c) {
            if (copy[c]) (void)hipStreamDestroy(copy[c]);
            copy[c] = nullptr;
        }
        compute = nullptr;
        if (exec) exec->release();
        device = -1;
    }
};

}  // namespace

struct hg_flow {
    std::unique_ptr<hg::TNode> root;
    Replica main;                                   // the handle's own device (hg_flow_to_device)
    std::vector<std::unique_ptr<Replica>> shards;   // replicas of hg_flow_execute_sharded, one per listed device
    std::unique_ptr<hg::Executor>& exec = main.exec;
    std::string fused_reject;  // why the fused plan was not chosen ("" if it was)
    int& device = main.device;
    int64_t flops = 0;
    bool profiling = false, force_generic = false;
    bool narrow = true;                  // HIGSFA_NO_NARROW, read once in hg_flow_load
    bool direct = true;                  // HIGSFA_HOST_DIRECT=0 (read once in hg_flow_load): host rows through the pinned ring and the copy
                                         // queues even where the host could store into device memory
    std::vector<hipEvent_t> events;
    std::vector<hg::StageProfile> prof;

    void need_device() const {
        if (device < 0) hg::fail(HG_ERR_DEVICE, "flow is not on a device: call hg_flow_to_device first");
    }
    void set_device() const { HG_HIP(hipSetDevice(device)); }
    void drop_events() {
        for (auto e : events) (void)hipEventDestroy(e);
        events.clear();
    }
    ~hg_flow() {
        if (device >= 0 && hipSetDevice(device) == hipSuccess) drop_events();
        main.destroy();
        for (auto& r : shards) r->destroy();
    }
};

namespace {

void check_exec_args(const hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx, const void* y, int y_dtype,
                     int64_t y_cols, int64_t ldy) {
    if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
    if (n < 0) hg::fail(HG_ERR_ARG, "negative row count");
    if (x_dtype != HG_U8 && x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "bad input dtype %d", x_dtype);
    if (y_dtype != HG_F32 && y_dtype != HG_F64) hg::fail(HG_ERR_ARG, "output dtype must be HG_F32 or HG_F64");
    if (y_cols <= 0 || y_cols > (int64_t)f->root->out_dim)
        hg::fail(HG_ERR_DIM, "y_cols %lld outside 1..output_dim (%u)", (long long)y_cols, f->root->out_dim);
    if (ldx < (int64_t)f->root->in_dim)
        hg::fail(HG_ERR_DIM, "x has %lld columns per row but the flow's input_dim is %u", (long long)ldx, f->root->in_dim);
    if (ldy < y_cols) hg::fail(HG_ERR_ARG, "ldy %lld < y_cols %lld", (long long)ldy, (long long)y_cols);
    if (n > 0 && (!x || !y)) hg::fail(HG_ERR_ARG, "null data pointer");
}

void run_on_device(hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                   int64_t ldy, hipStream_t st, Replica* rep = nullptr) {
    if (n == 0) return;
    if (rep && rep != &f->main) {      // shard replicas: never profiled (the benchmark= kwarg belongs to the handle's own device)
        rep->exec->run(x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy, st, nullptr);
        return;
    }
    hipEvent_t* ev = nullptr;
    const int ns = f->exec->n_stages();
    if (f->profiling) {
        if ((int)f->events.size() != ns + 1) {
            f->drop_events();
            f->events.resize(ns + 1);
            for (auto& e : f->events) HG_HIP(hipEventCreate(&e));
        }
        ev = f->events.data();
    }
    f->exec->run(x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy, st, ev);
    if (f->profiling) {
        // profiling is a diagnostic mode: it synchronises so that the event times can be read
        HG_HIP(hipStreamSynchronize(st));
        if ((int)f->prof.size() != ns) {
            f->prof.assign(ns, hg::StageProfile());
            for (int i = 0; i < ns; ++i) f->prof[i].name = f->exec->stage_name(i);
        }
        for (int i = 0; i < ns; ++i) {
            float ms = 0;
            HG_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            f->prof[i].total_ms += ms;
            f->prof[i].launches += 1;
        }
    }
}


// Device side of the host pipeline (hg_hostpipe.hpp): the copy queue fills the input buffer of pass P (slot P % NB) piece by
// piece, the compute queue runs the pass and brings its features back through a pinned slot.
//   slot reuse: dx[b] after ev_out[b] (kernels of the pass that used it: the copy queue waits); dy[b] in queue order;
//   hy[b] after the host has taken the features of the pass that used it (unpack, which waits for ev_out[b]).
struct HipSink final : hg::PipeSink {
    hg_flow* f;
    Replica& rep;
    char* y;
    int y_dtype;
    int64_t y_cols, ldy, in_dim;
    size_t ys;
    int wire_dtype = HG_U8;
    size_t row_wire = 0;
    int pass_base = 0;                                     // passes launched by earlier run_pipe invocations of this call
    int64_t row_base = 0;                                  // first row of this invocation within the call
    std::vector<std::pair<int64_t, int64_t>> pass_rows;    // (first row, rows) of every pass launched, in the call's rows
    int unpacked = 0;                                      // passes whose features are in the caller's y
    uint64_t mk_seq = 0, mk_done = 0;
    int n_queues = NC;                                     // HIGSFA_COPY_QUEUES=1: every piece through one queue
    uint64_t n_copies = 0;
    int waited[NC], used[NC], last_queue = 0;              // per copy queue: last pass whose buffer it has waited for / it has copied into
    bool timed = false;                                    // HIGSFA_HOST_TRACE: timed events around every pass
    std::vector<hipEvent_t> t_ev;                          // base, then (begin, end) per pass

    ~HipSink() {
        for (auto e : t_ev) (void)hipEventDestroy(e);
    }
    void stamp() {
        hipEvent_t e = nullptr;
        HG_HIP(hipEventCreate(&e));
        t_ev.push_back(e);
        HG_HIP(hipEventRecord(e, rep.compute));
    }
    HipSink(hg_flow* f_, Replica& r, void* y_, int ydt, int64_t yc, int64_t ldy_, int64_t in_dim_)
        : f(f_), rep(r), y((char*)y_), y_dtype(ydt), y_cols(yc), ldy(ldy_), in_dim(in_dim_), ys(hg::dtype_size(ydt)) {
        static const char* q = getenv("HIGSFA_COPY_QUEUES");
        if (q) n_queues = std::max(1, std::min(NC, atoi(q)));
        for (int c = 0; c < NC; ++c) waited[c] = used[c] = -1;
    }

    void unpack(int P) {      // features of pass P: pinned slot -> caller rows
        const int b = P % NB;
        // poll before blocking: hipEventSynchronize puts the thread to sleep and the wake-up costs 20-40 us — a third of a small call
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t q = hipEventQuery(rep.ev_out[b]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HG_HIP(q);
            (void)hipGetLastError();
            if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HG_HIP(hipEventSynchronize(rep.ev_out[b]));
                break;
            }
            hg::cpu_relax();
        }
        const int64_t r0 = pass_rows[(size_t)P].first, m = pass_rows[(size_t)P].second;
        const char* src = (const char*)rep.hy[b];
        char* dst = y + (size_t)r0 * ldy * ys;
        const size_t row = (size_t)y_cols * ys;
        if (ldy == y_cols) memcpy(dst, src, row * m);
        else for (int64_t r = 0; r < m; ++r) memcpy(dst + (size_t)r * ldy * ys, src + (size_t)r * row, row);
    }
    void unpack_upto(int P) {      // every launched pass below P
        for (; unpacked < P; ++unpacked) unpack(unpacked);
    }
    void copy(int pass, int64_t dst_row, const void* src, int64_t rows) override {
        const int P = pass_base + pass, b = P % NB, c = (int)(n_copies++ % (uint64_t)n_queues);
        if (waited[c] < P) {      // first piece of this pass on this queue: dx[b] is free once the kernels of pass P - NB are done
            if (P >= NB) HG_HIP(hipStreamWaitEvent(rep.copy[c], rep.ev_out[b], 0));
            waited[c] = P;
        }
        HG_HIP(hipMemcpyAsync((char*)rep.dx[b].p + (size_t)dst_row * row_wire, src, (size_t)rows * row_wire, hipMemcpyHostToDevice, rep.copy[c]));
        used[c] = P;
        last_queue = c;
    }
    void launch(int pass, int64_t r0, int64_t rows) override {
        const int P = pass_base + pass, b = P % NB;
        for (int c = 0; c < n_queues; ++c) {
            if (used[c] != P) continue;      // no piece of this pass went through queue c
            HG_HIP(hipEventRecord(rep.ev_h2d[b][c], rep.copy[c]));
            HG_HIP(hipStreamWaitEvent(rep.compute, rep.ev_h2d[b][c], 0));
        }
        if (P >= NB) unpack_upto(P - NB + 1);
        if (timed) stamp();
        // the last kernel of the pass stores the features straight into the pinned slot (device-visible host memory: 160 bytes per
        // row over PCIe), no copy command behind it
        run_on_device(f, rep.dx[b].p, wire_dtype, rows, in_dim, rep.hy[b], y_dtype, y_cols, y_cols, rep.compute, &rep);
        if (timed) stamp();
        HG_HIP(hipEventRecord(rep.ev_out[b], rep.compute));
        pass_rows.emplace_back(row_base + r0, rows);
    }
    uint64_t mark() override {
        const uint64_t s = ++mk_seq;
        hipEvent_t& e = rep.mk_ev[s % MK];
        if (!e) HG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        else
            while (s > MK && mk_done < s - MK) {      // the slot still carries mark s - MK, which nobody has seen completed yet
                HG_HIP(hipEventSynchronize(rep.mk_ev[(mk_done + 1) % MK]));      // in order: the marks sit on two queues
                ++mk_done;
            }
        HG_HIP(hipEventRecord(e, rep.copy[last_queue]));      // the queue the piece just went through
        return s;
    }
    bool finished(int pass) override {      // direct mode: only asked about launched passes, before their slot's event is recorded again
        const hipError_t q = hipEventQuery(rep.ev_out[(pass_base + pass) % NB]);
        if (q == hipSuccess) return true;
        if (q != hipErrorNotReady) HG_HIP(q);
        (void)hipGetLastError();
        return false;
    }
    bool reached(uint64_t m) override {      // polled oldest first
        if (m <= mk_done) return true;
        const hipError_t q = hipEventQuery(rep.mk_ev[m % MK]);
        if (q == hipSuccess) {
            mk_done = m;
            return true;
        }
        if (q != hipErrorNotReady) HG_HIP(q);
        (void)hipGetLastError();      // not ready is not an error: clear the sticky code
        return false;
    }
};

// What the pass planner assumes (hg_hostpipe.hpp plan_passes): microseconds per row until a row is on the device, and the cost
// of a pass.  Rates from tools/ubench/host_pack_bw.cpp on the pool's boxes (two sockets of EPYC 9575F, PCIe Gen5 x16): 16 threads
// on the rows' memory node narrow float64 at 288 GB/s of caller bytes, pinned memory crosses PCIe at 50 - 57 GB/s in 4 - 64 MiB
// copies; a pass costs ~8 us per launch plus its issued FLOPs at the ~80 TFLOP/s the fused plan sustains (one call
// = 100 us + 0.107 us per row on U11L-128).  The plan only has to be sensible, not exact: it decides pass boundaries, never results.
hg::PassModel pass_model(const hg_flow* f, const Replica& rep, size_t row_src, size_t row_wire, bool narrowing, int workers, int64_t max_pass_rows, bool direct) {
    hg::PassModel m;
    const double pack_Bpus = narrowing ? std::min(250e3, 20e3 * workers) : std::min(100e3, 10e3 * workers);      // bytes per microsecond
    m.arrive_us_per_row = std::max((double)row_src / pack_Bpus, (double)row_wire / (direct ? 42e3 : 50e3));
    m.lat_us = direct ? 10 : 30;
    m.c0_us = 8.0 * rep.exec->n_stages();
    const int64_t padded = rep.exec->padded_flops_per_row();
    m.c1_us_per_row = rep.exec->plan_kind() == HG_PLAN_FUSED ? (double)(padded > 0 ? padded : f->flops) / 80e6 : (double)f->flops / 1e6;
    m.max_pass_rows = max_pass_rows;
    static const char* price = getenv("HIGSFA_PASS_PRICE");      // experiments: microseconds one more pass must save (hg_hostpipe.hpp)
    if (price) m.per_pass_us = atof(price);
    return m;
}

// HIGSFA_PASSES=a,b,c (experiments): explicit pass sizes for calls of exactly a + b + c rows
std::vector<int64_t> passes_from_env(int64_t n, int64_t max_pass_rows) {
    std::vector<int64_t> out;
    static const char* e = getenv("HIGSFA_PASSES");
    if (!e) return out;
    int64_t sum = 0;
    for (const char* p = e; *p;) {
        char* q = nullptr;
        const long long v = strtoll(p, &q, 10);
        if (q == p || v <= 0 || v > max_pass_rows) return {};
        out.push_back(v);
        sum += v;
        p = *q == ',' ? q + 1 : q;
    }
    if (sum != n) out.clear();
    return out;
}

// Host rows -> device -> host rows through one replica (hg_hostpipe.hpp has the picture).  Values that are all integers 0..255
// (what images_asarray produces, face_analysis.py:786) cross PCIe as uint8 after an exact narrowing; the first row that is
// anything else switches the rest of the call to the caller's own type.
void run_host_rows_impl(hg_flow* f, Replica& rep, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                        int64_t ldy) {
    const size_t xs = hg::dtype_size(x_dtype), ys = hg::dtype_size(y_dtype);
    const int64_t in_dim = f->root->in_dim;
    const size_t row_src = (size_t)ldx * xs, row_given = (size_t)in_dim * xs, row_u8 = (size_t)in_dim;
    bool narrow = x_dtype != HG_U8 && f->narrow;
    // pass buffers: 16 MiB of wire bytes each (1024 uint8 rows of a 128x128 net), at least 16 rows of the caller's type;
    // pinned ring: the call's wire bytes up to 64 MiB, at least 64 rows
    const size_t pass_bytes = std::max<size_t>((size_t)16 << 20, 16 * row_given);
    auto pass_rows_for = [&](size_t row_wire) { return std::max<int64_t>(16, std::min<int64_t>(65536, (int64_t)(pass_bytes / row_wire) / 16 * 16)); };
    const int64_t widest_pass = std::min<int64_t>((n + 15) / 16 * 16, pass_rows_for(narrow ? row_u8 : row_given));
    // Large-BAR devices (every MI355X of this pool): the packers store wire rows straight into the pass's input buffer in HBM —
    // tools/ubench/bar_write_bw.cpp: 42 - 44 GB/s from four or more threads of either socket, a kernel launched afterwards sees
    // the bytes also in a buffer an earlier kernel has read — and there is no pinned ring and no copy queue: in the staged form
    // the DMA engine ran at 25 - 45 GB/s beside the packers' memory traffic and trailed them by up to 0.4 ms at the end of a call.
    // ... and only into buffers that the runtime confirms are mapped for the host at the device's own address (host_can_store)
    for (int b = 0; b < NB; ++b) {
        rep.dx[b].alloc(std::min<size_t>(pass_bytes, (size_t)((n + 15) / 16 * 16) * row_given));
    }
    const bool direct = rep.large_bar && f->direct && rep.inputs_host_writable();
    rep.last_transport = direct ? 1 : 0;
    const size_t ring_want = direct ? 0 : std::max<size_t>(64 * row_given, std::min<size_t>((size_t)64 << 20, (size_t)n * row_given));
    rep.need_streams();
    rep.need_pinned(ring_want, (size_t)widest_pass * y_cols * ys);
    rep.exec->reserve(widest_pass);

    HostPool* pool = rep.pool ? rep.pool.get() : &HostPool::get();
    const bool inline_pack = (size_t)n * row_given < ((size_t)1 << 20);      // under 1 MiB: the calling thread packs, nobody is woken
    if (!inline_pack) {
        const int a = node_of_address(x), b = node_of_address((const char*)x + (size_t)(n - 1) * row_src + row_given - 1);
        pool->bind_to_node(a == b ? a : -1);
    }
    HipSink sink(f, rep, y, y_dtype, y_cols, ldy, in_dim);
    static const bool want_trace = getenv("HIGSFA_HOST_TRACE") != nullptr;
    hg::PipeTrace trace;
    trace.t0 = std::chrono::steady_clock::now();
    if (want_trace) {
        sink.timed = true;
        sink.stamp();
    }
    for (int64_t r0 = 0; r0 < n;) {
        hg::PipeJob J;
        J.x = (const char*)x + (size_t)r0 * row_src;
        J.elem = (int)xs;
        J.n = n - r0;
        J.ldx = ldx;
        J.in_dim = in_dim;
        J.narrow = narrow;
        J.ring = (uint8_t*)rep.ring;
        J.ring_bytes = rep.ring_bytes;
        const size_t row_wire = narrow ? row_u8 : row_given;
        const int64_t max_pass = pass_rows_for(row_wire);
        J.passes = passes_from_env(J.n, max_pass);
        if (J.passes.empty()) J.passes = hg::plan_passes(J.n, pass_model(f, rep, row_src, row_wire, narrow, inline_pack ? 1 : pool->size(), max_pass, direct));
        for (int64_t pr : J.passes)      // the buffers below were sized for widest_pass rows: a wider pass would be written past them
            if (pr > max_pass || (size_t)pr * row_wire > rep.dx[0].bytes || (size_t)pr * y_cols * ys > rep.hy_bytes)
                hg::fail(HG_ERR_ARG, "internal: planned pass of %lld rows exceeds the pass buffers (%lld rows)", (long long)pr, (long long)max_pass);
        static const char* pk = getenv("HIGSFA_PIECE_KIB");      // experiments: "min,max" KiB of wire bytes per copy
        size_t piece_lo = (size_t)1 << 20, piece_hi = (size_t)8 << 20;
        if (pk) {
            unsigned long a = 0, b = 0;
            if (sscanf(pk, "%lu,%lu", &a, &b) == 2 && a > 0 && b >= a) piece_lo = a << 10, piece_hi = b << 10;
        }
        J.piece_min = std::max<int64_t>(1, (int64_t)(piece_lo / row_wire));
        J.piece_max = std::max<int64_t>(J.piece_min, (int64_t)(piece_hi / row_wire));
        J.copy_us_per_row = (double)row_wire / 57e3;
        static const char* tkb = getenv("HIGSFA_TICKET_KIB");      // experiments
        J.ticket_bytes = tkb ? (size_t)atol(tkb) << 10 : narrow ? (size_t)512 << 10 : (size_t)128 << 10;
        if (direct && !narrow) J.max_workers = 6;      // copying into device memory: the link is full with four to six writers
        if (direct) {
            J.direct_slots = NB;
            for (size_t p = 0; p < J.passes.size(); ++p) J.pass_dst.push_back((uint8_t*)rep.dx[(sink.pass_rows.size() + p) % NB].p);
        }
        sink.wire_dtype = narrow ? HG_U8 : x_dtype;
        sink.row_wire = row_wire;
        sink.pass_base = (int)sink.pass_rows.size();
        sink.row_base = r0;
        if (want_trace) J.trace = &trace;
        const hg::PipeResult res = hg::run_pipe(J, sink, pool, inline_pack);
        r0 += res.rows_done;
        if (!res.narrow_failed) break;
        narrow = false;      // a value that is not an integer 0..255: the rest of the call travels in the caller's type
        for (int c = 0; c < NC; ++c) HG_HIP(hipStreamSynchronize(rep.copy[c]));      // pieces of the abandoned pass must not land after the new ones
        // ... and the passes launched so far are taken home before the wide part starts: its first passes write the input buffers
        // straight away (the pipe assumes the buffers of its first NB passes free), and those may still be read by kernels in flight
        sink.unpack_upto((int)sink.pass_rows.size());
    }
    const float t_sub = want_trace ? trace.now() : 0.f;
    sink.unpack_upto((int)sink.pass_rows.size());
    rep.exec->check_errors();      // everything has completed: a poll that ran out in one of the kernels fails THIS call
    if (want_trace) {      // the LAST invocation's timeline (a call that fell through to the wide type shows its wide part)
        float pack_end = 0, pack_first = 1e30f;
        for (float t : trace.ticket_done) {
            pack_end = std::max(pack_end, t);
            if (t >= 0) pack_first = std::min(pack_first, t);
        }
        fprintf(stderr, "[host trace] memory nodes: caller rows %d, pinned ring %d, device %d; packers %d; %s\n", node_of_address(x), rep.ring ? node_of_address(rep.ring) : -1,
                node_of_device(rep.device), pool->size(), direct ? "direct stores into device memory (hsa_amd_pointer_info: host-mapped)" : "pinned ring + copy queues");
        fprintf(stderr, "[host trace] n %lld dtype %d: %zu tickets, first done %.0f us, last done %.0f us; submitted %.0f us; call %.0f us\n", (long long)n, x_dtype,
                trace.ticket_done.size(), pack_first, pack_end, t_sub, trace.now());
        size_t pi = 0;
        for (size_t k = 0; k < trace.passes.size(); ++k) {
            fprintf(stderr, "[host trace]   pass %zu (%lld rows): pieces", k, (long long)trace.passes[k].rows);
            for (; pi < trace.pieces.size() && trace.pieces[pi].pass == (int)k; ++pi)
                fprintf(stderr, " %lld@%.0f+%.0f", (long long)trace.pieces[pi].rows, trace.pieces[pi].t_begin, trace.pieces[pi].t_end - trace.pieces[pi].t_begin);
            float g0 = 0, g1 = 0;
            if (sink.t_ev.size() >= 2 * k + 3) {
                (void)hipEventElapsedTime(&g0, sink.t_ev[0], sink.t_ev[2 * k + 1]);
                (void)hipEventElapsedTime(&g1, sink.t_ev[0], sink.t_ev[2 * k + 2]);
            }
            fprintf(stderr, "; launch call %.0f .. %.0f us; kernels on the device %.0f .. %.0f us\n", trace.passes[k].t_begin, trace.passes[k].t_end, g0 * 1e3f, g1 * 1e3f);
        }
    }
}

// A call that fails half-way (a HIP error, a launch refused) must not leave copies and kernels in flight on the replica's
// staging buffers: the next call on the replica reuses them from the start without waiting for an event, and need_pinned may
// free and reallocate them.  Drain both streams before the error leaves.
void run_host_rows(hg_flow* f, Replica& rep, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                   int64_t ldy) {
    try {
        run_host_rows_impl(f, rep, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy);
    } catch (...) {
        for (int c = 0; c < NC; ++c)
            if (rep.copy[c]) (void)hipStreamSynchronize(rep.copy[c]);
        if (rep.compute) (void)hipStreamSynchronize(rep.compute);
        throw;
    }
}

}  // namespace

extern "C" {

int hg_version(void) { return HG_VERSION; }

const char* hg_last_error(void) { return g_last_error.c_str(); }

int hg_device_count(int* count) {
    return guarded([&] {
        if (!count) hg::fail(HG_ERR_ARG, "null count pointer");
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
        *count = c;
    });
}

int hg_flow_load(const void* blob, size_t nbytes, int flags, hg_flow** out) {
    return guarded([&] {
        if (!out) hg::fail(HG_ERR_ARG, "null output handle pointer");
        *out = nullptr;
        auto f = std::make_unique<hg_flow>();
        f->root = hg::parse_blob(blob, nbytes);
        f->flops = hg::tree_flops(*f->root);
        f->force_generic = flags & 1;
        f->narrow = getenv("HIGSFA_NO_NARROW") == nullptr;
        f->direct = !(getenv("HIGSFA_HOST_DIRECT") && atoi(getenv("HIGSFA_HOST_DIRECT")) == 0);
        if (!(flags & 1)) f->exec = hg::make_fused_executor(*f->root, &f->fused_reject);
        else f->fused_reject = "generic plan forced by caller";
        if (!f->exec) f->exec = hg::make_generic_executor(*f->root);
        *out = f.release();
    });
}

void hg_flow_free(hg_flow* f) { delete f; }

int hg_flow_info(const hg_flow* f, hg_info* info) {
    return guarded([&] {
        if (!f || !info) hg::fail(HG_ERR_ARG, "null argument");
        info->input_dim = f->root->in_dim;
        info->output_dim = f->root->out_dim;
        info->n_top_nodes = (int32_t)f->root->ch.size();
        info->plan_kind = f->exec->plan_kind();
        info->n_stages = f->exec->n_stages();
        info->device = f->device;
        info->weight_bytes = f->exec->weight_bytes();
        info->flops_per_row = f->flops;
        info->padded_flops_per_row = f->exec->padded_flops_per_row();
        info->workspace_bytes = f->exec->workspace_bytes();
    });
}

int hg_flow_describe(const hg_flow* f, char* buf, size_t cap, size_t* needed) {
    return guarded([&] {
        if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
        std::string s = "flow: " + std::to_string(f->root->in_dim) + " -> " + std::to_string(f->root->out_dim) + ", " +
                        std::to_string(f->root->ch.size()) + " top-level nodes\n";
        for (size_t i = 0; i < f->root->ch.size(); ++i) {
            const hg::TNode& c = *f->root->ch[i];
            s += "  node " + std::to_string(i) + ": " + hg::kind_name(c.kind) + " " + std::to_string(c.in_dim) + " -> " +
                 std::to_string(c.out_dim);
            if (c.kind == hg::K_LAYER) s += " (" + std::to_string(c.ch.size()) + " nodes)";
            if (c.kind == hg::K_CLONELAYER) s += " (" + std::to_string(c.aux) + " clones)";
            s += "\n";
        }
        s += f->exec->describe();
        if (!f->fused_reject.empty()) s += "fused plan not used: " + f->fused_reject + "\n";
        if (needed) *needed = s.size() + 1;
        if (buf && cap) {
            size_t m = std::min(cap - 1, s.size());
            memcpy(buf, s.data(), m);
            buf[m] = 0;
        }
    });
}

int hg_flow_to_device(hg_flow* f, int device) {
    return guarded([&] {
        if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
        if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range (0..%d)", device, count - 1);
        if (f->device >= 0 && f->device != device) hg::fail(HG_ERR_STATE, "flow already lives on device %d", f->device);
        if (f->device == device) return;
        f->main.create(device);
    });
}

int hg_flow_reserve(hg_flow* f, int64_t max_rows) {
    return guarded([&] {
        if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
        if (max_rows < 0) hg::fail(HG_ERR_ARG, "negative row count");
        f->need_device();
        f->set_device();
        f->exec->reserve(max_rows);
    });
}

int hg_flow_execute_device(hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype,
                           int64_t y_cols, int64_t ldy, void* stream) {
    return guarded([&] {
        check_exec_args(f, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy);
        f->need_device();
        f->set_device();
        run_on_device(f, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy, (hipStream_t)stream);
    });
}

int hg_flow_execute(hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                    int64_t ldy) {
    return guarded([&] {
        check_exec_args(f, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy);
        f->need_device();
        f->set_device();
        if (n == 0) return;
        run_host_rows(f, f->main, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy);
    });
}

int hg_flow_execute_sharded(hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                            int64_t ldy, const int* devices, int n_devices) {
    return guarded([&] {
        check_exec_args(f, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy);
        if (n_devices <= 0 || n_devices > 64) hg::fail(HG_ERR_ARG, "n_devices %d outside 1..64", n_devices);
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
        std::vector<int> devs(n_devices);
        for (int r = 0; r < n_devices; ++r) {
            devs[r] = devices ? devices[r] : r;
            if (devs[r] < 0 || devs[r] >= count) hg::fail(HG_ERR_DEVICE, "shard %d: device %d out of range (0..%d)", r, devs[r], count - 1);
        }
        if (n == 0) return;
        // creating / destroying replicas moves the CALLING thread's current device: put it back when this call ends, so that
        // the caller's later allocations and launches (torch) stay where they were
        struct DeviceRestore {
            int dev = -1;
            DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
            ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
        } restore_device;
        // replicas: one executor + streams + staging per listed device (a device may be listed more than once)
        if ((int)f->shards.size() > n_devices) {
            for (size_t r = n_devices; r < f->shards.size(); ++r) f->shards[r]->destroy();
            f->shards.resize(n_devices);
        }
        while ((int)f->shards.size() < n_devices) f->shards.emplace_back(new Replica());
        for (int r = 0; r < n_devices; ++r) {
            Replica& rep = *f->shards[r];
            if (rep.device == devs[r]) continue;
            rep.destroy();
            std::string why;
            if (!f->force_generic) rep.exec = hg::make_fused_executor(*f->root, &why);
            if (!rep.exec) rep.exec = hg::make_generic_executor(*f->root);
            rep.create(devs[r]);
        }
        // several blocks pack at the same time: each replica gets its share of the usable threads as a pool of its own (the
        // process-wide pool runs one region at a time); a single block uses the process-wide pool
        const int share = std::max(1, std::min(hg::usable_cpus(), 16) / n_devices);
        for (int r = 0; r < n_devices; ++r) {
            Replica& rep = *f->shards[r];
            if (n_devices == 1) rep.pool.reset();
            else if (!rep.pool || rep.pool->size() != share) rep.pool.reset(new HostPool(share));
        }
        // contiguous row blocks of ceil(n / n_devices) rows (pyfaceanalysis_amd/sharded.py shard_bounds), one host thread per
        // block; every block lands in the caller's y at its own rows, which IS the gather (host memory, no peer copy)
        const int64_t per = (n + n_devices - 1) / n_devices;
        const size_t xs = hg::dtype_size(x_dtype), ys = hg::dtype_size(y_dtype);
        std::vector<std::string> errs(n_devices);
        std::vector<int> codes(n_devices, HG_OK);
        std::vector<std::thread> th;
        for (int r = 0; r < n_devices; ++r) {
            const int64_t lo = std::min<int64_t>((int64_t)r * per, n), hi = std::min<int64_t>(lo + per, n);
            if (hi <= lo) continue;
            th.emplace_back([&, r, lo, hi] {
                try {
                    Replica& rep = *f->shards[r];
                    HG_HIP(hipSetDevice(rep.device));
                    run_host_rows(f, rep, (const char*)x + (size_t)lo * ldx * xs, x_dtype, hi - lo, ldx, (char*)y + (size_t)lo * ldy * ys,
                                  y_dtype, y_cols, ldy);
                } catch (const hg::Error& e) {
                    codes[r] = e.code;
                    errs[r] = e.what();
                } catch (const std::exception& e) {
                    codes[r] = HG_ERR_STATE;
                    errs[r] = e.what();
                }
            });
        }
        for (auto& t : th) t.join();
        for (int r = 0; r < n_devices; ++r)
            if (codes[r] != HG_OK) hg::fail(codes[r], "shard %d (device %d): %s", r, devs[r], errs[r].c_str());
    });
}

int hg_event_create(void** ev) {
    return guarded([&] {
        if (!ev) hg::fail(HG_ERR_ARG, "null event pointer");
        hipEvent_t e = nullptr;
        HG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
        *ev = (void*)e;
    });
}

int hg_event_create_on(void** ev, int device, int device_scope) {
    return guarded([&] {
        if (!ev) hg::fail(HG_ERR_ARG, "null event pointer");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) hg::fail(HG_ERR_DEVICE, "no HIP device available");
        if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range (0..%d)", device, count - 1);
        int cur = -1;
        HG_HIP(hipGetDevice(&cur));
        HG_HIP(hipSetDevice(device));      // an event belongs to the device that is current when it is created
        hipEvent_t e = nullptr;
        const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming | (device_scope ? hipEventDisableSystemFence : 0));
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
        HG_HIP(rc);
        *ev = (void*)e;
    });
}

void hg_event_destroy(void* ev) {
    if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}

int hg_event_record(void* ev, void* stream) {
    return guarded([&] {
        if (!ev) hg::fail(HG_ERR_ARG, "null event");
        HG_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    });
}

int hg_stream_wait_event(void* stream, void* ev) {
    return guarded([&] {
        if (!ev) hg::fail(HG_ERR_ARG, "null event");
        HG_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    });
}

int hg_event_query(void* ev) {
    int done = 0;
    const int rc = guarded([&] {
        if (!ev) hg::fail(HG_ERR_ARG, "null event");
        const hipError_t e = hipEventQuery((hipEvent_t)ev);
        if (e == hipSuccess) done = 1;
        else if (e == hipErrorNotReady) (void)hipGetLastError();      // not an error: clear the sticky code
        else HG_HIP(e);
    });
    return rc == HG_OK ? done : rc;
}

// ---- ceilings of the host path, MEASURED in the caller's process on the caller's box (bench.py host_path leg) ----
// (1) the packers alone: the same pool, placement, ticket size and narrowing / copy routines as hg_flow_execute, destinations in a
//     per-ticket scratch that stays in cache — the rate at which this host reads THIS array; (2) host stores into device memory
//     (only where hg_flow_execute would use them: large BAR + host_can_store); (3) a pinned-memory copy engine transfer.
int hg_host_pack_probe(const void* x, int x_dtype, int64_t n, int64_t ldx, int64_t in_dim, int reps, double* best_seconds) {
    return guarded([&] {
        if (!x || n <= 0 || in_dim <= 0 || ldx < in_dim || !best_seconds || reps < 1) hg::fail(HG_ERR_ARG, "bad argument");
        if (x_dtype != HG_U8 && x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "bad dtype");
        const size_t xs = hg::dtype_size(x_dtype), row_src = (size_t)ldx * xs;
        HostPool* pool = &HostPool::get();
        {
            const int a = node_of_address(x), b = node_of_address((const char*)x + (size_t)(n - 1) * row_src + (size_t)in_dim * xs - 1);
            pool->bind_to_node(a == b ? a : -1);
        }
        const size_t ticket_bytes = x_dtype == HG_U8 ? (size_t)128 << 10 : (size_t)512 << 10;      // as run_host_rows_impl
        const int64_t ticket_rows = std::max<int64_t>(1, (int64_t)(ticket_bytes / ((size_t)in_dim * xs)));
        const int n_tickets = (int)((n + ticket_rows - 1) / ticket_rows);
        std::atomic<int> bad{0};
        const std::function<void(int)> ticket = [&](int t) {
            thread_local std::vector<uint8_t> scratch;
            if (scratch.size() < (size_t)in_dim * xs) scratch.resize((size_t)in_dim * xs);
            const int64_t a = (int64_t)t * ticket_rows, e = std::min(n, a + ticket_rows);
            const char* src = (const char*)x + (size_t)a * row_src;
            for (int64_t r = a; r < e; ++r, src += row_src) {
                bool ok = true;
                if (x_dtype == HG_F64) ok = hg::narrow_row_f64((const double*)src, scratch.data(), in_dim);
                else if (x_dtype == HG_F32) ok = hg::narrow_row_f32((const float*)src, scratch.data(), in_dim);
                else memcpy(scratch.data(), src, (size_t)in_dim);
                if (!ok) bad.fetch_add(1, std::memory_order_relaxed);
            }
        };
        double best = 1e30;
        for (int rep = 0; rep < reps + 1; ++rep) {      // one untimed round first
            const auto t0 = std::chrono::steady_clock::now();
            pool->parallel_for(n_tickets, ticket);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0) best = std::min(best, dt);
        }
        *best_seconds = best;
    });
}

int hg_host_store_probe(int device, size_t bytes, int reps, double* best_seconds, int* direct) {
    return guarded([&] {
        if (!best_seconds || !direct || bytes < ((size_t)1 << 20) || reps < 1) hg::fail(HG_ERR_ARG, "bad argument");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "no such device");
        HG_HIP(hipSetDevice(device));
        int lb = 0;
        const bool large_bar = hipDeviceGetAttribute(&lb, hipDeviceAttributeIsLargeBar, device) == hipSuccess && lb != 0;
        hg::DevBuf dst;
        dst.alloc(bytes);
        *direct = (large_bar && host_can_store(dst.p, bytes)) ? 1 : 0;
        *best_seconds = 0.0;
        if (!*direct) return;
        std::vector<uint8_t> src(bytes);
        for (size_t i = 0; i < bytes; i += 4096) src[i] = (uint8_t)(i >> 12);
        HostPool* pool = &HostPool::get();
        pool->bind_to_node(node_of_address(src.data()));
        const size_t piece = (size_t)128 << 10;
        const int n_tickets = (int)((bytes + piece - 1) / piece);
        const std::function<void(int)> ticket = [&](int t) {
            const size_t o = (size_t)t * piece, m = std::min(piece, bytes - o);
            hg::stream_copy((uint8_t*)dst.p + o, src.data() + o, m);
            hg::store_fence();
        };
        double best = 1e30;
        for (int rep = 0; rep < reps + 1; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            pool->begin(n_tickets, ticket, 6);      // the writers run_host_rows_impl uses for rows that are only copied
            pool->end();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0) best = std::min(best, dt);
        }
        *best_seconds = best;
    });
}

int hg_host_dma_probe(int device, size_t bytes, int reps, double* best_seconds) {
    return guarded([&] {
        if (!best_seconds || bytes < ((size_t)1 << 20) || reps < 1) hg::fail(HG_ERR_ARG, "bad argument");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "no such device");
        HG_HIP(hipSetDevice(device));
        hg::DevBuf dst;
        dst.alloc(bytes);
        void* src = nullptr;
        HG_HIP(hipHostMalloc(&src, bytes, hipHostMallocDefault));
        memset(src, 1, bytes);
        hipStream_t st = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        double best = 1e30;
        for (int rep = 0; rep < reps + 1 && e == hipSuccess; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            e = hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep > 0) best = std::min(best, dt);
        }
        if (st) (void)hipStreamDestroy(st);
        (void)hipHostFree(src);
        HG_HIP(e);
        *best_seconds = best;
    });
}

int hg_flow_host_transport(const hg_flow* f, int* transport) {
    return guarded([&] {
        if (!f || !transport) hg::fail(HG_ERR_ARG, "null argument");
        *transport = f->main.last_transport;
    });
}

int hg_flow_set_profiling(hg_flow* f, int enabled) {
    return guarded([&] {
        if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
        f->profiling = enabled != 0;
    });
}

int hg_flow_stage_times(hg_flow* f, double* total_ms, int64_t* launches, int cap, int* n_stages) {
    return guarded([&] {
        if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
        const int ns = f->exec->n_stages();
        if (n_stages) *n_stages = ns;
        for (int i = 0; i < ns && i < cap; ++i) {
            bool have = i < (int)f->prof.size();
            if (total_ms) total_ms[i] = have ? f->prof[i].total_ms : 0.0;
            if (launches) launches[i] = have ? f->prof[i].launches : 0;
        }
    });
}

int hg_flow_stage_name(const hg_flow* f, int stage, char* buf, size_t cap) {
    return guarded([&] {
        if (!f || !buf || !cap) hg::fail(HG_ERR_ARG, "null argument");
        if (stage < 0 || stage >= f->exec->n_stages()) hg::fail(HG_ERR_ARG, "stage %d out of range", stage);
        std::string s = f->exec->stage_name(stage);
        size_t m = std::min(cap - 1, s.size());
        memcpy(buf, s.data(), m);
        buf[m] = 0;
    });
}

int hg_flow_reset_profile(hg_flow* f) {
    return guarded([&] {
        if (!f) hg::fail(HG_ERR_ARG, "null flow handle");
        f->prof.clear();
    });
}

}  // extern "C"
