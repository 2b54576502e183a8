// C ABI (include/higsfa.h): flow handle, host/device execute, profiling.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "hg_common.hpp"
#include "hg_hostpool.hpp"

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

}  // namespace

namespace hg {
// hg_hostpack.cpp (plain C++, built with g++ so that it can carry AVX2 clones): row of wide values -> uint8 when every
// value is an integer 0..255; false otherwise.
bool narrow_row_f64(const double* src, uint8_t* dst, int64_t n);
bool narrow_row_f32(const float* src, uint8_t* dst, int64_t n);
}  // namespace hg

namespace {

// One execution context: the executor on one device with its streams and staging buffers.
struct Replica {
    int device = -1;
    std::unique_ptr<hg::Executor> exec;
    hipStream_t compute = nullptr, copy = nullptr;
    void* hx[2] = {nullptr, nullptr};      // pinned host staging, two slots
    void* hy[2] = {nullptr, nullptr};
    size_t hx_bytes = 0, hy_bytes = 0;
    hg::DevBuf dx[2], dy[2];
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};

    void create(int dev) {
        HG_HIP(hipSetDevice(dev));
        exec->to_device();
        HG_HIP(hipStreamCreateWithFlags(&compute, hipStreamNonBlocking));
        HG_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HG_HIP(hipEventCreateWithFlags(&ev_h2d[b], hipEventDisableTiming));
            HG_HIP(hipEventCreateWithFlags(&ev_out[b], hipEventDisableTiming));
        }
        HG_HIP(hipDeviceSynchronize());
        device = dev;
    }
    void need_pinned(size_t xb, size_t yb) {
        if (xb > hx_bytes) {
            for (int b = 0; b < 2; ++b) {
                if (hx[b]) (void)hipHostFree(hx[b]);
                hx[b] = nullptr;
                HG_HIP(hipHostMalloc(&hx[b], xb, hipHostMallocDefault));
            }
            hx_bytes = xb;
        }
        if (yb > hy_bytes) {
            for (int b = 0; b < 2; ++b) {
                if (hy[b]) (void)hipHostFree(hy[b]);
                hy[b] = nullptr;
                HG_HIP(hipHostMalloc(&hy[b], yb, hipHostMallocDefault));
            }
            hy_bytes = yb;
        }
    }
    void destroy() {
        if (device < 0 || hipSetDevice(device) != hipSuccess) return;
        for (int b = 0; b < 2; ++b) {
            if (hx[b]) (void)hipHostFree(hx[b]);
            if (hy[b]) (void)hipHostFree(hy[b]);
            if (ev_h2d[b]) (void)hipEventDestroy(ev_h2d[b]);
            if (ev_out[b]) (void)hipEventDestroy(ev_out[b]);
            dx[b].free();
            dy[b].free();
            hx[b] = hy[b] = nullptr;
        }
        if (compute) (void)hipStreamDestroy(compute);
        if (copy) (void)hipStreamDestroy(copy);
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


// Host rows -> device -> host rows through one replica: chunks of rows are packed into pinned staging slot b
// (narrowed to uint8 when every value is an integer 0..255), copied on the copy stream while the previous chunk's
// kernels run on the compute stream, and the features come back through a pinned slot as well.
//   slot reuse: hx[b] after ev_h2d[b] (its H2D done); dx[b] / dy[b] / hy[b] after ev_out[b] (kernels + D2H of the
//   chunk that used the slot done).
void run_host_rows_impl(hg_flow* f, Replica& rep, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                        int64_t ldy, bool use_pool) {
    const size_t xs = hg::dtype_size(x_dtype), ys = hg::dtype_size(y_dtype);
    const int64_t in_dim = f->root->in_dim;
    // ~32 MiB of caller bytes per chunk (>= 256 rows): small enough to pipeline a 4096-row float64 batch in 16 chunks,
    // large enough that a chunk's kernels are past their launch-latency floor
    // uint8 rows cannot be narrowed and need no packing: they go straight from the caller's memory (the runtime stages pageable
    // memory itself; a second host copy into our pinned slot would only add a pass over the data) in 16 MiB chunks, so that the
    // copy of chunk i+1 still overlaps the kernels of chunk i
    const bool direct = x_dtype == HG_U8;
    int64_t chunk = std::max<int64_t>(256, ((direct ? 16ll : 32ll) << 20) / (in_dim * (int64_t)xs));
    chunk = std::min(chunk, n);
    chunk = (chunk + 15) / 16 * 16;
    const size_t x_slot = (size_t)chunk * in_dim * xs, y_slot = (size_t)chunk * y_cols * ys;
    rep.need_pinned(direct ? 0 : x_slot, y_slot);
    for (int b = 0; b < 2; ++b) {
        rep.dx[b].alloc(x_slot);
        rep.dy[b].alloc(y_slot);
    }
    rep.exec->reserve(chunk);
    const bool try_narrow = x_dtype != HG_U8 && f->narrow;
    HostPool& pool = HostPool::get();
    const int64_t n_chunks = (n + chunk - 1) / chunk;
    auto unpack = [&](int64_t ci) {      // features of chunk ci: pinned slot -> caller rows
        const int b = (int)(ci & 1);
        const int64_t r0 = ci * chunk, m = std::min(chunk, n - r0);
        HG_HIP(hipEventSynchronize(rep.ev_out[b]));
        const char* src = (const char*)rep.hy[b];
        char* dst = (char*)y + (size_t)r0 * ldy * ys;
        const size_t row = (size_t)y_cols * ys;
        if (ldy == y_cols) memcpy(dst, src, row * m);
        else for (int64_t r = 0; r < m; ++r) memcpy(dst + (size_t)r * ldy * ys, src + (size_t)r * row, row);
    };
    for (int64_t ci = 0; ci < n_chunks; ++ci) {
        const int b = (int)(ci & 1);
        const int64_t r0 = ci * chunk, m = std::min(chunk, n - r0);
        if (ci >= 2) {
            unpack(ci - 2);                               // frees hy[b] (and tells us dx[b] / dy[b] are free)
            if (!direct) HG_HIP(hipEventSynchronize(rep.ev_h2d[b]));   // hx[b] has left the host
        }
        // ---- pack (host threads), overlapping the GPU work of chunk ci - 1
        const char* xsrc = (const char*)x + (size_t)r0 * ldx * xs;
        int sent_dtype = x_dtype;
        if (direct) {
            if (ci >= 2) HG_HIP(hipStreamWaitEvent(rep.copy, rep.ev_out[b], 0));
            HG_HIP(hipMemcpy2DAsync(rep.dx[b].p, (size_t)in_dim * xs, xsrc, (size_t)ldx * xs, (size_t)in_dim * xs, (size_t)m, hipMemcpyHostToDevice, rep.copy));
            HG_HIP(hipEventRecord(rep.ev_h2d[b], rep.copy));
        }
        if (try_narrow) {
            std::atomic<int> ok{1};
            const int tasks = (int)std::min<int64_t>(m, use_pool ? 4 * pool.size() : 1);
            auto body = [&](int t) {
                const int64_t a = m * t / tasks, e = m * (t + 1) / tasks;
                for (int64_t r = a; r < e && ok.load(std::memory_order_relaxed); ++r) {
                    const bool good = x_dtype == HG_F64
                                          ? hg::narrow_row_f64((const double*)(xsrc + (size_t)r * ldx * xs), (uint8_t*)rep.hx[b] + (size_t)r * in_dim, in_dim)
                                          : hg::narrow_row_f32((const float*)(xsrc + (size_t)r * ldx * xs), (uint8_t*)rep.hx[b] + (size_t)r * in_dim, in_dim);
                    if (!good) ok.store(0, std::memory_order_relaxed);
                }
            };
            if (use_pool) pool.parallel_for(tasks, body);
            else body(0);
            if (ok.load()) sent_dtype = HG_U8;
        }
        const size_t ss = hg::dtype_size(sent_dtype);
        if (sent_dtype == x_dtype && !direct) {      // as given: rows copied into the pinned slot (strided source allowed)
            const int tasks = (int)std::min<int64_t>(m, use_pool ? 4 * pool.size() : 1);
            auto body = [&](int t) {
                const int64_t a = m * t / tasks, e = m * (t + 1) / tasks;
                if (ldx == in_dim) memcpy((char*)rep.hx[b] + (size_t)a * in_dim * xs, xsrc + (size_t)a * in_dim * xs, (size_t)(e - a) * in_dim * xs);
                else for (int64_t r = a; r < e; ++r) memcpy((char*)rep.hx[b] + (size_t)r * in_dim * xs, xsrc + (size_t)r * ldx * xs, (size_t)in_dim * xs);
            };
            if (use_pool) pool.parallel_for(tasks, body);
            else body(0);
        }
        // ---- copy stream: H2D once the slot's previous consumer is done
        if (!direct) {
            if (ci >= 2) HG_HIP(hipStreamWaitEvent(rep.copy, rep.ev_out[b], 0));
            HG_HIP(hipMemcpyAsync(rep.dx[b].p, rep.hx[b], (size_t)m * in_dim * ss, hipMemcpyHostToDevice, rep.copy));
            HG_HIP(hipEventRecord(rep.ev_h2d[b], rep.copy));
        }
        // ---- compute stream: kernels, features back
        HG_HIP(hipStreamWaitEvent(rep.compute, rep.ev_h2d[b], 0));
        run_on_device(f, rep.dx[b].p, sent_dtype, m, in_dim, rep.dy[b].p, y_dtype, y_cols, y_cols, rep.compute, &rep);
        HG_HIP(hipMemcpyAsync(rep.hy[b], rep.dy[b].p, (size_t)m * y_cols * ys, hipMemcpyDeviceToHost, rep.compute));
        HG_HIP(hipEventRecord(rep.ev_out[b], rep.compute));
    }
    if (n_chunks >= 2) unpack(n_chunks - 2);
    unpack(n_chunks - 1);
    rep.exec->check_errors();      // everything has completed: a poll that ran out in one of the kernels fails THIS call
}

// A call that fails half-way (a HIP error, a launch refused) must not leave copies and kernels in flight on the replica's
// staging slots: the next call on the replica starts at chunk 0 / 1, which reuse the slots without waiting for an event, and
// need_pinned may free and reallocate them.  Drain both streams before the error leaves.
void run_host_rows(hg_flow* f, Replica& rep, const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols,
                   int64_t ldy, bool use_pool) {
    try {
        run_host_rows_impl(f, rep, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy, use_pool);
    } catch (...) {
        if (rep.copy) (void)hipStreamSynchronize(rep.copy);
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
        run_host_rows(f, f->main, x, x_dtype, n, ldx, y, y_dtype, y_cols, ldy, true);
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
                                  y_dtype, y_cols, ldy, n_devices == 1);
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
