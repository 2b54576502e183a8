// C ABI (include/higsfa.h): flow handle, host/device execute, profiling.
#include <algorithm>
#include <mutex>

#include "hg_common.hpp"

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

struct hg_flow {
    std::unique_ptr<hg::TNode> root;
    std::unique_ptr<hg::Executor> exec;
    std::string fused_reject;  // why the fused plan was not chosen ("" if it was)
    int device = -1;
    int64_t flops = 0;
    bool profiling = false;
    std::vector<hipEvent_t> events;
    std::vector<hg::StageProfile> prof;
    hipStream_t own_stream = nullptr;
    hg::DevBuf stage_x, stage_y;

    void need_device() const {
        if (device < 0) hg::fail(HG_ERR_DEVICE, "flow is not on a device: call hg_flow_to_device first");
    }
    void set_device() const { HG_HIP(hipSetDevice(device)); }
    void drop_events() {
        for (auto e : events) (void)hipEventDestroy(e);
        events.clear();
    }
    ~hg_flow() {
        if (device >= 0 && hipSetDevice(device) == hipSuccess) {
            drop_events();
            if (own_stream) (void)hipStreamDestroy(own_stream);
            if (exec) exec->release();
            stage_x.free();
            stage_y.free();
        }
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
                   int64_t ldy, hipStream_t st) {
    if (n == 0) return;
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
        HG_HIP(hipSetDevice(device));
        f->exec->to_device();
        HG_HIP(hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking));
        HG_HIP(hipDeviceSynchronize());
        f->device = device;
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
        const size_t xs = hg::dtype_size(x_dtype), ys = hg::dtype_size(y_dtype);
        const int64_t in_dim = f->root->in_dim;
        // Row chunks of <= 256 MiB of input keep the staging buffers and activation workspace bounded.
        int64_t chunk = std::max<int64_t>(16, (256ll << 20) / (in_dim * (int64_t)xs));
        chunk = std::min(chunk, n);
        chunk = (chunk + 15) / 16 * 16;
        f->stage_x.alloc((size_t)chunk * in_dim * xs);
        f->stage_y.alloc((size_t)chunk * y_cols * ys);
        f->exec->reserve(chunk);
        hipStream_t st = f->own_stream;
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t m = std::min(chunk, n - r0);
            const char* xsrc = (const char*)x + (size_t)r0 * ldx * xs;
            HG_HIP(hipMemcpy2DAsync(f->stage_x.p, (size_t)in_dim * xs, xsrc, (size_t)ldx * xs, (size_t)in_dim * xs, (size_t)m,
                                    hipMemcpyHostToDevice, st));
            run_on_device(f, f->stage_x.p, x_dtype, m, in_dim, f->stage_y.p, y_dtype, y_cols, y_cols, st);
            char* ydst = (char*)y + (size_t)r0 * ldy * ys;
            HG_HIP(hipMemcpy2DAsync(ydst, (size_t)ldy * ys, f->stage_y.p, (size_t)y_cols * ys, (size_t)y_cols * ys, (size_t)m,
                                    hipMemcpyDeviceToHost, st));
            HG_HIP(hipStreamSynchronize(st));
        }
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
