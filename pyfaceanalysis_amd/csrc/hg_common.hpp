// Shared host-side definitions: error reporting, the parsed flow tree, the executor interface.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/higsfa.h"

namespace hg {

// ---- errors ---------------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define HG_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            ::hg::fail(HG_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                       __FILE__, __LINE__);                                                \
    } while (0)

// ---- parsed flow tree (mirrors pyfaceanalysis_amd/blob.py) -----------------------------------
enum NodeKind : uint32_t {
    K_FLOW = 1, K_SWITCHBOARD = 2, K_LAYER = 3, K_CLONELAYER = 4, K_AFFINE = 5, K_EXPANSION = 6,
    K_IGSFA = 7, K_IDENTITY = 8, K_HEAD = 9, K_CUTOFF = 10, K_FLOWNODE = 11
};
enum ExpKind : uint32_t { E_IDENTITY = 0, E_ABS_POW = 1, E_SIGNED_POW = 2, E_QUADRATIC = 3, E_PAIR_ADJ = 4, E_PAIR_BAND = 5 };
// E_PAIR_ADJ / E_PAIR_BAND: the two readings of cuicuilco's pair_prodsadj{k}_ex (nodes.pair_prodsadj_ex): x_i x_{i+k} only, or the
// reflexive band of offsets 0 .. k-1 stacked offset-major (squares first)

struct ExpFunc {
    uint32_t kind, sel, k;
    double expo;
    // columns the function reads: the first `sel` (cuicuilco's sel_exp(n, f) slices x[:, :n], which numpy clamps), 0 = all.
    // Compared unsigned: a huge sel clamps to d instead of turning negative.
    int used(int d) const { return sel > 0 && sel < (uint32_t)d ? (int)sel : d; }
    int out_dim(int d) const {
        int u = used(d);
        if (kind <= E_SIGNED_POW) return u;
        if (kind == E_QUADRATIC) return u * (u + 1) / 2;
        if (kind == E_PAIR_BAND) {
            int w = 0;
            for (int off = 0; off < (int)k && off < u; ++off) w += u - off;
            return w;
        }
        return u - (int)k > 0 ? u - (int)k : 0;
    }
};

struct TNode {
    uint32_t kind = 0, in_dim = 0, out_dim = 0, aux = 0;
    std::vector<std::unique_ptr<TNode>> ch;  // FLOW / LAYER / FLOWNODE children; CLONELAYER: one
    std::vector<int32_t> conn;               // SWITCHBOARD
    std::vector<double> a, W, b;             // AFFINE: y = (x - a) W + b, W row-major in x out
    std::vector<ExpFunc> funcs;              // EXPANSION
    std::vector<double> x_mean, magn;        // IGSFA
    std::unique_ptr<TNode> exp_node, sfa, lr, pca;
    double lo = 0, hi = 0;                   // CUTOFF
};

std::unique_ptr<TNode> parse_blob(const void* blob, size_t nbytes);
int64_t tree_flops(const TNode& n);
const char* kind_name(uint32_t kind);

// ---- executor interface ----------------------------------------------------------------------
struct StageProfile {
    std::string name;
    double total_ms = 0;
    int64_t launches = 0;
};

class Executor {
public:
    virtual ~Executor() {}
    virtual int plan_kind() const = 0;
    virtual int n_stages() const = 0;
    virtual std::string stage_name(int i) const = 0;
    virtual std::string describe() const = 0;
    virtual int64_t weight_bytes() const = 0;
    virtual int64_t padded_flops_per_row() const { return 0; }
    virtual int64_t workspace_bytes() const = 0;
    virtual void to_device() = 0;                 // current device already set
    virtual void reserve(int64_t rows) = 0;
    // Enqueue on `stream`.  When ev != nullptr it holds n_stages()+1 events to record around stages.
    virtual void run(const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype,
                     int64_t y_cols, int64_t ldy, hipStream_t stream, hipEvent_t* ev) = 0;
    virtual void check_errors() {}                // after a synchronisation: did a kernel of an earlier run report a failure?
    virtual void release() = 0;                   // free device memory
};

std::unique_ptr<Executor> make_generic_executor(const TNode& root);
// Returns nullptr (and a reason) when the flow does not have the regular structure the fused
// MFMA kernels need; the caller then falls back to the generic executor.
std::unique_ptr<Executor> make_fused_executor(const TNode& root, std::string* why_not);

inline size_t dtype_size(int dt) {
    switch (dt) {
        case HG_U8: return 1;
        case HG_F32: return 4;
        case HG_F64: return 8;
        default: fail(HG_ERR_ARG, "unknown dtype %d", dt);
    }
}

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void alloc(size_t n) {
        if (n <= bytes && p) return;
        free();
        if (n == 0) return;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            bytes = 0;
            fail(HG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
    }
    void upload(const void* src, size_t n) {
        alloc(n);
        if (n) HG_HIP(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
    }
    void free() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { free(); }
    DevBuf() = default;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

}  // namespace hg
