// Generic executor: any flow of the blob vocabulary, lowered to a sequence of column-parallel
// steps on row-major fp32 activations (N x width) in HBM.  Two kernels: an element-wise "map"
// (gather / copy / expansion / clip) and a per-node affine
// (rows a5, a7, and the affine pieces of a8).  This is the fallback for flows whose structure
// the fused MFMA plan (hg_fused.hip) does not cover, and an independent second HIP
// implementation the fused plan is cross-checked against.  Simple rather than fast.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <sstream>

#include "hg_common.hpp"

namespace hg {

namespace {

enum MapKind : int32_t { M_COPY = 0, M_ABS_POW = 1, M_SIGNED_POW = 2, M_PRODUCT = 3, M_CLIP = 4 };

struct MapEntry {
    int32_t i, j, kind;
    float p0, p1, p2;  // COPY: y = x_i - p1 | POW: |x_i - p1|^p0 | PRODUCT: (x_i-p1)(x_j-p2) | CLIP: clamp(x_i, p0, p1)
};

struct AffD {
    int in_off, in_dim, out_off, out_dim;
    std::vector<double> a, W, b;
};

struct Step {
    int in_w = 0, out_w = 0;
    std::vector<int32_t> map_cols;
    std::vector<MapEntry> map;
    std::vector<AffD> aff;
    std::string name;
};

struct Prog {
    int in_dim = 0, out_dim = 0;
    std::vector<Step> steps;
};

void add_copy(Step& s, int out_col, int in_col, float sub = 0.f) {
    s.map_cols.push_back(out_col);
    s.map.push_back(MapEntry{in_col, in_col, M_COPY, 0.f, sub, 0.f});
}

// Expansion entries for input block [in_off, in_off+d) -> output starting at out_off.
// `mean` (may be null): per-input pre-offset (iGSFA expands x - x_mean).
void add_expansion(Step& s, const std::vector<ExpFunc>& funcs, int d, int in_off, int out_off, const double* mean) {
    int o = out_off;
    auto m = [&](int i) { return mean ? (float)mean[i] : 0.f; };
    for (const ExpFunc& f : funcs) {
        int u = f.used(d);
        switch (f.kind) {
            case E_IDENTITY:
                for (int i = 0; i < u; ++i) add_copy(s, o++, in_off + i, m(i));
                break;
            case E_ABS_POW:
            case E_SIGNED_POW:
                for (int i = 0; i < u; ++i) {
                    s.map_cols.push_back(o++);
                    s.map.push_back(MapEntry{in_off + i, in_off + i, f.kind == E_ABS_POW ? M_ABS_POW : M_SIGNED_POW,
                                             (float)f.expo, m(i), 0.f});
                }
                break;
            case E_QUADRATIC:
                for (int i = 0; i < u; ++i)
                    for (int j = i; j < u; ++j) {
                        s.map_cols.push_back(o++);
                        s.map.push_back(MapEntry{in_off + i, in_off + j, M_PRODUCT, 0.f, m(i), m(j)});
                    }
                break;
            case E_PAIR_ADJ:
                for (int i = 0; i + (int)f.k < u; ++i) {
                    s.map_cols.push_back(o++);
                    s.map.push_back(MapEntry{in_off + i, in_off + i + (int)f.k, M_PRODUCT, 0.f, m(i), m(i + f.k)});
                }
                break;
            case E_PAIR_BAND:
                for (int off = 0; off < (int)f.k; ++off)
                    for (int i = 0; i + off < u; ++i) {
                        s.map_cols.push_back(o++);
                        s.map.push_back(MapEntry{in_off + i, in_off + i + off, M_PRODUCT, 0.f, m(i), m(i + off)});
                    }
                break;
        }
    }
}

Prog seq(Prog a, Prog b) {
    if (a.out_dim != b.in_dim) fail(HG_ERR_DIM, "internal: seq dims");
    for (auto& s : b.steps) a.steps.push_back(std::move(s));
    a.out_dim = b.out_dim;
    return a;
}

void shift_into(Step& dst, const Step& src, int in_off, int out_off) {
    for (size_t i = 0; i < src.map.size(); ++i) {
        MapEntry e = src.map[i];
        e.i += in_off;
        e.j += in_off;
        dst.map.push_back(e);
        dst.map_cols.push_back(src.map_cols[i] + out_off);
    }
    for (const AffD& a : src.aff) {
        AffD c = a;
        c.in_off += in_off;
        c.out_off += out_off;
        dst.aff.push_back(std::move(c));
    }
}

Prog par(const std::vector<const Prog*>& ps) {
    Prog out;
    size_t D = 0;
    for (auto* p : ps) {
        out.in_dim += p->in_dim;
        out.out_dim += p->out_dim;
        D = std::max(D, p->steps.size());
    }
    for (size_t t = 0; t < D; ++t) {
        Step st;
        int in_off = 0, out_off = 0;
        for (auto* p : ps) {
            size_t dk = p->steps.size();
            int w_in = t == 0 ? p->in_dim : (t <= dk ? p->steps[t - 1].out_w : p->out_dim);
            int w_out = t < dk ? p->steps[t].out_w : p->out_dim;
            if (t < dk)
                shift_into(st, p->steps[t], in_off, out_off);
            else
                for (int c = 0; c < w_out; ++c) add_copy(st, out_off + c, in_off + c);
            in_off += w_in;
            out_off += w_out;
        }
        st.in_w = in_off;
        st.out_w = out_off;
        out.steps.push_back(std::move(st));
    }
    return out;
}

Prog one_step(int in_dim, int out_dim) {
    Prog p;
    p.in_dim = in_dim;
    p.out_dim = out_dim;
    p.steps.emplace_back();
    p.steps[0].in_w = in_dim;
    p.steps[0].out_w = out_dim;
    return p;
}

AffD affine_of(const TNode& n, int in_off, int out_off) {
    AffD d{in_off, (int)n.in_dim, out_off, (int)n.out_dim, n.a, n.W, n.b};
    return d;
}

Prog lower(const TNode& n) {
    switch (n.kind) {
        case K_FLOW:
        case K_FLOWNODE: {
            Prog p = lower(*n.ch[0]);
            for (size_t i = 1; i < n.ch.size(); ++i) p = seq(std::move(p), lower(*n.ch[i]));
            return p;
        }
        case K_LAYER: {
            std::vector<Prog> ps;
            for (auto& c : n.ch) ps.push_back(lower(*c));
            std::vector<const Prog*> pp;
            for (auto& p : ps) pp.push_back(&p);
            return par(pp);
        }
        case K_CLONELAYER: {
            Prog c = lower(*n.ch[0]);
            std::vector<const Prog*> pp(n.aux, &c);
            return par(pp);
        }
        case K_SWITCHBOARD: {
            Prog p = one_step(n.in_dim, n.out_dim);
            for (uint32_t c = 0; c < n.out_dim; ++c) add_copy(p.steps[0], c, n.conn[c]);
            return p;
        }
        case K_AFFINE: {
            Prog p = one_step(n.in_dim, n.out_dim);
            p.steps[0].aff.push_back(affine_of(n, 0, 0));
            return p;
        }
        case K_EXPANSION: {
            Prog p = one_step(n.in_dim, n.out_dim);
            add_expansion(p.steps[0], n.funcs, n.in_dim, 0, 0, nullptr);
            return p;
        }
        case K_IDENTITY:
        case K_HEAD: {
            Prog p = one_step(n.in_dim, n.out_dim);
            for (uint32_t c = 0; c < n.out_dim; ++c) add_copy(p.steps[0], c, c);
            return p;
        }
        case K_CUTOFF: {
            Prog p = one_step(n.in_dim, n.out_dim);
            for (uint32_t c = 0; c < n.out_dim; ++c) {
                p.steps[0].map_cols.push_back(c);
                p.steps[0].map.push_back(MapEntry{(int)c, (int)c, M_CLIP, (float)n.lo, (float)n.hi, 0.f});
            }
            return p;
        }
        case K_IGSFA: {
            // SURVEY.md §8a row a8, in four column-parallel steps; x0 = x - x_mean is carried along.
            const int d = n.in_dim, S = n.sfa->out_dim, k = n.aux, Q = n.pca->out_dim;
            const int E = n.exp_node ? (int)n.exp_node->out_dim : d;
            Prog p;
            p.in_dim = d;
            p.out_dim = n.out_dim;
            Step s1;  // [e (E) | x0 (d)]
            s1.in_w = d;
            s1.out_w = E + d;
            if (n.exp_node)
                add_expansion(s1, n.exp_node->funcs, d, 0, 0, n.x_mean.data());
            else
                for (int c = 0; c < d; ++c) add_copy(s1, c, c, (float)n.x_mean[c]);
            for (int c = 0; c < d; ++c) add_copy(s1, E + c, c, (float)n.x_mean[c]);
            Step s2;  // [s (S) | x0 (d)],  s = ((e - a) W + b) * magn
            s2.in_w = E + d;
            s2.out_w = S + d;
            {
                AffD a = affine_of(*n.sfa, 0, 0);
                for (int r = 0; r < E; ++r)
                    for (int c = 0; c < S; ++c) a.W[(size_t)r * S + c] *= n.magn[c];
                for (int c = 0; c < S; ++c) a.b[c] *= n.magn[c];
                s2.aff.push_back(std::move(a));
            }
            for (int c = 0; c < d; ++c) add_copy(s2, S + c, E + c);
            Step s3;  // [s[:k] | r (d)],  r = x0 - lr(s) = [s, x0] @ [-Wlr; I] + (a_lr Wlr - b_lr)
            s3.in_w = S + d;
            s3.out_w = k + d;
            for (int c = 0; c < k; ++c) add_copy(s3, c, c);
            if (n.lr) {
                AffD a{0, S + d, k, d, {}, {}, {}};
                a.a.assign(S + d, 0.0);
                a.W.assign((size_t)(S + d) * d, 0.0);
                a.b.assign(d, 0.0);
                for (int r = 0; r < S; ++r) {
                    a.a[r] = n.lr->a[r];
                    for (int c = 0; c < d; ++c) a.W[(size_t)r * d + c] = -n.lr->W[(size_t)r * d + c];
                }
                for (int c = 0; c < d; ++c) {
                    a.W[(size_t)(S + c) * d + c] = 1.0;
                    a.b[c] = -n.lr->b[c];
                }
                s3.aff.push_back(std::move(a));
            } else {
                for (int c = 0; c < d; ++c) add_copy(s3, k + c, S + c);
            }
            Step s4;  // [s[:k] | q (Q)]
            s4.in_w = k + d;
            s4.out_w = k + Q;
            for (int c = 0; c < k; ++c) add_copy(s4, c, c);
            s4.aff.push_back(affine_of(*n.pca, k, k));
            p.steps.push_back(std::move(s1));
            p.steps.push_back(std::move(s2));
            p.steps.push_back(std::move(s3));
            p.steps.push_back(std::move(s4));
            return p;
        }
    }
    fail(HG_ERR_FORMAT, "internal: cannot lower node kind %u", n.kind);
}

// ---- device side ------------------------------------------------------------------------------
struct AffDev {
    int32_t in_off, in_dim, out_off, out_dim;
    int64_t w_off, a_off, b_off;
};

template <typename T>
__global__ void k_convert_in(const T* __restrict__ x, int64_t ldx, float* __restrict__ dst, int64_t n, int dim) {
    int64_t total = n * dim;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / dim;
        int c = (int)(idx - r * dim);
        dst[idx] = (float)x[r * ldx + c];
    }
}

template <typename T>
__global__ void k_convert_out(const float* __restrict__ src, int64_t ld_src, T* __restrict__ y, int64_t ldy, int64_t n, int cols) {
    int64_t total = n * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / cols;
        int c = (int)(idx - r * cols);
        y[r * ldy + c] = (T)src[r * ld_src + c];
    }
}

__device__ __forceinline__ float abs_pow(float v, float p) {
    // |v|^p = exp2(p * log2|v|), exact 0 -> 0 (SURVEY.md §7 "Hard parts")
    float a = fabsf(v);
    return a == 0.f ? 0.f : exp2f(p * log2f(a));
}

__global__ void k_map(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t ld_dst,
                      const int32_t* __restrict__ cols, const MapEntry* __restrict__ ent, int n_map, int64_t n) {
    int64_t total = n * n_map;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx / n_map;
        int c = (int)(idx - r * n_map);
        MapEntry e = ent[c];
        const float* row = src + r * ld_src;
        float xi = row[e.i], v;
        switch (e.kind) {
            case M_COPY: v = xi - e.p1; break;
            case M_ABS_POW: v = abs_pow(xi - e.p1, e.p0); break;
            case M_SIGNED_POW: {
                float t = xi - e.p1;
                v = copysignf(abs_pow(t, e.p0), t);
                break;
            }
            case M_PRODUCT: v = (xi - e.p1) * (row[e.j] - e.p2); break;
            default: v = fminf(fmaxf(xi, e.p0), e.p1); break;
        }
        dst[r * ld_dst + cols[c]] = v;
    }
}

// One block = one affine descriptor x R rows.  x - a staged in LDS; thread (r, jg) produces
// columns jg, jg + 256/R, ...
__global__ void __launch_bounds__(256)
k_affine(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t ld_dst,
         const AffDev* __restrict__ descs, const float* __restrict__ wts, int64_t n, int R, int kstride, int64_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int64_t bid = blockIdx.x;
    const AffDev d = descs[bid / n_tiles];
    const int64_t row0 = (bid % n_tiles) * R;
    const float* W = wts + d.w_off;
    const float* a = wts + d.a_off;
    const float* b = wts + d.b_off;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < R * d.in_dim; idx += 256) {
        int r = idx / d.in_dim, k = idx - r * d.in_dim;
        int64_t row = row0 + r;
        xs[r * kstride + k] = row < n ? src[row * ld_src + d.in_off + k] - a[k] : 0.f;
    }
    __syncthreads();
    const int r = tid % R, jg = tid / R, nj = 256 / R;
    const int64_t row = row0 + r;
    const float* xr = xs + r * kstride;
    for (int j = jg; j < d.out_dim; j += nj) {
        float acc = 0.f;
        for (int k = 0; k < d.in_dim; ++k) acc = fmaf(xr[k], W[(int64_t)k * d.out_dim + j], acc);
        if (row < n) dst[row * ld_dst + d.out_off + j] = acc + b[j];
    }
}

class GenericExecutor : public Executor {
public:
    explicit GenericExecutor(const TNode& root) {
        prog_ = lower(root);
        int idx = 0;
        max_w_ = prog_.in_dim;
        for (auto& s : prog_.steps) {
            std::ostringstream os;
            os << "generic step " << idx++ << ": " << s.in_w << " -> " << s.out_w << " (" << s.map.size() << " map cols, "
               << s.aff.size() << " affine nodes)";
            s.name = os.str();
            max_w_ = std::max(max_w_, std::max(s.in_w, s.out_w));
            if (s.map.size() + [&] { size_t t = 0; for (auto& a : s.aff) t += a.out_dim; return t; }() != (size_t)s.out_w)
                fail(HG_ERR_FORMAT, "internal: step %d does not cover its output frame", idx - 1);
        }
    }
    int plan_kind() const override { return HG_PLAN_GENERIC; }
    int n_stages() const override { return (int)prog_.steps.size() + 2; }
    std::string stage_name(int i) const override {
        if (i == 0) return "generic convert-in";
        if (i == n_stages() - 1) return "generic convert-out";
        return prog_.steps[i - 1].name;
    }
    std::string describe() const override {
        std::ostringstream os;
        os << "plan: GENERIC (row-major fp32 activations, map + affine kernels)\n";
        for (int i = 0; i < n_stages(); ++i) os << "  [" << i << "] " << stage_name(i) << "\n";
        return os.str();
    }
    int64_t weight_bytes() const override { return (int64_t)wts_.bytes + tables_bytes_; }
    int64_t workspace_bytes() const override { return (int64_t)(bufA_.bytes + bufB_.bytes); }

    void to_device() override {
        std::vector<float> w;
        dev_steps_.clear();
        dev_steps_.resize(prog_.steps.size());
        tables_bytes_ = 0;
        for (size_t si = 0; si < prog_.steps.size(); ++si) {
            const Step& s = prog_.steps[si];
            DevStep& ds = dev_steps_[si];
            ds.n_map = (int)s.map.size();
            if (ds.n_map) {
                ds.cols.upload(s.map_cols.data(), s.map_cols.size() * sizeof(int32_t));
                ds.ents.upload(s.map.data(), s.map.size() * sizeof(MapEntry));
                tables_bytes_ += ds.cols.bytes + ds.ents.bytes;
            }
            ds.n_aff = (int)s.aff.size();
            int kmax = 1;
            std::vector<AffDev> dd;
            for (const AffD& a : s.aff) {
                AffDev d{a.in_off, a.in_dim, a.out_off, a.out_dim, 0, 0, 0};
                d.w_off = (int64_t)w.size();
                for (double v : a.W) w.push_back((float)v);
                d.a_off = (int64_t)w.size();
                for (double v : a.a) w.push_back((float)v);
                d.b_off = (int64_t)w.size();
                for (double v : a.b) w.push_back((float)v);
                dd.push_back(d);
                kmax = std::max(kmax, a.in_dim);
            }
            if (ds.n_aff) {
                ds.descs.upload(dd.data(), dd.size() * sizeof(AffDev));
                tables_bytes_ += ds.descs.bytes;
                ds.kstride = kmax | 1;
                int R = 64;
                while (R > 1 && (size_t)R * ds.kstride * 4 > 48 * 1024) R >>= 1;
                if ((size_t)R * ds.kstride * 4 > 64 * 1024) fail(HG_ERR_FORMAT, "affine node with input_dim %d is too wide", kmax);
                ds.R = R;
            }
        }
        wts_.upload(w.data(), w.size() * sizeof(float));
    }

    void reserve(int64_t rows) override {
        size_t need = (size_t)rows * max_w_ * sizeof(float);
        bufA_.alloc(need);
        bufB_.alloc(need);
        cap_rows_ = std::max(cap_rows_, rows);
    }

    void run(const void* x, int x_dtype, int64_t n, int64_t ldx, void* y, int y_dtype, int64_t y_cols, int64_t ldy,
             hipStream_t st, hipEvent_t* ev) override {
        if (n > cap_rows_) reserve(n);
        float* cur = (float*)bufA_.p;
        float* nxt = (float*)bufB_.p;
        int e = 0;
        auto mark = [&] {
            if (ev) HG_HIP(hipEventRecord(ev[e++], st));
        };
        auto grid1d = [](int64_t total) { return (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 16); };
        mark();
        const int in_dim = prog_.in_dim;
        int64_t cur_ld = in_dim;
        switch (x_dtype) {
            case HG_U8: hipLaunchKernelGGL(k_convert_in<uint8_t>, grid1d(n * in_dim), 256, 0, st, (const uint8_t*)x, ldx, cur, n, in_dim); break;
            case HG_F32: hipLaunchKernelGGL(k_convert_in<float>, grid1d(n * in_dim), 256, 0, st, (const float*)x, ldx, cur, n, in_dim); break;
            default: hipLaunchKernelGGL(k_convert_in<double>, grid1d(n * in_dim), 256, 0, st, (const double*)x, ldx, cur, n, in_dim); break;
        }
        mark();
        for (size_t si = 0; si < prog_.steps.size(); ++si) {
            const Step& s = prog_.steps[si];
            DevStep& ds = dev_steps_[si];
            if (ds.n_map)
                hipLaunchKernelGGL(k_map, grid1d(n * ds.n_map), 256, 0, st, cur, cur_ld, nxt, (int64_t)s.out_w,
                                   (const int32_t*)ds.cols.p, (const MapEntry*)ds.ents.p, ds.n_map, n);
            if (ds.n_aff) {
                int64_t tiles = (n + ds.R - 1) / ds.R;
                int64_t blocks = tiles * ds.n_aff;
                if (blocks > 0x7fffffffll) fail(HG_ERR_ARG, "batch too large for the generic plan");
                hipLaunchKernelGGL(k_affine, (unsigned)blocks, 256, (size_t)ds.R * ds.kstride * 4, st, cur, cur_ld, nxt,
                                   (int64_t)s.out_w, (const AffDev*)ds.descs.p, (const float*)wts_.p, n, ds.R, ds.kstride, tiles);
            }
            std::swap(cur, nxt);
            cur_ld = s.out_w;
            mark();
        }
        switch (y_dtype) {
            case HG_F32: hipLaunchKernelGGL(k_convert_out<float>, grid1d(n * y_cols), 256, 0, st, cur, cur_ld, (float*)y, ldy, n, (int)y_cols); break;
            case HG_F64: hipLaunchKernelGGL(k_convert_out<double>, grid1d(n * y_cols), 256, 0, st, cur, cur_ld, (double*)y, ldy, n, (int)y_cols); break;
            default: fail(HG_ERR_ARG, "output dtype must be f32 or f64");
        }
        mark();
        HG_HIP(hipGetLastError());
    }

    void release() override {
        bufA_.free();
        bufB_.free();
        wts_.free();
        dev_steps_.clear();
        cap_rows_ = 0;
    }

private:
    struct DevStep {
        int n_map = 0, n_aff = 0, R = 64, kstride = 1;
        DevBuf cols, ents, descs;
    };
    Prog prog_;
    int max_w_ = 0;
    std::vector<DevStep> dev_steps_;
    DevBuf wts_, bufA_, bufB_;
    int64_t tables_bytes_ = 0, cap_rows_ = 0;
};

}  // namespace

std::unique_ptr<Executor> make_generic_executor(const TNode& root) { return std::make_unique<GenericExecutor>(root); }

}  // namespace hg
