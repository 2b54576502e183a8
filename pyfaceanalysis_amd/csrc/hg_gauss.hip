// Gaussian-classifier soft-label regression: the step right after the hot call,
//   reg_out = classifiers[k].regression(sl[:, 0:reg_num_signals], avg_labels)
// (FaceDetectUpdated.py:709-719; face_analysis.py:1068-1073, 1261-1290).  Parameters are the
// attributes stored in SavedClassifiers/*.pckl (means, inv_covs, _sqrt_def_covs, p, avg_labels).
//
//   post_c(x) ∝ p_c / sqrtdet_c * exp(-1/2 (x-m_c)' S_c^-1 (x-m_c)),  normalised over classes
//   reg(x)    = sum_c post_c(x) avg_labels[c]
//   std(x)    = sqrt(sum_c post_c(x) (avg_labels[c] - reg)^2)
// Evaluated in the log domain in fp64 (sqrtdet reaches 1e42 in the shipped classifiers).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "hg_common.hpp"
#include "hg_gauss_dev.hpp"

struct hg_gauss {
    int K = 0, d = 0, device = -1;
    hg::DevBuf means, inv_covs, logw, avg;  // logw[c] = log p_c - log sqrtdet_c
    hg::DevBuf sx, sreg, sstd;              // host-call staging
};

namespace hg {
GaussParams gauss_params(const hg_gauss* g) {
    GaussParams G;
    G.K = g->K;
    G.d = g->d;
    G.means = (const double*)g->means.p;
    G.inv_covs = (const double*)g->inv_covs.p;
    G.logw = (const double*)g->logw.p;
    G.avg = (const double*)g->avg.p;
    return G;
}
}  // namespace hg

namespace {

thread_local std::string g_gauss_error;

constexpr int kMaxClasses = hg::kGaussMaxClasses;

template <typename T>
__global__ void __launch_bounds__(64)
k_gauss_regression(const T* __restrict__ x, int64_t ldx, int64_t n, int K, int d, const double* __restrict__ means,
                   const double* __restrict__ inv_covs, const double* __restrict__ logw, const double* __restrict__ avg,
                   double* __restrict__ out_reg, double* __restrict__ out_std) {
    // one 64-lane wave per row; lane c handles classes c, c+64, ...
    __shared__ double xs[64];
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x;
    if (lane < d) xs[lane] = (double)x[row * ldx + lane];
    __syncthreads();
    double lmax = -INFINITY;
    // pass 1: log-weights, kept in registers for up to 16 classes per lane
    double lp[kMaxClasses / 64];
#pragma unroll
    for (int s = 0; s < kMaxClasses / 64; ++s) {
        int c = lane + 64 * s;
        double v = -INFINITY;
        if (c < K) {
            const double* m = means + (size_t)c * d;
            const double* S = inv_covs + (size_t)c * d * d;
            double q = 0;
            for (int i = 0; i < d; ++i) {
                double t = 0;
                for (int j = 0; j < d; ++j) t += S[i * d + j] * (xs[j] - m[j]);
                q += t * (xs[i] - m[i]);
            }
            v = logw[c] - 0.5 * q;
        }
        lp[s] = v;
        lmax = fmax(lmax, v);
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, o));
    double sw = 0, swa = 0, swa2 = 0;
#pragma unroll
    for (int s = 0; s < kMaxClasses / 64; ++s) {
        int c = lane + 64 * s;
        if (c < K) {
            double w = exp(lp[s] - lmax), a = avg[c];
            sw += w;
            swa += w * a;
            swa2 += w * a * a;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sw += __shfl_xor(sw, o);
        swa += __shfl_xor(swa, o);
        swa2 += __shfl_xor(swa2, o);
    }
    if (lane == 0) {
        double reg = swa / sw;
        out_reg[row] = reg;
        if (out_std) out_std[row] = sqrt(fmax(swa2 / sw - reg * reg, 0.0));
    }
}

// The workgroup form (round 4): hg_gauss_dev.hpp, shared with the cascade's stage kernel.
template <typename T, int R>
__global__ void __launch_bounds__(256)
k_gauss_regression_wg(const T* __restrict__ x, int64_t ldx, int64_t n, hg::GaussParams G, double* __restrict__ out_reg, double* __restrict__ out_std) {
    extern __shared__ double lds_g[];
    hg::gauss_rows_wg<T, R>(x, ldx, n, (int64_t)blockIdx.x * R, G, lds_g, out_reg, out_std);
}

// Several classifiers on the SAME feature rows in one launch (round 5: a cascade stage that owns a network and the "None" stages
// behind it, FaceDetectUpdated.py:678-682 / :704-706, read one sl): blockIdx.y picks the classifier, regressions of classifier s
// land in out_reg[s * out_stride + row].  The per-row arithmetic is gauss_rows_wg's, whatever the launch shape: the same bits
// as one launch per classifier (tests/test_cascade.py compares them).
struct GaussMulti {
    hg::GaussParams g[hg::kGaussMaxMulti];
};
template <typename T, int R>
__global__ void __launch_bounds__(256)
k_gauss_regression_wg_multi(const T* __restrict__ x, int64_t ldx, int64_t n, GaussMulti M, double* __restrict__ out_reg, int64_t out_stride) {
    extern __shared__ double lds_g[];
    // (selected with wave-uniform moves: indexing the by-value argument with blockIdx.y makes the compiler spill the whole table to scratch)
    const int y = blockIdx.y;
    hg::GaussParams G = M.g[0];
    if (y == 1) G = M.g[1];
    if (y == 2) G = M.g[2];
    if (y == 3) G = M.g[3];
    static_assert(hg::kGaussMaxMulti == 4, "selection above");
    hg::gauss_rows_wg<T, R>(x, ldx, n, (int64_t)blockIdx.x * R, G, lds_g, out_reg + (int64_t)blockIdx.y * out_stride, nullptr);
}

template <typename F>
int guarded(F&& fn) {
    try {
        fn();
        return HG_OK;
    } catch (const hg::Error& e) {
        g_gauss_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_gauss_error = e.what();
        return HG_ERR_STATE;
    }
}

void launch(hg_gauss* g, const void* x, int x_dtype, int64_t n, int64_t ldx, double* reg, double* sd, hipStream_t st) {
    if (n == 0) return;
    if (n > 0x7fffffffll) hg::fail(HG_ERR_ARG, "too many rows");
    if (x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "feature dtype must be HG_F32 or HG_F64");
    const bool no_wg = getenv("HIGSFA_GAUSS_WAVE") != nullptr;      // tests: the one-wave-per-row kernel only (read per call)
    if ((int64_t)g->K * g->d <= 4096 && !no_wg) {
        // four rows per workgroup from 256 rows on (fewer: one row per workgroup keeps more of the chip busy), while the LDS image fits
        const bool r4 = n >= 256 && (int64_t)g->K * g->d <= 1536;
        const int R = r4 ? 4 : 1;
        const size_t lds = (size_t)R * (64 + g->K * g->d) * 8;
        const unsigned grid = (unsigned)((n + R - 1) / R);
        const hg::GaussParams G = hg::gauss_params(g);
#define HG_GAUSS_WG(TT, RR) hipLaunchKernelGGL((k_gauss_regression_wg<TT, RR>), grid, 256, lds, st, (const TT*)x, ldx, n, G, reg, sd)
        if (x_dtype == HG_F32) {
            if (r4) HG_GAUSS_WG(float, 4);
            else HG_GAUSS_WG(float, 1);
        } else {
            if (r4) HG_GAUSS_WG(double, 4);
            else HG_GAUSS_WG(double, 1);
        }
#undef HG_GAUSS_WG
        HG_HIP(hipGetLastError());
        return;
    }
    if (x_dtype == HG_F32)
        hipLaunchKernelGGL(k_gauss_regression<float>, (unsigned)n, 64, 0, st, (const float*)x, ldx, n, g->K, g->d,
                           (const double*)g->means.p, (const double*)g->inv_covs.p, (const double*)g->logw.p,
                           (const double*)g->avg.p, reg, sd);
    else if (x_dtype == HG_F64)
        hipLaunchKernelGGL(k_gauss_regression<double>, (unsigned)n, 64, 0, st, (const double*)x, ldx, n, g->K, g->d,
                           (const double*)g->means.p, (const double*)g->inv_covs.p, (const double*)g->logw.p,
                           (const double*)g->avg.p, reg, sd);
    else
        hg::fail(HG_ERR_ARG, "feature dtype must be HG_F32 or HG_F64");
    HG_HIP(hipGetLastError());
}

}  // namespace

// hg_last_error() lives in hg_capi.cpp; route gauss errors through the same accessor.
extern "C" const char* hg_last_error(void);
namespace hg { void set_last_error(const std::string& s); }

extern "C" {

int hg_gauss_create(int32_t n_classes, int32_t dim, const double* means, const double* inv_covs, const double* sqrt_det_covs,
                    const double* priors, const double* avg_labels, int device, hg_gauss** out) {
    int rc = guarded([&] {
        if (!out) hg::fail(HG_ERR_ARG, "null output handle pointer");
        *out = nullptr;
        if (!means || !inv_covs || !sqrt_det_covs || !priors || !avg_labels) hg::fail(HG_ERR_ARG, "null parameter array");
        if (n_classes < 1 || n_classes > kMaxClasses) hg::fail(HG_ERR_ARG, "n_classes must be 1..%d", kMaxClasses);
        if (dim < 1 || dim > 64) hg::fail(HG_ERR_ARG, "dim must be 1..64");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
        if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range", device);
        HG_HIP(hipSetDevice(device));
        auto g = std::make_unique<hg_gauss>();
        g->K = n_classes;
        g->d = dim;
        g->device = device;
        std::vector<double> lw(n_classes);
        for (int c = 0; c < n_classes; ++c) {
            if (!(sqrt_det_covs[c] > 0) || !(priors[c] >= 0)) hg::fail(HG_ERR_ARG, "class %d: non-positive sqrt-det or negative prior", c);
            lw[c] = std::log(priors[c]) - std::log(sqrt_det_covs[c]);
        }
        g->means.upload(means, sizeof(double) * n_classes * dim);
        g->inv_covs.upload(inv_covs, sizeof(double) * n_classes * dim * dim);
        g->logw.upload(lw.data(), sizeof(double) * n_classes);
        g->avg.upload(avg_labels, sizeof(double) * n_classes);
        *out = g.release();
    });
    if (rc != HG_OK) hg::set_last_error(g_gauss_error);
    return rc;
}

void hg_gauss_free(hg_gauss* g) {
    if (g && g->device >= 0) (void)hipSetDevice(g->device);
    delete g;
}

int hg_gauss_regression_device(hg_gauss* g, const void* x, int x_dtype, int64_t n, int64_t ldx, double* out_reg,
                               double* out_std, void* stream) {
    int rc = guarded([&] {
        if (!g) hg::fail(HG_ERR_ARG, "null classifier handle");
        if (n < 0) hg::fail(HG_ERR_ARG, "negative row count");
        if (ldx < g->d) hg::fail(HG_ERR_DIM, "x has %lld columns but the classifier's input_dim is %d", (long long)ldx, g->d);
        if (n > 0 && (!x || !out_reg)) hg::fail(HG_ERR_ARG, "null data pointer");
        HG_HIP(hipSetDevice(g->device));
        launch(g, x, x_dtype, n, ldx, out_reg, out_std, (hipStream_t)stream);
    });
    if (rc != HG_OK) hg::set_last_error(g_gauss_error);
    return rc;
}

int hg_gauss_regression_multi_device(hg_gauss* const* gs, int m, const void* x, int x_dtype, int64_t n, int64_t ldx, double* out_reg,
                                     int64_t out_stride, void* stream) {
    int rc = guarded([&] {
        if (!gs || m < 1 || m > hg::kGaussMaxMulti) hg::fail(HG_ERR_ARG, "1..%d classifiers per launch", hg::kGaussMaxMulti);
        if (n < 0 || n > 0x7fffffffll) hg::fail(HG_ERR_ARG, "bad row count");
        if (out_stride < n) hg::fail(HG_ERR_ARG, "out_stride %lld < n %lld", (long long)out_stride, (long long)n);
        if (x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "feature dtype must be HG_F32 or HG_F64");
        // four rows per workgroup as soon as the launch has a chip's worth of workgroups WITHOUT it (m classifiers multiply the grid): at
        // one row per workgroup every workgroup pulls its classifier's K d d matrices (160 KB for the 50 x 20 pose regressors) through L2
        // for a single feature vector — 130 rows x 4 classifiers took 23.5 us that way, 348 x 4 at four rows per workgroup 14.5
        bool wg = getenv("HIGSFA_GAUSS_WAVE") == nullptr, r4 = n * (int64_t)m >= 128;
        size_t kd_max = 0;
        for (int s = 0; s < m; ++s) {
            if (!gs[s]) hg::fail(HG_ERR_ARG, "null classifier handle");
            if (gs[s]->device != gs[0]->device) hg::fail(HG_ERR_ARG, "classifiers of one launch must live on one device");
            if (ldx < gs[s]->d) hg::fail(HG_ERR_DIM, "x has %lld columns but a classifier's input_dim is %d", (long long)ldx, gs[s]->d);
            const size_t kd = (size_t)gs[s]->K * gs[s]->d;
            kd_max = std::max(kd_max, kd);
            wg = wg && kd <= 4096;
            r4 = r4 && kd <= 1536;
        }
        if (n > 0 && (!x || !out_reg)) hg::fail(HG_ERR_ARG, "null data pointer");
        HG_HIP(hipSetDevice(gs[0]->device));
        if (n == 0) return;
        if (!wg || m == 1) {      // shapes the workgroup form does not take: one launch per classifier
            for (int s = 0; s < m; ++s) launch(gs[s], x, x_dtype, n, ldx, out_reg + (int64_t)s * out_stride, nullptr, (hipStream_t)stream);
            return;
        }
        GaussMulti M;
        for (int s = 0; s < m; ++s) M.g[s] = hg::gauss_params(gs[s]);
        // eight rows per workgroup where the batch has them (every workgroup streams its classifier's matrices through L2 once for its
        // rows: 348 rows x 4 pose regressors moved 56 MB at four rows per workgroup and took 14 us whatever the batch)
        // (measured and not used, round 5: 16.0-17.9 us per launch against 14.1-15.2 at four rows — the launch is bound by the chain of
        // fp64 FMAs per thread, not by L2; the instantiation stays for HIGSFA_GAUSS_R8=1)
        const bool r8 = r4 && n >= 64 && kd_max <= 2048 && getenv("HIGSFA_GAUSS_R8") != nullptr;
        const int R = r8 ? 8 : r4 ? 4 : 1;
        const size_t lds = (size_t)R * (64 + kd_max) * 8;
        const dim3 grid((unsigned)((n + R - 1) / R), (unsigned)m);
        hipStream_t st = (hipStream_t)stream;
#define HG_GAUSS_WGM(TT, RR)                                                                                                                  \
    do {                                                                                                                                       \
        if (lds > 64 * 1024) {                                                                                                                 \
            static bool raised = false;      /* once per instantiation */                                                                      \
            if (!raised) HG_HIP(hipFuncSetAttribute((const void*)k_gauss_regression_wg_multi<TT, RR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            raised = true;                                                                                                                     \
        }                                                                                                                                      \
        hipLaunchKernelGGL((k_gauss_regression_wg_multi<TT, RR>), grid, dim3(256), lds, st, (const TT*)x, ldx, n, M, out_reg, out_stride);     \
    } while (0)
        if (x_dtype == HG_F32) {
            if (r8) HG_GAUSS_WGM(float, 8);
            else if (r4) HG_GAUSS_WGM(float, 4);
            else HG_GAUSS_WGM(float, 1);
        } else {
            if (r8) HG_GAUSS_WGM(double, 8);
            else if (r4) HG_GAUSS_WGM(double, 4);
            else HG_GAUSS_WGM(double, 1);
        }
#undef HG_GAUSS_WGM
        HG_HIP(hipGetLastError());
    });
    if (rc != HG_OK) hg::set_last_error(g_gauss_error);
    return rc;
}

int hg_gauss_regression(hg_gauss* g, const void* x, int x_dtype, int64_t n, int64_t ldx, double* out_reg, double* out_std) {
    int rc = guarded([&] {
        if (!g) hg::fail(HG_ERR_ARG, "null classifier handle");
        if (n < 0) hg::fail(HG_ERR_ARG, "negative row count");
        if (ldx < g->d) hg::fail(HG_ERR_DIM, "x has %lld columns but the classifier's input_dim is %d", (long long)ldx, g->d);
        if (n == 0) return;
        if (!x || !out_reg) hg::fail(HG_ERR_ARG, "null data pointer");
        if (x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "feature dtype must be HG_F32 or HG_F64");
        HG_HIP(hipSetDevice(g->device));
        const size_t es = hg::dtype_size(x_dtype);
        g->sx.alloc((size_t)n * g->d * es);
        g->sreg.alloc((size_t)n * 8);
        if (out_std) g->sstd.alloc((size_t)n * 8);
        HG_HIP(hipMemcpy2D(g->sx.p, g->d * es, x, (size_t)ldx * es, g->d * es, (size_t)n, hipMemcpyHostToDevice));
        launch(g, g->sx.p, x_dtype, n, g->d, (double*)g->sreg.p, out_std ? (double*)g->sstd.p : nullptr, nullptr);
        HG_HIP(hipMemcpy(out_reg, g->sreg.p, (size_t)n * 8, hipMemcpyDeviceToHost));
        if (out_std) HG_HIP(hipMemcpy(out_std, g->sstd.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    });
    if (rc != HG_OK) hg::set_last_error(g_gauss_error);
    return rc;
}

}  // extern "C"
