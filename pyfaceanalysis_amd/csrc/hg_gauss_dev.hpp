// Gaussian soft-label regression, workgroup form, as a device function (k_gauss_regression_wg in hg_gauss.hip; a header so that a
// kernel of another translation unit can run it too — round 4 tried that for the cascade's stage kernel, see hg_cascade.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace hg {

constexpr int kGaussMaxClasses = 1024;
constexpr int kGaussMaxMulti = 4;      // classifiers per hg_gauss_regression_multi_device launch

// What a classifier handle holds on the device (hg_gauss.hip): logw[c] = log p_c - log sqrtdet_c
struct GaussParams {
    int K = 0, d = 0;
    const double *means = nullptr, *inv_covs = nullptr, *logw = nullptr, *avg = nullptr;
};

// The regression with the quadratic forms spread over a workgroup (round 4).  In the one-wave form (hg_gauss.hip), lane c walks the d x d matrix of its
// class alone — 400 dependent steps for the pose regressors (50 classes, 20 features), with every lane of a load on another
// cache line: 25-32 us for a call on a handful of rows, nine times per frame.  Here thread (c, i) of four waves computes
// t_i = sum_j S_c[i][j] (x_j - m_c[j]) from one contiguous matrix row and leaves it in LDS; wave 0 then forms
// q += t_i (x_i - m_c[i]) over a class's d rows in the order — and with the very expression — of the loop above and finishes
// exactly as above: the same operations in the same order on every value, hence the same bits (tested).  For K d <= 4096
// (32 KiB of LDS).
// Rows row0 .. row0 + R - 1 (those below n) by the calling workgroup of 256 threads; lds: R * (64 + K d) doubles.  Every thread of
// the workgroup must call it (barriers inside); no thread returns early.
template <typename T, int R>
__device__ __forceinline__ void gauss_rows_wg(const T* __restrict__ x, int64_t ldx, int64_t n, int64_t row0, const GaussParams& G, double* lds_g,
                                              double* __restrict__ out_reg, double* __restrict__ out_std) {
    const int K = G.K, d = G.d;
    const double *means = G.means, *inv_covs = G.inv_covs, *logw = G.logw, *avg = G.avg;
    // R rows per workgroup: a matrix row read from L2 once serves R feature vectors (a launch on 348 rows of the 50-class
    // regressors moved 55 MB through L2 at one row per workgroup); wave r finishes row r
    double* xs = lds_g;                       // [R][64]
    double* term = lds_g + R * 64;            // [R][K * d]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kd = K * d;
    for (int e = tid; e < R * d; e += blockDim.x) {
        const int r = e / d, j = e - r * d;
        xs[r * 64 + j] = row0 + r < n ? (double)x[(row0 + r) * ldx + j] : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < kd; e += blockDim.x) {
        const int c = e / d, i = e - c * d;
        const double* m = means + (size_t)c * d;
        const double* S = inv_covs + (size_t)c * d * d + (size_t)i * d;
        double t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) t[r] = 0;
        // (matrix row and mean in chunks of eight, all loads of a chunk requested before the first multiply, was measured late in round 5:
        // 14.8 -> 22 us for the four-classifier launch.  Not kept.)
        for (int j = 0; j < d; ++j) {
            const double sj = S[j], mj = m[j];
#pragma unroll
            for (int r = 0; r < R; ++r) t[r] += sj * (xs[r * 64 + j] - mj);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) term[r * kd + e] = t[r];
    }
    __syncthreads();
    for (int rr = wave; rr < R; rr += (int)(blockDim.x >> 6)) {      // wave w finishes rows w, w + 4, ... of the workgroup's R (R <= 4: one each)
        if (row0 + rr >= n) break;
        const double* xr = xs + rr * 64;
        const double* tr = term + rr * kd;
        double lmax = -INFINITY;
        double lp[kGaussMaxClasses / 64];
#pragma unroll
        for (int s = 0; s < kGaussMaxClasses / 64; ++s) {
            const int c = lane + 64 * s;
            double v = -INFINITY;
            if (c < K) {
                const double* m = means + (size_t)c * d;
                double q = 0;
                for (int i = 0; i < d; ++i) q += tr[c * d + i] * (xr[i] - m[i]);      // the expression of the one-wave kernel, term = its t
                v = logw[c] - 0.5 * q;
            }
            lp[s] = v;
            lmax = fmax(lmax, v);
        }
        for (int o = 32; o > 0; o >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, o));
        double sw = 0, swa = 0, swa2 = 0;
#pragma unroll
        for (int s = 0; s < kGaussMaxClasses / 64; ++s) {
            const int c = lane + 64 * s;
            if (c < K) {
                const double w = exp(lp[s] - lmax), a = avg[c];
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
            const double reg = swa / sw;
            out_reg[row0 + rr] = reg;
            if (out_std) out_std[row0 + rr] = sqrt(fmax(swa2 / sw - reg * reg, 0.0));
        }
    }
}

// hg_gauss.hip: the device-side parameters of a classifier handle (for kernels of other translation units)
}  // namespace hg

struct hg_gauss;
namespace hg {
GaussParams gauss_params(const hg_gauss* g);
}  // namespace hg
