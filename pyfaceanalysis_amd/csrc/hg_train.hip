// SFA training step for one layer of nodes (BASELINE.json configs[4]).  The reference never
// trains (flows arrive pre-trained, face_analysis.py:451-479; the statistics it once needed are the
// `cov_mtx` / `dcov_mtx` mentioned at face_analysis.py:463-467); this restates what mdp.nodes.SFANode.train +
// stop_training compute, per node k over its input columns conn[k]:
//     mean = E[x],   B = Cov(x),   A = Cov(x[t+1] - x[t]),   solve A w = lambda B w,
//     eigenvalues ascending (slowest first), eigenvectors normalised w' B w = 1.
// Statistics: hand-written HIP kernel, fp64 accumulation (MDP accumulates in float64 too), one workgroup per
// (node, sample split), partial sums reduced in a fixed order (bit-reproducible).  Solve: rocSOLVER
// dsygvd_strided_batched (a plain library call on small dense matrices).
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "hg_common.hpp"

namespace hg { void set_last_error(const std::string& s); }

namespace {

constexpr int kTS = 64;   // samples staged per step

// partial[(split * n_nodes + node)] = { sum x (d) | sum x x' (d*d) | sum dx dx' (d*d) }
template <typename XT>
__global__ void __launch_bounds__(256) k_sfa_stats(const XT* __restrict__ x, int64_t ldx, int64_t n, const int32_t* __restrict__ conn, int d,
                                                    int n_nodes, int n_splits, double* __restrict__ partial) {
    extern __shared__ double xs[];   // [(kTS + 1)][d]
    const int node = blockIdx.x % n_nodes, split = blockIdx.x / n_nodes;
    const int tid = threadIdx.x;
    const int64_t per = (n + n_splits - 1) / n_splits;
    const int64_t t0 = split * per, t1 = min(n, t0 + per);
    const int32_t* cols = conn + (size_t)node * d;
    const int dd = d * d;
    // each thread owns entries e = tid, tid + 256, ... of the d x d matrices (<= 16 for d = 64)
    double sxx[16], sdd[16], sx = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sxx[k] = sdd[k] = 0.0;
    for (int64_t tb = t0; tb < t1; tb += kTS) {
        const int m = (int)min<int64_t>(kTS, t1 - tb);
        const int mm = (tb + m < n) ? m + 1 : m;            // one sample ahead for the last difference of the chunk
        __syncthreads();
        for (int idx = tid; idx < mm * d; idx += 256) {
            const int tt = idx / d, c = idx - tt * d;
            xs[idx] = (double)x[(tb + tt) * ldx + cols[c]];
        }
        __syncthreads();
        const int nd = mm - 1;   // differences x[t+1] - x[t] that START in this chunk (the last sample overall starts none)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = tid + k * 256;
            if (e < dd) {
                const int i = e / d, j = e - i * d;
                double a = sxx[k], b = sdd[k];
                for (int tt = 0; tt < m; ++tt) a += xs[tt * d + i] * xs[tt * d + j];
                for (int tt = 0; tt < nd; ++tt) {
                    const double di = xs[(tt + 1) * d + i] - xs[tt * d + i], dj = xs[(tt + 1) * d + j] - xs[tt * d + j];
                    b += di * dj;
                }
                sxx[k] = a;
                sdd[k] = b;
            }
        }
        if (tid < d)
            for (int tt = 0; tt < m; ++tt) sx += xs[tt * d + tid];
    }
    double* out = partial + ((size_t)split * n_nodes + node) * (size_t)(d + 2 * dd);
    if (tid < d) out[tid] = sx;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = tid + k * 256;
        if (e < dd) {
            out[d + e] = sxx[k];
            out[d + dd + e] = sdd[k];
        }
    }
}

// fixed-order reduction over splits, then mean / covariance / difference covariance per node
__global__ void k_sfa_finish(const double* __restrict__ partial, int n_nodes, int n_splits, int d, int64_t n, double* __restrict__ mean,
                             double* __restrict__ B, double* __restrict__ A) {
    const int node = blockIdx.x, dd = d * d;
    const size_t rec = (size_t)(d + 2 * dd);
    for (int e = threadIdx.x; e < dd; e += blockDim.x) {
        const int i = e / d, j = e - i * d;
        double sxx = 0, sdd = 0, si = 0, sj = 0;
        for (int s = 0; s < n_splits; ++s) {
            const double* p = partial + ((size_t)s * n_nodes + node) * rec;
            sxx += p[d + e];
            sdd += p[d + dd + e];
            si += p[i];
            sj += p[j];
        }
        B[(size_t)node * dd + e] = (sxx - si * sj / (double)n) / (double)(n - 1);
        A[(size_t)node * dd + e] = sdd / (double)(n - 1);          // MDP: derivative covariance, not mean-centred
        if (j == 0) mean[(size_t)node * d + i] = si / (double)n;
    }
}

template <typename F>
int guarded(F&& fn) {
    try {
        fn();
        return HG_OK;
    } catch (const hg::Error& e) {
        hg::set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        hg::set_last_error(e.what());
        return HG_ERR_STATE;
    }
}

}  // namespace

extern "C" int hg_sfa_train_layer(const void* x_in, int x_on_host, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host, int32_t n_nodes,
                                  int32_t d, int device, double* evals_host, double* evecs_host, double* mean_host,
                                  double* timings_ms) {
    return guarded([&] {
        if (!x_in || !conn_host || !evals_host || !evecs_host || !mean_host) hg::fail(HG_ERR_ARG, "null pointer");
        if (n < 3 || n_nodes < 1 || d < 1 || d > 64) hg::fail(HG_ERR_ARG, "need n >= 3 samples, 1..64 inputs per node");
        if (x_dtype != HG_U8 && x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "bad dtype");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
        if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range", device);
        HG_HIP(hipSetDevice(device));
        for (int64_t i = 0; i < (int64_t)n_nodes * d; ++i)
            if (conn_host[i] < 0 || conn_host[i] >= ldx) hg::fail(HG_ERR_ARG, "connection %d out of range", conn_host[i]);
        hg::DevBuf xbuf;
        const void* x_dev = x_in;
        if (x_on_host) {
            xbuf.upload(x_in, (size_t)n * ldx * hg::dtype_size(x_dtype));
            x_dev = xbuf.p;
        }
        const int dd = d * d;
        int n_splits = (int)std::max<int64_t>(1, std::min<int64_t>(64, (2048 + n_nodes - 1) / n_nodes));
        n_splits = (int)std::min<int64_t>(n_splits, std::max<int64_t>(1, n / 256));
        hg::DevBuf conn, partial, mean, A, B, W, E, info;
        conn.upload(conn_host, (size_t)n_nodes * d * 4);
        partial.alloc((size_t)n_splits * n_nodes * (d + 2 * dd) * 8);
        mean.alloc((size_t)n_nodes * d * 8);
        A.alloc((size_t)n_nodes * dd * 8);
        B.alloc((size_t)n_nodes * dd * 8);
        W.alloc((size_t)n_nodes * d * 8);
        E.alloc((size_t)n_nodes * d * 8);
        info.alloc((size_t)n_nodes * 4);
        hipEvent_t e0, e1, e2;
        HG_HIP(hipEventCreate(&e0));
        HG_HIP(hipEventCreate(&e1));
        HG_HIP(hipEventCreate(&e2));
        HG_HIP(hipEventRecord(e0, nullptr));
        const size_t lds = (size_t)(kTS + 1) * d * 8;
        const unsigned grid = (unsigned)(n_nodes * n_splits);
        if (x_dtype == HG_U8)
            hipLaunchKernelGGL(k_sfa_stats<uint8_t>, grid, 256, lds, nullptr, (const uint8_t*)x_dev, ldx, n, (const int32_t*)conn.p, d, n_nodes,
                               n_splits, (double*)partial.p);
        else if (x_dtype == HG_F32)
            hipLaunchKernelGGL(k_sfa_stats<float>, grid, 256, lds, nullptr, (const float*)x_dev, ldx, n, (const int32_t*)conn.p, d, n_nodes,
                               n_splits, (double*)partial.p);
        else
            hipLaunchKernelGGL(k_sfa_stats<double>, grid, 256, lds, nullptr, (const double*)x_dev, ldx, n, (const int32_t*)conn.p, d, n_nodes,
                               n_splits, (double*)partial.p);
        hipLaunchKernelGGL(k_sfa_finish, (unsigned)n_nodes, 256, 0, nullptr, (const double*)partial.p, n_nodes, n_splits, d, n, (double*)mean.p,
                           (double*)B.p, (double*)A.p);
        HG_HIP(hipGetLastError());
        HG_HIP(hipEventRecord(e1, nullptr));
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success) hg::fail(HG_ERR_DEVICE, "rocblas_create_handle failed");
        rocblas_status rs = rocsolver_dsygvd_strided_batched(h, rocblas_eform_ax, rocblas_evect_original, rocblas_fill_upper, d, (double*)A.p, d,
                                                            dd, (double*)B.p, d, dd, (double*)W.p, d, (double*)E.p, d, (rocblas_int*)info.p,
                                                            n_nodes);
        HG_HIP(hipEventRecord(e2, nullptr));
        HG_HIP(hipDeviceSynchronize());
        rocblas_destroy_handle(h);
        if (rs != rocblas_status_success) hg::fail(HG_ERR_DEVICE, "rocsolver_dsygvd_strided_batched failed (%d)", (int)rs);
        std::vector<int> hinfo(n_nodes);
        HG_HIP(hipMemcpy(hinfo.data(), info.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < n_nodes; ++k)
            if (hinfo[k] != 0) hg::fail(HG_ERR_STATE, "node %d: sygvd info = %d (covariance not positive definite or no convergence)", k, hinfo[k]);
        HG_HIP(hipMemcpy(evals_host, W.p, (size_t)n_nodes * d * 8, hipMemcpyDeviceToHost));
        HG_HIP(hipMemcpy(evecs_host, A.p, (size_t)n_nodes * dd * 8, hipMemcpyDeviceToHost));   // column-major (LAPACK) per node
        HG_HIP(hipMemcpy(mean_host, mean.p, (size_t)n_nodes * d * 8, hipMemcpyDeviceToHost));
        if (timings_ms) {
            float a = 0, b = 0;
            HG_HIP(hipEventElapsedTime(&a, e0, e1));
            HG_HIP(hipEventElapsedTime(&b, e1, e2));
            timings_ms[0] = a;
            timings_ms[1] = b;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipEventDestroy(e2);
    });
}
