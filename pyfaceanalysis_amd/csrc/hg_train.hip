// SFA training step for one layer of nodes (BASELINE.json configs[4]).  The reference never
// trains (flows arrive pre-trained, face_analysis.py:451-479; the statistics it once needed are the
// `cov_mtx` / `dcov_mtx` mentioned at face_analysis.py:463-467); this restates what mdp.nodes.SFANode.train +
// stop_training compute, per node k over its input columns conn[k]:
//     mean = E[x],   B = Cov(x),   A = Cov(x[t+1] - x[t]),   solve A w = lambda B w,
//     eigenvalues ascending (slowest first), eigenvectors normalised w' B w = 1.
// Statistics: hand-written HIP kernel, fp64 accumulation (MDP accumulates in float64 too), one workgroup per
// (node, sample split), partial sums reduced in a fixed order (bit-reproducible); nodes of <= 16 inputs on the fp64 matrix
// cores (v_mfma_f64_16x16x4_f64).  Solve: a hand-written one-wave-per-node kernel for nodes of <= 16 inputs (Cholesky of B,
// C = L^-1 A L^-T, cyclic Jacobi, back-substitution, sort: < 0.2 ms for 1024 nodes), rocSOLVER dsygvj for wider nodes
// (HIGSFA_SYGVJ / HIGSFA_SYGVD force the library solvers); include/higsfa.h and DESIGN.md §8 say the same.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>
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
    // each thread owns entries e = e_base + tid, + 256, ... of the d x d matrices: 16 per pass over the samples, so nodes of up
    // to 64 inputs take one pass and wider ones (up to 128: the upper layers of an 11-layer net) up to four
    double* out = partial + ((size_t)split * n_nodes + node) * (size_t)(d + 2 * dd);
    for (int e_base = 0; e_base < dd; e_base += 16 * 256) {
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
                const int e = e_base + tid + k * 256;
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
            if (e_base == 0 && tid < d)
                for (int tt = 0; tt < m; ++tt) sx += xs[tt * d + tid];
        }
        if (e_base == 0 && tid < d) out[tid] = sx;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = e_base + tid + k * 256;
            if (e < dd) {
                out[d + e] = sxx[k];
                out[d + dd + e] = sdd[k];
            }
        }
    }
}

// Nodes of up to 16 inputs (the first layer: 4 x 4 fields) on the fp64 matrix cores.  With A = X' and B = X
// (X = 4 samples x 16 inputs) the two operands of v_mfma_f64_16x16x4_f64 are the SAME register: lane
// (k = lane / 16, i = lane % 16) holds x[t + k][input i], and one instruction adds the 16 x 16 outer-product
// sum of four samples.  A workgroup owns a chunk of 16 consecutive nodes (two per wave) and a slice of the
// samples; the chunk's distinct input columns are staged once per 32/64 samples into LDS with row-segment
// reads (adjacent nodes read adjacent columns: full lines instead of the 16-byte pieces one node alone sees).
// partial layout as k_sfa_stats; accumulation order is fixed (bit-reproducible).
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename XT, typename ST, int S>
__global__ void __launch_bounds__(512) k_sfa_stats16(const XT* __restrict__ x, int64_t ldx, int64_t n, const int32_t* __restrict__ ucols,
                                                      const int32_t* __restrict__ chunk_off, const int32_t* __restrict__ lidx, int d,
                                                      int n_nodes, int n_chunks, int n_splits, int row_stride, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ST* tile = (ST*)lds_raw;                       // [S + 1][row_stride]
    const int chunk = blockIdx.x % n_chunks, split = blockIdx.x / n_chunks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane >> 4, i = lane & 15;
    const int c0 = chunk_off[chunk], nc = chunk_off[chunk + 1] - c0;
    const int64_t per = (n + n_splits - 1) / n_splits;
    const int64_t t0 = split * per, t1 = min(n, t0 + per);
    int node[2], li[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        node[sl] = chunk * 16 + wave * 2 + sl;
        li[sl] = node[sl] < n_nodes ? lidx[(size_t)node[sl] * 16 + i] : -1;
    }
    f64x4 ax[2], ad[2];
    double sx[2] = {0.0, 0.0};
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) ax[sl] = ad[sl] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int col = tid & 255, r0 = tid >> 8;      // loader: one column, rows r0, r0 + 2, ..
    const int64_t src_col = col < nc ? ucols[c0 + col] : 0;
    for (int64_t tb = t0; tb < t1; tb += S) {
        const int m = (int)min<int64_t>(S, t1 - tb);
        const int mm = (tb + m < n) ? m + 1 : m;    // one sample ahead for the last difference of the block
        const int nd = mm - 1;                      // differences x[t+1] - x[t] that START in this block
        __syncthreads();
        if (col < nc)
            for (int r = r0; r < mm; r += 2) tile[r * row_stride + col] = (ST)x[(tb + r) * ldx + src_col];
        __syncthreads();
        for (int q = 0; q < m; q += 4) {
            const int tt = q + k;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                double v = 0.0, dv = 0.0;
                if (li[sl] >= 0) {
                    if (tt < m) v = (double)tile[tt * row_stride + li[sl]];
                    if (tt < nd) dv = (double)tile[(tt + 1) * row_stride + li[sl]] - v;
                }
                ax[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, ax[sl], 0, 0, 0);
                ad[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(dv, dv, ad[sl], 0, 0, 0);
                sx[sl] += v;
            }
        }
    }
    const int dd = d * d;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        double tot = sx[sl];
        tot += __shfl_xor(tot, 16);
        tot += __shfl_xor(tot, 32);
        if (node[sl] >= n_nodes) continue;
        double* out = partial + ((size_t)split * n_nodes + node[sl]) * (size_t)(d + 2 * dd);
        if (k == 0 && i < d) out[i] = tot;
#pragma unroll
        for (int r = 0; r < 4; ++r) {               // C/D of the f64 form: col = lane & 15, row = lane / 16 + 4 r
            const int row = k + 4 * r;
            if (row < d && i < d) {
                out[d + row * d + i] = ax[sl][r];
                out[d + dd + row * d + i] = ad[sl][r];
            }
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

// Generalized symmetric-definite eigenproblem A w = lambda B w for nodes of up to 16 inputs, one wave per node,
// everything in LDS in float64:  B = L L' (Cholesky),  C = L^-1 A L^-T,  C = V diag(lambda) V' by cyclic Jacobi
// with the round-robin ordering (8 disjoint rotations per step, 15 steps per sweep),  W = L^-T V,  columns
// sorted by ascending eigenvalue.  W' B W = V' V = I by construction.  Matrices narrower than 16 are padded
// with a decoupled diagonal block of huge eigenvalues (no rotation ever mixes it in; it sorts to the end).
// rocSOLVER's batched dsygvj needs 8 ms for 1024 such 16 x 16 problems, dsygvd 31 ms; this takes < 0.2 ms.
// A, B: [node][d*d] symmetric (row- or column-major alike).  evecs: column-major per node like LAPACK.
constexpr int kJ = 16, kJS = 17;   // padded size and LDS row stride (doubles)

__global__ void __launch_bounds__(64) k_sygv16(const double* __restrict__ A, const double* __restrict__ B, int d, double* __restrict__ evals,
                                               double* __restrict__ evecs, int* __restrict__ info) {
    __shared__ double L[kJ * kJS], Cm[kJ * kJS], V[kJ * kJS], X[kJ * kJS];
    __shared__ double rc[8], rs[8], lam[kJ];
    __shared__ int rp[8], rq[8], ord[kJ], bad;
    const int node = blockIdx.x, tid = threadIdx.x, dd = d * d;
    const double* An = A + (size_t)node * dd;
    const double* Bn = B + (size_t)node * dd;
    if (tid == 0) bad = 0;
    // load (padding: identity in B, a huge decoupled diagonal in A)
    for (int e = tid; e < kJ * kJ; e += 64) {
        const int i = e >> 4, j = e & 15;
        const bool in = i < d && j < d;
        L[i * kJS + j] = in ? Bn[i * d + j] : (i == j ? 1.0 : 0.0);
        Cm[i * kJS + j] = in ? An[i * d + j] : (i == j ? 1e280 * (1.0 + i) : 0.0);
        V[i * kJS + j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    // ---- Cholesky, lower triangle of L in place (right-looking)
    for (int j = 0; j < kJ; ++j) {
        if (tid == 0) {
            const double djj = L[j * kJS + j];
            if (!(djj > 0.0)) bad = j + 1;
            L[j * kJS + j] = sqrt(djj > 0.0 ? djj : 1.0);
        }
        __syncthreads();
        const double ljj = L[j * kJS + j];
        if (tid > j && tid < kJ) L[tid * kJS + j] /= ljj;
        __syncthreads();
        for (int e = tid; e < kJ * kJ; e += 64) {
            const int i = e >> 4, k = e & 15;
            if (i > j && k > j && k <= i) L[i * kJS + k] -= L[i * kJS + j] * L[k * kJS + j];
        }
        __syncthreads();
    }
    // ---- X = L^-1 A (forward substitution, one column per lane), then C = L^-1 X' (A symmetric => C = L^-1 A L^-T)
    for (int pass = 0; pass < 2; ++pass) {
        if (tid < kJ) {
            const int c = tid;
            double x[kJ];
#pragma unroll
            for (int i = 0; i < kJ; ++i) {
                double v = pass == 0 ? Cm[i * kJS + c] : X[c * kJS + i];
                for (int k = 0; k < i; ++k) v -= L[i * kJS + k] * x[k];
                x[i] = v / L[i * kJS + i];
            }
#pragma unroll
            for (int i = 0; i < kJ; ++i) (pass == 0 ? X : Cm)[i * kJS + c] = x[i];
        }
        __syncthreads();
    }
    // symmetrise (the two passes leave rounding-level asymmetry)
    for (int e = tid; e < kJ * kJ; e += 64) {
        const int i = e >> 4, j = e & 15;
        if (i < j) {
            const double m = 0.5 * (Cm[i * kJS + j] + Cm[j * kJS + i]);
            X[i * kJS + j] = m;
            X[j * kJS + i] = m;
        } else if (i == j) {
            X[i * kJS + i] = Cm[i * kJS + i];
        }
    }
    __syncthreads();
    for (int e = tid; e < kJ * kJ; e += 64) Cm[(e >> 4) * kJS + (e & 15)] = X[(e >> 4) * kJS + (e & 15)];
    __syncthreads();
    // ---- cyclic Jacobi, round-robin pairs
    for (int sweep = 0; sweep < 30; ++sweep) {
        // convergence: off-diagonal mass of the real block against its diagonal
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < kJ * kJ; e += 64) {
            const int i = e >> 4, j = e & 15;
            if (i < d && j < d) {
                const double v = Cm[i * kJS + j];
                if (i == j) dia += v * v; else off += v * v;
            }
        }
        for (int m = 32; m >= 1; m >>= 1) {
            off += __shfl_xor(off, m);
            dia += __shfl_xor(dia, m);
        }
        if (off <= 1e-30 * dia || off == 0.0) break;
        for (int step = 0; step < kJ - 1; ++step) {
            if (tid < 8) {
                const int k = tid;
                int p = k == 0 ? kJ - 1 : (step + k) % (kJ - 1);
                int q = (step - k + (kJ - 1)) % (kJ - 1);
                if (p > q) { const int t2 = p; p = q; q = t2; }
                const double apq = Cm[p * kJS + q], app = Cm[p * kJS + p], aqq = Cm[q * kJS + q];
                double c = 1.0, sn = 0.0;
                if (apq != 0.0 && fabs(apq) > 1e-300) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    sn = t * c;
                }
                rc[k] = c;
                rs[k] = sn;
                rp[k] = p;
                rq[k] = q;
            }
            __syncthreads();
            // rows: (row p, row q) <- (c p - s q, s p + c q)
            for (int e = tid; e < 8 * kJ; e += 64) {
                const int k = e >> 4, j = e & 15, p = rp[k], q = rq[k];
                const double c = rc[k], sn = rs[k], a = Cm[p * kJS + j], b = Cm[q * kJS + j];
                Cm[p * kJS + j] = c * a - sn * b;
                Cm[q * kJS + j] = sn * a + c * b;
            }
            __syncthreads();
            // columns of C and of V
            for (int e = tid; e < 8 * kJ; e += 64) {
                const int k = e >> 4, i = e & 15, p = rp[k], q = rq[k];
                const double c = rc[k], sn = rs[k];
                double a = Cm[i * kJS + p], b = Cm[i * kJS + q];
                Cm[i * kJS + p] = c * a - sn * b;
                Cm[i * kJS + q] = sn * a + c * b;
                a = V[i * kJS + p];
                b = V[i * kJS + q];
                V[i * kJS + p] = c * a - sn * b;
                V[i * kJS + q] = sn * a + c * b;
            }
            __syncthreads();
        }
    }
    // ---- W = L^-T V (back substitution, one column per lane), eigenvalues, ascending order
    if (tid < kJ) {
        const int c = tid;
        double x[kJ];
#pragma unroll
        for (int i = kJ - 1; i >= 0; --i) {
            double v = V[i * kJS + c];
            for (int k = i + 1; k < kJ; ++k) v -= L[k * kJS + i] * x[k];
            x[i] = v / L[i * kJS + i];
        }
#pragma unroll
        for (int i = 0; i < kJ; ++i) X[i * kJS + c] = x[i];
        lam[c] = Cm[c * kJS + c];
    }
    __syncthreads();
    if (tid < kJ) {
        int rank = 0;
        for (int j = 0; j < kJ; ++j) rank += (lam[j] < lam[tid] || (lam[j] == lam[tid] && j < tid)) ? 1 : 0;
        ord[rank] = tid;
    }
    __syncthreads();
    if (tid < d) evals[(size_t)node * d + tid] = lam[ord[tid]];
    for (int e = tid; e < dd; e += 64) {
        const int col = e / d, row = e - col * d;
        evecs[(size_t)node * dd + e] = X[row * kJS + ord[col]];
    }
    if (tid == 0) info[node] = bad;
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

__global__ void k_fill_identity(double* __restrict__ m, int d, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int r = (int)(e % ((int64_t)d * d));
    m[e] = (r / d == r % d) ? 1.0 : 0.0;
}

// out[t, node * width + f * p + j] = func_f(z_j),  z = (x[t, conn[node]] - mean[node]) W[node]   — float64 throughout: the data a
// layer passes to the statistics of the next one during training (MDP trains in float64).  One thread per (row, node, j).
template <typename XT>
__global__ void __launch_bounds__(256) k_train_apply(const XT* __restrict__ x, int64_t ldx, int64_t n, const int32_t* __restrict__ conn, int d,
                                                      int n_nodes, const double* __restrict__ mean, const double* __restrict__ W, int p, int nf,
                                                      const int* __restrict__ kinds, const double* __restrict__ expos, double* __restrict__ out,
                                                      int64_t ldo) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n * n_nodes * p;
    if (id >= total) return;
    const int j = (int)(id % p);
    const int node = (int)((id / p) % n_nodes);
    const int64_t t = id / ((int64_t)p * n_nodes);
    const int32_t* cols = conn + (size_t)node * d;
    const double* mu = mean + (size_t)node * d;
    const double* Wn = W + (size_t)node * d * p;
    const XT* xr = x + t * ldx;
    double z = 0.0;
    for (int i = 0; i < d; ++i) z = __fma_rn((double)xr[cols[i]] - mu[i], Wn[(size_t)i * p + j], z);
    double* o = out + t * ldo + (size_t)node * (nf > 0 ? nf : 1) * p;
    if (nf <= 0) {
        o[j] = z;
        return;
    }
    for (int f = 0; f < nf; ++f) {
        double v = z;
        if (kinds[f] == (int)hg::E_ABS_POW) v = pow(fabs(z), expos[f]);
        else if (kinds[f] == (int)hg::E_SIGNED_POW) v = copysign(pow(fabs(z), expos[f]), z);
        o[(size_t)f * p + j] = v;
    }
}

int train_layer_impl(const void* x_in, int x_on_host, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host, int32_t n_nodes,
                     int32_t d, int mode, int device, double* evals_host, double* evecs_host, double* mean_host, double* timings_ms) {
    return guarded([&] {
        if (mode != 0 && mode != 1) hg::fail(HG_ERR_ARG, "mode must be 0 (SFA) or 1 (PCA)");
        if (!x_in || !conn_host || !evals_host || !evecs_host || !mean_host) hg::fail(HG_ERR_ARG, "null pointer");
        if (n < 3 || n_nodes < 1 || d < 1 || d > 128) hg::fail(HG_ERR_ARG, "need n >= 3 samples, 1..128 inputs per node");
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
        // d <= 16: matrix-core kernel over chunks of 16 nodes; plan the chunks' distinct columns on the host
        const bool mfma16 = d <= 16;
        const int n_chunks = (n_nodes + 15) / 16;
        std::vector<int32_t> ucols, chunk_off(1, 0), lidx;
        int max_nc = 0;
        if (mfma16) {
            lidx.assign((size_t)n_nodes * 16, -1);
            for (int c = 0; c < n_chunks; ++c) {
                const int k0 = c * 16, k1 = std::min(n_nodes, k0 + 16);
                std::vector<int32_t> u(conn_host + (size_t)k0 * d, conn_host + (size_t)k1 * d);
                std::sort(u.begin(), u.end());
                u.erase(std::unique(u.begin(), u.end()), u.end());
                for (int k = k0; k < k1; ++k)
                    for (int i = 0; i < d; ++i)
                        lidx[(size_t)k * 16 + i] = (int32_t)(std::lower_bound(u.begin(), u.end(), conn_host[(size_t)k * d + i]) - u.begin());
                ucols.insert(ucols.end(), u.begin(), u.end());
                chunk_off.push_back((int32_t)ucols.size());
                max_nc = std::max(max_nc, (int)u.size());
            }
        }
        int n_splits;
        if (mfma16) n_splits = (int)std::max<int64_t>(1, std::min<int64_t>(64, (1024 + n_chunks - 1) / n_chunks));
        else n_splits = (int)std::max<int64_t>(1, std::min<int64_t>(64, (2048 + n_nodes - 1) / n_nodes));
        n_splits = (int)std::min<int64_t>(n_splits, std::max<int64_t>(1, n / 256));
        hg::DevBuf conn, partial, mean, A, B, W, E, info, d_ucols, d_choff, d_lidx;
        conn.upload(conn_host, (size_t)n_nodes * d * 4);
        if (mfma16) {
            d_ucols.upload(ucols.data(), ucols.size() * 4);
            d_choff.upload(chunk_off.data(), chunk_off.size() * 4);
            d_lidx.upload(lidx.data(), lidx.size() * 4);
        }
        partial.alloc((size_t)n_splits * n_nodes * (d + 2 * dd) * 8);
        mean.alloc((size_t)n_nodes * d * 8);
        A.alloc((size_t)n_nodes * dd * 8);
        B.alloc((size_t)n_nodes * dd * 8);
        W.alloc((size_t)n_nodes * d * 8);
        E.alloc((size_t)n_nodes * d * 8);
        info.alloc((size_t)n_nodes * 4);
        hg::DevBuf resid, sweeps;       // sygvj work outputs
        resid.alloc((size_t)n_nodes * 8);
        sweeps.alloc((size_t)n_nodes * 4);
        hipEvent_t e0, e1, e2;
        HG_HIP(hipEventCreate(&e0));
        HG_HIP(hipEventCreate(&e1));
        HG_HIP(hipEventCreate(&e2));
        HG_HIP(hipEventRecord(e0, nullptr));
        const size_t lds = (size_t)(kTS + 1) * d * 8;
        const unsigned grid = (unsigned)(n_nodes * n_splits);
        if (mfma16) {
            // row stride: odd number of words, so the four sample rows of a lane group start on different banks
            const int rs = (max_nc | 1) + 2;
            const unsigned g16 = (unsigned)(n_chunks * n_splits);
            const int32_t *uc = (const int32_t*)d_ucols.p, *co = (const int32_t*)d_choff.p, *li = (const int32_t*)d_lidx.p;
            double* pp = (double*)partial.p;
            if (x_dtype == HG_U8) {
                auto fn = k_sfa_stats16<uint8_t, float, 64>;
                const size_t l2 = (size_t)65 * rs * 4;
                if (l2 > 64 * 1024) HG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
                hipLaunchKernelGGL(fn, g16, 512, l2, nullptr, (const uint8_t*)x_dev, ldx, n, uc, co, li, d, n_nodes, n_chunks, n_splits, rs, pp);
            } else if (x_dtype == HG_F32) {
                auto fn = k_sfa_stats16<float, float, 64>;
                const size_t l2 = (size_t)65 * rs * 4;
                if (l2 > 64 * 1024) HG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
                hipLaunchKernelGGL(fn, g16, 512, l2, nullptr, (const float*)x_dev, ldx, n, uc, co, li, d, n_nodes, n_chunks, n_splits, rs, pp);
            } else {
                auto fn = k_sfa_stats16<double, double, 32>;
                const size_t l2 = (size_t)33 * rs * 8;
                if (l2 > 64 * 1024) HG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2));
                hipLaunchKernelGGL(fn, g16, 512, l2, nullptr, (const double*)x_dev, ldx, n, uc, co, li, d, n_nodes, n_chunks, n_splits, rs, pp);
            }
        } else if (lds > 64 * 1024 && (hipFuncSetAttribute((const void*)k_sfa_stats<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                                       hipFuncSetAttribute((const void*)k_sfa_stats<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                                       hipFuncSetAttribute((const void*)k_sfa_stats<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)) {
            hg::fail(HG_ERR_DEVICE, "cannot raise the LDS limit of the statistics kernel");
        } else if (x_dtype == HG_U8)
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
        // Solver: nodes of <= 16 inputs go to the hand-written one-wave-per-node Jacobi kernel; wider nodes (or
        // HIGSFA_SYGVJ=1 / HIGSFA_SYGVD=1) to rocSOLVER's batched Jacobi / divide-and-conquer routines.
        if (mode == 1) {      // PCA: eigen-decomposition of the covariance itself = the same solver on (Cov, I)
            HG_HIP(hipMemcpyAsync(A.p, B.p, (size_t)n_nodes * dd * 8, hipMemcpyDeviceToDevice, nullptr));
            const int64_t total = (int64_t)n_nodes * dd;
            hipLaunchKernelGGL(k_fill_identity, (unsigned)((total + 255) / 256), 256, 0, nullptr, (double*)B.p, d, total);
        }
        rocblas_status rs = rocblas_status_success;
        hg::DevBuf Wv;     // eigenvectors of the hand-written solver (rocSOLVER overwrites A instead)
        static const int solver_env = getenv("HIGSFA_SYGVD") ? 2 : getenv("HIGSFA_SYGVJ") ? 1 : 0;   // read once per process
        const bool own_solver = d <= 16 && solver_env == 0;
        rocblas_handle h = nullptr;
        if (!own_solver && rocblas_create_handle(&h) != rocblas_status_success) hg::fail(HG_ERR_DEVICE, "rocblas_create_handle failed");
        HG_HIP(hipEventRecord(e1, nullptr));
        if (own_solver) {
            Wv.alloc((size_t)n_nodes * dd * 8);
            hipLaunchKernelGGL(k_sygv16, (unsigned)n_nodes, 64, 0, nullptr, (const double*)A.p, (const double*)B.p, d, (double*)W.p,
                               (double*)Wv.p, (int*)info.p);
            HG_HIP(hipGetLastError());
        } else if (solver_env == 2) {
            rs = rocsolver_dsygvd_strided_batched(h, rocblas_eform_ax, rocblas_evect_original, rocblas_fill_upper, d, (double*)A.p, d, dd,
                                                  (double*)B.p, d, dd, (double*)W.p, d, (double*)E.p, d, (rocblas_int*)info.p, n_nodes);
        } else {
            rs = rocsolver_dsygvj_strided_batched(h, rocblas_eform_ax, rocblas_evect_original, rocblas_fill_upper, d, (double*)A.p, d, dd,
                                                  (double*)B.p, d, dd, 0.0, (double*)resid.p, 100, (rocblas_int*)sweeps.p, (double*)W.p, d,
                                                  (rocblas_int*)info.p, n_nodes);
        }
        HG_HIP(hipEventRecord(e2, nullptr));
        HG_HIP(hipDeviceSynchronize());
        if (h) rocblas_destroy_handle(h);
        if (rs != rocblas_status_success) hg::fail(HG_ERR_DEVICE, "rocSOLVER generalized eigen-solve failed (%d)", (int)rs);
        std::vector<int> hinfo(n_nodes);
        HG_HIP(hipMemcpy(hinfo.data(), info.p, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < n_nodes; ++k)
            if (hinfo[k] != 0) hg::fail(HG_ERR_STATE, "node %d: eigen-solve info = %d (covariance not positive definite or no convergence)", k, hinfo[k]);
        HG_HIP(hipMemcpy(evals_host, W.p, (size_t)n_nodes * d * 8, hipMemcpyDeviceToHost));
        HG_HIP(hipMemcpy(evecs_host, own_solver ? Wv.p : A.p, (size_t)n_nodes * dd * 8, hipMemcpyDeviceToHost));   // column-major (LAPACK) per node
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

}  // namespace

extern "C" {

int hg_sfa_train_layer(const void* x, int x_on_host, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host, int32_t n_nodes, int32_t d,
                       int device, double* evals_host, double* evecs_host, double* mean_host, double* timings_ms) {
    return train_layer_impl(x, x_on_host, x_dtype, n, ldx, conn_host, n_nodes, d, 0, device, evals_host, evecs_host, mean_host, timings_ms);
}

int hg_pca_train_layer(const void* x, int x_on_host, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host, int32_t n_nodes, int32_t d,
                       int device, double* evals_host, double* evecs_host, double* mean_host, double* timings_ms) {
    return train_layer_impl(x, x_on_host, x_dtype, n, ldx, conn_host, n_nodes, d, 1, device, evals_host, evecs_host, mean_host, timings_ms);
}

int hg_train_apply_device(const void* x_dev, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host, int32_t n_nodes, int32_t d,
                          const double* mean_host, const double* w_host, int32_t p, int32_t n_funcs, const int32_t* func_kinds,
                          const double* func_expos, double* out_dev, int64_t ldo, int device) {
    return guarded([&] {
        if (!x_dev || !conn_host || !mean_host || !w_host || !out_dev) hg::fail(HG_ERR_ARG, "null pointer");
        if (n < 1 || n_nodes < 1 || d < 1 || p < 1 || n_funcs < 0 || n_funcs > 8) hg::fail(HG_ERR_ARG, "bad sizes");
        if (x_dtype != HG_U8 && x_dtype != HG_F32 && x_dtype != HG_F64) hg::fail(HG_ERR_ARG, "bad dtype");
        if (ldo < (int64_t)n_nodes * std::max(1, n_funcs) * p) hg::fail(HG_ERR_ARG, "ldo too small");
        for (int f = 0; f < n_funcs; ++f)
            if (!func_kinds || func_kinds[f] < 0 || func_kinds[f] > (int)hg::E_SIGNED_POW || (func_kinds[f] != 0 && !func_expos))
                hg::fail(HG_ERR_ARG, "only element-wise expansion functions");
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
        if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range", device);
        HG_HIP(hipSetDevice(device));
        for (int64_t i = 0; i < (int64_t)n_nodes * d; ++i)
            if (conn_host[i] < 0 || conn_host[i] >= ldx) hg::fail(HG_ERR_ARG, "connection %d out of range", conn_host[i]);
        hg::DevBuf conn, mean, W, kinds, expos;
        conn.upload(conn_host, (size_t)n_nodes * d * 4);
        mean.upload(mean_host, (size_t)n_nodes * d * 8);
        W.upload(w_host, (size_t)n_nodes * d * p * 8);
        if (n_funcs) {
            kinds.upload(func_kinds, (size_t)n_funcs * 4);
            std::vector<double> ex(n_funcs, 1.0);
            for (int f = 0; f < n_funcs; ++f)
                if (func_expos) ex[f] = func_expos[f];
            expos.upload(ex.data(), (size_t)n_funcs * 8);
        }
        const int64_t total = n * n_nodes * p;
        if ((total + 255) / 256 > 0x7fffffffll) hg::fail(HG_ERR_ARG, "too much work for one launch");
        const unsigned grid = (unsigned)((total + 255) / 256);
#define HG_APPLY(XT) hipLaunchKernelGGL(k_train_apply<XT>, grid, 256, 0, nullptr, (const XT*)x_dev, ldx, n, (const int32_t*)conn.p, d, n_nodes, \
                                        (const double*)mean.p, (const double*)W.p, p, n_funcs, (const int*)kinds.p, (const double*)expos.p, out_dev, ldo)
        if (x_dtype == HG_U8) HG_APPLY(uint8_t);
        else if (x_dtype == HG_F32) HG_APPLY(float);
        else HG_APPLY(double);
#undef HG_APPLY
        HG_HIP(hipGetLastError());
        HG_HIP(hipDeviceSynchronize());
    });
}

}  // extern "C"
