/* TEST ORACLE — NOT PRODUCT CODE.  *** parity unpinned *** (see oracle/mdp_restate.py header)
 *
 * Second, independently written CPU restatement (plain C, float64, straightforward loops) of
 * the leaf arithmetic on the reference's hot path
 *     sl = networks[k].execute(subimages_arr, benchmark=benchmark)   (FaceDetectUpdated.py:699)
 * The arithmetic itself lives in mdp-toolkit / cuicuilco (absent from /root/reference,
 * SURVEY.md §0 F2); each function restates the published behaviour of one node class and
 * cites the SURVEY.md §8a row it follows.  oracle/ref_c.py walks a flow and calls these; tests
 * require it to agree with the numpy restatement to 1e-12.  Nothing here is linked into the
 * product library.
 */
#include <math.h>
#include <stddef.h>

/* mdp.hinet.Switchboard._execute: y = x[:, connections]    */
void ref_gather_f64(const double* x, long n, long ldx, const long* idx, long m, double* y, long ldy) {
    for (long r = 0; r < n; ++r)
        for (long c = 0; c < m; ++c) y[r * ldy + c] = x[r * ldx + idx[c]];
}

/* PCANode / WhiteningNode / SFANode / GSFANode / LinearRegressionNode in the common form
 * y = (x - a) W + b   (rows a5, a7).  W is d_in x d_out row-major. */
void ref_affine_f64(const double* x, long n, long ldx, long d_in, const double* a, const double* W, const double* b,
                    long d_out, double* y, long ldy) {
    for (long r = 0; r < n; ++r)
        for (long j = 0; j < d_out; ++j) {
            double acc = 0.0;
            for (long k = 0; k < d_in; ++k) acc += (x[r * ldx + k] - a[k]) * W[k * d_out + j];
            y[r * ldy + j] = acc + b[j];
        }
}

/* One cuicuilco.nonlinear_expansion function (row a6).  kind: 0 identity, 1 |x|^p,
 * 2 sign(x)|x|^p, 3 x_i x_j (i<=j, i-major), 4 x_i x_{i+k}, 5 x_i x_{i+off} for off = 0..k-1 (offset-major; the
 * other reading of pair_prodsadj{k}_ex).  d = columns actually used.
 * Returns the number of output columns written starting at y[.., 0]. */
long ref_expfunc_f64(const double* x, long n, long ldx, long d, int kind, double expo, long k, double* y, long ldy) {
    long m = 0;
    switch (kind) {
        case 0: m = d; break;
        case 1: m = d; break;
        case 2: m = d; break;
        case 3: m = d * (d + 1) / 2; break;
        case 4: m = d - k > 0 ? d - k : 0; break;
        case 5: for (long off = 0; off < k && off < d; ++off) m += d - off; break;
        default: return -1;
    }
    for (long r = 0; r < n; ++r) {
        const double* xr = x + r * ldx;
        double* yr = y + r * ldy;
        long o = 0;
        switch (kind) {
            case 0: for (long i = 0; i < d; ++i) yr[o++] = xr[i]; break;
            case 1: for (long i = 0; i < d; ++i) yr[o++] = pow(fabs(xr[i]), expo); break;
            case 2:
                for (long i = 0; i < d; ++i) {
                    double v = pow(fabs(xr[i]), expo);
                    yr[o++] = xr[i] > 0 ? v : (xr[i] < 0 ? -v : 0.0);
                }
                break;
            case 3:
                for (long i = 0; i < d; ++i)
                    for (long j = i; j < d; ++j) yr[o++] = xr[i] * xr[j];
                break;
            case 4: for (long i = 0; i + k < d; ++i) yr[o++] = xr[i] * xr[i + k]; break;
            case 5:
                for (long off = 0; off < k; ++off)
                    for (long i = 0; i + off < d; ++i) yr[o++] = xr[i + off] * xr[i];
                break;
        }
    }
    return m;
}

/* mdp.nodes.GaussianClassifier class posteriors + cuicuilco's regression()  (SURVEY.md §8f-2;
 * FaceDetectUpdated.py:709-719).  MDP form: prob_c = p_c (2 pi)^(-d/2) / sqrtdet_c
 * exp(-1/2 (x-m_c)' S_c^-1 (x-m_c)), normalised over c; here in the log domain. */
void ref_gauss_regression_f64(const double* x, long n, long ldx, long K, long d, const double* means, const double* inv_covs,
                              const double* sqrt_det, const double* prior, const double* avg_labels, double* reg,
                              double* sd) {
    for (long r = 0; r < n; ++r) {
        double lmax = -INFINITY;
        double lp[4096];
        for (long c = 0; c < K; ++c) {
            double q = 0.0;
            for (long i = 0; i < d; ++i) {
                double t = 0.0;
                for (long j = 0; j < d; ++j) t += inv_covs[(c * d + i) * d + j] * (x[r * ldx + j] - means[c * d + j]);
                q += t * (x[r * ldx + i] - means[c * d + i]);
            }
            lp[c] = log(prior[c]) - log(sqrt_det[c]) - 0.5 * q;
            if (lp[c] > lmax) lmax = lp[c];
        }
        double sw = 0, swa = 0, swa2 = 0;
        for (long c = 0; c < K; ++c) {
            double w = exp(lp[c] - lmax);
            sw += w;
            swa += w * avg_labels[c];
            swa2 += w * avg_labels[c] * avg_labels[c];
        }
        reg[r] = swa / sw;
        if (sd) {
            double v = swa2 / sw - reg[r] * reg[r];
            sd[r] = v > 0 ? sqrt(v) : 0.0;
        }
    }
}
