/* TEST ORACLE — NOT PRODUCT CODE.  *** parity unpinned *** (see oracle/mdp_restate.py header)
 *
 * "Good CPU" point for bench.py's cpu_baseline leg (SURVEY.md §8d "CPU baseline beside it": the
 * MDP-structured numpy restatement is what the reference does; this is what a careful CPU
 * implementation of the same flow would do).  float64, same arithmetic as oracle/ref_c.c, but the
 * whole flow runs from one flat op list (built by oracle/fast_cpu.py) over cache-sized row chunks,
 * rows in parallel with OpenMP, inner loops written so that gcc vectorises them.
 * Ops follow SURVEY.md §8a rows a3 (gather), a5/a7 (affine), a6 (expansion).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    long kind;          /* 0 gather, 1 affine, 2 expand(identity), 3 expand(|x|^p), 4 expand(sgn|x|^p) */
    long src, src_off;  /* buffer id, column offset */
    long dst, dst_off;
    long d_in, d_out;
    long p0;            /* offset into the double pool (affine: a[d_in], W[d_in*d_out], b[d_out]) or the long pool (gather) */
    double expo;
} fc_op;

#define FC_CHUNK 16

/* |x|^p over a row: lives in fast_cpu_pow.c, compiled with -ffast-math so that gcc calls glibc's vector
 * pow (libmvec, <= 4 ulp); libm's scalar pow would otherwise be most of the whole flow's CPU time. */
void fc_abs_pow_row(const double* s, double* d, long n, double expo);
#define abs_pow_row fc_abs_pow_row

/* y[r][:] = b + (x[r][:] - a) W for 4 rows at a time: every W row is loaded once per 4 rows and the
 * j loops vectorise (d_out <= FC_MAXOUT keeps the accumulators on the stack, in L1). */
#define FC_MAXOUT 256
static void affine_rows(const double* S, long ws, double* D, long wd, long rows, long d_in, long d_out,
                        const double* a, const double* W, const double* b) {
    long r = 0;
    if (d_out <= FC_MAXOUT)
        for (; r + 4 <= rows; r += 4) {
            double y0[FC_MAXOUT], y1[FC_MAXOUT], y2[FC_MAXOUT], y3[FC_MAXOUT];
            for (long j = 0; j < d_out; ++j) y0[j] = y1[j] = y2[j] = y3[j] = b[j];
            const double *x0 = S + r * ws, *x1 = x0 + ws, *x2 = x1 + ws, *x3 = x2 + ws;
            for (long k = 0; k < d_in; ++k) {
                const double v0 = x0[k] - a[k], v1 = x1[k] - a[k], v2 = x2[k] - a[k], v3 = x3[k] - a[k];
                const double* w = W + k * d_out;
                for (long j = 0; j < d_out; ++j) {
                    y0[j] += v0 * w[j];
                    y1[j] += v1 * w[j];
                    y2[j] += v2 * w[j];
                    y3[j] += v3 * w[j];
                }
            }
            memcpy(D + r * wd, y0, sizeof(double) * (size_t)d_out);
            memcpy(D + (r + 1) * wd, y1, sizeof(double) * (size_t)d_out);
            memcpy(D + (r + 2) * wd, y2, sizeof(double) * (size_t)d_out);
            memcpy(D + (r + 3) * wd, y3, sizeof(double) * (size_t)d_out);
        }
    for (; r < rows; ++r) {
        double* y = D + r * wd;
        for (long j = 0; j < d_out; ++j) y[j] = b[j];
        for (long k = 0; k < d_in; ++k) {
            const double xv = S[r * ws + k] - a[k];
            const double* w = W + k * d_out;
            for (long j = 0; j < d_out; ++j) y[j] += xv * w[j];
        }
    }
}

static void run_chunk(const fc_op* ops, long n_ops, const double* dpool, const long* lpool, double** buf,
                      const long* width, long rows) {
    for (long i = 0; i < n_ops; ++i) {
        const fc_op* o = &ops[i];
        const long ws = width[o->src], wd = width[o->dst];
        const double* S = buf[o->src] + o->src_off;
        double* D = buf[o->dst] + o->dst_off;
        switch (o->kind) {
            case 0: {
                const long* idx = lpool + o->p0;
                for (long r = 0; r < rows; ++r)
                    for (long c = 0; c < o->d_out; ++c) D[r * wd + c] = S[r * ws + idx[c]];
            } break;
            case 1: {
                const double* a = dpool + o->p0;
                const double* W = a + o->d_in;
                const double* b = W + o->d_in * o->d_out;
                affine_rows(S, ws, D, wd, rows, o->d_in, o->d_out, a, W, b);
            } break;
            case 2:
                for (long r = 0; r < rows; ++r) memcpy(D + r * wd, S + r * ws, sizeof(double) * (size_t)o->d_in);
                break;
            case 3:
                for (long r = 0; r < rows; ++r) abs_pow_row(S + r * ws, D + r * wd, o->d_in, o->expo);
                break;
            case 4:
                for (long r = 0; r < rows; ++r) {
                    abs_pow_row(S + r * ws, D + r * wd, o->d_in, o->expo);
                    for (long c = 0; c < o->d_in; ++c) {
                        const double x = S[r * ws + c];
                        D[r * wd + c] = x > 0 ? D[r * wd + c] : (x < 0 ? -D[r * wd + c] : 0.0);
                    }
                }
                break;
        }
    }
}

/* x: n x width[0] input rows (copied into buffer 0 chunk by chunk); result rows are read from buffer
 * `out_buf`, columns [0, y_cols).  Returns 0, or -1 when a scratch allocation fails. */
int fc_run(const fc_op* ops, long n_ops, const double* dpool, const long* lpool, const long* width, long n_buf,
           long out_buf, const double* x, long n, double* y, long y_cols, int threads) {
    int bad = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double** buf = (double**)calloc((size_t)n_buf, sizeof(double*));
        int ok = buf != NULL;
        for (long b = 0; ok && b < n_buf; ++b) {
            buf[b] = (double*)malloc(sizeof(double) * (size_t)width[b] * FC_CHUNK);
            ok = buf[b] != NULL;
        }
        if (!ok) {
#pragma omp atomic write
            bad = 1;
        }
#pragma omp barrier
        if (!bad) {
#pragma omp for schedule(dynamic, 1)
            for (long r0 = 0; r0 < n; r0 += FC_CHUNK) {
                const long rows = n - r0 < FC_CHUNK ? n - r0 : FC_CHUNK;
                memcpy(buf[0], x + r0 * width[0], sizeof(double) * (size_t)(rows * width[0]));
                run_chunk(ops, n_ops, dpool, lpool, buf, width, rows);
                for (long r = 0; r < rows; ++r)
                    memcpy(y + (r0 + r) * y_cols, buf[out_buf] + r * width[out_buf], sizeof(double) * (size_t)y_cols);
            }
        }
        if (buf) {
            for (long b = 0; b < n_buf; ++b) free(buf[b]);
            free(buf);
        }
    }
    return bad ? -1 : 0;
}
