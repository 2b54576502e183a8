// On-device sub-image extraction: the producer of the hot call's input,
//   subimages_arr = load_network_subimages(images, ..., curr_subimage_coordinates, curr_angles, w, h, NEAREST)
//                                                        (FaceDetectUpdated.py:686; face_analysis.py:775-800)
// which the reference does per patch with PIL: Image.transform((w, h), Image.EXTENT, (x0, y0, x1, y1),
// Image.NEAREST) inside cuicuilco.image_loader.extract_subimages_rotate.  This restates PIL's
// EXTENT/NEAREST index rule (ImagingScaleAffine) exactly, including its additive accumulation of the
// source coordinate in double precision:
//     a = (x1 - x0) / w;  xo = x0 + a/2;  for x in 0..w-1: xin = xo < 0 ? -1 : (int)xo;  xo += a
// (same for y); source pixels outside the frame leave the output pixel 0.
//
// Rotated windows (delta_ang != 0; the reference passes -1 * curr_angles, face_analysis.py:782, non-zero after the first
// PAng stage).  cuicuilco's own composition is not available ([K]); the rule implemented — and tested bit for bit
// against PIL — is
//     window = frame.rotate(delta_ang, NEAREST, center = ((x0 + x1) / 2, (y0 + y1) / 2)).transform((w, h), EXTENT, box, NEAREST)
// composed per output pixel: the EXTENT tables give the pixel (xr, yr) of the ROTATED frame, and PIL's affine_fixed()
// (Geometry.c) gives that pixel's source: 16.16 fixed-point coefficients A0..A5 from the matrix Image.rotate builds
// (cos / sin of -radians(angle % 360) rounded to 15 decimals), xs = (A2 + yr A1 + xr A0) >> 16, ys likewise.  The integer
// sums are PIL's running sums in closed form.  Multiples of 180 degrees take PIL's "scaling" branch (sin rounds to 0) with
// its floating running sums.  The coefficients are computed on the device in double precision without contraction; the
// one place this can differ from the host's libm is the last bit of cos / sin before the 15-decimal rounding, which
// moves a fixed-point coefficient only if it sits within 7e-11 of a rounding boundary.  Rotated frames must stay inside
// PIL's fixed-point range (|source coordinate| < 32768), otherwise the window is left 0.
#include <hip/hip_runtime.h>

#include "hg_common.hpp"

struct hg_patcher {
    int device = -1;
    hg::DevBuf tabs, boxes, frame, out, rot, angles;
    // hg_patcher_extract_keyed_device: index tables kept per key (boxes the CALLER declares unchanged: the prescale's whole-frame box, the
    // first-stage grid of a frame size) — the table kernel then runs once, not once per frame
    struct Keyed {
        uint64_t key = 0;
        int64_t n = 0;
        int out_w = 0, out_h = 0, frame_w = 0, frame_h = 0;
        hg::DevBuf tabs;
    };
    Keyed keyed[4];
    int keyed_next = 0;
};

namespace hg { void set_last_error(const std::string& s); }

namespace {

// Source index tables, one entry per thread.  PIL steps the source coordinate by repeated addition
// (o += a per output pixel); to land on the same pixel in every case each thread repeats that sum from the start
// of its axis — O(m) dependent adds for the last entry instead of one thread walking all m entries with a
// conversion, two compares and a store per step.
__device__ __forceinline__ int32_t extent_entry(const double* __restrict__ box, int e, int w, int h, int fw, int fh) {
    const int axis = e >= w ? 1 : 0, i = axis ? e - w : e;
    const double lo = box[axis], hi = box[2 + axis];
    const int m = axis ? h : w, lim = axis ? fh : fw;
    const double a = (hi - lo) / m;
    double o = lo + a * 0.5;
    // PIL's running sum, one addition per output pixel before this one (rounded one by one: no contraction, no reassociation);
    // eight per trip, so that the loop's compare-and-branch is not what the chain waits for (the 1562-entry table of the prescale
    // took 26.8 us with one addition per trip)
    int k = 0;
    for (; k + 8 <= i; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) o = __dadd_rn(o, a);
    }
    for (; k < i; ++k) o = __dadd_rn(o, a);
    const int v = o < 0.0 ? -1 : (int)o;
    return (v >= 0 && v < lim) ? v : -1;
}

struct RotCoef;
__device__ __forceinline__ void rot_coef_store(const double* __restrict__ boxes, const double* __restrict__ angs, int64_t b, int fw, int fh, RotCoef* out);

// angs != nullptr: the thread of a box's first table entry also computes the box's rotation coefficients (a launch of its own —
// k_rot_coefs — cost the cascade 4.8 us nine times per frame; here it runs beside the other threads' addition chains)
__global__ void k_extent_tables(const double* __restrict__ boxes, int64_t n, int w, int h, int fw, int fh, int32_t* __restrict__ tabs,
                                const double* __restrict__ angs, RotCoef* __restrict__ rot) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wh = w + h;
    if (id >= n * wh) return;
    const int64_t b = id / wh;
    const int e = (int)(id - b * wh);
    tabs[b * wh + e] = extent_entry(boxes + b * 4, e, w, h, fw, fh);
    if (angs && e == 0) rot_coef_store(boxes, angs, b, fw, fh, rot);
}

// One workgroup per (group of output rows, box): no index divisions, the row's source line and the column table are
// read once; four output pixels per thread and one vector store when the output is uint8.
template <typename FT, typename OT>
__global__ void __launch_bounds__(256) k_extent_gather(const FT* __restrict__ frame, int64_t ld, const int32_t* __restrict__ tabs, int64_t n,
                                                        int w, int h, OT* __restrict__ out, int64_t ldo) {
    const int y = blockIdx.x * blockDim.y + threadIdx.y;     // blockDim.y output rows per workgroup
    if (y >= h) return;
    for (int64_t b = blockIdx.y; b < n; b += gridDim.y) {
        const int32_t* t = tabs + b * (w + h);
        const int ys = t[w + y];
        const FT* src = frame + (int64_t)(ys >= 0 ? ys : 0) * ld;
        OT* dst = out + b * ldo + (int64_t)y * w;
        if constexpr (sizeof(OT) == 1) {
            if ((w & 3) == 0 && (ldo & 3) == 0 && ((uintptr_t)out & 3) == 0) {
                for (int x = threadIdx.x * 4; x < w; x += blockDim.x * 4) {
                    uint32_t pk = 0;
                    int xs4[4];
                    if (((w + h) & 3) == 0) {            // table rows 16-byte aligned: one vector read
                        const int4 v4 = *(const int4*)(t + x);
                        xs4[0] = v4.x; xs4[1] = v4.y; xs4[2] = v4.z; xs4[3] = v4.w;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) xs4[k] = t[x + k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int xs = xs4[k];
                        const uint32_t v = (xs >= 0 && ys >= 0) ? (uint32_t)(uint8_t)(OT)src[xs] : 0u;
                        pk |= v << (8 * k);
                    }
                    *(uint32_t*)(dst + x) = pk;
                }
                continue;
            }
        }
        for (int x = threadIdx.x; x < w; x += blockDim.x) {
            const int xs = t[x];
            dst[x] = (xs >= 0 && ys >= 0) ? (OT)src[xs] : (OT)0;
        }
    }
}

// uint8 frame -> uint8 windows whose rows are a multiple of 16 pixels (the cascade's first stage: 1738 windows of 128 x 128 from a
// 1000 x 562 frame, 28 MB of output).  k_extent_gather above issues, per four pixels, one table load, four byte loads and one
// 4-byte store: 24 memory instructions per 16 pixels, and it is their issue that bounds it (35 us = 0.8 TB/s written).  Here a
// thread owns sixteen output columns of its window for all the rows of its workgroup: the column indices are read once and kept in
// registers; a window at most twice as wide as its output (most of a pyramid: the number of windows falls with the square of
// their size) has the four source pixels of four output pixels inside eight consecutive bytes, so ONE unaligned 8-byte load
// (gfx950 runs in unaligned-access mode: the compiler itself emits global_load_dwordx2 for a byte-aligned 8-byte copy) replaces
// four byte loads; one 16-byte store per row.  5 memory instructions per 16 pixels.  Same bytes: every pixel is still
// frame[ytab[y]][xtab[x]] or 0.
__global__ void __launch_bounds__(256) k_extent_gather_u8x16(const uint8_t* __restrict__ frame, int64_t ld, int fw, const int32_t* __restrict__ tabs,
                                                             int64_t n, int w, int h, uint8_t* __restrict__ out, int64_t ldo, int rows_per_wg) {
    const int tpr = w >> 4;                                   // threads per output row (a power of two, <= 256)
    const int tx = threadIdx.x & (tpr - 1), ty = threadIdx.x / tpr, rows_per_pass = 256 / tpr;
    const int y_begin = blockIdx.x * rows_per_wg, y_end = min(h, y_begin + rows_per_wg);
    for (int64_t b = blockIdx.y; b < n; b += gridDim.y) {
        const int32_t* t = tabs + b * (w + h);
        int xs[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 v = *(const int4*)(t + tx * 16 + q * 4);      // (table rows are 16-byte aligned: checked by the launcher)
            xs[q * 4] = v.x; xs[q * 4 + 1] = v.y; xs[q * 4 + 2] = v.z; xs[q * 4 + 3] = v.w;
        }
        // a group of four columns can come out of one 8-byte load if all four are inside the frame, ascending within 8 bytes of the
        // first, and the 8 bytes end inside the frame's row.  The choice is made per WAVE and the loads carry no per-lane branch
        // (a branch per load puts a wait behind each one: the loads of a row, and of the four rows of the unrolled loop, must overlap)
        bool all8 = true;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int x0 = xs[g * 4];
            bool asc = x0 >= 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) asc = asc && xs[g * 4 + k] >= x0;
            all8 = all8 && asc && xs[g * 4 + 3] < x0 + 8 && xs[g * 4 + 2] < x0 + 8 && xs[g * 4 + 1] < x0 + 8 && x0 + 8 <= fw;
        }
        const bool wave8 = __builtin_amdgcn_ballot_w64(!all8) == 0;
        uint8_t* dst = out + b * ldo + tx * 16;
        if (wave8) {
#pragma unroll 4
            for (int y = y_begin + ty; y < y_end; y += rows_per_pass) {      // (unrolled: the rows' loads overlap, one latency for four rows)
                const int ys = t[w + y];
                const uint8_t* src = frame + (int64_t)(ys >= 0 ? ys : 0) * ld;
                uint32_t pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint64_t c;
                    __builtin_memcpy(&c, src + xs[g * 4], 8);
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v |= (uint32_t)((c >> (8 * (xs[g * 4 + k] - xs[g * 4]))) & 0xffull) << (8 * k);
                    pk[g] = ys >= 0 ? v : 0u;
                }
                *(uint4*)(dst + (int64_t)y * w) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        } else {
            // wide windows, windows that leave the frame: a byte load per pixel from a clamped address, the value masked afterwards
            // (a middle tier — one 16-byte load per four pixels for windows up to four times as wide as their output — was measured:
            // 23.7 us against 20.2 for the first stage's 1738 windows; the third code path costs registers every wave pays for)
#pragma unroll 2
            for (int y = y_begin + ty; y < y_end; y += rows_per_pass) {
                const int ys = t[w + y];
                const uint8_t* src = frame + (int64_t)(ys >= 0 ? ys : 0) * ld;
                uint32_t pk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t v = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int x = xs[g * 4 + k];
                        const uint32_t px = src[x >= 0 ? x : 0];
                        v |= ((x >= 0 && ys >= 0) ? px : 0u) << (8 * k);
                    }
                    pk[g] = v;
                }
                *(uint4*)(dst + (int64_t)y * w) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
}

// Per-box rotation record (Image.rotate's matrix as affine_fixed / ImagingScaleAffine use it).
struct RotCoef {
    int32_t mode;            // 0: no rotation, 1: fixed point, 2: scaling branch (sin rounds to 0), 3: outside the fixed-point range
    int32_t A[6];
    int32_t pad;
    double m0, m2, m4, m5;   // scaling branch
};

__device__ __forceinline__ double round15(double v) {       // Python round(v, 15) for |v| <= 1
    return __ddiv_rn(rint(__dmul_rn(v, 1e15)), 1e15);
}
__device__ __forceinline__ int32_t fix16(double v) {        // Geometry.c FIX(): FLOOR(v * 65536 + 0.5)
    const double x = __dadd_rn(__dmul_rn(v, 65536.0), 0.5);
    return x >= 0.0 ? (int32_t)x : (int32_t)floor(x);
}

__device__ __forceinline__ RotCoef rot_coef_of(const double* __restrict__ boxes, const double* __restrict__ angs, int64_t b, int fw, int fh) {
    RotCoef rc{};
    double a = fmod(angs[b], 360.0);                 // Python's float %: result carries the divisor's sign
    if (a < 0.0) a += 360.0;
    if (a == 360.0) a = 0.0;
    if (a != 0.0 && isfinite(a)) {
        const double cx = __dmul_rn(__dadd_rn(boxes[b * 4], boxes[b * 4 + 2]), 0.5), cy = __dmul_rn(__dadd_rn(boxes[b * 4 + 1], boxes[b * 4 + 3]), 0.5);
        const double r = -__dmul_rn(a, 3.14159265358979323846 / 180.0);       // -math.radians(angle)
        const double c = round15(cos(r)), sn = round15(sin(r));
        const double m0 = c, m1 = sn, m3 = -sn, m4 = c;
        // transform(-cx, -cy, matrix) = (a x + b y) + c, then += center: every operation rounded on its own
        const double m2 = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m0, -cx), __dmul_rn(m1, -cy)), 0.0), cx);
        const double m5 = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m3, -cx), __dmul_rn(m4, -cy)), 0.0), cy);
        if (m1 == 0.0 && m3 == 0.0) {
            rc.mode = 2;
            rc.m0 = m0; rc.m2 = m2; rc.m4 = m4; rc.m5 = m5;
        } else {
            // check_fixed() on the four corners of the rotated frame (same size as the frame)
            bool ok = true;
            const double xs[2] = {0.0, (double)fw}, ys[2] = {0.0, (double)fh};
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) {
                    const double u = __dadd_rn(__dadd_rn(__dmul_rn(xs[i], m0), __dmul_rn(ys[j], m1)), m2);
                    const double v = __dadd_rn(__dadd_rn(__dmul_rn(xs[i], m3), __dmul_rn(ys[j], m4)), m5);
                    ok = ok && fabs(u) < 32768.0 && fabs(v) < 32768.0;
                }
            rc.mode = ok ? 1 : 3;
            rc.A[0] = fix16(m0); rc.A[1] = fix16(m1); rc.A[3] = fix16(m3); rc.A[4] = fix16(m4);
            rc.A[2] = fix16(__dadd_rn(__dadd_rn(m2, __dmul_rn(m0, 0.5)), __dmul_rn(m1, 0.5)));
            rc.A[5] = fix16(__dadd_rn(__dadd_rn(m5, __dmul_rn(m3, 0.5)), __dmul_rn(m4, 0.5)));
        }
    }
    return rc;
}

__device__ __forceinline__ void rot_coef_store(const double* __restrict__ boxes, const double* __restrict__ angs, int64_t b, int fw, int fh, RotCoef* out) {
    out[b] = rot_coef_of(boxes, angs, b, fw, fh);
}

// Rotated windows: per pixel, EXTENT table -> pixel of the rotated frame -> source pixel.  The frame (<= a few MB)
// sits in L2; the access pattern is a rotated scan line.
// One output row y of one box: EXTENT table (t[0..w) columns, t[w..w+h) rows) -> pixel of the rotated frame -> source pixel.
// uint8 windows: four pixels per thread and one 32-bit store (round 4: byte stores made the first stage's 1738 windows 79 us).
template <typename FT, typename OT>
__device__ __forceinline__ void gather_rot_row(const FT* __restrict__ frame, int64_t ld, int fw, int fh, const int32_t* t, const RotCoef& rc, int y,
                                               int w, OT* __restrict__ dst, bool packed, int tx, int ntx) {
    const int yr = t[w + y];
    double yo = 0.0;
    int ys2 = -1;
    if (rc.mode == 2 && yr >= 0) {      // ImagingScaleAffine on the rotated frame: yo = a5 + a4 / 2, then += a4 per row
        yo = __dadd_rn(rc.m5, __dmul_rn(rc.m4, 0.5));
        for (int k = 0; k < yr; ++k) yo = __dadd_rn(yo, rc.m4);
        ys2 = yo < 0.0 ? -1 : (int)yo;
        if (ys2 >= fh) ys2 = -1;
    }
    auto pixel = [&](int x) -> OT {
        const int xr = t[x];
        int xs = -1, ys = -1;
        if (xr >= 0 && yr >= 0) {
            if (rc.mode == 0) {
                xs = xr; ys = yr;
            } else if (rc.mode == 1) {
                const int64_t xx = (int64_t)rc.A[2] + (int64_t)yr * rc.A[1] + (int64_t)xr * rc.A[0];
                const int64_t yy = (int64_t)rc.A[5] + (int64_t)yr * rc.A[4] + (int64_t)xr * rc.A[3];
                xs = (int)(xx >> 16); ys = (int)(yy >> 16);
            } else if (rc.mode == 2) {
                double xo = __dadd_rn(rc.m2, __dmul_rn(rc.m0, 0.5));
                for (int k = 0; k < xr; ++k) xo = __dadd_rn(xo, rc.m0);
                xs = xo < 0.0 ? -1 : (int)xo;
                ys = ys2;
            }
        }
        const bool in = xs >= 0 && xs < fw && ys >= 0 && ys < fh;
        return in ? (OT)frame[(int64_t)ys * ld + xs] : (OT)0;
    };
    if constexpr (sizeof(OT) == 1) {
        if (packed) {
            // sixteen pixels per thread where the row allows (the launcher then gives a row w / 16 threads): the sixteen source
            // addresses first, then sixteen independent loads in flight and ONE 16-byte store — a thread with four pixels spent
            // its time in three dependent round trips (table -> frame -> store): 51 us for the first stage's 1738 windows
            if ((w & 15) == 0 && (((uintptr_t)dst) & 15) == 0 && rc.mode != 2) {
                // (round 5: no branch between the loads — the sixteen table entries as four 16-byte loads where the table row allows it,
                // the rotation mode tested once per box, every pixel load unconditional from a clamped address and masked afterwards;
                // with a branch around each load the compiler put a wait behind each one)
                const bool tab16 = ((uintptr_t)t & 15) == 0;
                const int mode = rc.mode;
                for (int x = tx * 16; x < w; x += ntx * 16) {
                    int xr[16];
                    if (tab16) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int4 v = *(const int4*)(t + x + q * 4);
                            xr[q * 4] = v.x; xr[q * 4 + 1] = v.y; xr[q * 4 + 2] = v.z; xr[q * 4 + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) xr[q] = t[x + q];
                    }
                    int64_t off[16];
                    bool in[16];
                    if (mode == 1) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int64_t xx = (int64_t)rc.A[2] + (int64_t)yr * rc.A[1] + (int64_t)xr[q] * rc.A[0];
                            const int64_t yy = (int64_t)rc.A[5] + (int64_t)yr * rc.A[4] + (int64_t)xr[q] * rc.A[3];
                            const int xs = (int)(xx >> 16), ys = (int)(yy >> 16);
                            in[q] = (xr[q] >= 0) & (yr >= 0) & (xs >= 0) & (xs < fw) & (ys >= 0) & (ys < fh);      // (&: no short-circuit branches)
                            off[q] = in[q] ? (int64_t)ys * ld + xs : 0;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            in[q] = (mode == 0) & (xr[q] >= 0) & (yr >= 0) & (xr[q] < fw) & (yr < fh);      // (mode 3: outside the fixed-point range, the window stays 0)
                            off[q] = in[q] ? (int64_t)yr * ld + xr[q] : 0;
                        }
                    }
                    uint32_t px[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) px[q] = (uint32_t)(uint8_t)frame[off[q]];
                    uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < 16; ++q) pk[q >> 2] |= (in[q] ? px[q] : 0u) << (8 * (q & 3));
                    *(uint4*)(dst + x) = uint4{pk[0], pk[1], pk[2], pk[3]};
                }
                return;
            }
            for (int x = tx * 4; x < w; x += ntx * 4) {
                uint32_t pk = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk |= (uint32_t)(uint8_t)pixel(x + q) << (8 * q);
                *(uint32_t*)(dst + x) = pk;
            }
            return;
        }
    }
    for (int x = tx; x < w; x += ntx) dst[x] = pixel(x);
}

// Rotated windows, tables from memory (k_extent_tables, k_rot_coefs): the frame (<= a few MB) sits in L2; the access pattern is a
// rotated scan line.
template <typename FT, typename OT>
__global__ void __launch_bounds__(256) k_extent_gather_rot(const FT* __restrict__ frame, int64_t ld, int fw, int fh, const int32_t* __restrict__ tabs,
                                                            const RotCoef* __restrict__ rot, int64_t n, int w, int h, OT* __restrict__ out, int64_t ldo) {
    const int y = blockIdx.x * blockDim.y + threadIdx.y;
    if (y >= h) return;
    const bool packed = sizeof(OT) == 1 && (w & 3) == 0 && (ldo & 3) == 0 && ((uintptr_t)out & 3) == 0;
    for (int64_t b = blockIdx.y; b < n; b += gridDim.y) {
        const RotCoef rc = rot[b];
        gather_rot_row<FT, OT>(frame, ld, fw, fh, tabs + b * (w + h), rc, y, w, out + b * ldo + (int64_t)y * w, packed, threadIdx.x, blockDim.x);
    }
}

// (Round 4 also measured tables + coefficients + gather as ONE launch, a workgroup per box building its tables and coefficients in
// LDS: 48.6 us against 38.5 for the first stage's 1738 windows and 25 against 21 for a few hundred — one thread's double-precision
// fmod / cos / sin and the dependent additions of the tables sit on every workgroup's critical path instead of being spread over
// a launch of their own.  Not kept.)

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

template <typename FT>
void launch_gather(const void* frame, int64_t ld, int fw, const int32_t* tabs, int64_t n, int w, int h, void* out, int out_dtype, int64_t ldo,
                   hipStream_t st) {
    if constexpr (sizeof(FT) == 1) {
        const int tpr = w >> 4;
        static const bool v1 = getenv("HIGSFA_EXTENT_V1") != nullptr;      // (A/B: the four-pixels-per-thread kernel for every shape)
        if (!v1 && out_dtype == HG_U8 && (w & 15) == 0 && tpr >= 1 && tpr <= 256 && (tpr & (tpr - 1)) == 0 && ((w + h) & 3) == 0 && (ldo & 15) == 0 &&
            ((uintptr_t)out & 15) == 0 && ((uintptr_t)tabs & 15) == 0) {
            // whole windows per workgroup while that leaves >= 1024 workgroups, else row chunks (the column indices are read once per chunk)
            const int rows_per_pass = 256 / tpr;
            int passes = (h + rows_per_pass - 1) / rows_per_pass;
            while (passes > 1 && n * ((h + rows_per_pass * passes - 1) / (rows_per_pass * passes)) < 1024) passes = (passes + 1) / 2;
            const int rows_per_wg = rows_per_pass * passes;
            const dim3 grid((unsigned)((h + rows_per_wg - 1) / rows_per_wg), (unsigned)std::min<int64_t>(n, 65535));
            hipLaunchKernelGGL(k_extent_gather_u8x16, grid, 256, 0, st, (const uint8_t*)frame, ld, fw, tabs, n, w, h, (uint8_t*)out, ldo, rows_per_wg);
            return;
        }
    }
    const unsigned tx = w >= 1024 ? 256 : w >= 256 ? 64 : 32;       // four pixels per thread on the uint8 path
    const dim3 thr(tx, 256 / tx);
    const dim3 grid((unsigned)((h + thr.y - 1) / thr.y), (unsigned)std::min<int64_t>(n, 65535));
    switch (out_dtype) {
        case HG_U8: hipLaunchKernelGGL((k_extent_gather<FT, uint8_t>), grid, thr, 0, st, (const FT*)frame, ld, tabs, n, w, h, (uint8_t*)out, ldo); break;
        case HG_F32: hipLaunchKernelGGL((k_extent_gather<FT, float>), grid, thr, 0, st, (const FT*)frame, ld, tabs, n, w, h, (float*)out, ldo); break;
        default: hipLaunchKernelGGL((k_extent_gather<FT, double>), grid, thr, 0, st, (const FT*)frame, ld, tabs, n, w, h, (double*)out, ldo); break;
    }
}

template <typename FT>
void launch_gather_rot(const void* frame, int64_t ld, int fw, int fh, const int32_t* tabs, const RotCoef* rot, int64_t n, int w, int h, void* out,
                       int out_dtype, int64_t ldo, hipStream_t st) {
    // uint8 windows: a thread packs sixteen pixels where the row is a multiple of 16 (a 128-pixel row takes 8 threads and a
    // workgroup 32 rows), four otherwise
    const unsigned tx = out_dtype == HG_U8 ? ((w & 15) == 0 ? (w >= 2048 ? 128 : w >= 1024 ? 64 : w >= 512 ? 32 : w >= 256 ? 16 : w >= 128 ? 8 : 4)
                                                            : (w >= 512 ? 128 : w >= 256 ? 64 : 32))
                                           : (w >= 128 ? 128 : w >= 64 ? 64 : 32);
    const dim3 thr(tx, 256 / tx);
    const dim3 grid((unsigned)((h + thr.y - 1) / thr.y), (unsigned)std::min<int64_t>(n, 65535));
    switch (out_dtype) {
        case HG_U8: hipLaunchKernelGGL((k_extent_gather_rot<FT, uint8_t>), grid, thr, 0, st, (const FT*)frame, ld, fw, fh, tabs, rot, n, w, h, (uint8_t*)out, ldo); break;
        case HG_F32: hipLaunchKernelGGL((k_extent_gather_rot<FT, float>), grid, thr, 0, st, (const FT*)frame, ld, fw, fh, tabs, rot, n, w, h, (float*)out, ldo); break;
        default: hipLaunchKernelGGL((k_extent_gather_rot<FT, double>), grid, thr, 0, st, (const FT*)frame, ld, fw, fh, tabs, rot, n, w, h, (double*)out, ldo); break;
    }
}

void check_args(const hg_patcher* p, const void* frame, int frame_dtype, int fh, int fw, int64_t ld, const double* boxes, int64_t n,
                int w, int h, const void* out, int out_dtype, int64_t ldo) {
    if (!p) hg::fail(HG_ERR_ARG, "null patcher handle");
    if (frame_dtype != HG_U8 && frame_dtype != HG_F32) hg::fail(HG_ERR_ARG, "frame dtype must be HG_U8 or HG_F32");
    if (out_dtype != HG_U8 && out_dtype != HG_F32 && out_dtype != HG_F64) hg::fail(HG_ERR_ARG, "bad output dtype");
    if (fh <= 0 || fw <= 0 || ld < fw) hg::fail(HG_ERR_ARG, "bad frame geometry");
    if (w <= 0 || h <= 0 || w > 4096 || h > 4096) hg::fail(HG_ERR_ARG, "bad sub-image size");
    if (n < 0 || ldo < (int64_t)w * h) hg::fail(HG_ERR_ARG, "bad batch geometry");
    if (n > 0 && (!frame || !boxes || !out)) hg::fail(HG_ERR_ARG, "null data pointer");
}

}  // namespace

extern "C" {

int hg_patcher_create(int device, hg_patcher** out) {
    return guarded([&] {
        if (!out) hg::fail(HG_ERR_ARG, "null output handle pointer");
        *out = nullptr;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
            hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
        if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range", device);
        auto p = std::make_unique<hg_patcher>();
        p->device = device;
        *out = p.release();
    });
}

void hg_patcher_free(hg_patcher* p) {
    if (p && p->device >= 0) (void)hipSetDevice(p->device);
    delete p;
}

int hg_patcher_extract_rotate_device(hg_patcher* p, const void* frame_dev, int frame_dtype, int frame_h, int frame_w, int64_t ld,
                                     const double* boxes_dev, const double* delta_angs_dev, int64_t n, int out_w, int out_h, void* out_dev,
                                     int out_dtype, int64_t ldo, void* stream) {
    return guarded([&] {
        check_args(p, frame_dev, frame_dtype, frame_h, frame_w, ld, boxes_dev, n, out_w, out_h, out_dev, out_dtype, ldo);
        if (n == 0) return;
        HG_HIP(hipSetDevice(p->device));
        hipStream_t st = (hipStream_t)stream;
        p->tabs.alloc((size_t)n * (out_w + out_h) * 4);
        const int64_t n_ent = n * (out_w + out_h);
        if ((n_ent + 255) / 256 > 0x7fffffffll) hg::fail(HG_ERR_ARG, "too many boxes");
        if (delta_angs_dev) {
            if (frame_w >= 32768 || frame_h >= 32768) hg::fail(HG_ERR_ARG, "rotated windows need a frame smaller than 32768 pixels per side");
            p->rot.alloc((size_t)n * sizeof(RotCoef));
        }
        hipLaunchKernelGGL(k_extent_tables, (unsigned)((n_ent + 255) / 256), 256, 0, st, boxes_dev, n, out_w, out_h, frame_w, frame_h,
                           (int32_t*)p->tabs.p, delta_angs_dev, delta_angs_dev ? (RotCoef*)p->rot.p : nullptr);
        if (delta_angs_dev) {
            if (frame_dtype == HG_U8)
                launch_gather_rot<uint8_t>(frame_dev, ld, frame_w, frame_h, (const int32_t*)p->tabs.p, (const RotCoef*)p->rot.p, n, out_w, out_h, out_dev,
                                           out_dtype, ldo, st);
            else
                launch_gather_rot<float>(frame_dev, ld, frame_w, frame_h, (const int32_t*)p->tabs.p, (const RotCoef*)p->rot.p, n, out_w, out_h, out_dev,
                                         out_dtype, ldo, st);
        } else if (frame_dtype == HG_U8) {
            launch_gather<uint8_t>(frame_dev, ld, frame_w, (const int32_t*)p->tabs.p, n, out_w, out_h, out_dev, out_dtype, ldo, st);
        } else {
            launch_gather<float>(frame_dev, ld, frame_w, (const int32_t*)p->tabs.p, n, out_w, out_h, out_dev, out_dtype, ldo, st);
        }
        HG_HIP(hipGetLastError());
    });
}

int hg_patcher_extract_keyed_device(hg_patcher* p, uint64_t key, const void* frame_dev, int frame_dtype, int frame_h, int frame_w, int64_t ld,
                                    const double* boxes_dev, int64_t n, int out_w, int out_h, void* out_dev, int out_dtype, int64_t ldo, void* stream) {
    if (key == 0) return hg_patcher_extract_rotate_device(p, frame_dev, frame_dtype, frame_h, frame_w, ld, boxes_dev, nullptr, n, out_w, out_h, out_dev, out_dtype, ldo, stream);
    return guarded([&] {
        check_args(p, frame_dev, frame_dtype, frame_h, frame_w, ld, boxes_dev, n, out_w, out_h, out_dev, out_dtype, ldo);
        if (n == 0) return;
        HG_HIP(hipSetDevice(p->device));
        hipStream_t st = (hipStream_t)stream;
        const int64_t n_ent = n * (out_w + out_h);
        if ((n_ent + 255) / 256 > 0x7fffffffll) hg::fail(HG_ERR_ARG, "too many boxes");
        hg_patcher::Keyed* K = nullptr;
        for (auto& k : p->keyed)
            if (k.key == key && k.n == n && k.out_w == out_w && k.out_h == out_h && k.frame_w == frame_w && k.frame_h == frame_h) K = &k;
        if (!K) {      // first use of this key (or its shape changed): build the tables, in stream order, into a buffer of their own
            K = &p->keyed[p->keyed_next];
            p->keyed_next = (p->keyed_next + 1) % 4;
            if (K->tabs.p && (size_t)n_ent * 4 > K->tabs.bytes) HG_HIP(hipStreamSynchronize(st));      // a launch in flight may still read the old buffer
            K->tabs.alloc((size_t)n_ent * 4);
            K->key = key; K->n = n; K->out_w = out_w; K->out_h = out_h; K->frame_w = frame_w; K->frame_h = frame_h;
            hipLaunchKernelGGL(k_extent_tables, (unsigned)((n_ent + 255) / 256), 256, 0, st, boxes_dev, n, out_w, out_h, frame_w, frame_h, (int32_t*)K->tabs.p,
                               (const double*)nullptr, (RotCoef*)nullptr);
        }
        if (frame_dtype == HG_U8) launch_gather<uint8_t>(frame_dev, ld, frame_w, (const int32_t*)K->tabs.p, n, out_w, out_h, out_dev, out_dtype, ldo, st);
        else launch_gather<float>(frame_dev, ld, frame_w, (const int32_t*)K->tabs.p, n, out_w, out_h, out_dev, out_dtype, ldo, st);
        HG_HIP(hipGetLastError());
    });
}

int hg_patcher_extract_device(hg_patcher* p, const void* frame_dev, int frame_dtype, int frame_h, int frame_w, int64_t ld,
                              const double* boxes_dev, int64_t n, int out_w, int out_h, void* out_dev, int out_dtype, int64_t ldo,
                              void* stream) {
    return hg_patcher_extract_rotate_device(p, frame_dev, frame_dtype, frame_h, frame_w, ld, boxes_dev, nullptr, n, out_w, out_h, out_dev, out_dtype,
                                            ldo, stream);
}

int hg_patcher_extract_rotate(hg_patcher* p, const void* frame, int frame_dtype, int frame_h, int frame_w, int64_t ld, const double* boxes,
                              const double* delta_angs, int64_t n, int out_w, int out_h, void* out, int out_dtype, int64_t ldo) {
    return guarded([&] {
        check_args(p, frame, frame_dtype, frame_h, frame_w, ld, boxes, n, out_w, out_h, out, out_dtype, ldo);
        if (n == 0) return;
        HG_HIP(hipSetDevice(p->device));
        const size_t fs = hg::dtype_size(frame_dtype), os = hg::dtype_size(out_dtype);
        p->frame.alloc((size_t)frame_h * frame_w * fs);
        HG_HIP(hipMemcpy2D(p->frame.p, (size_t)frame_w * fs, frame, (size_t)ld * fs, (size_t)frame_w * fs, (size_t)frame_h, hipMemcpyHostToDevice));
        p->boxes.upload(boxes, (size_t)n * 4 * 8);
        if (delta_angs) p->angles.upload(delta_angs, (size_t)n * 8);
        const size_t row = (size_t)out_w * out_h;
        p->out.alloc((size_t)n * row * os);
        int rc = hg_patcher_extract_rotate_device(p, p->frame.p, frame_dtype, frame_h, frame_w, frame_w, (const double*)p->boxes.p,
                                                  delta_angs ? (const double*)p->angles.p : nullptr, n, out_w, out_h, p->out.p, out_dtype,
                                                  (int64_t)row, nullptr);
        if (rc != HG_OK) hg::fail(rc, "%s", hg_last_error());
        HG_HIP(hipMemcpy2D(out, (size_t)ldo * os, p->out.p, row * os, row * os, (size_t)n, hipMemcpyDeviceToHost));
    });
}

int hg_patcher_extract(hg_patcher* p, const void* frame, int frame_dtype, int frame_h, int frame_w, int64_t ld, const double* boxes,
                       int64_t n, int out_w, int out_h, void* out, int out_dtype, int64_t ldo) {
    return hg_patcher_extract_rotate(p, frame, frame_dtype, frame_h, frame_w, ld, boxes, nullptr, n, out_w, out_h, out, out_dtype, ldo);
}

}  // extern "C"
