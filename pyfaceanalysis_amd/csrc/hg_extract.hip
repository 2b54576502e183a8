// On-device sub-image extraction: the producer of the hot call's input,
//   subimages_arr = load_network_subimages(images, ..., curr_subimage_coordinates, curr_angles, w, h, NEAREST)
//                                                        (FaceDetectUpdated.py:686; face_analysis.py:775-800)
// which the reference does per patch with PIL: Image.transform((w, h), Image.EXTENT, (x0, y0, x1, y1),
// Image.NEAREST) inside cuicuilco.image_loader.extract_subimages_rotate.  This restates PIL's
// EXTENT/NEAREST index rule (ImagingScaleAffine) exactly, including its additive accumulation of the
// source coordinate in double precision:
//     a = (x1 - x0) / w;  xo = x0 + a/2;  for x in 0..w-1: xin = xo < 0 ? -1 : (int)xo;  xo += a
// (same for y); source pixels outside the frame leave the output pixel 0.  Angles other than 0 are not
// covered (the rotation rule lives in cuicuilco, which is not available): callers must pass boxes only.
#include <hip/hip_runtime.h>

#include "hg_common.hpp"

struct hg_patcher {
    int device = -1;
    hg::DevBuf tabs, boxes, frame, out;
};

namespace hg { void set_last_error(const std::string& s); }

namespace {

// Source index tables, one entry per thread.  PIL steps the source coordinate by repeated addition
// (o += a per output pixel); to land on the same pixel in every case each thread repeats that sum from the start
// of its axis — O(m) dependent adds for the last entry instead of one thread walking all m entries with a
// conversion, two compares and a store per step.
__global__ void k_extent_tables(const double* __restrict__ boxes, int64_t n, int w, int h, int fw, int fh, int32_t* __restrict__ tabs) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wh = w + h;
    if (id >= n * wh) return;
    const int64_t b = id / wh;
    const int e = (int)(id - b * wh);
    const int axis = e >= w ? 1 : 0, i = axis ? e - w : e;
    const double lo = boxes[b * 4 + axis], hi = boxes[b * 4 + 2 + axis];
    const int m = axis ? h : w, lim = axis ? fh : fw;
    const double a = (hi - lo) / m;
    double o = lo + a * 0.5;
    for (int k = 0; k < i; ++k) o += a;
    const int v = o < 0.0 ? -1 : (int)o;
    tabs[b * wh + e] = (v >= 0 && v < lim) ? v : -1;
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
void launch_gather(const void* frame, int64_t ld, const int32_t* tabs, int64_t n, int w, int h, void* out, int out_dtype, int64_t ldo,
                   hipStream_t st) {
    const unsigned tx = w >= 1024 ? 256 : w >= 256 ? 64 : 32;       // four pixels per thread on the uint8 path
    const dim3 thr(tx, 256 / tx);
    const dim3 grid((unsigned)((h + thr.y - 1) / thr.y), (unsigned)std::min<int64_t>(n, 65535));
    switch (out_dtype) {
        case HG_U8: hipLaunchKernelGGL((k_extent_gather<FT, uint8_t>), grid, thr, 0, st, (const FT*)frame, ld, tabs, n, w, h, (uint8_t*)out, ldo); break;
        case HG_F32: hipLaunchKernelGGL((k_extent_gather<FT, float>), grid, thr, 0, st, (const FT*)frame, ld, tabs, n, w, h, (float*)out, ldo); break;
        default: hipLaunchKernelGGL((k_extent_gather<FT, double>), grid, thr, 0, st, (const FT*)frame, ld, tabs, n, w, h, (double*)out, ldo); break;
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

int hg_patcher_extract_device(hg_patcher* p, const void* frame_dev, int frame_dtype, int frame_h, int frame_w, int64_t ld,
                              const double* boxes_dev, int64_t n, int out_w, int out_h, void* out_dev, int out_dtype, int64_t ldo,
                              void* stream) {
    return guarded([&] {
        check_args(p, frame_dev, frame_dtype, frame_h, frame_w, ld, boxes_dev, n, out_w, out_h, out_dev, out_dtype, ldo);
        if (n == 0) return;
        HG_HIP(hipSetDevice(p->device));
        hipStream_t st = (hipStream_t)stream;
        p->tabs.alloc((size_t)n * (out_w + out_h) * 4);
        const int64_t n_ent = n * (out_w + out_h);
        if ((n_ent + 255) / 256 > 0x7fffffffll) hg::fail(HG_ERR_ARG, "too many boxes");
        hipLaunchKernelGGL(k_extent_tables, (unsigned)((n_ent + 255) / 256), 256, 0, st, boxes_dev, n, out_w, out_h, frame_w, frame_h,
                           (int32_t*)p->tabs.p);
        if (frame_dtype == HG_U8)
            launch_gather<uint8_t>(frame_dev, ld, (const int32_t*)p->tabs.p, n, out_w, out_h, out_dev, out_dtype, ldo, st);
        else
            launch_gather<float>(frame_dev, ld, (const int32_t*)p->tabs.p, n, out_w, out_h, out_dev, out_dtype, ldo, st);
        HG_HIP(hipGetLastError());
    });
}

int hg_patcher_extract(hg_patcher* p, const void* frame, int frame_dtype, int frame_h, int frame_w, int64_t ld, const double* boxes,
                       int64_t n, int out_w, int out_h, void* out, int out_dtype, int64_t ldo) {
    return guarded([&] {
        check_args(p, frame, frame_dtype, frame_h, frame_w, ld, boxes, n, out_w, out_h, out, out_dtype, ldo);
        if (n == 0) return;
        HG_HIP(hipSetDevice(p->device));
        const size_t fs = hg::dtype_size(frame_dtype), os = hg::dtype_size(out_dtype);
        p->frame.alloc((size_t)frame_h * frame_w * fs);
        HG_HIP(hipMemcpy2D(p->frame.p, (size_t)frame_w * fs, frame, (size_t)ld * fs, (size_t)frame_w * fs, (size_t)frame_h, hipMemcpyHostToDevice));
        p->boxes.upload(boxes, (size_t)n * 4 * 8);
        const size_t row = (size_t)out_w * out_h;
        p->out.alloc((size_t)n * row * os);
        int rc = hg_patcher_extract_device(p, p->frame.p, frame_dtype, frame_h, frame_w, frame_w, (const double*)p->boxes.p, n, out_w, out_h,
                                           p->out.p, out_dtype, (int64_t)row, nullptr);
        if (rc != HG_OK) hg::fail(rc, "%s", hg_last_error());
        HG_HIP(hipMemcpy2D(out, (size_t)ldo * os, p->out.p, row * os, row * os, (size_t)n, hipMemcpyDeviceToHost));
    });
}

}  // extern "C"
