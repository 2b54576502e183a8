// Cascade glue on the device: what the reference's stage loop does between two calls of the hot path,
//     reg_out = classifiers[k].regression(sl[:, 0:d], avg_labels)                          FaceDetectUpdated.py:719
//     curr_subimage_coordinates, curr_angles = update_current_subimage_coordinates(...)    :728   (face_analysis.py:803-840)
//     new_wrong_images = identify_patches_to_discard(...)                                  :733   (face_analysis.py:842-887)
//     boolean-mask compaction of coordinates, angles, indices, sl, subimages_arr           :739-759
// so that extract -> execute -> regression -> update -> discard -> compaction chain on one stream without a host copy of any
// per-candidate array (the host reads back ONE integer, the survivor count, where it needs a launch size).
// All arithmetic is float64 in the reference's operation order (no contraction), so coordinates follow the numpy path bit for
// bit given the same regression outputs.  Candidates of all pyramid levels may share one batch (the author's note
// FaceDetectUpdated.py:599): the per-level constants max_Dx_diff, max_Dy_diff, base_side travel per ORIGINAL window.
#include <hip/hip_runtime.h>

#include "hg_common.hpp"

namespace hg { void set_last_error(const std::string& s); }

namespace {

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

__global__ void k_cascade_update(int type, hg_cascade_consts c, int64_t n, double* __restrict__ coords, double* __restrict__ angles,
                                 const double* __restrict__ reg, const int32_t* __restrict__ orig_index, const double* __restrict__ orig_coords,
                                 const double* __restrict__ orig_angles, const double* __restrict__ orig_level, uint8_t* __restrict__ discard) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x0 = coords[i * 4], y0 = coords[i * 4 + 1], x1 = coords[i * 4 + 2], y1 = coords[i * 4 + 3], ang = angles[i];
    const double r = reg[i];
    const int32_t oi = orig_index[i];
    bool wrong = false;
    switch (type) {
        case HG_STAGE_DISC:        // coordinates untouched (face_analysis.py:804-805); discard on the cut-off (:881-882)
            wrong = r >= c.cut_off_face;
            break;
        case HG_STAGE_POSX: {      // :806-812, :846-856
            const double ro = __ddiv_rn(__dmul_rn(r, __dsub_rn(x1, x0)), c.regression_width);
            x0 = __dsub_rn(x0, ro);
            x1 = __dsub_rn(x1, ro);
            const double d = __dsub_rn(__ddiv_rn(__dadd_rn(x1, x0), 2.0), __ddiv_rn(__dadd_rn(orig_coords[oi * 4 + 2], orig_coords[oi * 4]), 2.0));
            wrong = fabs(d) > __dmul_rn(orig_level[oi * 3], c.tolerance_posxy_deviation);
            break;
        }
        case HG_STAGE_POSY: {      // :813-819, :857-867
            const double ro = __ddiv_rn(__dmul_rn(r, __dsub_rn(y1, y0)), c.regression_height);
            y0 = __dsub_rn(y0, ro);
            y1 = __dsub_rn(y1, ro);
            const double d = __dsub_rn(__ddiv_rn(__dadd_rn(y1, y0), 2.0), __ddiv_rn(__dadd_rn(orig_coords[oi * 4 + 3], orig_coords[oi * 4 + 1]), 2.0));
            wrong = fabs(d) > __dmul_rn(orig_level[oi * 3 + 1], c.tolerance_posxy_deviation);
            break;
        }
        case HG_STAGE_PANG: {      // :820-821, :868-872
            ang = __dadd_rn(ang, r);
            const double lim = __dmul_rn(c.net_Dang, c.tolerance_angle_deviation), oa = orig_angles[oi];
            wrong = ang > __dadd_rn(oa, lim) || ang < __dsub_rn(oa, lim);
            break;
        }
        case HG_STAGE_SCALE: {     // :822-835, :873-880
            const double ow = __dsub_rn(x1, x0), oh = __dsub_rn(y1, y0);
            const double xc = __ddiv_rn(__dadd_rn(x1, x0), 2.0), yc = __ddiv_rn(__dadd_rn(y1, y0), 2.0);
            const double w = __dmul_rn(__ddiv_rn(ow, r), c.desired_sampling), h = __dmul_rn(__ddiv_rn(oh, r), c.desired_sampling);
            x0 = __dsub_rn(xc, __ddiv_rn(w, 2.0));
            x1 = __dadd_rn(xc, __ddiv_rn(w, 2.0));
            y0 = __dsub_rn(yc, __ddiv_rn(h, 2.0));
            y1 = __dadd_rn(yc, __ddiv_rn(h, 2.0));
            const double dx = __dsub_rn(x0, x1), dy = __dsub_rn(y0, y1);
            const double side = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
            const double ratio = __ddiv_rn(side, orig_level[oi * 3 + 2]);
            wrong = ratio > __dmul_rn(c.max_scale_radio, c.tolerance_scale_deviation) || ratio < __ddiv_rn(c.min_scale_radio, c.tolerance_scale_deviation);
            break;
        }
    }
    coords[i * 4] = x0; coords[i * 4 + 1] = y0; coords[i * 4 + 2] = x1; coords[i * 4 + 3] = y1;
    angles[i] = ang;
    discard[i] = wrong ? 1 : 0;
}

// map[j] = index of the j-th kept row, *count = number of kept rows.  One workgroup walks the flags in chunks of its size
// with a running offset (a real frame has <= a few thousand candidates: one or two chunks).
__global__ void __launch_bounds__(1024) k_cascade_compact(const uint8_t* __restrict__ discard, int64_t n, int32_t* __restrict__ map,
                                                           int32_t* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n; i0 += blockDim.x) {
        const int64_t i = i0 + tid;
        const int keep = (i < n && discard[i] == 0) ? 1 : 0;
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) map[off + before] = (int32_t)i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *count = base;
}

// dst[j, :] = src[map[j], :] for j < *count; rows of row_bytes bytes (multiple of 4), 16-byte vectors where alignment allows.
__global__ void __launch_bounds__(256) k_gather_rows(const char* __restrict__ src, char* __restrict__ dst, int64_t row_bytes,
                                                      const int32_t* __restrict__ map, const int32_t* __restrict__ count, int vec16) {
    const int cnt = *count;
    for (int64_t j = blockIdx.x; j < cnt; j += gridDim.x) {
        const char* s = src + (int64_t)map[j] * row_bytes;
        char* d = dst + j * row_bytes;
        if (vec16) {
            for (int64_t o = (int64_t)threadIdx.x * 16; o < row_bytes; o += (int64_t)blockDim.x * 16) *(int4*)(d + o) = *(const int4*)(s + o);
        } else {
            for (int64_t o = (int64_t)threadIdx.x * 4; o < row_bytes; o += (int64_t)blockDim.x * 4) *(int32_t*)(d + o) = *(const int32_t*)(s + o);
        }
    }
}

void set_dev(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        hg::fail(HG_ERR_DEVICE, "no HIP device available (this library has no CPU execution path)");
    if (device < 0 || device >= count) hg::fail(HG_ERR_DEVICE, "device %d out of range (0..%d)", device, count - 1);
    HG_HIP(hipSetDevice(device));
}

}  // namespace

extern "C" {

int hg_cascade_update_device(int device, int stage_type, const hg_cascade_consts* c, int64_t n, double* coords_dev, double* angles_dev,
                             const double* reg_dev, const int32_t* orig_index_dev, const double* orig_coords_dev,
                             const double* orig_angles_dev, const double* orig_level_dev, uint8_t* discard_dev, void* stream) {
    return guarded([&] {
        if (!c) hg::fail(HG_ERR_ARG, "null constants");
        if (stage_type < HG_STAGE_DISC || stage_type > HG_STAGE_SCALE) hg::fail(HG_ERR_ARG, "unknown stage type %d", stage_type);
        if (n < 0) hg::fail(HG_ERR_ARG, "negative candidate count");
        if (n > 0 && (!coords_dev || !angles_dev || !reg_dev || !orig_index_dev || !orig_coords_dev || !orig_angles_dev || !orig_level_dev || !discard_dev))
            hg::fail(HG_ERR_ARG, "null data pointer");
        set_dev(device);
        if (n == 0) return;
        hipLaunchKernelGGL(k_cascade_update, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream, stage_type, *c, n, coords_dev, angles_dev, reg_dev,
                           orig_index_dev, orig_coords_dev, orig_angles_dev, orig_level_dev, discard_dev);
        HG_HIP(hipGetLastError());
    });
}

int hg_cascade_compact_device(int device, const uint8_t* discard_dev, int64_t n, int32_t* map_dev, int32_t* count_dev, void* stream) {
    return guarded([&] {
        if (n < 0 || n > 0x7fffffffll) hg::fail(HG_ERR_ARG, "bad candidate count");
        if (!count_dev || (n > 0 && (!discard_dev || !map_dev))) hg::fail(HG_ERR_ARG, "null data pointer");
        set_dev(device);
        hipLaunchKernelGGL(k_cascade_compact, 1, 1024, 0, (hipStream_t)stream, discard_dev, n, map_dev, count_dev);
        HG_HIP(hipGetLastError());
    });
}

int hg_gather_rows_device(int device, const void* src_dev, void* dst_dev, int64_t row_bytes, const int32_t* map_dev,
                          const int32_t* count_dev, int64_t n_max, void* stream) {
    return guarded([&] {
        if (row_bytes <= 0 || (row_bytes & 3)) hg::fail(HG_ERR_ARG, "row_bytes must be a positive multiple of 4");
        if (n_max < 0) hg::fail(HG_ERR_ARG, "negative row count");
        if (!count_dev || (n_max > 0 && (!src_dev || !dst_dev || !map_dev))) hg::fail(HG_ERR_ARG, "null data pointer");
        if (src_dev == dst_dev && n_max > 0) hg::fail(HG_ERR_ARG, "gather must be out of place");
        set_dev(device);
        if (n_max == 0) return;
        const int vec16 = (row_bytes % 16 == 0 && ((uintptr_t)src_dev % 16) == 0 && ((uintptr_t)dst_dev % 16) == 0) ? 1 : 0;
        const unsigned threads = row_bytes >= 4096 ? 256 : 64;
        hipLaunchKernelGGL(k_gather_rows, (unsigned)std::min<int64_t>(n_max, 16384), threads, 0, (hipStream_t)stream, (const char*)src_dev, (char*)dst_dev,
                           row_bytes, map_dev, count_dev, vec16);
        HG_HIP(hipGetLastError());
    });
}

}  // extern "C"
