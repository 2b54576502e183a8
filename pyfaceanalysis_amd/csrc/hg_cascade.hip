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

#include <algorithm>
#include <chrono>
#include <cstring>

#include "hg_common.hpp"
#include "hg_gauss_dev.hpp"

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

// One cascade stage's glue in ONE launch of ONE workgroup (a frame has at most a few thousand candidates): update + discard
// test per candidate (same arithmetic as k_cascade_update), order-preserving compaction, and the gather of every small
// per-candidate array into the other half of its ping-pong pair.  The candidate count comes from device memory (the previous
// stage's output), so stages chain without a host round trip; the new count goes to device memory and, for the caller that
// wants it, to a pinned host word.
struct StageArrays {
    double *coords[2], *angles[2], *conf[2], *neg_angles;   // [cur / nxt]
    int32_t* oidx[2];
    float* sl[2];
    const double *reg, *orig_coords, *orig_angles, *orig_level;
    uint8_t* discard;
    int32_t* map;
    int32_t *count_in, *count_out;
    int32_t k_feat, cur, n_max;
};

__device__ __forceinline__ bool update_one(int type, const hg_cascade_consts& c, double& x0, double& y0, double& x1, double& y1, double& ang,
                                            double r, const double* oc, double oa, const double* lvl) {
    switch (type) {
        case HG_STAGE_DISC: return r >= c.cut_off_face;
        case HG_STAGE_POSX: {
            const double ro = __ddiv_rn(__dmul_rn(r, __dsub_rn(x1, x0)), c.regression_width);
            x0 = __dsub_rn(x0, ro);
            x1 = __dsub_rn(x1, ro);
            const double d = __dsub_rn(__ddiv_rn(__dadd_rn(x1, x0), 2.0), __ddiv_rn(__dadd_rn(oc[2], oc[0]), 2.0));
            return fabs(d) > __dmul_rn(lvl[0], c.tolerance_posxy_deviation);
        }
        case HG_STAGE_POSY: {
            const double ro = __ddiv_rn(__dmul_rn(r, __dsub_rn(y1, y0)), c.regression_height);
            y0 = __dsub_rn(y0, ro);
            y1 = __dsub_rn(y1, ro);
            const double d = __dsub_rn(__ddiv_rn(__dadd_rn(y1, y0), 2.0), __ddiv_rn(__dadd_rn(oc[3], oc[1]), 2.0));
            return fabs(d) > __dmul_rn(lvl[1], c.tolerance_posxy_deviation);
        }
        case HG_STAGE_PANG: {
            ang = __dadd_rn(ang, r);
            const double lim = __dmul_rn(c.net_Dang, c.tolerance_angle_deviation);
            return ang > __dadd_rn(oa, lim) || ang < __dsub_rn(oa, lim);
        }
        default: {
            const double ow = __dsub_rn(x1, x0), oh = __dsub_rn(y1, y0);
            const double xc = __ddiv_rn(__dadd_rn(x1, x0), 2.0), yc = __ddiv_rn(__dadd_rn(y1, y0), 2.0);
            const double w = __dmul_rn(__ddiv_rn(ow, r), c.desired_sampling), h = __dmul_rn(__ddiv_rn(oh, r), c.desired_sampling);
            x0 = __dsub_rn(xc, __ddiv_rn(w, 2.0));
            x1 = __dadd_rn(xc, __ddiv_rn(w, 2.0));
            y0 = __dsub_rn(yc, __ddiv_rn(h, 2.0));
            y1 = __dadd_rn(yc, __ddiv_rn(h, 2.0));
            const double dx = __dsub_rn(x0, x1), dy = __dsub_rn(y0, y1);
            const double ratio = __ddiv_rn(__dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy))), lvl[2]);
            return ratio > __dmul_rn(c.max_scale_radio, c.tolerance_scale_deviation) || ratio < __ddiv_rn(c.min_scale_radio, c.tolerance_scale_deviation);
        }
    }
}

// What the host polls instead of synchronising the stream (pinned, device-visible): word 0 = survivor count, word 1 = sequence
// number of the stage that wrote it (written last, system scope).  A stream synchronisation costs the caller 20-40 us before it
// has caught up with the device again; a poll sees the word 2-3 us after the store.
__device__ __forceinline__ void publish_count(int32_t* host_count, int cnt, int seq) {
    host_count[0] = cnt;
    __threadfence_system();
    __hip_atomic_store(host_count + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The final survivors for the host, written by the last stage's kernel into pinned memory (a handful of rows): the call ends with
// a poll of the count word instead of four device-to-host copies and a stream synchronisation.
struct HostResults {
    double* coords;      // [cap][4]
    double* angles;      // [cap]
    double* conf;        // [cap]
    int32_t* oidx;       // [cap]
    int32_t cap;
};

// Start of a frame: candidate i = original window i (the host copied the boxes into orig_coords); replaces five small copies /
// memsets and the synchronisation that kept their host temporaries alive.
__global__ void k_cascade_init(int n, const double* __restrict__ orig_coords, double* __restrict__ coords, double* __restrict__ angles,
                               double* __restrict__ neg, double* __restrict__ conf, int32_t* __restrict__ oidx, int32_t* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = n;
    if (i >= n) return;
    for (int q = 0; q < 4; ++q) coords[(size_t)i * 4 + q] = orig_coords[(size_t)i * 4 + q];
    angles[i] = 0.0;
    neg[i] = 0.0;
    conf[i] = 0.0;
    oidx[i] = i;
}

// The same start with the windows computed HERE from the grid's closed form (round 5; the boxes and level constants of a 1080p
// frame were 97 KB of pageable host memory copied per frame): level L holds ny x nx windows, y-major, at
// numpy.linspace(0, stop, n) positions — i * (stop / (n - 1)), the last one `stop` itself, a single one 0.0 — and a window is
// (posX, posY, posX + pw - 1, posY + ph - 1)  (face_analysis.py:630-646, :661-669; grid.level_boxes).  float64 in numpy's
// operation order, no contraction: the boxes equal the host's bit for bit (tests/test_cascade.py).
constexpr int kMaxLevels = 32;
struct LevelTable {
    int32_t n_levels, pad;
    int32_t first[kMaxLevels + 1];      // first window of every level, and the total
    hg_cascade_level lv[kMaxLevels];
};

__global__ void k_cascade_init_grid(LevelTable T, double* __restrict__ orig_coords, double* __restrict__ orig_level, double* __restrict__ coords,
                                    double* __restrict__ angles, double* __restrict__ neg, double* __restrict__ conf, int32_t* __restrict__ oidx,
                                    int32_t* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = T.first[T.n_levels];
    if (i == 0 && count) *count = n;
    if (i >= n) return;
    int L = 0;
    while (L + 1 < T.n_levels && i >= T.first[L + 1]) ++L;
    const hg_cascade_level& V = T.lv[L];
    const int k = i - T.first[L], iy = k / V.nx, ix = k - iy * V.nx;
    auto lin = [](int j, int num, double stop) -> double {
        if (num <= 1 || j == 0) return 0.0;
        if (j == num - 1) return stop;
        return __dmul_rn((double)j, __ddiv_rn(stop, (double)(num - 1)));
    };
    const double x0 = lin(ix, V.nx, V.x_stop), y0 = lin(iy, V.ny, V.y_stop);
    const double b[4] = {x0, y0, __dsub_rn(__dadd_rn(x0, V.patch_w), 1.0), __dsub_rn(__dadd_rn(y0, V.patch_h), 1.0)};
    for (int q = 0; q < 4; ++q) orig_coords[(size_t)i * 4 + q] = b[q];
    orig_level[(size_t)i * 3] = V.max_dx;
    orig_level[(size_t)i * 3 + 1] = V.max_dy;
    orig_level[(size_t)i * 3 + 2] = V.base_side;
    if (coords) {
        for (int q = 0; q < 4; ++q) coords[(size_t)i * 4 + q] = b[q];
        angles[i] = 0.0;
        neg[i] = 0.0;
        conf[i] = 0.0;
        oidx[i] = i;
    }
}

// A GROUP of stages (round 5): a stage that owns a network and the stages behind it whose network is None read the SAME sl
// (FaceDetectUpdated.py:678-682, :704-706; Pipelines/Pipeline_experimental.txt:8-19: PosX owns the flow, PosY / PAng / Scale
// reuse it).  Rows are independent and a discard is per row, so the reference's m rounds of (regression, update, discard,
// compaction) are ONE regression launch on the group's rows (hg_gauss_regression_multi_device: reg[s * stride + i]) and ONE glue
// launch that applies the m updates and discard tests to every row in stage order — a row discarded at stage s takes no part in
// the later ones, exactly as if it had been compacted away — and compacts once: the same survivors in the same order with the
// same bits.  Survivor counts after every stage of the group are still produced (the reference counts them, :707).
constexpr int kMaxGroup = hg::kGaussMaxMulti;
struct GroupDesc {
    int32_t m, pad;
    int32_t type[kMaxGroup];
    double cut[kMaxGroup];      // cut_offs_face[serial] of a Disc stage
};

// candidate i through the group's stages; returns whether it survives all of them, alive_after bit s = alive after stage s
// (requesting the regression values of ALL the group's stages and the original window's nine numbers up front, so that the loads
// overlap, was measured late in round 5: k_cascade_group 62.8 -> 70.1 us per frame — most candidates die at the group's first stage
// and never needed them.  Not kept.)
__device__ __forceinline__ bool group_one(const GroupDesc& G, const hg_cascade_consts& c0, const StageArrays& A, int64_t reg_stride, int i, double& x0,
                                          double& y0, double& x1, double& y1, double& ang, double& cf, int32_t oi, unsigned& alive_after) {
    bool alive = true;
    alive_after = 0;
    for (int s = 0; s < G.m; ++s) {
        if (alive) {
            const double r = A.reg[(size_t)s * reg_stride + i];
            hg_cascade_consts c = c0;
            c.cut_off_face = G.cut[s];
            const bool wrong = update_one(G.type[s], c, x0, y0, x1, y1, ang, r, A.orig_coords + (size_t)oi * 4, A.orig_angles[oi], A.orig_level + (size_t)oi * 3);
            if (G.type[s] == HG_STAGE_DISC) cf = r;      // FaceDetectUpdated.py:758-759
            alive = !wrong;
        }
        alive_after |= (alive ? 1u : 0u) << s;
    }
    return alive;
}

// host_count (pinned): {count, sequence number, survivors after stage 0 .. m-1 of the group}
__device__ __forceinline__ void publish_group(int32_t* host_count, const int* scount, int m, int cnt, int seq) {
    for (int s = 0; s + 1 < m; ++s) host_count[2 + s] = scount[s];
    host_count[2 + m - 1] = cnt;
    publish_count(host_count, cnt, seq);
}

// One workgroup of 1024 threads (a frame has at most a few thousand candidates).
__global__ void __launch_bounds__(1024) k_cascade_group(GroupDesc G, hg_cascade_consts c, StageArrays A, int64_t reg_stride, int32_t* host_count, int seq, HostResults H) {
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ int scount[kMaxGroup];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(*A.count_in, A.n_max);
    const int cur = A.cur, nxt = 1 - cur;
    if (tid == 0) base = 0;
    if (tid < kMaxGroup) scount[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
        const int i = i0 + tid;
        int keep = 0;
        double x0 = 0, y0 = 0, x1 = 0, y1 = 0, ang = 0, cf = 0;
        int32_t oi = 0;
        unsigned after = 0;
        if (i < n) {
            x0 = A.coords[cur][i * 4]; y0 = A.coords[cur][i * 4 + 1]; x1 = A.coords[cur][i * 4 + 2]; y1 = A.coords[cur][i * 4 + 3];
            ang = A.angles[cur][i];
            cf = A.conf[cur][i];
            oi = A.oidx[cur][i];
            keep = group_one(G, c, A, reg_stride, i, x0, y0, x1, y1, ang, cf, oi, after) ? 1 : 0;
        }
        for (int s = 0; s + 1 < G.m; ++s) {      // survivors after the group's inner stages (the last stage's count is the compaction's)
            const unsigned long long ms = __ballot((after >> s) & 1u);
            if (lane == 0 && ms) atomicAdd(&scount[s], (int)__popcll(ms));
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            const int j = off + before;
            A.map[j] = i;
            A.coords[nxt][j * 4] = x0; A.coords[nxt][j * 4 + 1] = y0; A.coords[nxt][j * 4 + 2] = x1; A.coords[nxt][j * 4 + 3] = y1;
            A.angles[nxt][j] = ang;
            A.neg_angles[j] = -ang;                                         // what the next extraction is called with (face_analysis.py:782)
            A.oidx[nxt][j] = oi;
            A.conf[nxt][j] = cf;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    const int cnt = base;
    // features of the survivors (map is complete: barriers above)
    const int kf = A.k_feat;
    for (int e = tid; e < cnt * kf; e += blockDim.x) {
        const int j = e / kf, f = e - j * kf;
        A.sl[nxt][(size_t)j * kf + f] = A.sl[cur][(size_t)A.map[j] * kf + f];
    }
    if (H.cap > 0 && cnt <= H.cap) {      // last group: the survivors themselves, for the host (nxt[] is complete: barrier)
        __syncthreads();
        for (int j = tid; j < cnt; j += blockDim.x) {
            for (int q = 0; q < 4; ++q) H.coords[(size_t)j * 4 + q] = A.coords[nxt][(size_t)j * 4 + q];
            H.angles[j] = A.angles[nxt][j];
            H.conf[j] = A.conf[nxt][j];
            H.oidx[j] = A.oidx[nxt][j];
        }
        __threadfence_system();
        __syncthreads();
    }
    if (tid == 0) {
        *A.count_out = cnt;
        if (host_count) publish_group(host_count, scount, G.m, cnt, seq);
    }
}

// (Round 4 measured regression AND glue of a stage as ONE launch — a workgroup per R candidates runs the regression of its rows,
// publishes them with write-through stores, draws a ticket, and the workgroup with the last ticket runs the glue: 16 us per stage
// against 8.7 + 5.4 for the two launches (1.57 against 1.49 ms per frame; with a device-scope fence instead of write-through
// stores 1.66: the fence writes back the whole L2 of the workgroup's XCD).  Not kept.)

// The same group for batches of many frames' windows (the single workgroup above walks 1024 candidates per step): two launches
// of one workgroup per kChunk candidates.  `mark` updates every candidate IN PLACE in the cur arrays (nobody else reads them any
// more), writes its keep flag and the chunk's survivor counts (one per stage of the group); `scatter` turns the chunk counts into
// its base offset (a sum over the chunks before it), repeats the chunk-local scan on the flags and moves the survivors, order
// preserved, into the nxt arrays.
constexpr int kChunk = 4096;

__global__ void __launch_bounds__(1024) k_cascade_group_mark(GroupDesc G, hg_cascade_consts c, StageArrays A, int64_t reg_stride, int32_t* __restrict__ chunk_count) {
    __shared__ int scount[kMaxGroup];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = min(*A.count_in, A.n_max), cur = A.cur;
    const int lo = blockIdx.x * kChunk, hi = min(n, lo + kChunk);
    if (tid < kMaxGroup) scount[tid] = 0;
    __syncthreads();
    for (int i0 = lo; i0 < hi; i0 += blockDim.x) {
        const int i = i0 + tid;
        unsigned after = 0;
        if (i < hi) {
            double x0 = A.coords[cur][(size_t)i * 4], y0 = A.coords[cur][(size_t)i * 4 + 1], x1 = A.coords[cur][(size_t)i * 4 + 2], y1 = A.coords[cur][(size_t)i * 4 + 3];
            double ang = A.angles[cur][i], cf = A.conf[cur][i];
            const int32_t oi = A.oidx[cur][i];
            const bool keep = group_one(G, c, A, reg_stride, i, x0, y0, x1, y1, ang, cf, oi, after);
            A.coords[cur][(size_t)i * 4] = x0; A.coords[cur][(size_t)i * 4 + 1] = y0; A.coords[cur][(size_t)i * 4 + 2] = x1; A.coords[cur][(size_t)i * 4 + 3] = y1;
            A.angles[cur][i] = ang;
            A.conf[cur][i] = cf;
            A.discard[i] = keep ? 0 : 1;
        }
        for (int s = 0; s < G.m; ++s) {
            const unsigned long long ms = __ballot((after >> s) & 1u);
            if (lane == 0 && ms) atomicAdd(&scount[s], (int)__popcll(ms));
        }
    }
    __syncthreads();
    if (tid < kMaxGroup) chunk_count[blockIdx.x * kMaxGroup + tid] = scount[tid];
}

__global__ void __launch_bounds__(1024) k_cascade_group_scatter(int m, StageArrays A, const int32_t* __restrict__ chunk_count, int32_t* host_count, int seq) {
    __shared__ int wsum[16];
    __shared__ int base;
    __shared__ int stot[kMaxGroup];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(*A.count_in, A.n_max), cur = A.cur, nxt = 1 - cur;
    const int lo = blockIdx.x * kChunk, hi = min(n, lo + kChunk);
    // base = survivors of the chunks before this one (the last workgroup also learns the totals)
    int part = 0;
    for (int b = tid; b < (int)blockIdx.x; b += blockDim.x) part += chunk_count[b * kMaxGroup + m - 1];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
        base = t;
    }
    __syncthreads();
    const int kf = A.k_feat;
    for (int i0 = lo; i0 < hi; i0 += blockDim.x) {
        const int i = i0 + tid;
        const int keep = (i < hi && A.discard[i] == 0) ? 1 : 0;
        const unsigned long long mk = __ballot(keep);
        const int before = __popcll(mk & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(mk);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (keep) {
            const int j = off + before;
            const double ang = A.angles[cur][i];
            A.map[j] = i;
            for (int q = 0; q < 4; ++q) A.coords[nxt][(size_t)j * 4 + q] = A.coords[cur][(size_t)i * 4 + q];
            A.angles[nxt][j] = ang;
            A.neg_angles[j] = -ang;
            A.oidx[nxt][j] = A.oidx[cur][i];
            A.conf[nxt][j] = A.conf[cur][i];
            for (int f = 0; f < kf; ++f) A.sl[nxt][(size_t)j * kf + f] = A.sl[cur][(size_t)i * kf + f];
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1) {      // the last chunk's end offset is the total
        if (host_count) {      // totals of the inner stages, for the host
            if (tid < kMaxGroup) stot[tid] = 0;
            __syncthreads();
            for (int s = 0; s + 1 < m; ++s) {
                int t = 0;
                for (int b = tid; b < (int)gridDim.x; b += blockDim.x) t += chunk_count[b * kMaxGroup + s];
                for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                if (lane == 0 && t) atomicAdd(&stot[s], t);
            }
            __syncthreads();
        }
        if (tid == 0) {
            *A.count_out = base;
            if (host_count) publish_group(host_count, stot, m, base, seq);
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

// ---- the whole stage loop as one host call ---------------------------------------------------------------------------
struct hg_cascade {
    int device = 0, w = 0, h = 0, k = 0;
    std::vector<hg_cascade_stage> stages;
    hg_cascade_consts base{};
    double cut_offs[10];
    hg_patcher* patcher = nullptr;
    int64_t cap = 0;
    hg::DevBuf coords[2], angles[2], conf[2], oidx[2], sl[2], subs[2], neg, reg, discard, map, count, orig_coords, orig_level, orig_angles, chunk_count;
    hg::DevBuf pre_box, pre_frame;       // hg_cascade_detect_frame_device: the box of the whole frame and the prescaled frame
    int pre_src_w = 0, pre_src_h = 0;    // ... the frame size pre_box was written for
    int32_t* host_count = nullptr;       // pinned, device-visible: {count, sequence number} (publish_count)
    int32_t seq = 0;                     // sequence number of the last read-back asked for
    char* host_res = nullptr;            // pinned: the final survivors (HostResults), kResCap rows
    static constexpr int kResCap = 4096;

    void reserve(int64_t n0) {
        if (n0 <= cap) return;
        for (int b = 0; b < 2; ++b) {
            coords[b].alloc((size_t)n0 * 32);
            angles[b].alloc((size_t)n0 * 8);
            conf[b].alloc((size_t)n0 * 8);
            oidx[b].alloc((size_t)n0 * 4);
            sl[b].alloc((size_t)n0 * k * 4);
            subs[b].alloc((size_t)n0 * w * h);
        }
        neg.alloc((size_t)n0 * 8);
        reg.alloc((size_t)n0 * 8 * kMaxGroup);      // regressions of a group of stages: [stage of the group][candidate], stride = cap
        discard.alloc((size_t)n0);
        map.alloc((size_t)n0 * 4);
        count.alloc(16);
        orig_coords.alloc((size_t)n0 * 32);
        orig_level.alloc((size_t)n0 * 24);
        orig_angles.alloc((size_t)n0 * 8);
        HG_HIP(hipMemset(orig_angles.p, 0, (size_t)n0 * 8));
        for (auto& st : stages)
            if (st.flow && hg_flow_reserve(st.flow, n0) != HG_OK) hg::fail(HG_ERR_NOMEM, "%s", hg_last_error());
        cap = n0;
    }
};

int hg_cascade_create(const hg_cascade_stage* stages, int n_stages, int sub_w, int sub_h, int n_features, const hg_cascade_consts* consts,
                      const double* cut_offs_face, int n_cut_offs, int device, hg_cascade** out) {
    return guarded([&] {
        if (!out) hg::fail(HG_ERR_ARG, "null output handle pointer");
        *out = nullptr;
        if (!stages || n_stages <= 0 || n_stages > 64 || !consts || !cut_offs_face) hg::fail(HG_ERR_ARG, "bad stage list");
        if (sub_w <= 0 || sub_h <= 0 || (sub_w * sub_h) % 4 || n_features <= 0 || n_features > 256) hg::fail(HG_ERR_ARG, "bad sub-image size / feature count");
        for (int k = 0; k < n_stages; ++k) {
            if (stages[k].type < HG_STAGE_DISC || stages[k].type > HG_STAGE_SCALE) hg::fail(HG_ERR_ARG, "stage %d: unknown type %d", k, stages[k].type);
            if (stages[k].serial < 0 || stages[k].serial >= n_cut_offs || n_cut_offs > 10) hg::fail(HG_ERR_ARG, "stage %d: serial %d has no cut-off", k, stages[k].serial);
            if (!stages[k].classifier) hg::fail(HG_ERR_ARG, "stage %d: no classifier", k);
            if (k == 0 && !stages[k].flow) hg::fail(HG_ERR_ARG, "the first stage needs a network");
        }
        set_dev(device);
        auto c = std::make_unique<hg_cascade>();
        c->device = device;
        c->w = sub_w;
        c->h = sub_h;
        c->k = n_features;
        c->stages.assign(stages, stages + n_stages);
        c->base = *consts;
        for (int i = 0; i < 10; ++i) c->cut_offs[i] = i < n_cut_offs ? cut_offs_face[i] : 0.0;
        if (hg_patcher_create(device, &c->patcher) != HG_OK) hg::fail(HG_ERR_DEVICE, "%s", hg_last_error());
        HG_HIP(hipHostMalloc((void**)&c->host_count, 64, hipHostMallocDefault));
        c->host_count[0] = c->host_count[1] = 0;
        HG_HIP(hipHostMalloc((void**)&c->host_res, (size_t)hg_cascade::kResCap * 52, hipHostMallocDefault));
        *out = c.release();
    });
}

void hg_cascade_free(hg_cascade* c) {
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->patcher) hg_patcher_free(c->patcher);
        if (c->host_count) (void)hipHostFree(c->host_count);
        if (c->host_res) (void)hipHostFree(c->host_res);
    }
    delete c;
}

}  // extern "C"

namespace {

LevelTable make_level_table(const hg_cascade_level* levels, int n_levels) {
    if (!levels || n_levels < 1 || n_levels > kMaxLevels) hg::fail(HG_ERR_ARG, "1..%d pyramid levels", kMaxLevels);
    LevelTable T{};
    T.n_levels = n_levels;
    int64_t tot = 0;
    for (int L = 0; L < n_levels; ++L) {
        const hg_cascade_level& v = levels[L];
        if (v.nx < 1 || v.ny < 1 || (int64_t)v.nx * v.ny > 0x7fffffffll / 64) hg::fail(HG_ERR_ARG, "level %d: bad grid %d x %d", L, v.nx, v.ny);
        T.first[L] = (int32_t)tot;
        tot += (int64_t)v.nx * v.ny;
        if (tot > 0x7fffffffll / 64) hg::fail(HG_ERR_ARG, "too many windows");
        T.lv[L] = v;
    }
    T.first[n_levels] = (int32_t)tot;
    return T;
}

// The stage loop.  `T`: the windows come from the grid's closed form (k_cascade_init_grid), else from boxes_host / level_host.
// FNV-1a over a plain struct: the key under which the patcher keeps index tables of boxes that depend on sizes only
uint64_t key_of(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char*)p)[i]) * 1099511628211ull;
    return h ? h : 1;
}

void detect_impl(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld, const double* boxes_host, const double* level_host,
                 const LevelTable* T, int64_t n0, double* out_coords, double* out_angles, int32_t* out_orig_index, double* out_confidence,
                 int64_t out_cap, int64_t* n_out, int32_t* stage_counts, int64_t* rows_executed, void* stream) {
    if (!c || !n_out) hg::fail(HG_ERR_ARG, "null argument");
    if (n0 < 0 || n0 > 0x7fffffffll / 64) hg::fail(HG_ERR_ARG, "bad window count");
    if (n0 > 0 && (!frame_dev || (!T && (!boxes_host || !level_host)))) hg::fail(HG_ERR_ARG, "null data pointer");
    set_dev(c->device);
    hipStream_t st = (hipStream_t)stream;
    const int ns = (int)c->stages.size();
    *n_out = 0;
    if (rows_executed) *rows_executed = 0;
    if (n0 == 0) {
        for (int k = 0; k < ns && stage_counts; ++k) stage_counts[k] = 0;
        return;
    }
    c->reserve(n0);
    if (T) {
        hipLaunchKernelGGL(k_cascade_init_grid, (unsigned)((n0 + 255) / 256), 256, 0, st, *T, (double*)c->orig_coords.p, (double*)c->orig_level.p,
                           (double*)c->coords[0].p, (double*)c->angles[0].p, (double*)c->neg.p, (double*)c->conf[0].p, (int32_t*)c->oidx[0].p,
                           (int32_t*)c->count.p);
    } else {
        HG_HIP(hipMemcpyAsync(c->orig_coords.p, boxes_host, (size_t)n0 * 32, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(c->orig_level.p, level_host, (size_t)n0 * 24, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_cascade_init, (unsigned)((n0 + 255) / 256), 256, 0, st, (int)n0, (const double*)c->orig_coords.p, (double*)c->coords[0].p,
                           (double*)c->angles[0].p, (double*)c->neg.p, (double*)c->conf[0].p, (int32_t*)c->oidx[0].p, (int32_t*)c->count.p);
    }
    HG_HIP(hipGetLastError());
    // the count word is polled, never waited for with a stream synchronisation (publish_count); a deadline guards against a
    // device that never answers
    auto poll_count = [&](int32_t seq) -> int64_t {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(c->host_count + 1, __ATOMIC_ACQUIRE) == seq) return c->host_count[0];
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
                HG_HIP(hipStreamSynchronize(st));
                if (__atomic_load_n(c->host_count + 1, __ATOMIC_ACQUIRE) == seq) return c->host_count[0];
                hg::fail(HG_ERR_DEVICE, "cascade stage did not report its survivor count");
            }
            __builtin_ia32_pause();
        }
    };
    HostResults H{};
    H.coords = (double*)c->host_res;
    H.angles = (double*)(c->host_res + (size_t)hg_cascade::kResCap * 32);
    H.conf = (double*)(c->host_res + (size_t)hg_cascade::kResCap * 40);
    H.oidx = (int32_t*)(c->host_res + (size_t)hg_cascade::kResCap * 48);
    bool results_on_host = false;
    int cur = 0, cnt_slot = 0;
    int sb = 0;                                    // which of subs[] holds the extracted sub-images, row-aligned with the candidates
    int64_t n_bound = n0, rows = 0;                // n_bound: host-side upper bound of the candidate count (exact after a Disc stage)
    const size_t row = (size_t)c->w * c->h;
    // The reference compacts subimages_arr after EVERY stage (FaceDetectUpdated.py:753) and reuses it in a stage that follows a
    // Disc stage and owns a network (:674-681).  Here the rows travel only while a later stage will read them before the next
    // extraction replaces them: carry[k] = "stage k's survivors' sub-images are needed again".  (In the shipped pipeline that is
    // after Disc1 / Disc3 / Disc5 / Disc7 only; a pipeline such as PosX(net), Disc(None), PosX(net) carries them through two stages.)
    std::vector<char> carry((size_t)ns, 0);
    {
        bool need = false;                         // need at the entry of stage k + 1
        for (int k = ns - 1; k >= 0; --k) {
            carry[(size_t)k] = need;
            const bool prev_disc = k > 0 && c->stages[k - 1].type == HG_STAGE_DISC;
            if (c->stages[k].flow) need = prev_disc;          // reuses them (needs them at entry) or extracts afresh (does not)
        }
    }
    const bool no_groups = getenv("HIGSFA_CASCADE_NO_GROUPS") != nullptr;      // every stage a group of its own (tests, A/B); read per frame
    for (int k = 0; k < ns;) {
        // the group: stage k and the stages behind it that have no network of their own (they read the same sl), at most kMaxGroup
        int m = 1;
        while (!no_groups && k + m < ns && m < kMaxGroup && !c->stages[k + m].flow) ++m;
        const hg_cascade_stage& S = c->stages[k];
        if (n_bound == 0) {
            for (int s = 0; s < m && stage_counts; ++s) stage_counts[k + s] = 0;
            k += m;
            continue;
        }
        const bool skip_extract = (k > 0 && c->stages[k - 1].type == HG_STAGE_DISC) || !S.flow;      // FaceDetectUpdated.py:674-681
        if (!skip_extract) {
            // (the first stage's angles are all zero: the plain EXTENT kernel — a window with delta_ang == 0 is cut from the frame itself —
            // and where the windows come from the level table they are a function of that table alone: their index tables are kept)
            const int rc = k == 0 ? hg_patcher_extract_keyed_device(c->patcher, T ? key_of(T, sizeof *T) : 0, frame_dev, HG_U8, frame_h, frame_w, ld,
                                                                    (const double*)c->coords[cur].p, n_bound, c->w, c->h, c->subs[sb].p, HG_U8, (int64_t)row, st)
                                  : hg_patcher_extract_rotate_device(c->patcher, frame_dev, HG_U8, frame_h, frame_w, ld, (const double*)c->coords[cur].p,
                                                                     (const double*)c->neg.p, n_bound, c->w, c->h, c->subs[sb].p, HG_U8, (int64_t)row, st);
            if (rc != HG_OK) hg::fail(HG_ERR_DEVICE, "%s", hg_last_error());
        }
        if (S.flow) {
            if (hg_flow_execute_device(S.flow, c->subs[sb].p, HG_U8, n_bound, (int64_t)row, c->sl[cur].p, HG_F32, c->k, c->k, st) != HG_OK)
                hg::fail(HG_ERR_DEVICE, "%s", hg_last_error());
            rows += n_bound;
        }
        GroupDesc G{};
        G.m = m;
        hg_gauss* clf[kMaxGroup] = {};
        bool any_disc = false;
        for (int s = 0; s < m; ++s) {
            const hg_cascade_stage& Ss = c->stages[k + s];
            G.type[s] = Ss.type;
            G.cut[s] = c->cut_offs[Ss.serial];
            clf[s] = Ss.classifier;
            any_disc = any_disc || Ss.type == HG_STAGE_DISC;
        }
        // the group's regressions, all on the rows of sl[cur]: reg[s * cap + i]
        if (hg_gauss_regression_multi_device(clf, m, c->sl[cur].p, HG_F32, n_bound, c->k, (double*)c->reg.p, c->cap, st) != HG_OK)
            hg::fail(HG_ERR_DEVICE, "%s", hg_last_error());
        StageArrays A{};
        for (int b = 0; b < 2; ++b) {
            A.coords[b] = (double*)c->coords[b].p;
            A.angles[b] = (double*)c->angles[b].p;
            A.conf[b] = (double*)c->conf[b].p;
            A.oidx[b] = (int32_t*)c->oidx[b].p;
            A.sl[b] = (float*)c->sl[b].p;
        }
        A.neg_angles = (double*)c->neg.p;
        A.reg = (const double*)c->reg.p;
        A.orig_coords = (const double*)c->orig_coords.p;
        A.orig_angles = (const double*)c->orig_angles.p;
        A.orig_level = (const double*)c->orig_level.p;
        A.discard = (uint8_t*)c->discard.p;
        A.map = (int32_t*)c->map.p;
        A.count_in = (int32_t*)c->count.p + cnt_slot;
        A.count_out = (int32_t*)c->count.p + (1 - cnt_slot);
        A.k_feat = c->k;
        A.cur = cur;
        A.n_max = (int32_t)n_bound;
        // the host needs the exact count where it shrinks and sizes the next launches: after every group with a Disc stage (the
        // read-back is a poll of a pinned word, cheap enough for the small stages too), and at the end
        const bool last = k + m == ns;
        const bool want_count = any_disc || last;
        const int32_t seq = want_count ? ++c->seq : 0;
        HostResults Hk{};
        if (n_bound > 2 * kChunk) {      // many frames' windows: one workgroup per kChunk candidates, two launches
            const unsigned chunks = (unsigned)((n_bound + kChunk - 1) / kChunk);
            c->chunk_count.alloc((size_t)chunks * kMaxGroup * 4);
            hipLaunchKernelGGL(k_cascade_group_mark, chunks, 1024, 0, st, G, c->base, A, c->cap, (int32_t*)c->chunk_count.p);
            hipLaunchKernelGGL(k_cascade_group_scatter, chunks, 1024, 0, st, m, A, (const int32_t*)c->chunk_count.p, want_count ? c->host_count : nullptr, seq);
        } else {
            if (last && n_bound <= hg_cascade::kResCap) {
                Hk = H;
                Hk.cap = hg_cascade::kResCap;
                results_on_host = true;
            }
            hipLaunchKernelGGL(k_cascade_group, 1, 1024, 0, st, G, c->base, A, c->cap, want_count ? c->host_count : nullptr, seq, Hk);
        }
        if (carry[(size_t)(k + m - 1)]) {       // the group's compaction applied to the sub-images as well (:753)
            const int vec16 = row % 16 == 0 ? 1 : 0;
            hipLaunchKernelGGL(k_gather_rows, (unsigned)std::min<int64_t>(n_bound, 16384), 256, 0, st, (const char*)c->subs[sb].p, (char*)c->subs[1 - sb].p,
                               (int64_t)row, (const int32_t*)c->map.p, (const int32_t*)A.count_out, vec16);
            sb = 1 - sb;
        }
        HG_HIP(hipGetLastError());
        cur = 1 - cur;
        cnt_slot = 1 - cnt_slot;
        if (want_count) {
            n_bound = poll_count(seq);
            for (int s = 0; s < m && stage_counts; ++s) stage_counts[k + s] = c->host_count[2 + s];      // written before the sequence number
        } else {
            for (int s = 0; s < m && stage_counts; ++s) stage_counts[k + s] = -1;      // not read back (no stage of the group discards by a cut-off: bound of the last Disc stage)
        }
        k += m;
    }
    if (n_bound > out_cap) hg::fail(HG_ERR_ARG, "%lld detections but room for %lld", (long long)n_bound, (long long)out_cap);
    if (n_bound > 0 && results_on_host) {      // written by the last group's kernel before it published the count
        if (out_coords) memcpy(out_coords, H.coords, (size_t)n_bound * 32);
        if (out_angles) memcpy(out_angles, H.angles, (size_t)n_bound * 8);
        if (out_orig_index) memcpy(out_orig_index, H.oidx, (size_t)n_bound * 4);
        if (out_confidence) memcpy(out_confidence, H.conf, (size_t)n_bound * 8);
    } else if (n_bound > 0) {
        if (out_coords) HG_HIP(hipMemcpyAsync(out_coords, c->coords[cur].p, (size_t)n_bound * 32, hipMemcpyDeviceToHost, st));
        if (out_angles) HG_HIP(hipMemcpyAsync(out_angles, c->angles[cur].p, (size_t)n_bound * 8, hipMemcpyDeviceToHost, st));
        if (out_orig_index) HG_HIP(hipMemcpyAsync(out_orig_index, c->oidx[cur].p, (size_t)n_bound * 4, hipMemcpyDeviceToHost, st));
        if (out_confidence) HG_HIP(hipMemcpyAsync(out_confidence, c->conf[cur].p, (size_t)n_bound * 8, hipMemcpyDeviceToHost, st));
        HG_HIP(hipStreamSynchronize(st));
    }
    *n_out = n_bound;
    if (rows_executed) *rows_executed = rows;
}

}  // namespace

extern "C" {

int hg_cascade_detect_device(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld, const double* boxes_host,
                             const double* level_host, int64_t n0, double* out_coords, double* out_angles, int32_t* out_orig_index,
                             double* out_confidence, int64_t out_cap, int64_t* n_out, int32_t* stage_counts, int64_t* rows_executed,
                             void* stream) {
    return guarded([&] {
        detect_impl(c, frame_dev, frame_h, frame_w, ld, boxes_host, level_host, nullptr, n0, out_coords, out_angles, out_orig_index, out_confidence,
                    out_cap, n_out, stage_counts, rows_executed, stream);
    });
}

int hg_cascade_detect_levels_device(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld, const hg_cascade_level* levels,
                                    int n_levels, double* out_coords, double* out_angles, int32_t* out_orig_index, double* out_confidence,
                                    int64_t out_cap, int64_t* n_out, int32_t* stage_counts, int64_t* rows_executed, void* stream) {
    return guarded([&] {
        const LevelTable T = make_level_table(levels, n_levels);
        detect_impl(c, frame_dev, frame_h, frame_w, ld, nullptr, nullptr, &T, T.first[T.n_levels], out_coords, out_angles, out_orig_index,
                    out_confidence, out_cap, n_out, stage_counts, rows_executed, stream);
    });
}

int hg_cascade_detect_frame_device(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld, int prescale_w, int prescale_h,
                                   const hg_cascade_level* levels, int n_levels, double* out_coords, double* out_angles, int32_t* out_orig_index,
                                   double* out_confidence, int64_t out_cap, int64_t* n_out, int32_t* stage_counts, int64_t* rows_executed, void* stream) {
    return guarded([&] {
        if (!c) hg::fail(HG_ERR_ARG, "null argument");
        const LevelTable T = make_level_table(levels, n_levels);
        const void* fr = frame_dev;
        int fh = frame_h, fw = frame_w;
        int64_t fld = ld;
        if (prescale_w > 0 || prescale_h > 0) {
            // FaceDetectUpdated.py:551-561: im.resize((w, h), NEAREST) before the grid is laid out — PIL's nearest resize is the EXTENT
            // rule over the whole frame (tested against PIL), so the patcher does it, into a buffer that lives with the cascade
            if (prescale_w <= 0 || prescale_h <= 0 || !frame_dev) hg::fail(HG_ERR_ARG, "bad prescale size %d x %d", prescale_w, prescale_h);
            set_dev(c->device);
            if (c->pre_src_w != frame_w || c->pre_src_h != frame_h) {
                const double box[4] = {0.0, 0.0, (double)frame_w, (double)frame_h};
                c->pre_box.upload(box, sizeof box);
                c->pre_src_w = frame_w;
                c->pre_src_h = frame_h;
            }
            c->pre_frame.alloc((size_t)prescale_w * prescale_h);
            const int32_t pre_shape[4] = {frame_w, frame_h, prescale_w, prescale_h};      // the whole-frame box depends on these alone
            if (hg_patcher_extract_keyed_device(c->patcher, key_of(pre_shape, sizeof pre_shape, 0x9e3779b97f4a7c15ull), frame_dev, HG_U8, frame_h, frame_w, ld,
                                                (const double*)c->pre_box.p, 1, prescale_w, prescale_h, c->pre_frame.p, HG_U8,
                                                (int64_t)prescale_w * prescale_h, stream) != HG_OK)
                hg::fail(HG_ERR_DEVICE, "%s", hg_last_error());
            fr = c->pre_frame.p;
            fh = prescale_h;
            fw = prescale_w;
            fld = prescale_w;
        }
        detect_impl(c, fr, fh, fw, fld, nullptr, nullptr, &T, T.first[T.n_levels], out_coords, out_angles, out_orig_index, out_confidence, out_cap,
                    n_out, stage_counts, rows_executed, stream);
    });
}

int hg_cascade_grid_device(int device, const hg_cascade_level* levels, int n_levels, double* boxes_dev, double* level_dev, int64_t cap, int64_t* n0,
                           void* stream) {
    return guarded([&] {
        const LevelTable T = make_level_table(levels, n_levels);
        const int64_t n = T.first[T.n_levels];
        if (n0) *n0 = n;
        if (!boxes_dev && !level_dev) return;      // a query of the window count
        if (!boxes_dev || !level_dev || cap < n) hg::fail(HG_ERR_ARG, "room for %lld windows, the grid has %lld", (long long)cap, (long long)n);
        set_dev(device);
        hipLaunchKernelGGL(k_cascade_init_grid, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream, T, boxes_dev, level_dev, (double*)nullptr,
                           (double*)nullptr, (double*)nullptr, (double*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
        HG_HIP(hipGetLastError());
    });
}

}  // extern "C"

