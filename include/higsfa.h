/*
 * higsfa.h — C ABI of the MI355X-native HiGSFA inference path.
 *
 * Drop-in boundary for PyFaceAnalysis' hot call
 *
 *     sl = networks[num_network].execute(subimages_arr, benchmark=benchmark)
 *                                      (reference: FaceDetectUpdated.py:699;
 *                                       also face_analysis.py:1064 and :1257)
 *
 * The reference has no native code and no FFI; these entry points are what a
 * ctypes/cffi binding for that one call needs: load a flow description, run batches of
 * flattened sub-images through it on one GPU, read results and per-stage timings back.
 * Plain C types only.  All functions return 0 (HG_OK) on success or a negative hg_status;
 * hg_last_error() returns a thread-local message for the last failure on the calling thread.
 *
 * Threading: one hg_flow may be used by one thread at a time; different flows are
 * independent.  The library keeps no global mutable state besides the last-error string.
 * There is NO CPU execution path: without a usable HIP device hg_flow_to_device /
 * hg_flow_execute* fail with HG_ERR_DEVICE.
 */
#ifndef HIGSFA_H
#define HIGSFA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HG_VERSION 100 /* 0.1.0 */

typedef struct hg_flow hg_flow;

/* Element types accepted for the sub-image matrix / produced for the feature matrix.
 * The reference passes float64 arrays holding integer pixel values 0..255
 * (images_asarray, face_analysis.py:786) and receives float64 (MDP node dtype). */
enum hg_dtype { HG_U8 = 0, HG_F32 = 1, HG_F64 = 2 };

enum hg_status {
    HG_OK = 0,
    HG_ERR_ARG = -1,      /* null pointer, bad dtype, bad leading dimension, ...            */
    HG_ERR_FORMAT = -2,   /* malformed / truncated / unsupported blob                        */
    HG_ERR_DIM = -3,      /* input_dim / output_dim mismatch (MDP raises on these too)        */
    HG_ERR_DEVICE = -4,   /* no HIP device, HIP call failed, flow not on a device; also: a persistent kernel of an
                           * earlier call on this flow reported a failed internal hand-off (its features are invalid) */
    HG_ERR_NOMEM = -5,
    HG_ERR_STATE = -6     /* call made in the wrong state (e.g. timings without profiling)    */
};

/* Which execution plan the loader chose for a flow. */
enum hg_plan_kind {
    HG_PLAN_GENERIC = 0,  /* step-by-step kernels on row-major activations (any flow)         */
    HG_PLAN_FUSED = 1     /* per-layer fused gather+affine+expansion+affine MFMA kernels       */
};

typedef struct hg_info {
    int64_t input_dim;        /* columns of x  (mdp Flow: flow[0].input_dim)                   */
    int64_t output_dim;       /* columns of y  (flow[-1].output_dim)                           */
    int32_t n_top_nodes;      /* len(flow) — top-level nodes as the reference's flow has them  */
    int32_t plan_kind;        /* enum hg_plan_kind                                             */
    int32_t n_stages;         /* kernels (fused) or steps (generic) launched per execute       */
    int32_t device;           /* HIP device ordinal, -1 while host-only                        */
    int64_t weight_bytes;     /* device bytes held by weights/descriptors                      */
    int64_t flops_per_row;    /* algorithmic FLOPs per sub-image: sum 2*in*out over affines    */
    int64_t padded_flops_per_row; /* FLOPs the fused MFMA tiling actually issues per row       */
    int64_t workspace_bytes;  /* device bytes currently reserved for activations              */
} hg_info;

int hg_version(void);
const char* hg_last_error(void);

/* Number of visible HIP devices (0 and HG_OK when there are none / no driver). */
int hg_device_count(int* count);

/* Parse a flow blob ("HGSFAFL1", see pyfaceanalysis_amd/blob.py) and build the execution
 * plan on the host.  Replaces: cache_obj.load_obj_from_cache (pickle) in
 * face_analysis.py:457.  Touches no GPU.  `flags`: bit0 = force the generic plan. */
int hg_flow_load(const void* blob, size_t nbytes, int flags, hg_flow** out);
void hg_flow_free(hg_flow* f);

int hg_flow_info(const hg_flow* f, hg_info* info);

/* Human-readable plan listing (counterpart of more_nodes.describe_flow,
 * FaceDetectUpdated.py:193).  Writes at most cap-1 chars + NUL; returns needed length in
 * *needed when non-null. */
int hg_flow_describe(const hg_flow* f, char* buf, size_t cap, size_t* needed);

/* Upload weights/descriptors to HIP device `device` (one device per flow handle). */
int hg_flow_to_device(hg_flow* f, int device);

/* Pre-size the activation workspace for batches of up to `max_rows` rows so that
 * hg_flow_execute_device performs no allocation. */
int hg_flow_reserve(hg_flow* f, int64_t max_rows);

/* y[n, 0:y_cols] = flow(x[n, :]) with HOST buffers (the ndarray-in / ndarray-out call of
 * FaceDetectUpdated.py:699).  x: n rows of input_dim elements, row stride ldx elements;
 * y: n rows, the first y_cols (<= output_dim) columns, row stride ldy elements.  The caller
 * usually wants only the first classifier.input_dim columns (FaceDetectUpdated.py:709,719).
 * n == 0 is valid and a no-op (reference guards len(subimages_arr) > 0 at :694).
 * Synchronous: returns after y is complete.  A pool of host threads on the memory node of the rows packs them while the
 * device works on earlier rows (passes of a few hundred rows, sized by a planner): straight into device memory on large-BAR
 * devices, through a pinned ring and two copy queues otherwise; float32 / float64 rows whose values are all integers 0..255 —
 * what images_asarray produces (face_analysis.py:786) — cross PCIe as uint8, which changes no output bit; from the first
 * row that holds anything else the rest of the call travels in the caller's own type. */
int hg_flow_execute(hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx,
                    void* y, int y_dtype, int64_t y_cols, int64_t ldy);

/* The same call over several devices of this process (sub-images are independent, so the batch is
 * cut into n_devices contiguous row blocks of ceil(n / n_devices) rows; block r runs on devices[r] with replicated
 * weights, its own streams and staging buffers, driven by its own host thread; every block's features land in the
 * caller's y at the block's rows — with host buffers that IS the gather, no peer copy or collective is needed).
 * devices == NULL means 0 .. n_devices-1; a device may be listed more than once (each entry is a replica).
 * Replicas are created on first use and kept in the handle; hg_flow_to_device is not required.
 * Multi-process callers (one rank per GPU, RCCL all-gather of device-resident features) use
 * hg_flow_execute_device per rank instead: pyfaceanalysis_amd/sharded.py. */
int hg_flow_execute_sharded(hg_flow* f, const void* x, int x_dtype, int64_t n, int64_t ldx,
                            void* y, int y_dtype, int64_t y_cols, int64_t ldy,
                            const int* devices, int n_devices);

/* Same with DEVICE buffers on the flow's device, enqueued on `stream` (hipStream_t, may be
 * null = default stream); returns without synchronising. */
int hg_flow_execute_device(hg_flow* f, const void* x_dev, int x_dtype, int64_t n, int64_t ldx,
                           void* y_dev, int y_dtype, int64_t y_cols, int64_t ldy, void* stream);

/* Device-scope events for callers that overlap a collective with the next batch (pyfaceanalysis_amd/sharded.py: the RCCL
 * all-gather of the features on a side stream, ordered against the kernels in both directions).  Created with
 * hipEventDisableTiming | hipEventDisableSystemFence: recording one does not write the caches back to system scope the way
 * a default event does (measured on an MI355X: 13 us between two launches of the stream it is recorded on, against < 2).
 * They order device work only — nothing the host reads may depend on them.  `ev` is an opaque handle (hipEvent_t),
 * streams are hipStream_t of the calling thread's current device. */
int hg_event_create(void** ev);
/* The same on a stated device, whatever the calling thread's current device is (which it leaves unchanged), and with the
 * choice of scope: device_scope 0 creates an ordinary (hipEventDisableTiming) event whose record also publishes to system
 * scope — what pyfaceanalysis_amd/sharded.py uses for the hand-off that carries DATA to the collective unless a run on more
 * than one GPU has verified the device-scope form (bench.py does that before it times anything). */
int hg_event_create_on(void** ev, int device, int device_scope);
void hg_event_destroy(void* ev);
int hg_event_record(void* ev, void* stream);
int hg_stream_wait_event(void* stream, void* ev);
/* 1 when everything the event was last recorded behind has completed, 0 when not yet (never blocks); < 0: hg_status.
 * Lets the host skip a wait that would only put a barrier packet in front of the next launch (measured: 4.5 us). */
int hg_event_query(void* ev);

/* How the last hg_flow_execute of this handle moved the caller's rows (FaceDetectUpdated.py:699 hands over a host ndarray,
 * face_analysis.py:786): *transport = 1 — packer threads stored the wire rows straight into device memory, which is done only
 * where the device reports a large BAR AND hsa_amd_pointer_info confirms every input buffer host-mapped at its device address;
 * 0 — pinned ring + copy queues (HIGSFA_HOST_DIRECT=0, or any other answer of the probe); -1 — no such call yet. */
int hg_flow_host_transport(const hg_flow* f, int* transport);
/* What bounds that call on THIS box, measured in the caller's process (bench.py puts them beside the host-path figures):
 * hg_host_pack_probe — the library's packer threads alone over the caller's array (same pool, placement and routines as
 * hg_flow_execute; destinations stay in cache): the rate at which the host can read and narrow / copy these rows;
 * hg_host_store_probe — the packers storing `bytes` from host memory straight into device memory (*direct = 0 and no time where
 * hg_flow_execute would not do that either); hg_host_dma_probe — one pinned-memory copy of `bytes` through the copy engine.
 * best_seconds: best of `reps` rounds after one untimed round. */
int hg_host_pack_probe(const void* x, int x_dtype, int64_t n, int64_t ldx, int64_t in_dim, int reps, double* best_seconds);
int hg_host_store_probe(int device, size_t bytes, int reps, double* best_seconds, int* direct);
int hg_host_dma_probe(int device, size_t bytes, int reps, double* best_seconds);
/* Per-stage timing (the `benchmark=` kwarg of the reference call; benchmarking.py:39-58).
 * When enabled every stage launch is bracketed by hipEvents on the execution stream. */
int hg_flow_set_profiling(hg_flow* f, int enabled);
/* After a profiled execute has completed: accumulated ms and launch count per stage since
 * the last reset; `cap` entries at most; *n_stages receives the stage count. */
int hg_flow_stage_times(hg_flow* f, double* total_ms, int64_t* launches, int cap, int* n_stages);
int hg_flow_stage_name(const hg_flow* f, int stage, char* buf, size_t cap);
int hg_flow_reset_profile(hg_flow* f);

/* --- Gaussian-classifier soft-label regression -------------------------
 * The step right after the hot call: classifiers[k].regression(sl[:, 0:d], avg_labels)
 * (FaceDetectUpdated.py:709-719).  Parameters as stored in the SavedClassifiers pickles:
 * means (K,d), inv_covs (K,d,d), sqrt_det_covs (K), priors p (K), avg_labels (K); float64.
 * out_reg[n] = sum_c post_c(x_n) avg_labels[c]; out_std (optional) the posterior std. */
typedef struct hg_gauss hg_gauss;
int hg_gauss_create(int32_t n_classes, int32_t dim, const double* means, const double* inv_covs,
                    const double* sqrt_det_covs, const double* priors, const double* avg_labels,
                    int device, hg_gauss** out);
void hg_gauss_free(hg_gauss* g);
/* x_dev: (n, >=dim) device matrix of dtype x_dtype (F32/F64), row stride ldx; outputs are
 * device float64 arrays of n elements (out_std may be null).  Enqueued on `stream`. */
int hg_gauss_regression_device(hg_gauss* g, const void* x_dev, int x_dtype, int64_t n, int64_t ldx,
                               double* out_reg_dev, double* out_std_dev, void* stream);
/* m <= 4 classifiers on the SAME feature rows in one launch — a cascade stage that owns a network and the stages behind it whose
 * network is None reuse one sl (FaceDetectUpdated.py:678-682, :704-706; Pipelines/Pipeline_experimental.txt:8-19): regressions of
 * classifier s go to out_reg_dev[s * out_stride + row].  Same bits as m calls of hg_gauss_regression_device. */
int hg_gauss_regression_multi_device(hg_gauss* const* gs, int m, const void* x_dev, int x_dtype, int64_t n, int64_t ldx,
                                     double* out_reg_dev, int64_t out_stride, void* stream);
/* Host-buffer convenience wrapper (synchronous). */
int hg_gauss_regression(hg_gauss* g, const void* x, int x_dtype, int64_t n, int64_t ldx,
                        double* out_reg, double* out_std);

/* --- On-device sub-image extraction ---------------------------------------
 * The producer of the hot call's input: load_network_subimages -> extract_subimages_rotate
 * (face_analysis.py:775-800; FaceDetectUpdated.py:686), which crops every window with PIL's
 * Image.transform((w, h), EXTENT, (x0, y0, x1, y1), NEAREST).  Same index rule, bit for bit
 * (tested against PIL).  The *_rotate entries also take delta_angs (degrees, counter-clockwise, the
 * reference passes -1 * curr_angles, face_analysis.py:782; null = no rotation): a window with
 * delta_ang != 0 is cut from frame.rotate(delta_ang, NEAREST, center = centre of its box) — PIL's
 * Image.rotate arithmetic (16.16 fixed point), composed with the EXTENT rule per pixel; cuicuilco's own
 * composition is not available, this rule is the build's and is tested against PIL.
 * frame: (frame_h, frame_w) pixels, HG_U8 or HG_F32, row stride ld elements.  boxes: n x 4 doubles
 * (x0, y0, x1, y1) in frame coordinates.  out: n rows of out_w*out_h elements (row-major pixels,
 * the layout flow.execute expects), row stride ldo elements, dtype U8 / F32 / F64. */
typedef struct hg_patcher hg_patcher;
int hg_patcher_create(int device, hg_patcher** out);
void hg_patcher_free(hg_patcher* p);
int hg_patcher_extract_device(hg_patcher* p, const void* frame_dev, int frame_dtype, int frame_h, int frame_w,
                              int64_t ld, const double* boxes_dev, int64_t n, int out_w, int out_h,
                              void* out_dev, int out_dtype, int64_t ldo, void* stream);
/* The same for boxes the caller declares UNCHANGED between calls (key != 0; e.g. the prescale's whole-frame box, the first-stage
 * grid of a frame size, which depend on the frame's size only — face_analysis.py:630-669): the index tables are built on the first
 * call with a key (and whenever n or a size differs from what the key was built for) and reused afterwards.  Unrotated windows.
 * key = 0: hg_patcher_extract_device. */
int hg_patcher_extract_keyed_device(hg_patcher* p, uint64_t key, const void* frame_dev, int frame_dtype, int frame_h, int frame_w,
                                    int64_t ld, const double* boxes_dev, int64_t n, int out_w, int out_h, void* out_dev,
                                    int out_dtype, int64_t ldo, void* stream);
int hg_patcher_extract(hg_patcher* p, const void* frame, int frame_dtype, int frame_h, int frame_w, int64_t ld,
                       const double* boxes, int64_t n, int out_w, int out_h, void* out, int out_dtype,
                       int64_t ldo);
int hg_patcher_extract_rotate_device(hg_patcher* p, const void* frame_dev, int frame_dtype, int frame_h, int frame_w,
                                     int64_t ld, const double* boxes_dev, const double* delta_angs_dev, int64_t n,
                                     int out_w, int out_h, void* out_dev, int out_dtype, int64_t ldo, void* stream);
int hg_patcher_extract_rotate(hg_patcher* p, const void* frame, int frame_dtype, int frame_h, int frame_w, int64_t ld,
                              const double* boxes, const double* delta_angs, int64_t n, int out_w, int out_h,
                              void* out, int out_dtype, int64_t ldo);

/* --- Cascade glue on the device (the reference's stage loop between two hot calls) -----------------
 * update_current_subimage_coordinates (face_analysis.py:803-840) + identify_patches_to_discard (:842-887) for n
 * candidates from their regression outputs, then the boolean-mask compaction of FaceDetectUpdated.py:739-759 as
 * "index map + row gather", so that extract -> execute -> regression -> update -> discard -> compaction chain on one
 * stream.  float64, the reference's operation order.  Candidates of all pyramid levels may share a batch: the per-level
 * constants travel per ORIGINAL window in orig_level (n0, 3) = (max_Dx_diff, max_Dy_diff, base_side),
 * FaceDetectUpdated.py:595,604-605; orig_index maps a candidate to its original window (:621). */
enum hg_stage_type { HG_STAGE_DISC = 0, HG_STAGE_POSX = 1, HG_STAGE_POSY = 2, HG_STAGE_PANG = 3, HG_STAGE_SCALE = 4 };
typedef struct hg_cascade_consts {
    double regression_width, regression_height;   /* Pipeline header (face_analysis.py:395-400)            */
    double desired_sampling;                      /* 0.825 (FaceDetectUpdated.py:729)                        */
    double tolerance_posxy_deviation, tolerance_scale_deviation, tolerance_angle_deviation; /* :113-115      */
    double max_scale_radio, min_scale_radio;      /* net_maxs / 0.825, net_mins / 0.825 (:596-597)           */
    double net_Dang;
    double cut_off_face;                          /* cut_offs_face[network_serial] (:98, :672), Disc stages  */
} hg_cascade_consts;
/* coords (n,4) and angles (n) are updated in place; discard[i] = 1 where the reference sets new_wrong_images. */
int hg_cascade_update_device(int device, int stage_type, const hg_cascade_consts* c, int64_t n, double* coords_dev,
                             double* angles_dev, const double* reg_dev, const int32_t* orig_index_dev,
                             const double* orig_coords_dev, const double* orig_angles_dev, const double* orig_level_dev,
                             uint8_t* discard_dev, void* stream);
/* map_dev[j] = index of the j-th candidate with discard == 0 (order kept), *count_dev = how many. */
int hg_cascade_compact_device(int device, const uint8_t* discard_dev, int64_t n, int32_t* map_dev, int32_t* count_dev,
                              void* stream);
/* dst[j, :] = src[map[j], :] for j < *count_dev (count read on the device); rows of row_bytes bytes (multiple of 4);
 * n_max bounds the launch (the candidate count before compaction).  Out of place. */
int hg_gather_rows_device(int device, const void* src_dev, void* dst_dev, int64_t row_bytes, const int32_t* map_dev,
                          const int32_t* count_dev, int64_t n_max, void* stream);

/* The whole stage loop (FaceDetectUpdated.py:665-766) of one batch of first-stage windows as ONE host call: for every stage
 * extract (rotated by -angle, unless the previous stage was a Disc stage or the stage has no network, :674-681) -> the stage's
 * flow (hg_flow_execute_device; NULL = reuse the previous features, the pipeline's "None0") -> regression -> coordinate update,
 * discard test, compaction (one fused kernel).  A stage that owns a network and the stages behind it whose network is NULL (at most
 * four) read the same sl: they run as ONE regression launch (hg_gauss_regression_multi_device) and ONE glue launch that applies
 * their updates and discard tests per row in stage order — same survivors, same order, same bits (HIGSFA_CASCADE_NO_GROUPS=1:
 * one stage per launch).  No per-candidate array visits the host; the host reads the survivor count only
 * after Disc stages (where it shrinks a lot and sizes the next launches) — between them launches are sized by the last count
 * read and the kernels take the exact count from device memory.  All pyramid levels may be one batch (:599).
 * flow / classifier handles stay owned by the caller and must live on `device`. */
typedef struct hg_gauss hg_gauss;
typedef struct hg_cascade hg_cascade;
typedef struct hg_cascade_stage {
    int32_t type;            /* enum hg_stage_type                                                   */
    int32_t serial;          /* trailing digit of the stage name: index into cut_offs_face (:669-672) */
    hg_flow* flow;           /* NULL: networks[k] is None                                            */
    hg_gauss* classifier;
} hg_cascade_stage;
int hg_cascade_create(const hg_cascade_stage* stages, int n_stages, int sub_w, int sub_h, int n_features,
                      const hg_cascade_consts* consts, const double* cut_offs_face, int n_cut_offs, int device,
                      hg_cascade** out);
void hg_cascade_free(hg_cascade* c);
/* frame_dev: (frame_h, frame_w) uint8 on the device, row stride ld; boxes_host (n0, 4) / level_host (n0, 3): the first-stage
 * windows and their level constants.  Outputs (host, room for out_cap detections): final coordinates, angles, index of the
 * original window, Disc confidence; *n_out detections; stage_counts[n_stages] survivors after each stage (-1 where the count
 * was not read back); *rows_executed rows pushed through flows.  Synchronous. */
int hg_cascade_detect_device(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld,
                             const double* boxes_host, const double* level_host, int64_t n0, double* out_coords,
                             double* out_angles, int32_t* out_orig_index, double* out_confidence, int64_t out_cap,
                             int64_t* n_out, int32_t* stage_counts, int64_t* rows_executed, void* stream);

/* The same with the first-stage windows computed ON THE DEVICE from the grid's closed form (face_analysis.py:630-646, :661-669):
 * pyramid level L holds ny x nx windows, y-major, at numpy.linspace(0, stop, n) positions; a window is
 * (posX, posY, posX + patch_w - 1, posY + patch_h - 1); every window of the level carries (max_dx, max_dy, base_side)
 * (face_analysis.py:651-652, FaceDetectUpdated.py:604-605).  float64 in numpy's operation order: the boxes equal the host
 * formulas' bit for bit (hg_cascade_grid_device writes them to device arrays for inspection; boxes_dev == NULL: only *n0). */
typedef struct hg_cascade_level {
    int32_t nx, ny;                      /* grid points along x / y (face_analysis.py:640-641)            */
    double x_stop, y_stop;               /* im_width - patch_w, im_height - patch_h (end points, :645-646) */
    double patch_w, patch_h;             /* subimage size x sampling value (:630-631)                      */
    double max_dx, max_dy, base_side;    /* net_Dx * patch_w / regression_width, ..., sqrt(pw^2 + ph^2)    */
} hg_cascade_level;
int hg_cascade_detect_levels_device(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld,
                                    const hg_cascade_level* levels, int n_levels, double* out_coords, double* out_angles,
                                    int32_t* out_orig_index, double* out_confidence, int64_t out_cap, int64_t* n_out,
                                    int32_t* stage_counts, int64_t* rows_executed, void* stream);
/* One frame, one host call: the reference's prescale (im.resize((w, h), NEAREST), FaceDetectUpdated.py:551-561; prescale_w = 0: none)
 * into a buffer owned by the cascade, the grid of the PRESCALED frame from `levels`, the stage loop. */
int hg_cascade_detect_frame_device(hg_cascade* c, const void* frame_dev, int frame_h, int frame_w, int64_t ld, int prescale_w,
                                   int prescale_h, const hg_cascade_level* levels, int n_levels, double* out_coords,
                                   double* out_angles, int32_t* out_orig_index, double* out_confidence, int64_t out_cap,
                                   int64_t* n_out, int32_t* stage_counts, int64_t* rows_executed, void* stream);
int hg_cascade_grid_device(int device, const hg_cascade_level* levels, int n_levels, double* boxes_dev, double* level_dev,
                           int64_t cap, int64_t* n0, void* stream);

/* --- SFA training step for one layer of nodes (BASELINE.json configs[4]) -----
 * Not on the reference's path (it never trains, face_analysis.py:451-479); restates
 * mdp.nodes.SFANode train/stop_training per node k over input columns conn[k*d .. (k+1)*d):
 * mean, B = Cov(x), A = Cov(x[t+1]-x[t]) accumulated in fp64 by a HIP kernel, then
 * A w = lambda B w (eigenvalues ascending, w' B w = 1): hand-written Jacobi kernel for d <= 16, rocSOLVER dsygvj
 * for wider nodes (HIGSFA_SYGVJ=1 / HIGSFA_SYGVD=1 force rocSOLVER's dsygvj / dsygvd).
 * x: (n, ldx) matrix in time order, a device pointer or (x_on_host != 0) a host pointer that is
 * copied to the device first.  Host outputs: evals (n_nodes, d), evecs
 * (n_nodes, d, d) column-major per node (column i = eigenvector i), mean (n_nodes, d);
 * timings_ms[2] (optional): statistics kernels, solve. */
int hg_sfa_train_layer(const void* x, int x_on_host, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host,
                       int32_t n_nodes, int32_t d, int device, double* evals_host, double* evecs_host,
                       double* mean_host, double* timings_ms);
/* The PCA / whitening step of a node (mdp.nodes.PCANode / WhiteningNode train + stop_training): same statistics kernel, then the
 * eigen-decomposition of Cov(x) itself (the same solver on (Cov, I)): evals ascending, orthonormal eigenvectors. */
int hg_pca_train_layer(const void* x, int x_on_host, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host,
                       int32_t n_nodes, int32_t d, int device, double* evals_host, double* evecs_host,
                       double* mean_host, double* timings_ms);
/* What a trained step passes on during training, in float64 like MDP: out[t, node * width + f * p + j] =
 * func_f(((x[t, conn[node]] - mean[node]) W[node])_j), width = max(1, n_funcs) * p; n_funcs = 0: the affine map alone.
 * func_kinds: 0 identity, 1 |z|^e, 2 sgn(z)|z|^e.  x_dev / out_dev: device; conn / mean / W (n_nodes, d, p row-major): host. */
int hg_train_apply_device(const void* x_dev, int x_dtype, int64_t n, int64_t ldx, const int32_t* conn_host, int32_t n_nodes,
                          int32_t d, const double* mean_host, const double* w_host, int32_t p, int32_t n_funcs,
                          const int32_t* func_kinds, const double* func_expos, double* out_dev, int64_t ldo, int device);

#ifdef __cplusplus
}
#endif
#endif /* HIGSFA_H */
