/*
 * cofusion_hip.h -- C-ABI of the MI355X-native Co-Fusion hot path (libcofusion_hip.so).
 *
 * Drop-in boundary.  The reference's lower seam is the set of free
 * functions in Core/Cuda/cudafuncs.cuh:64-193 (called from Core/Utils/RGBDOdometry.cpp and
 * Core/Model/Model.cpp:341-343) plus the GLSL passes driven by Core/Model/Model.cpp and
 * Core/Model/ModelProjection.cpp.  Every entry point below names the reference interface it
 * replaces.  Signatures are plain C: device pointers, sizes, host PODs; no torch, no Eigen,
 * no OpenGL.  All functions return 0 on success or a negative CF_E* code; cf_last_error()
 * gives the message (the reference printf+exit(-1)s instead, Core/Cuda/convenience.cuh:74-83).
 *
 * Layouts
 *   depth           f32  [H*W] metres, 0 = invalid            (FrameData.depth, CV_32FC1)
 *   rgba            u8x4 [H*W] R,G,B,A                        (GL_RGBA texture of CoFusion.cpp:179)
 *   vertex4/normal4 f32x4 [H*W]                               (RGBA32F splat outputs)
 *   planar map      f32  [3*H*W]: x rows, y rows, z rows      (DeviceArray2D<float>(3*rows, cols))
 *   pose            f32[16] ROW-major T(model <- camera)      (Eigen::Matrix4f in the reference)
 *   SE3 sums        int64[32]: 27 upper-triangular products row_i*row_j (i<6, i<=j<7) in
 *                   Q31.32 fixed point, [27] = sum r^2 (Q31.32), [28] = inlier count.  The RGB step's sums use
 *                   CF_FIX_RGB_BITS(sigma) fraction bits (see below)
 *   surfel          12 f32 (48 B): [x y z conf][colour24 0 initTime lastTime][nx ny nz radius]
 *                   (Core/Shaders/Vertex.cpp:21-43)
 */
#ifndef COFUSION_HIP_H_
#define COFUSION_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_OK 0
#define CF_EINVAL (-1)
#define CF_EHIP (-2)
#define CF_ENOMEM (-3)
#define CF_ESTATE (-4)

#define CF_NUM_PYRS 3        /* RGBDOdometry::NUM_PYRS, RGBDOdometry.h:69 */
#define CF_SE3_WORDS 32
#define CF_SO3_WORDS 16
#define CF_FIX_ICP 32
/* Fraction bits of the RGB step's sums are NOT a constant: they follow the sigma handed to rgbStep --
 * 8 + 2*floor(log2 sigma) capped at 32, and 8 for sigma in {-1, < 2}.  Readers of cf_rgb_step's sums_host must use
 * CF_FIX_RGB_BITS(sigma); CF_FIX_RGB_MAX is only the cap (reached for sigma >= 4096). */
#define CF_FIX_RGB_MAX 32
static inline int CF_FIX_RGB_BITS(float sigma)
{
    union { float f; uint32_t u; } v;
    int F;
    if (sigma == -1.0f || !(sigma >= 2.0f)) return 8;
    v.f = sigma;
    F = 8 + 2 * ((int)((v.u >> 23) & 255u) - 127);
    return F > CF_FIX_RGB_MAX ? CF_FIX_RGB_MAX : F;
}
#define CF_FIX_SO3 12

typedef struct cf_ctx cf_ctx;
typedef struct cf_odom cf_odom;
typedef struct cf_model cf_model;

typedef struct { float fx, fy, cx, cy; } cf_cam; /* CameraModel, Core/Cuda/types.cuh:83-99 */

/* one surfel of a model map, byte-compatible with the reference's VBO vertex (Core/Shaders/Vertex.cpp:21-43):
 * what cf_model_download_map / cf_model_upload_map move */
typedef struct {
    float x, y, z, conf;
    float colour24, unused, init_time, last_time; /* colour: r<<16 | g<<8 | b stored as a float (color_encoding.glsl) */
    float nx, ny, nz, radius;
} cf_surfel;
#ifdef __cplusplus
static_assert(sizeof(cf_surfel) == 48, "surfel record is 3 x float4");
#else
_Static_assert(sizeof(cf_surfel) == 48, "surfel record is 3 x float4");
#endif

/* DataTerm, Core/Cuda/types.cuh:75-81 */
typedef struct {
    int16_t zero_x, zero_y;
    int16_t one_x, one_y;
    float diff;
    int32_t valid;
} cf_dataterm;

typedef struct {
    int width, height;
    float fx, fy, cx, cy;     /* Intrinsics singleton, Core/Utils/Intrinsics.h */
    int device;               /* HIP device ordinal */
    int max_models;           /* concurrently tracked models (background + objects) */
    int max_surfels;          /* per model; reference default 3072*3072 (Model.cpp:92-98) */
} cf_config;

/* ------------------------------------------------------------------ context ---- */
int cf_create(const cf_config *cfg, cf_ctx **out);
void cf_destroy(cf_ctx *ctx);
const char *cf_last_error(const cf_ctx *ctx);
/* all work is enqueued on one stream: by default a ctx-owned non-blocking stream; cf_set_stream
 * switches to a caller-owned hipStream_t (NULL = the legacy default stream), cf_use_own_stream back */
int cf_set_stream(cf_ctx *ctx, void *hip_stream);
int cf_use_own_stream(cf_ctx *ctx);
void *cf_get_stream(cf_ctx *ctx);
int cf_synchronize(cf_ctx *ctx);
/* Overlap independent work of one frame (e.g. the surfel passes of different models): cf_fork(lane) routes the following
 * calls to auxiliary stream `lane` (0..7), ordered after everything enqueued so far on the context's stream; cf_main
 * returns to that stream without waiting (the lanes keep running beside what follows); cf_join returns to it and orders
 * it after all lanes used since the last join.  Calls routed to a lane must not wait on the host. */
int cf_fork(cf_ctx *ctx, int lane);
int cf_main(cf_ctx *ctx);
int cf_join(cf_ctx *ctx);
/* the same for ONE lane: the stream waits for that lane only (the other lanes keep running beside what follows) */
int cf_join_lane(cf_ctx *ctx, int lane);
/* Enqueue from several host threads: the owning thread forks the lanes (cf_fork(lane) for each, then cf_main), helper threads
 * call cf_thread_lane(ctx, lane) and then issue the cf_model_* calls of ONE model each -- those go to the lane, the context's
 * current stream is untouched -- and unbind with lane < 0; the owning thread waits for the helpers and calls cf_join.  Only the
 * cf_model_* entry points honour the binding. */
int cf_thread_lane(cf_ctx *ctx, int lane);
/* cf_mark(slot 0..3) remembers the current point of the stream; cf_fork_after(lane, slot) routes the following calls to `lane`
 * ordered after that point only (slot < 0: after nothing): for work that does not depend on what the stream still has queued,
 * e.g. filtering the next frame while the previous frame's fusion passes run.  cf_join orders the stream after the lane. */
int cf_mark(cf_ctx *ctx, int slot);
int cf_event_wait_host(cf_ctx *ctx, int slot); /* block the host until the stream has passed cf_mark(slot) */
int cf_fork_after(cf_ctx *ctx, int lane, int slot);
/* device memory helpers for hosts that do not link HIP themselves */
int cf_malloc(cf_ctx *ctx, uint64_t bytes, void **dptr);
int cf_free(cf_ctx *ctx, void *dptr);
int cf_memcpy_h2d(cf_ctx *ctx, void *dst, const void *src, uint64_t bytes);
int cf_memcpy_d2h(cf_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* pinned host memory + copies that are only ENQUEUED on the context's stream (the host side of the buffer must stay
 * untouched until the stream has passed the copy: cf_mark / cf_synchronize) */
int cf_malloc_host(cf_ctx *ctx, uint64_t bytes, void **hptr);
int cf_free_host(cf_ctx *ctx, void *hptr);
int cf_memcpy_h2d_async(cf_ctx *ctx, void *dst, const void *src_pinned, uint64_t bytes);
int cf_memcpy_d2h_async(cf_ctx *ctx, void *dst_pinned, const void *src, uint64_t bytes);
int cf_memcpy_d2d_async(cf_ctx *ctx, void *dst, const void *src, uint64_t bytes);
/* FrameData.rgb (3 B/pixel, device copy) -> RGBA8 with alpha 255: the GL_RGBA upload of CoFusion.cpp:179 */
int cf_rgb_to_rgba(cf_ctx *ctx, const uint8_t *rgb_dev, int cols, int rows, uint8_t *rgba_dev);

/* --------------------------------------------- map preparation (cudafuncs.cuh) ---- */
/* createVMap  cudafuncs.cuh:125-130 */
int cf_create_vmap(cf_ctx *ctx, const float *depth, int cols, int rows, cf_cam intr, float depth_cutoff, float *vmap);
/* createNMap  cudafuncs.cuh:132-133 */
int cf_create_nmap(cf_ctx *ctx, const float *vmap, int cols, int rows, float *nmap);
/* copyMaps    cudafuncs.cuh:142-145 */
int cf_copy_maps(cf_ctx *ctx, const float *vertex4, const float *normal4, int cols, int rows, float *vmap, float *nmap);
/* resizeVMap / resizeNMap  cudafuncs.cuh:147-151 */
int cf_resize_map(cf_ctx *ctx, const float *in, int in_cols, int in_rows, float *out, int normalize);
/* tranformMaps cudafuncs.cuh:135-140 (in place) */
int cf_transform_maps(cf_ctx *ctx, float *vmap, float *nmap, int cols, int rows, const float R[9], const float t[3]);
/* verticesToDepth cudafuncs.cuh:156-158 */
int cf_vertices_to_depth(cf_ctx *ctx, const float *vertex4, int cols, int rows, float cutoff, float *depth);
/* pyrDownGaussF cudafuncs.cuh:174-175, pyrDownUcharGauss :177-178 */
int cf_pyrdown_gauss_f32(cf_ctx *ctx, const float *src, int src_cols, int src_rows, float *dst);
int cf_pyrdown_gauss_u8(cf_ctx *ctx, const uint8_t *src, int src_cols, int src_rows, uint8_t *dst);
/* imageBGRToIntensity cudafuncs.cuh:153-154 */
int cf_rgba_to_intensity(cf_ctx *ctx, const uint8_t *rgba, int cols, int rows, uint8_t *dst);
/* computeDerivativeImages cudafuncs.cuh:184-186 */
int cf_sobel(cf_ctx *ctx, const uint8_t *src, int cols, int rows, int16_t *dx, int16_t *dy);
/* projectToPointCloud cudafuncs.cuh:161-164 (intr already divided by 2^level) */
int cf_project_cloud(cf_ctx *ctx, const float *depth, int cols, int rows, cf_cam intr_level, float *cloud3);

/* ------------------------------------------------- reductions (cudafuncs.cuh) ---- */
/* icpStep cudafuncs.cuh:64-82.  Device maps in, host A[36] (row-major, symmetric filled),
 * b[6], residual[2] = {sum r^2, inliers} out (synchronous, like the reference).  sums_host
 * (nullable) additionally receives the exact int64[32] sums.  err_surface: device f32 [rows*cols]
 * or NULL (the cudaSurfaceObject_t of the reference). */
int cf_icp_step(cf_ctx *ctx, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                const float *nmap_curr, const float Rprev_inv[9], const float tprev[3], cf_cam intr,
                const float *vmap_g_prev, const float *nmap_g_prev, float dist_thres, float angle_thres,
                int cols, int rows, float *A_host, float *b_host, float *residual_host, int64_t *sums_host,
                float *err_surface);
/* The same reduction over the row band [row_begin, row_end) only: a rank's share when ONE model's reduction is split
 * over several GPUs.  The int64 sums of the bands add up to the full-image sums exactly (integer arithmetic), so an
 * all-reduce(SUM) of sums_host over the ranks reproduces cf_icp_step bit for bit.  A/b/residual are the band's own. */
int cf_icp_step_band(cf_ctx *ctx, const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                     const float *nmap_curr, const float Rprev_inv[9], const float tprev[3], cf_cam intr,
                     const float *vmap_g_prev, const float *nmap_g_prev, float dist_thres, float angle_thres,
                     int cols, int rows, int row_begin, int row_end, float *A_host, float *b_host,
                     float *residual_host, int64_t *sums_host, float *err_surface);
/* computeRgbResidual cudafuncs.cuh:102-121 */
int cf_rgb_residual(cf_ctx *ctx, float min_scale, const int16_t *dIdx, const int16_t *dIdy, const float *last_depth,
                    const float *next_depth, const uint8_t *last_image, const uint8_t *next_image,
                    cf_dataterm *corres, float max_depth_delta, const float kt[3], const float krkinv[9],
                    int cols, int rows, int *sigma_sum_host, int *count_host);
/* rgbStep cudafuncs.cuh:84-97.  sigma: the correspondence count (RGBDOdometry.cpp:373-374), 1 or -1 (rgbOnly);
 * sums_host are fixed point with 8 + 2*floor(log2 sigma) fraction bits (max 32; 8 for sigma -1 or < 2) */
int cf_rgb_step(cf_ctx *ctx, const cf_dataterm *corres, float sigma, const float *cloud3, float fx, float fy,
                const int16_t *dIdx, const int16_t *dIdy, float sobel_scale, int cols, int rows, float *A_host,
                float *b_host, int64_t *sums_host);
/* so3Step cudafuncs.cuh:99-110 */
int cf_so3_step(cf_ctx *ctx, const uint8_t *last_image, const uint8_t *next_image, const float image_basis[9],
                const float kinv[9], const float krlr[9], int cols, int rows, float *A_host, float *b_host,
                float *residual_host, int64_t *sums_host);

/* ------------------------------- RGBDOdometry (Core/Utils/RGBDOdometry.h:42-60) ---- */
typedef struct {
    int rgb_only, pyramid, fast_odom, so3;
    float icp_weight;
} cf_track_opts;
typedef struct {
    float last_icp_error, last_icp_count, last_rgb_error, last_rgb_count, last_so3_error, last_so3_count;
    double lastA[36], lastb[6];
    int so3_iterations;
    int fault;              /* non-zero: a bounded device-side wait expired (cf_odom_fetch_result returns CF_ESTATE) */
    int cull_box[4];        /* cf_odom_set_culling: level-0 pixel rectangle [x0, y0, x1, y1] (inclusive, may exceed the image) the last ICP
                             * iteration was restricted to -- pixels outside cannot find a correspondence; x0 > x1: empty; the whole
                             * image when culling is off */
} cf_track_stats;

int cf_odom_create(cf_ctx *ctx, cf_odom **out);
void cf_odom_destroy(cf_odom *od);
/* initICPModel(predictedVertices, predictedNormals, depthCutoff, modelPose) RGBDOdometry.h:53 */
int cf_odom_init_icp_model(cf_odom *od, const float *pred_vertex4, const float *pred_normal4, const float pose[16]);
/* initRGBModel(rgb) :57 / initRGB(rgb) :55 / initFirstRGB(rgb) :59 -- rgba is a device RGBA8 image */
int cf_odom_init_rgb_model(cf_odom *od, const uint8_t *pred_rgba);
int cf_odom_init_rgb(cf_odom *od, const uint8_t *rgba);
int cf_odom_init_first_rgb(cf_odom *od, const uint8_t *rgba);
/* initICPModel + initRGBModel(pred_rgba[k]) + initRGB(frame_rgba) of `n` trackers at once: one launch per preparation
 * kernel with one grid row per tracker instead of seven launches per tracker (what a frame with several active models
 * needs before cf_odom_track_batch_async); identical results to the three single calls per tracker */
int cf_odom_init_models_batch(cf_ctx *ctx, cf_odom *const *ods, int n, const float *const *pred_vertex4,
                              const float *const *pred_normal4, const uint8_t *const *pred_rgba, const float *const *poses /* n x [16] */,
                              const uint8_t *frame_rgba);
/* ... with one frame image per tracker (frame_rgba[k] for ods[k]): the trackers of several sequences prepared by the same launches */
int cf_odom_init_models_batch_frames(cf_ctx *ctx, cf_odom *const *ods, int n, const float *const *pred_vertex4,
                                     const float *const *pred_normal4, const uint8_t *const *pred_rgba, const float *const *poses /* n x [16] */,
                                     const uint8_t *const *frame_rgba /* n */);
/* ... and with the choice between two sets of predictions per tracker made ON THE DEVICE: Model::initICP (Model.cpp:350-367) tracks against
 * the fill-in images when CoFusion::requiresFillIn (CoFusion.cpp:547-565) says so, and that answer comes from a count over the
 * previous frame's last prediction -- a host that asks for it (cf_model_requires_fill_in) has to wait for the previous frame to
 * drain before it can enqueue this one.  Here tracker k uses alt_*[k] instead of pred_*[k] when
 * (float)counts[0] / (float)counts[1] < ratio with counts = fill_counts[k] (device, from cf_model_fill_ratio_device; the very
 * expression of cf_model_requires_fill_in); fill_counts[k] == NULL (or the arrays NULL): no choice, pred_*[k]. */
int cf_odom_init_models_batch_select(cf_ctx *ctx, cf_odom *const *ods, int n, const float *const *pred_vertex4,
                                     const float *const *pred_normal4, const uint8_t *const *pred_rgba,
                                     const float *const *alt_vertex4, const float *const *alt_normal4, const uint8_t *const *alt_rgba,
                                     const uint32_t *const *fill_counts, float ratio, const float *const *poses /* n x [16] */,
                                     const uint8_t *const *frame_rgba /* n */);
/* initICP(depthPyramid, maskPyramid, depthCutoff) :48-49 (frame -> model); the mask pyramid is dead in the
 * reference (cudafuncs.cu:119) and therefore not part of the ABI */
int cf_odom_init_icp(cf_odom *od, const float *const depth_pyr[CF_NUM_PYRS], float depth_cutoff);
/* getIncrementalTransformation :62-64.  trans/rot host in-out.  The whole Gauss-Newton loop (SO3
 * pre-alignment, 4/5/10 pyramid schedule, f64 6x6 solve, SE3 update) runs device-resident; the host
 * waits once at the end.  icp_err_surface: device f32 [H*W] or NULL. */
int cf_odom_get_incremental_transformation(cf_odom *od, float trans[3], float rot[9], const cf_track_opts *opts,
                                           float *icp_err_surface, cf_track_stats *stats);
/* batched, asynchronous flavour used by the orchestrator: all `n` models advance through the
 * GN schedule in lock-step inside the same launches (grid.y = model).  Poses are read from /
 * written to the odom objects' device state; call cf_odom_fetch_result after cf_synchronize. */
int cf_odom_track_batch_async(cf_ctx *ctx, cf_odom *const *ods, int n, const float *const *poses_in /* n x [16] */,
                              const cf_track_opts *opts, float *const *icp_err_surfaces /* nullable entries */);
int cf_odom_fetch_result(cf_odom *od, float trans[3], float rot[9], cf_track_stats *stats);
/* RGBDOdometry::getCovariance (RGBDOdometry.h:60, RGBDOdometry.cpp:479: lastA.cast<double>().lu().inverse()) of a tracking call's
 * statistics: partial-pivot LU inverse of the 6x6 normal matrix, row-major f64 [36], on the host (the caller already holds lastA).
 * A singular lastA gives inf / NaN entries, as Eigen's does. */
int cf_odom_get_covariance(const cf_track_stats *stats, double cov[36]);
/* pixels the level-0 {ICP reduction || RGB residual} launch of the last fetched tracking call visited for this tracker: the 64-pixel
 * runs inside its final screen box and the record slots between its first and last RGB candidate (the whole image for a tracker
 * that is not culled) -- the physical byte count of a roofline figure, as opposed to SURVEY 8(d)'s every-pixel-to-every-tracker */
int cf_odom_level0_visited(cf_odom *od, uint64_t *icp_pixels, uint64_t *residual_pixels);
/* test access to internal device pyramids (same `which` numbering as the oracle's orc_odom_buffer) */
/* share the frame-wide current vertex/normal pyramids between models (all models track the same frame,
 * cudafuncs.cu:119); pass NULL arrays to return to the odom-private maps written by cf_odom_init_icp */
/* Skip the model-map gathers of pixels that project into empty 4x4 blocks of the prediction (an occupancy bitmap written by
 * initICPModel) and, since round 5, every pixel outside the screen box of the prediction's frustum piece.  Results are unchanged; it
 * pays for models that cover a small part of the image (object models) and costs a dependent look-up for one that covers all of it
 * (background).  Default off.  Precondition of the screen box: the tracking call starts from the pose the model maps were prepared with
 * (cf_odom_init_icp_model's `pose` == the pose handed to cf_odom_track_batch_async / _get_incremental_transformation, bit for bit -- what
 * Model::performTracking does); a call that starts elsewhere is tracked WITHOUT the screen box (checked per call; the occupancy look-up
 * does not depend on it). */
int cf_odom_set_culling(cf_odom *od, int on);
/* Multi-GPU hooks.  cf_set_collective registers the caller's in-place all-reduce over its ranks (RCCL): op 0 = SUM of int64 words,
 * op 1 = MIN of unsigned 64-bit words; dev_buf is a device address, the call must only ENQUEUE on `hip_stream` (or order itself
 * after it and before later work on it); 0 = success.  cf_odom_set_band splits one model's reductions over the ranks by image
 * rows (level-0 rows, multiples of 4; add_counts = 1 on exactly one rank of the split): inside the device-resident loop the
 * collective sums that model's normal equations after every {ICP || residual} launch -- 32 words (256 bytes): the library folds the
 * rank's grouped partial sums first (round 6; 16 KB until then) --, so all ranks solve the same system. */
typedef int (*cf_collective_fn)(void *user, int op, void *dev_buf, uint64_t words, void *hip_stream);
int cf_set_collective(cf_ctx *ctx, cf_collective_fn fn, void *user);
int cf_odom_set_band(cf_odom *od, int row_begin, int row_end, int add_counts);
/* RCCL inside the library (one process per GPU; north_star: "RCCL all-reduce of the 6x6 system over xGMI").  cf_rccl_unique_id
 * wraps ncclGetUniqueId (rank 0 creates the 128-byte id, the caller hands it to the other ranks over any side channel);
 * cf_rccl_init creates the context's own ncclComm_t on the context's device (collective call: every rank of `world`) and registers
 * the library's own collective in place of cf_set_collective: the split reductions then run ncclAllReduce in place on the stream
 * their kernels are enqueued on -- no staging copy, no callback.  cf_rccl_allreduce: op 0 = SUM of int64 words, op 1 = MIN of
 * unsigned 64-bit words, in place, enqueued on hip_stream (NULL: the context's stream); cf_rccl_broadcast: `bytes` bytes from
 * rank `root` (a frame from the ingest GPU).  cf_destroy releases the communicator. */
#define CF_RCCL_ID_BYTES 128
int cf_rccl_unique_id(void *id128);
int cf_rccl_init(cf_ctx *ctx, const void *id128, int rank, int world);
int cf_rccl_allreduce(cf_ctx *ctx, void *dev_buf, uint64_t words, int op, void *hip_stream);
int cf_rccl_broadcast(cf_ctx *ctx, void *dev_buf, uint64_t bytes, int root, void *hip_stream);
int cf_rccl_info(const cf_ctx *ctx, int *rank, int *world, int *rccl_version);
int cf_rccl_destroy(cf_ctx *ctx);
int cf_odom_bind_frame_maps(cf_odom *od, const float *const vmaps[CF_NUM_PYRS], const float *const nmaps[CF_NUM_PYRS]);
/* ... or, when another tracker of the same context computed them with cf_odom_init_icp: share that tracker's current-frame pyramids
 * (and the per-run depth intervals the culled reduction uses) */
int cf_odom_share_frame_maps(cf_odom *od, cf_odom *owner);
int cf_odom_buffer(cf_odom *od, int which, int level, void **dptr, uint64_t *bytes);
/* Model::generateCUDATextures depth half (Model.cpp:341-343): l1/l2 device outputs */
int cf_depth_pyramid(cf_ctx *ctx, const float *depth_filtered, int cols, int rows, float *l1, float *l2);

/* ------------------------------------------------ frame pre-processing + surfel Model ---- */
/* CoFusion::filterDepth + depth_bilateral_metric.frag (CoFusion.cpp:567-574): 13x13 bilateral, zero outside [0.3, maxD] */
int cf_bilateral(cf_ctx *ctx, const float *depth, int cols, int rows, float maxD, float *out);

/* Model (Core/Model/Model.h:117-235).  The pose lives with the caller (the CoFusion facade) and is passed
 * explicitly, ROW-major T(model <- camera).  Two surfel buffers are ping-ponged exactly like Model::vbos[2]. */
int cf_model_create(cf_ctx *ctx, int max_surfels, cf_model **out);
void cf_model_destroy(cf_model *m);
/* computeFeedbackBuffers + Model::initialise (CoFusion.cpp:157-169, Model.cpp:227-272); frame 1 only */
int cf_model_initialise(cf_model *m, const uint8_t *rgba, const float *depth_raw, const float *depth_filtered, int time,
                        float maxDepth);
/* Model::lastCount (Model.cpp:815) */
int cf_model_count(cf_model *m, uint32_t *count);
/* Model::predictIndices -> ModelProjection::predictIndices (ModelProjection.cpp:105-157) */
int cf_model_predict_indices(cf_model *m, const float pose[16], int time, float maxDepth, int timeDelta);
/* predictIndices in two halves for a surfel map sharded over GPUs: rasterise the surfels [surfel_begin, surfel_end) into
 * keys_dev (u64 [H*W], filled by the call; smaller key = nearer surfel, ties -> lower id, empty = all ones), MIN-all-reduce
 * the key maps over the ranks as unsigned 64-bit integers, then resolve the reduced map into the model's index textures. */
int cf_model_index_keys(cf_model *m, const float pose[16], int time, float maxDepth, int timeDelta, uint32_t surfel_begin,
                        uint32_t surfel_end, uint64_t *keys_dev);
int cf_model_index_resolve(cf_model *m, const float pose[16], uint64_t *keys_dev);
/* the three steps in one call for a REPLICATED map whose rasterisation is split over `nshards` ranks: this rank's range of the exact
 * surfel count, the registered collective (cf_set_collective, op 1) on the key map, resolve */
int cf_model_predict_indices_sharded(cf_model *m, const float pose[16], int time, float maxDepth, int timeDelta, int shard, int nshards);
/* Model::combinedPredict -> ModelProjection::combinedPredict (ModelProjection.cpp:192-273), ACTIVE prediction */
int cf_model_combined_predict(cf_model *m, const float pose[16], float maxDepth, float confThreshold, int time, int maxTime,
                              int timeDelta);
/* optional: count the covered pixels of the latest splat prediction now and read them back asynchronously, so that the
 * next cf_model_requires_fill_in does not wait (call it after the frame's last cf_model_combined_predict) */
int cf_model_prefetch_fill_ratio(cf_model *m);
/* Model::performFillIn (Model.cpp:901-909) */
int cf_model_perform_fill_in(cf_model *m, const uint8_t *rgba, const float *depth_filtered, int passthrough_geom,
                             int passthrough_rgb);
/* CoFusion::requiresFillIn (CoFusion.cpp:547-565); out = 1 when fewer than `ratio` of the sampled pixels are set */
int cf_model_requires_fill_in(cf_model *m, float ratio, int *out);
/* device address of the two counts (covered, total) the last cf_model_prefetch_fill_ratio left, for cf_odom_init_models_batch_select;
 * *counts_dev = NULL when no prefetch is pending (the caller then asks cf_model_requires_fill_in).  Does not wait. */
int cf_model_fill_ratio_device(cf_model *m, const uint32_t **counts_dev);
/* Model::fuse (Model.cpp:408-563); weighting = Model::computeFusionWeight; maxDepth = min(depthCutoff, model maxDepth) */
int cf_model_fuse(cf_model *m, const float pose[16], int time, const uint8_t *rgba, const uint8_t *mask, const float *depth_raw,
                  const float *depth_filtered, float maxDepth, float weighting, int maskID);
/* Model::clean (Model.cpp:565-697); count_out = surfels written (the GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN query) */
int cf_model_clean(cf_model *m, const float pose[16], int time, float confThreshold, float outlierCoeff, int timeDelta,
                   const float *depth_filtered, const uint8_t *mask, int maskID, uint32_t *count_out);
/* The second half of a frame for SEVERAL models in lock-step (CoFusion.cpp:316-330 and 346 / 533-545): what the reference runs as six
 * loops over the models -- predictIndices, fuse, predictIndices, clean for every model that fuses, combinedPredict(time, time) for all --
 * as ONE chain of launches whose workgroups are dealt to the models (the way the tracking launches carry all trackers): an object
 * model's share of a pass is a few dozen workgroups, and a chain of ~16 launch-floor kernels per model, one model after another,
 * was a third of a multi-object frame.  The items may belong to different sequences (own frame images, own clock).  Identical
 * results to the per-model calls in the same order; the surfel counts travel back asynchronously as with cf_model_clean. */
typedef struct {
    cf_model *model;
    const float *pose;                       /* [16] row-major T(model <- camera) */
    const uint8_t *rgba, *mask;              /* the model's frame: RGBA8, label mask (device) */
    const float *depth_raw, *depth_filtered;
    int do_fuse;                             /* 0: prediction only (CoFusion.cpp:463: !rgbOnly && trackingOk && !lost) */
    int time;                                /* the sequence's tick */
    float fuse_max_depth;                    /* min(depthCutoff, model maxDepth) (Model.cpp:443) */
    float weighting;                         /* Model::computeFusionWeight */
    int mask_id;
    float conf_threshold;
} cf_model_pass;
int cf_models_frame_passes(cf_ctx *ctx, const cf_model_pass *items, int n, float depth_cutoff, float outlier_coeff, int time_delta);
/* The first index pass of that chain (Model::predictIndices before Model::fuse, CoFusion.cpp:316-318) for models that were tracked this
 * frame, enqueued BEFORE the host waits for the tracking results: the pose comes from the tracker's device state (inverted by a small
 * kernel), so the GPU rasterises the index maps while the host wakes up, reads poses and segmentation decisions and prepares the rest
 * of the chain (a 45-60 us hole in the queue per frame until round 5).  cf_models_frame_passes skips its first index pass for such a
 * model when the pose it is given equals the tracker's result bit for bit, and rasterises again otherwise (overridden pose).  A model
 * that is not fused afterwards (tracking lost, deactivated) just had its index map overwritten early: nothing reads it in between. */
typedef struct {
    cf_model *model;
    const cf_odom *odom;   /* the tracker whose result is the model's pose for this frame */
    int time;              /* the sequence's tick, as cf_model_pass::time */
} cf_model_preindex;
int cf_models_preindex(cf_ctx *ctx, const cf_model_preindex *items, int n, float depth_cutoff, int time_delta);
/* Model::downloadMap (Model.cpp:867-899): `count` surfels of 12 floats */
int cf_model_download_map(cf_model *m, float *host_surfels, uint32_t capacity, uint32_t *count);
int cf_model_upload_map(cf_model *m, const float *host_surfels, uint32_t count);
/* device views of the projection outputs.  which: 0 index(u32) 1 vertConf 2 colorTime 3 normRad (f32x4) | 4 splat image
 * (rgba8) 5 splat vertexConf 6 splat normalRad (f32x4) 7 splat time (u16) | 8 fill vertex 9 fill normal (f32x4)
 * 10 fill image (rgba8) | 11 surfels */
int cf_model_buffer(cf_model *m, int which, void **dptr, uint64_t *bytes);
/* Model::computeFusionWeight (Model.cpp:391-406); pure host math */
float cf_fusion_weight(const float pose[16], const float lastPose[16], float weightMultiplier);

/* ------------------------------------------------------------- motion segmentation ---- */
/* GPU side of Core/Segmentation (SLIC -> per-superpixel sums -> dense-CRF mean field -> up-sampling).
 * gSLICr / densecrf are third party and absent from the reference tree (Scripts/install.sh:84-85); they are
 * replaced by the published algorithms stated in oracle/orc_segment.c.  Superpixels are 16x16 on a regular
 * grid (Slic.cpp:33-43): K = (width/16)*(height/16) nodes. */
typedef struct cf_segmenter cf_segmenter;
int cf_seg_create(cf_ctx *ctx, cf_segmenter **out);
void cf_seg_destroy(cf_segmenter *s);
/* Slic::setInputImage + processFrame (Slic.cpp:48-81) */
int cf_seg_slic(cf_segmenter *s, const uint8_t *rgba);
/* Slic::downsample / downsampleThresholded sums (Slic.h:48-120) as exact Q32 fixed point: per superpixel
 * pixel count, count and sum of depth > 0.02, and per model the sums of the ICP error surface and of the
 * splat confidence (vertexConf.w).  Host outputs: [K], [K], [K], [n_models*K], [n_models*K], [K]. */
int cf_seg_accumulate(cf_segmenter *s, const float *depth, int n_models, const float *const *icp_err,
                      const float *const *vertconf4, uint32_t *spix_count_host, uint32_t *depth_count_host,
                      int64_t *depth_sum_host, int64_t *icp_sum_host, int64_t *conf_sum_host, int32_t *resample_labels_host);
/* DenseCRF2D inference of Segmentation.cpp:436-480: unary [K*L], Gaussian features [K*2] and bilateral
 * features [K*6] from the host, marginals Q [K*L] back to the host. */
int cf_seg_crf(cf_segmenter *s, const float *unary_host, int L, const float *feat_smooth_host, const float *feat_app_host,
               float w_smooth, float w_app, int iterations, float *Q_host);
/* Slic::upsample<unsigned char> (Slic.h:127-139): low_map [K] host -> full-resolution mask on the device */
int cf_seg_upsample(cf_segmenter *s, const uint8_t *low_map_host, uint8_t *full_dev);
/* Device-resident flavour of the three calls above (what the facade uses): per-superpixel sums stay on the device
 * (cf_seg_sums), unary construction, the mean field, arg-max, connected components (ConnectedLabels.hpp:50-172), the
 * largest-component / size / border gates, bounding boxes, depth statistics and the up-sampling into full_dev all run as
 * kernels (cf_seg_infer, Segmentation.cpp:160-706); only the decisions come back (cf_seg_fetch, the one host wait). */
typedef struct {
    float unaryWeightError, unaryKError, unaryThresholdNew;        /* GUI defaults 75, 0.0375, 5.5 (GUI.h:222-224) */
    float weightAppearance, weightSmoothness;                      /* 7, 2 */
    float scaleFeaturesRGB, scaleFeaturesDepth, scaleFeaturesPos;  /* 1/10, 1/0.9, 1/1.8 */
    float minRelSizeNew, maxRelSizeNew;                            /* 0.015, 0.4 */
    int crfIterations;                                             /* 10 */
} cf_seg_params;
#define CF_SEG_MAX_ENTRIES 256   /* model ids are 8 bits (CoFusion.cpp:631-634) and 255 marks a rejected superpixel: at most 255 models + the new label */
typedef struct {  /* SegmentationResult::ModelData (Segmentation.h:45-71) */
    uint32_t id, superPixelCount;
    float avgConfidence, depthMean, depthStd;
    int32_t top, right, bottom, left;
} cf_seg_model;
typedef struct {
    int32_t has_new_label, n_models;   /* n_models: rows of model[] (the new label's row is dropped when it got no superpixel) */
    float depth_range;
    cf_seg_model model[CF_SEG_MAX_ENTRIES];   /* rows [0, n_models) are valid */
} cf_seg_result;
int cf_seg_sums(cf_segmenter *s, const float *depth, int n_models, const float *const *icp_err, const float *const *vertconf4,
                int64_t **sums_dev, uint64_t *sums_words);
int cf_seg_infer(cf_segmenter *s, const cf_seg_params *params, const uint8_t *rgba, int n_models, const uint32_t *model_ids,
                 uint32_t next_model_id, int allow_new, uint8_t *full_dev);
int cf_seg_fetch(cf_segmenter *s, cf_seg_result *out, uint8_t *low_map_host);
/* cf_seg_sums + cf_seg_infer of SEVERAL segmenters of one context -- the sequences of a lock-step group (the reference runs one
 * performSegmentation per CoFusion instance, Segmentation.cpp:124-706) -- through shared launches: the chain of ~30 small kernels is
 * issued once per 8 segmenters (all of one image size, <= 16 labels each; otherwise one chain per segmenter) instead of once per
 * segmenter.  Per segmenter the results are those of the two single calls, bit for bit; each segmenter's decisions arrive with its
 * own cf_seg_fetch.  Single-process callers only: no collective can sit between the sums and the inference. */
typedef struct cf_seg_job {
    cf_segmenter *seg;
    const float *depth;                 /* cf_seg_sums' arguments */
    int32_t n_models;
    const float *const *icp_err;
    const float *const *vertconf4;
    const uint8_t *rgba;                /* cf_seg_infer's arguments */
    const uint32_t *model_ids;
    uint32_t next_model_id;
    int32_t allow_new;
    uint8_t *full_dev;
} cf_seg_job;
int cf_seg_run_batch(cf_ctx *ctx, const cf_seg_params *params, const cf_seg_job *jobs, int n_jobs);
/* Model-parallel callers (one process per GPU, each tracking some of the models): the block cf_seg_sums hands out ends in a tail of
 * max_models x 18 words.  cf_seg_publish_poses (between cf_seg_sums and the caller's in-place SUM all-reduce of the block) writes there, for
 * every model tracked by THIS process (trackers[m] != NULL), the tracked pose (row-major 4x4) + ICP error + ICP inlier count as f32
 * bit patterns, zeros for the others; the all-reduce then leaves every model's pose on every rank, and cf_seg_fetch_poses hands
 * them out after cf_seg_infer / cf_seg_fetch ([n_models][18] words) -- no separate pose exchange, no extra host wait. */
int cf_seg_publish_poses(cf_segmenter *s, int n_models, cf_odom *const *trackers);
int cf_seg_fetch_poses(cf_segmenter *s, int n_models, int64_t *words_host);
/* device view of the SLIC labels, int32 [H*W] */
int cf_seg_labels(cf_segmenter *s, void **dptr, uint64_t *bytes);

/* data path between the RGB residual pass and the RGB step inside the device-resident Gauss-Newton loop.  1 (default): the
 * residual pass packs the valid correspondences of each workgroup into 8 B records, RGB step and 6x6 solve are launches of their
 * own (three launches per iteration); 2: the same records, the RGB step's workgroups of a tracker share one XCD and the last of
 * them to finish runs the solve (two launches per iteration; measured slower at 640x480, kept as an option); 0: the reference's
 * 16 B DataTerm record per pixel (diagnostic / comparison).  Results are bit-identical in all three. */
/* (mode 2 needs hardware workgroup b of a launch to run on XCD b mod 8; checked on the device once per context -- CF_ESTATE and the mode
 * unchanged where that does not hold.  Whatever the mode: a tracking call one of whose solves never ran is reported by
 * cf_odom_fetch_result as CF_ESTATE, not returned as a pose.) */
int cf_set_gn_mode(cf_ctx *ctx, int mode);
/* launch-shape tuning of the ICP reduction (GPUConfig.h:51-58 in the reference): threads per workgroup (64..1024, default 256) and
 * pixels per lane of trackers that reduce their whole image (1, 2, 4; 0 = default: two at pyramid level 0, one below).  The sums are
 * integers: every shape gives the same bits. */
int cf_set_icp_launch(cf_ctx *ctx, int threads, int pixels_per_thread);
/* Rounding specification of the ICP normal equations (icpStep's 27 products + residual, reduce.cu:334-394), a property of the context:
 *   CF_ICP_ARITH_PRODUCT (default): each product row_i*row_j is formed exactly in f64 and rounded once to 2^-CF_FIX_ICP;
 *   CF_ICP_ARITH_GRAM: each row ENTRY is rounded once to a fixed-point grid (2^-20 normal, 2^-17 moment, 2^-22 residual) and the
 *     integer products are summed exactly -- the Gram matrix of an integer matrix, contracted over the pixels on the matrix cores
 *     (signed 8-bit limbs, v_mfma_i32_32x32x32_i8).  Fewer vector instructions per pixel; both forms are exact integer sums
 *     (independent of launch shape and GPU count) and both have an oracle (oracle/orc.h: orc_set_icp_arith).
 * The 64-bit sums cf_icp_step returns are in the units of the chosen form.  Also: CF_ICP_ARITH=product|gram|reference in the environment. */
#define CF_ICP_ARITH_PRODUCT 0
#define CF_ICP_ARITH_GRAM 1
/*   CF_ICP_ARITH_REFERENCE: the reference's OWN order, for every reduction of the tracker (ICP, RGB, SO3), not only the ICP sums:
 *     thread-strided f32 partial sums, 32-lane shuffle-down tree, block tree and second-stage reduceSum of Core/Cuda/reduce.cu:90-185,
 *     396-417, 475-499 at the launch shapes of Core/Utils/GPUConfig.h:51-58, and the host loop of RGBDOdometry.cpp:217-477 around them
 *     (device synchronisation and a read-back per step, host LDL^T).  Not launch-shape independent -- that is the point: its poses,
 *     and with them whole trajectories and surfel counts, equal those of the reference's RGBDOdometry class bit for bit
 *     (tests/test_refpin_gpu.py).  A parity mode: synchronous, ~60 host round trips per tracker and frame; no culling, no split
 *     reductions; cf_icp_step / cf_rgb_step / cf_so3_step return A, b, residual of the f32 trees (sums_host is zero-filled). */
#define CF_ICP_ARITH_REFERENCE 2
int cf_set_icp_arith(cf_ctx *ctx, int mode);
int cf_get_icp_arith(cf_ctx *ctx);

/* micro-benchmark of the ICP reduction on the state of the last tracking call (level 0..2) */
int cf_odom_bench_icp(cf_odom *od, int level, int iters, float *avg_us);

/* timing of the most recent reductions, measured with hipEvents on the ctx stream */
typedef struct {
    double icp_ms_total;   /* accumulated GPU time of ICP-reduce launches since last reset */
    uint64_t icp_launches;
    uint64_t icp_bytes;    /* algorithmic bytes of those launches: per level-0 pixel (24 + 24*M) for the ICP reduction of the M
                            * lock-step models + 11*M for the residual passes that share the launch (27*M with
                            * cf_set_gn_mode 0, which writes the 16 B DataTerm records) */
    double surfel_ms_total;   /* stream time of the sampled cf_models_frame_passes chains (index maps, fuse, clean, compactions, prediction) */
    uint64_t surfel_calls;
    uint64_t surfel_bytes;    /* algorithmic bytes of those chains: 8 passes x 48 B per surfel of every fusing model (2 index + 2
                               * splat reads, update read + write, clean read + write), 2 x 48 B per surfel of a model that is only predicted */
} cf_profile;
/* on = 0: off; on = N >= 1: attach begin/end events to the level-0 launches of every N-th tracking call (sampling keeps the host cost
 * of the event pairs out of the measured frame rate) */
int cf_profile_enable(cf_ctx *ctx, int on);
int cf_profile_read(cf_ctx *ctx, cf_profile *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* COFUSION_HIP_H_ */
