/*
 * orc.h -- public declarations of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (co_fusion_amd/, the
 * C-ABI library, the C++ facade) may include, link or call anything in oracle/.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * PARITY: the reference (martinruenz/co-fusion) ships no tests, golden vectors or
 * fixtures and its build cannot run here.  This oracle is a
 * from-scratch restatement of the reference arithmetic, function by function, each
 * citing the reference file:line it follows.
 *   PINNED   orc_track.c kernels and orc_surfel.c passes: against the reference's own
 *            CUDA kernels / GLSL shaders compiled for the CPU (oracle/ref_shim ->
 *            oracle/_ref; fixtures tests/golden/ref_v1.npz, ref_surfel_v1.npz;
 *            tests/test_cpu_refpin.py).
 *            The host Gauss-Newton loop of orc_track.c: against the reference's own RGBDOdometry class compiled
 *            with a fixed-size Eigen stand-in (Eigen is absent; its rounding conventions are stated in
 *            oracle/ref_shim/eigen_fixed/Eigen/Core): counts identical, poses within 5e-6 (ref_odo_v1.npz).
 *            orc_segment.c's host logic: against Segmentation.cpp / Slic.* / ConnectedLabels.hpp (ref_seg_v1.npz).
 *            The frame loop (tests/orc_multi.py, tests/orc_pipeline.py): against the text of CoFusion::processFrame and its
 *            helpers (ref_cofusion_v1.json): identical bits.
 *   UNPINNED the two third-party algorithms that are not in the tree (gSLICr, densecrf): stated in orc_segment.c.
 *
 * Data layouts (identical to the HIP C-ABI in include/cofusion_hip.h):
 *   depth            f32  [H*W] metres, 0 = invalid
 *   rgba image       u8x4 [H*W] (R,G,B,A)
 *   vertex/normal 4  f32x4 [H*W]  (GL RGBA32F texture layout of the reference)
 *   planar map       f32  [3*H*W] rows 0..H-1 = x, H..2H-1 = y, 2H..3H-1 = z
 *                    (reference DeviceArray2D<float>(3*rows, cols), unpitched)
 *   surfel           12 f32: [x y z conf][colour24 0 initTime lastTime][nx ny nz radius]
 */
#ifndef ORC_H_
#define ORC_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_PYRS 3
#define ORC_SE3_WORDS 32 /* 27 products, residual, inliers, 3 pad */
#define ORC_SO3_WORDS 16 /* 9 products, residual, inliers, pad */
#define ORC_FIX_ICP 32
#define ORC_FIX_RGB 32 /* for sigma >= 4096; in general orc_rgb_fix_bits(sigma), see orc_math.h */
#define ORC_FIX_SO3 12

/* Rounding specification of the ICP normal equations (the RGB and SO3 sums always use the first):
 *   ORC_ICP_ARITH_PRODUCT (default): every product row_i*row_j is formed exactly and rounded ONCE to 2^-32 (orc_fix_prod);
 *   ORC_ICP_ARITH_GRAM: every row ENTRY is rounded once to a fixed-point grid (orc_gram_quant: 2^-20 for the normal,
 *     2^-17 for the moment, 2^-22 for the residual) and the products of those integers are summed exactly -- the sums are
 *     the Gram matrix of an integer matrix, which the HIP kernels contract on the matrix cores (signed 8-bit limbs,
 *     v_mfma_i32_32x32x32_i8; cf_set_icp_arith).
 * Process-global switch: test infrastructure. */
#define ORC_ICP_ARITH_PRODUCT 0
#define ORC_ICP_ARITH_GRAM 1
/*   ORC_ICP_ARITH_REFERENCE: no fixed point at all -- EVERY reduction of the tracker (ICP, RGB, SO3) is the reference's own
 *     launch-shape dependent f32 tree at GPUConfig.h's default shapes and the host algebra follows the order of the classes the
 *     reference instantiates (orc_track.c: odom_track_reference_order); the mode in which whole trajectories and surfel counts
 *     equal those of the reference's RGBDOdometry class bit for bit (cf_set_icp_arith 2 on the HIP side). */
#define ORC_ICP_ARITH_REFERENCE 2
void orc_set_icp_arith(int mode);
int orc_get_icp_arith(void);

typedef struct { float fx, fy, cx, cy; } orc_cam;

/* reference DataTerm, Core/Cuda/types.cuh:75-81 (16 bytes; `valid` is a bool + pad) */
typedef struct {
    int16_t zero_x, zero_y;
    int16_t one_x, one_y;
    float diff;
    int32_t valid;
} orc_dataterm;

/* ------------------------------ map preparation ------------------------------ */
void orc_create_vmap(const float *depth, int cols, int rows, orc_cam intr, float depth_cutoff, float *vmap);
void orc_create_nmap(const float *vmap, int cols, int rows, float *nmap);
void orc_copy_maps(const float *v4, const float *n4, int cols, int rows, float *vmap, float *nmap);
void orc_resize_map(const float *in, int in_cols, int in_rows, float *out, int normalize);
void orc_transform_maps(float *vmap, float *nmap, int cols, int rows, const float R[9], const float t[3]);
void orc_vertices_to_depth(const float *v4, int cols, int rows, float cutoff, float *depth);
void orc_pyrdown_gauss_f32(const float *src, int src_cols, int src_rows, float *dst);
void orc_pyrdown_gauss_u8(const uint8_t *src, int src_cols, int src_rows, uint8_t *dst);
void orc_rgba_to_intensity(const uint8_t *rgba, int cols, int rows, uint8_t *dst);
void orc_sobel(const uint8_t *src, int cols, int rows, int16_t *dx, int16_t *dy);
void orc_project_cloud(const float *depth, int cols, int rows, orc_cam intr_level, float *cloud3);

/* -------------------------------- reductions --------------------------------- */
/* Exact (fixed-point) statement: sums[0..26] upper-triangular products row_i*row_j
 * (i<=j<7, i<6) in Q(63-F).F, sums[27] = residual^2 (Q.F), sums[28] = inliers (integer). */
void orc_icp_step(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                  const float Rprev_inv[9], const float tprev[3], orc_cam intr, const float *vmap_g_prev,
                  const float *nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows,
                  int64_t sums[ORC_SE3_WORDS], float *err_surface /* nullable, [rows*cols] */);
/* reference summation order in f32 (grid-stride + warp32 shuffle tree), reduce.cu:90-185,396-417 */
void orc_icp_step_f32tree(const float Rcurr[9], const float tcurr[3], const float *vmap_curr, const float *nmap_curr,
                          const float Rprev_inv[9], const float tprev[3], orc_cam intr, const float *vmap_g_prev,
                          const float *nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows,
                          int threads, int blocks, float out29[29]);
void orc_rgb_residual(float min_scale, const int16_t *dIdx, const int16_t *dIdy, const float *last_depth,
                      const float *next_depth, const uint8_t *last_image, const uint8_t *next_image,
                      orc_dataterm *corres, float max_depth_delta, const float kt[3], const float krkinv[9],
                      int cols, int rows, int *sigma_sum, int *count);
void orc_rgb_step(const orc_dataterm *corres, float sigma, const float *cloud3, float fx, float fy,
                  const int16_t *dIdx, const int16_t *dIdy, float sobel_scale, int cols, int rows,
                  int64_t sums[ORC_SE3_WORDS]);
void orc_so3_step(const uint8_t *last_image, const uint8_t *next_image, const float image_basis[9],
                  const float kinv[9], const float krlr[9], int cols, int rows, int64_t sums[ORC_SO3_WORDS]);
/* sums -> the reference's host outputs (reduce.cu:481-498, 1158-1175) */
int orc_rgb_fix_bits_of(float sigma);
void orc_rgb_step_f32tree(const orc_dataterm *corres, float sigma, const float *cloud3, float fx, float fy, const int16_t *dIdx,
                          const int16_t *dIdy, float sobel_scale, int cols, int rows, int threads, int blocks, float out29[29]);
void orc_so3_step_f32tree(const uint8_t *last_image, const uint8_t *next_image, const float image_basis[9], const float kinv[9],
                          const float krlr[9], int cols, int rows, int threads, int blocks, float out11[11]);
void orc_se3_sums_to_host(const int64_t sums[ORC_SE3_WORDS], int F, float A[36], float b[6], float residual[2]);
/* the ICP sums under the current rounding specification (orc_set_icp_arith): F = ORC_FIX_ICP, or the per-entry scales of the Gram form */
void orc_icp_sums_to_host(const int64_t sums[ORC_SE3_WORDS], float A[36], float b[6], float residual[2]);
void orc_so3_sums_to_host(const int64_t sums[ORC_SO3_WORDS], int F, float A[9], float b[3], float residual[2]);

/* ----------------------- RGBDOdometry (Core/Utils/RGBDOdometry.*) ------------- */
typedef struct orc_odometry orc_odometry;
orc_odometry *orc_odom_create(int width, int height, float cx, float cy, float fx, float fy);
void orc_odom_destroy(orc_odometry *o);
void orc_odom_init_icp_model(orc_odometry *o, const float *pred_v4, const float *pred_n4, const float pose[16]);
void orc_odom_init_rgb_model(orc_odometry *o, const uint8_t *pred_rgba);
void orc_odom_init_icp(orc_odometry *o, const float *const depth_pyr[ORC_NUM_PYRS], float depth_cutoff);
void orc_odom_init_rgb(orc_odometry *o, const uint8_t *rgba);
void orc_odom_init_first_rgb(orc_odometry *o, const uint8_t *rgba);
typedef struct {
    int rgb_only, pyramid, fast_odom, so3;
    float icp_weight;
} orc_track_opts;
typedef struct {
    float last_icp_error, last_icp_count, last_rgb_error, last_rgb_count, last_so3_error, last_so3_count;
    double lastA[36], lastb[6];
    int so3_iterations;
} orc_track_stats;
/* trans[3]/rot[9] (row-major) in-out; err_surface nullable [H*W] */
void orc_odom_get_incremental_transformation(orc_odometry *o, float trans[3], float rot[9], const orc_track_opts *opts,
                                             float *icp_err_surface, orc_track_stats *stats);
/* RGBDOdometry::getCovariance (RGBDOdometry.cpp:479): lastA.cast<double>().lu().inverse(), row-major 6x6 in and out */
void orc_covariance(const double lastA[36], double cov[36]);
/* test access to internal pyramids: which = 0 vmap_curr,1 nmap_curr,2 vmap_g_prev,3 nmap_g_prev (planar f32),
 * 4 lastDepth,5 nextDepth (f32), 6 lastImage,7 nextImage,8 lastNextImage (u8), 9 dIdx,10 dIdy (s16) */
const void *orc_odom_buffer(const orc_odometry *o, int which, int level);

/* ------------------------------ surfel path (orc_surfel.c) --------------------------------- */
void orc_inverse_pose(const float pose[16], float out[16]);
void orc_bilateral(const float *depth, int cols, int rows, float maxD, float *out);
int orc_vertex_feedback(const uint8_t *rgba, const float *depth, int cols, int rows, orc_cam cam, int time, float maxDepth, float *out);
int orc_model_initialise(const float *raw_fb, int raw_count, const float *filtered_fb, float *surfels);
void orc_predict_indices(const float *surfels, int count, const float pose[16], orc_cam cam, int cols, int rows, float maxDepth,
                         int time, int timeDelta, uint32_t *index, float *vertConf4, float *colorTime4, float *normRad4);
void orc_combined_predict(const float *surfels, int count, const float pose[16], orc_cam cam, int cols, int rows, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, uint8_t *image_rgba, float *vertexConf4,
                          float *normalRad4, uint16_t *time16);
void orc_fill_in(const float *pred_vertex4, const float *pred_normal4, const uint8_t *pred_image, const float *depth,
                 const uint8_t *rgba, int cols, int rows, orc_cam cam, int passthrough_geom, int passthrough_rgb,
                 float *out_vertex4, float *out_normal4, uint8_t *out_image);
int orc_requires_fill_in(const uint8_t *pred_image, int cols, int rows, float ratio);
void orc_fuse(const float *surfels_in, int count, const uint32_t *index, const float *vertConf4, const float *normRad4,
              const uint8_t *rgba, const float *depth_raw, const float *depth_filt, const uint8_t *mask, const float pose[16],
              orc_cam cam, int cols, int rows, int time, float weighting, int maskID, float maxDepth, float *surfels_out,
              float *new_unstable, int *n_new);
int orc_clean(const float *surfels_in, int count, const float *new_unstable, int n_new, const uint32_t *index, const float *vertConf4,
              const float *colorTime4, const float *depth_filt, const uint8_t *mask, const float pose[16], orc_cam cam, int cols,
              int rows, int time, float confThreshold, float outlierCoeff, int timeDelta, int maskID, float *surfels_out);
float orc_fusion_weight(const float pose[16], const float lastPose[16], float weightMultiplier);

/* ------------------------------ segmentation (orc_segment.c) -------------------------------- */
typedef struct {
    float unaryWeightError, unaryKError, unaryThresholdNew; /* GUI defaults 75, 0.0375, 5.5 (GUI.h:222-224) */
    float weightAppearance, weightSmoothness;               /* 7, 2 */
    float scaleFeaturesRGB, scaleFeaturesDepth, scaleFeaturesPos; /* 1/10, 1/0.9, 1/1.8 */
    float minRelSizeNew, maxRelSizeNew;                      /* 0.015, 0.4 */
    int crfIterations;                                       /* 10 */
} orc_seg_params;
typedef struct {
    unsigned id, superPixelCount;
    float avgConfidence, depthMean, depthStd;
    int top, right, bottom, left;
} orc_seg_model;
int orc_connected_labels(const uint8_t *in, int cols, int rows, int *comp, int *stats6, int max_stats);
void orc_slic(const uint8_t *rgba, int cols, int rows, int32_t *labels);
void orc_crf_kernel(const float *feat, int D, int n, float *Kn);
void orc_crf_exp_and_normalize(const float *in, float *out, int L, int n);
void orc_crf_apply(const float *Kn, int n, int L, float w, const float *Q, float *out);
void orc_crf_meanfield(const float *unary, int L, int n, const float *feat_smooth, const float *feat_app, float w_smooth,
                       float w_app, int iterations, float *Q);
int orc_segment_crf(const orc_seg_params *P, int cols, int rows, const uint8_t *rgba, const float *depth, int n_models,
                    const unsigned *model_ids, const float *const *icp_err, const float *const *vertconf4, unsigned nextModelID,
                    int allowNew, uint8_t *full_seg, orc_seg_model *out_models, int *n_out, int *hasNewLabel, float *depthRange_out,
                    int32_t *labels_out, uint8_t *low_map_out);
int orc_segment_gt(const uint8_t *gt_mask, const float *depth, int cols, int rows, int n_models, const unsigned *model_ids,
                   unsigned nextModelID, int allowNew, uint8_t *mapping, uint8_t *full_seg, orc_seg_model *out_models, int *n_out,
                   int *hasNewLabel);

/* Model::generateCUDATextures depth pyramid (Model.cpp:319-348) */
void orc_depth_pyramid(const float *depth_filtered, int cols, int rows, float *l1, float *l2);

#ifdef __cplusplus
}
#endif
#endif /* ORC_H_ */
