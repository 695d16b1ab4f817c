/*
 * cofusion.h -- flat C wrapper of the C++ facade (co_fusion_amd/host/CoFusion.h) for language bindings
 * (ctypes in bench.py / tests).  Mirrors the calls GUI/MainController.cpp makes on the reference's CoFusion
 * object: construct (MainController.cpp:328-331), setters (:449-473), processFrame (:390), getters.
 * Poses are ROW-major float[16]; 0 = success, negative = error (message via cofusion_last_error()).
 */
#ifndef COFUSION_H_
#define COFUSION_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct cofusion_handle cofusion_handle;

typedef struct {
    int width, height;
    float fx, fy, cx, cy;
    int device, max_surfels, max_models;
    float conf_global_init, conf_object_init, depth_cutoff, icp_weight, outlier_coefficient;
    int fast_odom, so3, frame_to_frame_rgb, pyramid, rgb_only;
    unsigned model_spawn_offset;
    int enable_multiple_models;
} cofusion_config;

void cofusion_default_config(cofusion_config *cfg);
int cofusion_create(const cofusion_config *cfg, cofusion_handle **out);
void cofusion_destroy(cofusion_handle *h);
const char *cofusion_last_error(void);
/* work is enqueued on this hipStream_t (NULL = legacy default stream) */
int cofusion_set_stream(cofusion_handle *h, void *hip_stream);
/* CoFusion::processFrame with host buffers (rgb 3 B/px, depth f32 metres, mask u8 or NULL, in_pose or NULL) */
int cofusion_process_frame(cofusion_handle *h, int64_t timestamp, const uint8_t *rgb, const float *depth, const uint8_t *mask,
                           const float *in_pose);
/* same with the frame already resident in HBM (depth f32, rgba u8x4) */
int cofusion_process_frame_device(cofusion_handle *h, int64_t timestamp, const float *depth_dev, const uint8_t *rgba_dev,
                                  const float *in_pose);
int cofusion_num_models(cofusion_handle *h);
int cofusion_tick(cofusion_handle *h);
/* per model (list order, 0 = background): id, surfel count, pose T(model <- camera), confidence threshold */
int cofusion_model_info(cofusion_handle *h, int index, unsigned *id, unsigned *count, float pose[16], float *conf_threshold);
int cofusion_model_download(cofusion_handle *h, int index, float *surfels, uint32_t capacity, uint32_t *count);
int cofusion_model_icp_stats(cofusion_handle *h, int index, float *icp_error, float *icp_count);
/* device pointer of the full-resolution label mask (u8 [H*W]) */
const uint8_t *cofusion_mask_device(cofusion_handle *h);
/* the underlying C-ABI context (profiling hooks etc.) */
void *cofusion_context(cofusion_handle *h);
/* CRF / segmentation parameters (CoFusion.h:205-248 setters) */
int cofusion_set_crf(cofusion_handle *h, float unary_weight_error, float unary_k_error, float threshold_new, float weight_appearance,
                     float weight_smoothness, float sigma_rgb, float sigma_depth, float sigma_pos, float min_rel_size_new,
                     float max_rel_size_new, unsigned iterations);

#ifdef __cplusplus
}
#endif
#endif
