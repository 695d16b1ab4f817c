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
    int device, max_surfels, max_models; /* at most min(max_models, 255) models are active at a time (default 16): the context's tracker
                                          * staging and the segmentation's label dimension are sized by it; ids are 8 bits and 255 marks
                                          * a rejected superpixel (the reference's own limit, CoFusion.cpp:631-634).  Beyond the cap new
                                          * objects are not spawned (reported once on stderr) */
    float conf_global_init, conf_object_init, depth_cutoff, icp_weight, outlier_coefficient;
    int fast_odom, so3, frame_to_frame_rgb, pyramid, rgb_only;
    unsigned model_spawn_offset;
    int enable_multiple_models;
    int enable_pose_logging; /* CoFusion ctor argument enablePoseLogging (CoFusion.h:59); needed by cofusion_export_poses */
    int rank, world;         /* model-parallel operation over `world` processes / GPUs (default 0, 1); see cofusion_set_allreduce */
    int device_frames_complete; /* 1: buffers given to cofusion_process_frame_device are complete at call time (uploaded ahead), so the
                                 * new frame's depth filter may run beside the previous frame's fusion passes; 0 (default): they may be
                                 * produced by work queued on the context's stream and are consumed in stream order */
    int mid_frame_predict;      /* 1: also run the reference's prediction between tracking and fusion (CoFusion.cpp:346); its outputs are
                                 * overwritten by the end-of-frame prediction before the frame loop reads them (GUI only).  Default 0. */
    int shard_background;       /* model-parallel operation (world > 1): 1 = every rank keeps a replica of the background map and takes a
                                 * share of its index-map rasterisation (surfel range, MIN all-reduce of the z-keys) and of its ICP
                                 * reduction (image rows, SUM all-reduce of the accumulators after every launch of the Gauss-Newton
                                 * loop); needs cf_set_collective on the context (cofusion_context).  Default 0. */
    int enqueue_threads;        /* model-parallel operation (world > 1) only: helper threads that enqueue the per-model surfel passes (one
                                 * model's launch chain each) beside the calling thread; 0 = none (default).  A single process runs the
                                 * passes of all models as one chain of batched launches and ignores it.  Results do not depend on it. */
    int colocate_background;    /* model-parallel operation: 1 = object models round-robin over ALL ranks, the background shares rank 0
                                 * (BASELINE.json configs[3]: one object model per GPU); 0 (default) = the background alone on rank 0 */
    int reloc;                  /* CoFusion's `reloc` constructor argument (CoFusion.h:47): failure detection of the frame loop -- frames
                                 * with a background ICP error >= 1e-4 or a pose-covariance diagonal entry > 1e-4 are not fused, after
                                 * more than ten in a row the camera is lost (no fusion, the clock stops; CoFusion.cpp:225,301-338).
                                 * cofusion_is_lost reports it.  Default 0. */
    int early_index_maps;       /* 1 (default): the index maps of the tracked models are rasterised with the poses the trackers left ON THE
                                 * DEVICE, behind the segmentation and before the frame's host wait (cf_models_preindex), so the GPU works
                                 * while the host reads poses and decisions; 0: with the rest of the surfel chain, after the wait.  Results are
                                 * identical either way. */
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
/* CoFusion::getLost (CoFusion.h:183-185): 1 while the camera is lost (cofusion_config.reloc) */
int cofusion_is_lost(cofusion_handle *h);
/* per model (list order, 0 = background): id, surfel count, pose T(model <- camera), confidence threshold */
int cofusion_model_info(cofusion_handle *h, int index, unsigned *id, unsigned *count, float pose[16], float *conf_threshold);
int cofusion_model_download(cofusion_handle *h, int index, float *surfels, uint32_t capacity, uint32_t *count);
int cofusion_model_icp_stats(cofusion_handle *h, int index, float *icp_error, float *icp_count);
/* the level-0 pixel rectangle [x0, y0, x1, y1] the model's last ICP iteration was restricted to (cf_track_stats::cull_box) */
int cofusion_model_cull_box(cofusion_handle *h, int index, int box[4]);
/* pixels the level-0 {ICP || residual} launch of the model's last tracking call visited for it (cf_odom_level0_visited) */
int cofusion_model_level0_visited(cofusion_handle *h, int index, uint64_t *icp_pixels, uint64_t *residual_pixels);
/* host copies of what the NEXT frame's tracking of this model reads (Model::initICP, Model.cpp:350-367): the predicted
 * vertex+conf / normal+radius maps (f32x4 [H*W]) and the predicted image (rgba8 [H*W]); any pointer may be NULL */
int cofusion_model_tracking_inputs(cofusion_handle *h, int index, float *vertex4, float *normal4, uint8_t *image_rgba);
/* device pointer of the full-resolution label mask (u8 [H*W]) */
const uint8_t *cofusion_mask_device(cofusion_handle *h);
/* the underlying C-ABI context (profiling hooks etc.) */
void *cofusion_context(cofusion_handle *h);
/* CRF / segmentation parameters (CoFusion.h:205-248 setters) */
int cofusion_set_crf(cofusion_handle *h, float unary_weight_error, float unary_k_error, float threshold_new, float weight_appearance,
                     float weight_smoothness, float sigma_rgb, float sigma_depth, float sigma_pos, float min_rel_size_new,
                     float max_rel_size_new, unsigned iterations);

/* Several independent RGB-D sequences on ONE GPU in lock-step (throughput mode): the sequences share one context, and every set of
 * tracking launches (map preparation, SO(3) pre-alignment, the Gauss-Newton loop) carries the trackers of all of them, up to 16 per
 * launch -- the kernels of a 640x480 frame run at their launch floor, more trackers per launch is what fills the GPU.  Each sequence
 * keeps its own maps, models, segmentation and clock; its results are those of a cofusion_handle of its own, bit for bit.  One
 * configuration for all sequences (cfg->max_models is per sequence).  cofusion_group_sequence returns a BORROWED handle for the
 * per-sequence getters (model info, download, mask, export ...): do not pass it to cofusion_destroy / cofusion_process_frame*. */
typedef struct cofusion_group cofusion_group;
int cofusion_group_create(const cofusion_config *cfg, int sequences /* 1..16 */, cofusion_group **out);
void cofusion_group_destroy(cofusion_group *g);
int cofusion_group_size(cofusion_group *g);
cofusion_handle *cofusion_group_sequence(cofusion_group *g, int s);
int cofusion_group_set_stream(cofusion_group *g, void *hip_stream);
/* one frame of EVERY sequence: entry s of each array belongs to sequence s (timestamps / mask: nullable, mask entries nullable) */
int cofusion_group_process_frames(cofusion_group *g, const int64_t *timestamps, const uint8_t *const *rgb, const float *const *depth_m,
                                  const uint8_t *const *mask);
int cofusion_group_process_frames_device(cofusion_group *g, const int64_t *timestamps, const float *const *depth_dev,
                                         const uint8_t *const *rgba_dev);

/* Model-parallel mode (cfg.world > 1, one process per GPU, every rank fed the same frames): the object models are placed
 * round-robin on ranks 1.., the background on rank 0; every rank runs the same frame loop and keeps data-less shadows of
 * the models it does not own.  All inter-rank traffic (poses after tracking, per-superpixel ICP / confidence sums for
 * the CRF, surfel counts at retirement) goes through this one callback: an in-place SUM all-reduce of `n` int64 values
 * over all ranks (e.g. ncclAllReduce / torch.distributed.all_reduce); return 0 on success.  Register before frame 1. */
typedef int (*cofusion_allreduce_i64_fn)(int64_t *buf, uint64_t n, void *user);
int cofusion_set_allreduce(cofusion_handle *h, cofusion_allreduce_i64_fn fn, void *user);
/* Optional second form of the same collective for buffers that live in HBM (the per-superpixel segmentation sums, 2*16*K
 * int64): an in-place SUM all-reduce of `n` int64 values at device address `dev_buf`, ENQUEUED on `hip_stream` (ncclAllReduce on
 * that stream, or on a stream ordered after it and before whatever is enqueued next) -- no host visit.  Without it such
 * buffers are staged through the host callback. */
typedef int (*cofusion_allreduce_dev_fn)(int64_t *dev_buf, uint64_t n, void *hip_stream, void *user);
int cofusion_set_allreduce_device(cofusion_handle *h, cofusion_allreduce_dev_fn fn, void *user);
/* RCCL inside the library: instead of the two callbacks above, give the instance its own ncclComm_t (created on the instance's
 * device, one process per GPU).  Rank 0 creates the 128-byte ncclUniqueId (cofusion_rccl_unique_id) and hands it to the other
 * ranks over any side channel (a file, MPI, a torch.distributed broadcast); then EVERY rank of cfg.world calls cofusion_init_rccl
 * (collective).  From then on the segmentation sums / tracked poses, the split reductions of a sharded background and the host-side
 * exchanges run ncclAllReduce in place on the context's stream -- no staging copies, no callback into the host language.
 * cofusion_broadcast sends a device buffer (a frame: depth + colour) from rank `root` to every rank with ncclBroadcast on the same
 * stream, so a following cofusion_process_frame_device consumes it in stream order (device_frames_complete = 0). */
int cofusion_rccl_unique_id(void *id128 /* 128 bytes */);
int cofusion_init_rccl(cofusion_handle *h, const void *id128);
int cofusion_broadcast(cofusion_handle *h, void *dev_buf, uint64_t bytes, int root);
/* 1 if the model at `index` lives on this rank, 0 if it is a shadow (count reads 0, download is empty) */
int cofusion_model_owned(cofusion_handle *h, int index);
/* diagnostics: accumulated host wall-clock (ms) per processFrame phase on the calling thread -- prepare, track, slic+sums,
 * unaries, crf, segmentation post-processing, model logic, fuse+clean, predict; returns the number of phases */
int cofusion_debug_phase_ms(double *out, int n, long *frames, int reset);
/* CoFusion::savePly / exportPoses (CoFusion.cpp:646-783): writes <prefix>cloud-<id>.ply / <prefix>poses-<id>.txt; returns the
 * number of files written or a negative error */
int cofusion_save_ply(cofusion_handle *h, const char *export_dir_prefix);
int cofusion_export_poses(cofusion_handle *h, const char *export_dir_prefix);
/* exportSegmentation (CoFusion.cpp:235-240): every segmented frame writes <prefix>Segmentation<tick>.png; NULL / "" switches it off */
int cofusion_set_export_segmentation(cofusion_handle *h, const char *export_dir_prefix);

/* .klg RGB-D logs (GUI/Tools/KlgLogReader.cpp:22-87): u16-mm depth raw or zlib, 8-bit x3 colour raw (JPEG frames are
 * rejected: no libjpeg in this build).  depth_m [H*W] metres, rgb [H*W*3]. */
typedef struct cofusion_klg_reader cofusion_klg_reader;
typedef struct cofusion_klg_writer cofusion_klg_writer;
int cofusion_klg_open(const char *file, int width, int height, int flip_colors, cofusion_klg_reader **out, int *num_frames);
int cofusion_klg_next(cofusion_klg_reader *r, int64_t *timestamp, float *depth_m, uint8_t *rgb);
/* 1: stop one frame early like the reference's KlgLogReader::hasMore() (`currentFrame + 1 < numFrames`); default 0 = play every frame */
int cofusion_klg_set_reference_compatible(cofusion_klg_reader *r, int on);
void cofusion_klg_close(cofusion_klg_reader *r);
int cofusion_klg_create(const char *file, int width, int height, int compress_depth, cofusion_klg_writer **out);
int cofusion_klg_write(cofusion_klg_writer *w, int64_t timestamp, const float *depth_m, const uint8_t *rgb);
int cofusion_klg_finish(cofusion_klg_writer *w);

#ifdef __cplusplus
}
#endif
#endif
