// cf_host.h -- host-side object definitions behind the opaque C-ABI handles.
#pragma once

#include <mutex>
#include <string>

#include "cf_kernels.h"

struct cf_ctx;
// cf_thread_lane: a helper thread of the host enqueues one model's passes on a lane while the owning thread does the same for
// another model; the binding is per thread, so the context's own `stream` field is never written from two threads
struct cf_thread_binding { const cf_ctx* ctx = nullptr; hipStream_t stream = nullptr; };
extern thread_local cf_thread_binding cf_tls_binding;

struct cf_ctx {
    cf_config cfg{};
    hipStream_t stream = nullptr;
    hipStream_t cur() const { return cf_tls_binding.ctx == this ? cf_tls_binding.stream : stream; }  // stream of the calling thread
    hipStream_t own_stream = nullptr;
    std::string last_error;
    std::mutex error_mutex;            // helper threads bound with cf_thread_lane may fail at the same time
    cf::IcpLaunch icp_launch{256, 0};
    int icp_arith = 0;                 // cf_set_icp_arith: 0 product, 1 Gram (= icp_launch.gram), 2 the reference's own f32 trees + host loop (track_ref.hip)
    float* d_ref = nullptr;            // scratch of the reference-order tracker (block partials, totals, counts), created on first use
    float* h_ref = nullptr;            // ... and its pinned read-back
    // scratch for the stand-alone reduction steps
    unsigned long long* d_acc_a = nullptr;
    unsigned long long* d_acc_b = nullptr;
    unsigned long long* d_out = nullptr;
    unsigned long long* h_out = nullptr;  // pinned
    cf::OdomDev* d_scratch_state = nullptr;
    cf::OdomDev* h_scratch_state = nullptr;  // pinned
    uint8_t* d_cand_scratch = nullptr;
    cf::So3Sync* d_so3_sync = nullptr;  // [max_models]
    unsigned gn_mode_epoch = 0;         // bumped by every change of gn_mode: a tracker that last ran under another epoch clears its RGB accumulators first
    int xcd_round_robin = -1;           // probe_xcd_round_robin of this device: -1 not probed yet, 0 / 1
    int gn_mode = 1;                    // launch_gn_track mode (1: record slots between the residual pass and the RGB step; 2: ... whose last
                                        // workgroup per tracker runs the solve -- measured slower, DESIGN-NOTES R5; 0: DataTerm image)
    // device / pinned-host pools of the trackers' state structs: a batch of trackers is uploaded / read back with ONE
    // copy over its slot range instead of one copy per tracker
    static constexpr int kStateSlots = 256;  // (up to 255 models per sequence; trackers beyond the pool keep state blocks of their own)
    cf::OdomDev* d_state_pool = nullptr;
    cf::OdomDev* h_state_pool = nullptr;  // pinned
    bool slot_used[kStateSlots]{};
    hipEvent_t batch_event = nullptr;     // cf_models_frame_passes: ONE event behind a batch's compactions marks all its models' counts
    // auxiliary streams for independent per-model work of one frame (cf_fork / cf_join)
    static constexpr int kLanes = 8;
    hipStream_t lanes[kLanes]{};
    hipEvent_t lane_done[kLanes]{};
    hipEvent_t fork_point = nullptr;
    static constexpr int kMarks = 64;  // (a group of sequences sharing the context uses four per sequence)
    hipEvent_t marks[kMarks]{};       // cf_mark / cf_fork_after: points of the main stream a detached lane waits for
    bool mark_set[kMarks]{};
    hipStream_t forked_from = nullptr;
    bool forked = false;
    unsigned lanes_used = 0;
    // collective of a multi-GPU caller (cf_set_collective): op 0 = in-place SUM all-reduce of int64 words, op 1 = in-place MIN
    // all-reduce of unsigned 64-bit words; enqueued on the given stream
    int (*collective)(void* user, int op, void* dev_buf, uint64_t words, void* stream) = nullptr;
    void* collective_user = nullptr;
    void* rccl = nullptr;              // the library's own RCCL communicator (cf_rccl_init, rccl_comm.hip); its collective replaces the caller's
    unsigned prof_calls = 0;           // tracking calls seen while profiling is on (events are attached to every prof.enabled-th call)
    cf::ProfSink prof{};
    double prof_ms_accum = 0;
    // ... and event pairs around the sampled surfel chains (cf_models_frame_passes)
    static constexpr int kSurfEvents = 64;
    hipEvent_t surf_events[kSurfEvents]{};
    int surf_used = 0; unsigned surf_calls_seen = 0;
    double surf_ms_accum = 0; uint64_t surf_calls = 0, surf_bytes = 0;
    void set_error(const std::string& m);
};
extern "C" int cf_wait_stream(cf_ctx* ctx);   // the frame's host wait (cabi.hip)
struct cf_odom;
namespace cf {
// the reference-order tracker (track_ref.hip): RGBDOdometry::getIncrementalTransformation for one prepared tracker, host loop included
int ref_track(cf_ctx* ctx, cf_odom* od, const float pose[16], const cf_track_opts* opts, float* err_surface);
// ... and its single reductions in the reference's own f32 order (the C-ABI steps under cf_set_icp_arith 2)
int ref_icp_step(cf_ctx* ctx, const float Rcurr[9], const float tcurr[3], const float* vc, const float* nc, const float Rprev_inv[9], const float tprev[3],
                 cf_cam intr, const float* vp, const float* np, float dist_thres, float angle_thres, int cols, int rows, float* err, float out29[29]);
int ref_rgb_step(cf_ctx* ctx, const cf_dataterm* corres, float sigma, const float* cloud3, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                 float sobel_scale, int cols, int rows, float out29[29]);
int ref_so3_step(cf_ctx* ctx, const uint8_t* last_image, const uint8_t* next_image, const float basis[9], const float kinv[9], const float krlr[9],
                 int cols, int rows, float out11[11]);
}

// Device-resident RGBDOdometry (Core/Utils/RGBDOdometry.h:78-137)
struct cf_odom {
    cf_ctx* ctx = nullptr;
    float* vmaps_tmp = nullptr;
    float* nmaps_tmp = nullptr;
    float* vmap_g_prev[3]{};
    float* nmap_g_prev[3]{};
    float* vmap_curr[3]{};
    float* nmap_curr[3]{};
    const float* ext_vmap_curr[3]{};  // frame-shared current maps (all models track the same frame)
    const float* ext_nmap_curr[3]{};
    float2* zrange[3]{};              // depth interval of every 64-pixel run of vmap_curr (written with the frame maps, cf_odom_init_icp)
    bool zrange_valid = false;        // cf_odom_init_icp wrote zrange for the current contents of vmap_curr (cleared when a caller takes the map's pointer)
    const float2* ext_zrange[3]{};    // ... of the shared frame maps (cf_odom_share_frame_maps), null when bound without them
    float* lastDepth[3]{};
    float* nextDepth[3]{};
    // cf_odom_init_models_batch: initRGB takes its depth from the same snapshot of the predicted vertex map as initRGBModel
    // (RGBDOdometry.cpp:179), so the "next" depth pyramid IS the "last" one; the batch path builds it once and reads lastDepth for both
    bool next_depth_is_last = false;
    float* next_depth(int i) const { return next_depth_is_last ? lastDepth[i] : nextDepth[i]; }
    uint8_t* lastImage[3]{};
    uint8_t* nextImage[3]{};
    uint8_t* lastNextImage[3]{};
    int16_t* dIdx[3]{};
    int16_t* dIdy[3]{};
    float* cloud[3]{};
    cf_dataterm* corres[3]{};
    uint8_t* cand[3]{};
    unsigned char* occ = nullptr;    // occupancy map of the model maps (written by model_maps_kernel)
    bool occ_valid = false;          // the map describes the current model maps
    bool use_occ = false;            // cf_odom_set_culling: occupancy look-up + screen-box culling of the ICP reduction
    unsigned* aabb = nullptr;        // 8 words: bounding-box accumulator of the model maps (OdomDev::aabb_acc)
    float map_pose[12]{};            // R (9) | t (3) of the pose the model maps were prepared with: the camera frame of the bounding frustum
    int box_hint[4] = {0x7fffffff, 0, 0, 0};  // level-0 screen box at the end of the previous tracking call (kNoBoxHint: none): sizes the culled launches
    unsigned* res_range = nullptr;   // 8 words: first / last candidate chunk per level of the RGB residual pass (OdomDev::res_range)
    int res_hint[3] = {-1, -1, -1};  // record slots per level between them in the previous tracking call (-1: unknown): sizes the residual launches
    bool box_valid = false;          // the model-map pass of this frame fed the accumulator
    int band_begin = 0, band_end = 0;  // cf_odom_set_band: this rank's rows of the model's reductions (0, 0: all rows)
    bool band_counts = true;           // this rank adds the residual pass's count / sigma (exactly one rank of a split does)            // cf_odom_set_culling: worth it for models that cover a small part of the image
    unsigned long long* icp_acc = nullptr;
    unsigned long long* rgb_acc = nullptr;
    cf::OdomDev* d_state = nullptr;
    cf::OdomDev* h_state = nullptr;  // pinned
    int slot = -1;                   // index into the context's state pools, -1: own allocations
    float distThres = 0, angleThres = 0, sobelScale = 0, maxDepthDeltaRGB = 0, maxDepthRGB = 0;
    float angleSqLt = 0, distSqLe = 0;  // exact radicand bounds of the two ICP gates
    float minGrad[3]{};
    bool pending_so3_swap = false;
    size_t rgb_acc_words = 0;        // size of rgb_acc (kGroups rows, or one row per RGB-step workgroup of cf_set_gn_mode 2)
    unsigned gn_epoch_seen = 0;      // cf_ctx::gn_mode_epoch of the last tracking call
    bool result_pending = false;     // a tracking call of this tracker is in flight / not fetched: its pinned host state must not be rewritten yet
    int expected_solves = -1;        // Gauss-Newton iterations the pending tracking call enqueued (-1: not checked), against OdomDev::solves
};
