// cf_kernels.h -- internal launcher declarations + device-resident tracker state.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/cofusion_hip.h"

namespace cf {

constexpr int kGroups = 64;  // accumulation groups for the grouped integer atomics

// HOT STATE of a tracker (round 5): everything the workgroups of the per-iteration launches read from the device-resident state,
// packed into three 64-byte scalar-cache lines.  The solve of the previous launch wrote the state on another XCD, so the first wave on
// every CU takes its scalar loads of it all the way to memory; spread over the 1 KB OdomDev, and tested field by field (`!st->icp ||
// st->level_done`, then the pose, then the box), those were three to five DEPENDENT round trips in front of every first-round wave.
// Now: one clause of s_load_dwordx16 (lines 0 + 1 for the ICP reduction, line 2 for the residual pass and the RGB step), one wait.
// A derived copy: the fields of OdomDev stay the source of truth; refresh_hot() re-derives it wherever the state is written (host
// preparation, the first launch of the schedule, every solve).
struct alignas(64) GnHot {
    // line 0
    int icp, level_done; float cull_z[2];
    float Rcurr[9], tcurr[3];
    // line 1
    float Rprev_inv[9], tprev[3];
    int cull_box[4];
    // line 2
    int rgb, rgbOnly, level_done2, pad;
    float krkInv[9], kt[3];
};
static_assert(sizeof(GnHot) == 192, "three 64-byte lines");

// Device-resident mirror of RGBDOdometry's members (Core/Utils/RGBDOdometry.h:78-137) plus the
// Gauss-Newton state that the reference keeps in host locals (RGBDOdometry.cpp:217-477).
struct OdomDev;
struct OdomDev {
    // pyramids (level 0..2)
    const float* vmap_curr[3];
    const float* nmap_curr[3];
    const float* vmap_g_prev[3];
    const float* nmap_g_prev[3];
    const float* lastDepth[3];
    const float* nextDepth[3];
    const uint8_t* lastImage[3];
    const uint8_t* nextImage[3];
    const uint8_t* lastNextImage[3];
    const int16_t* dIdx[3];
    const int16_t* dIdy[3];
    const float* cloud[3];
    cf_dataterm* corres[3];
    const uint8_t* cand[3];  // iteration-invariant part of RGBResidual's validity test
    // grouped accumulators: [kGroups][32] u64 each
    unsigned long long* icp_acc;
    unsigned long long* rgb_acc;
    float* err_surface;  // nullable; written on the last L0 iteration only
    // configuration
    cf_cam intr;
    int width, height;
    float distThres, angleThres, sobelScale, maxDepthDeltaRGB;
    float minGrad[3];
    float icpWeight;
    int icp, rgb, rgbOnly;
    // screen-box culling of the ICP reduction (cf_odom_set_culling): the bounding frustum of the model's predicted vertices (pixel
    // rectangle + depth interval in the prediction camera) is accumulated by the model-map pass into aabb_acc (6 order-preserving keys,
    // zero = empty), latched into box_lo / box_hi by the first kernel of the Gauss-Newton loop and re-projected into the current camera
    // after every pose update (screen_box)
    unsigned* aabb_acc;
    int cull;
    float box_R[9], box_t[3];   // pose of the camera the prediction was rendered from: the frame box_lo / box_hi are expressed in
    // culled trackers: first / last 256-pixel chunk with an RGB candidate per level (RgbPrepArgs::res_range; null: not tracked) --
    // the residual workgroups of the loop cover the record slots between them only
    unsigned* res_range;
    OdomDev* host_twin;          // nullable: the pinned host copy of this state; the LAST solve of a schedule publishes its result there
    // Gauss-Newton state
    float Rprev[9], tprev[3], Rprev_inv[9], Rcurr[9], tcurr[3];
    double resultRt[16];
    float krkInv[9], kt[3];
    float lastRGBError;
    int level_done;
    int solves;                  // Gauss-Newton solves this tracking call has run (every solve adds one; the host zeroes it): cf_odom_fetch_result
                                 // compares it with the launches it enqueued -- a solve that never ran (cf_set_gn_mode 2 on a part whose workgroups
                                 // do not land on XCD b mod 8: nobody draws the last ticket) is reported instead of returning a stale pose
    float residual[2];
    float box_lo[3], box_hi[3];  // bounding frustum of the model's predicted vertices in the prediction camera (box_R / box_t): level-0 pixel
                                 // rectangle [0..1] and depth interval [2] (box_lo[0] > box_hi[0]: no valid vertex)
    int res_seen[3];             // record slots between the first and the last RGB candidate of each level in this call (-1: not
                                 // tracked): sizes the residual part of the next call's launches (the last solve writes it)
    float cull_z[2];             // depth interval (current camera) of that box dilated by distThres: a 64-pixel run of the current frame
                                 // whose valid depths all lie outside cannot find a correspondence (-inf, +inf: no depth culling)
    float pose_inv[16];          // inverse of the tracked pose [Rcurr | tcurr] (row-major 4x4, inv44f), written by the LAST solve of a schedule: the index
                                 // pass enqueued before the host has seen the pose reads it here (cf_models_preindex; a launch of its own until round 6)
    // (the screen box itself -- the level-0 pixel rectangle outside of which no pixel of the current frame can find a correspondence
    // under Rcurr / tcurr -- is stats.cull_box)
    // outputs
    cf_track_stats stats;
    GnHot hot;   // (last: 64-byte aligned, inside the region the solve writes back)
};
__host__ __device__ inline void refresh_hot(OdomDev* od)
{
    GnHot& h = od->hot;
    h.icp = od->icp; h.level_done = od->level_done; h.cull_z[0] = od->cull_z[0]; h.cull_z[1] = od->cull_z[1];
    for (int k = 0; k < 9; k++) { h.Rcurr[k] = od->Rcurr[k]; h.Rprev_inv[k] = od->Rprev_inv[k]; h.krkInv[k] = od->krkInv[k]; }
    for (int k = 0; k < 3; k++) { h.tcurr[k] = od->tcurr[k]; h.tprev[k] = od->tprev[k]; h.kt[k] = od->kt[k]; }
    for (int k = 0; k < 4; k++) h.cull_box[k] = od->stats.cull_box[k];
    h.rgb = od->rgb; h.rgbOnly = od->rgbOnly; h.level_done2 = od->level_done; h.pad = 0;
}

// ---- prep launchers (track_prep.hip) ----
void launch_vmap(hipStream_t s, const float* depth, int cols, int rows, cf_cam intr, float cutoff, float* vmap);
void launch_nmap(hipStream_t s, const float* vmap, int cols, int rows, float* nmap);
void launch_copy_maps(hipStream_t s, const float* v4, const float* n4, int cols, int rows, float* vmap, float* nmap);
void launch_resize_map(hipStream_t s, const float* in, int in_cols, int in_rows, float* out, bool normalize);
void launch_transform_maps(hipStream_t s, float* vmap, float* nmap, int cols, int rows, const float R[9], const float t[3]);
void launch_vertices_to_depth(hipStream_t s, const float* v4, int cols, int rows, float cutoff, float* depth);
void launch_pyrdown_f32(hipStream_t s, const float* src, int scols, int srows, float* dst);
void launch_pyrdown_u8(hipStream_t s, const uint8_t* src, int scols, int srows, uint8_t* dst);
void launch_intensity(hipStream_t s, const uint8_t* rgba, int cols, int rows, uint8_t* dst);
void launch_rgb_expand(hipStream_t s, const uint8_t* rgb, int n, uint8_t* rgba);
void launch_sobel(hipStream_t s, const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy);
void launch_cloud(hipStream_t s, const float* depth, int cols, int rows, cf_cam il, float* cloud3);

// fused per-frame preparation used by the tracker object (all pyramid levels per launch)
struct Level3 { int cols[3], rows[3], blk_end[3]; };  // blk_end: cumulative workgroup counts per level
struct FrameMapsArgs {
    Level3 L;
    const float* depth[3]; float* vmap[3]; float* nmap[3];
    float fx_inv[3], fy_inv[3], cx[3], cy[3];
    float cutoff;
    float2* zrange[3];           // nullable: (min, max) valid depth of every run of 64 consecutive pixels (flat index / 64) of a level
};
struct RgbPrepArgs {
    Level3 L;
    const uint8_t* nextImage[3]; const float* nextDepth[3]; const float* lastDepth[3];
    int16_t* dIdx[3]; int16_t* dIdy[3]; uint8_t* cand[3]; float* cloud[3];
    float minScale[3], fx_inv[3], fy_inv[3], cx[3], cy[3];
    unsigned* res_range;         // nullable, [3][2]: per level ~(first chunk), last chunk + 1 of the 256-pixel chunks that hold a candidate
                                 // (atomicMax; zero = none; cleared by the last solve of the schedule)
};
struct ModelMapsArgs {
    const float* pred_v4; const float* pred_n4;  // RGBA32F prediction (vertex+conf, normal+radius)
    const float* alt_v4; const float* alt_n4;    // ... replaced by these when (float)sel[0] / (float)sel[1] < sel_ratio (sel nullable: no choice)
    const unsigned* sel; float sel_ratio;
    float* snapshot;                             // copy of pred_v4 kept by the tracker (RGBDOdometry::vmaps_tmp)
    float* vmap[3]; float* nmap[3];
    int cols, rows;
    float R[9], t[3];
    unsigned char* occ;                          // nullable: occupancy map of the prediction (1 byte per 4x4 block)
    unsigned* aabb;                              // nullable: bounding frustum of the valid level-0 vertices, camera frame (OdomDev::aabb_acc)
};
// batches: one grid row per tracked model (<= kPrepBatch), so that a frame with several models still issues each
// preparation kernel once
constexpr int kPrepBatch = 8;
struct ModelMapsBatch { ModelMapsArgs m[kPrepBatch]; };
struct RgbPrepBatch { RgbPrepArgs m[kPrepBatch]; };
struct RgbdChain {  // verticesToDepth + intensity + pyramids
    const float* v4; const uint8_t* rgba; float* depth[3]; uint8_t* image[3];
    const float* alt_v4; const uint8_t* alt_rgba; const unsigned* sel; float sel_ratio;  // the choice of ModelMapsArgs
};
struct RgbdBatch {
    RgbdChain c[2 * kPrepBatch];
    // Trackers that track the SAME frame (all models of a sequence) need the same intensity pyramid of it, each in its own buffers: one
    // chain computes it and stores every level to all of them (late in round 6; until then one chain -- a full-resolution grid row -- per
    // tracker).  fan_owner[t]: 1 + the chain that also writes tracker t's pyramid fan_image[level][t]; 0: nobody (its own chain does).
    uint8_t* fan_image[3][kPrepBatch];
    signed char fan_owner[kPrepBatch];
};
void launch_model_maps(hipStream_t s, const ModelMapsBatch& b, int n);  // needs cols % 4 == 0 && rows % 4 == 0
void launch_frame_maps(hipStream_t s, FrameMapsArgs a, int W, int H);
void launch_rgbd_pyramids(hipStream_t s, const RgbdBatch& b, int n_chains, int W, int H, float cutoff);
void launch_model_maps_and_pyramids(hipStream_t s, const ModelMapsBatch& mb, int n, const RgbdBatch& rb, int n_chains, int W, int H, float cutoff);
void launch_rgb_prep(hipStream_t s, RgbPrepBatch b, int n, int W, int H);

// ---- reduction launchers (track_reduce.hip) ----
// Division of a pixel index by the image width without the ~25-instruction integer-division sequence: for 0 <= n < 2^31 and d >= 2,
// n / d == umulhi(n, M) >> s with M = ceil(2^(31 + l) / d), s = l - 1, l = ceil(log2 d) (M * d - 2^(31+l) < d <= 2^l, so the error term
// n * (M d - 2^(31+l)) stays below 2^(31+l)).  Filled by the launchers from `cols`.
struct IDiv { unsigned M, s; };
inline IDiv make_idiv(int d)
{
    if (d < 2) return IDiv{0x80000000u, 0u};  // (d == 1 would need n itself; no image is one pixel wide -- gives n / 2, asserted against in the launchers)
    unsigned l = 0;
    while ((1u << l) < (unsigned)d) l++;
    return IDiv{(unsigned)((((unsigned long long)1 << (31 + l)) + (unsigned)d - 1) / (unsigned)d), l - 1};
}
// Gram form of the ICP sums (cf_set_icp_arith 1, cf_device.h): fraction bits of the fixed-point grid of row entry i (7 = the inlier flag)
constexpr int kGramBits[8] = {20, 20, 20, 17, 17, 17, 22, 0};
struct IcpLaunch { int threads; int ppt; int gram; };  // threads per workgroup, pixels per thread, rounding specification of the sums (cf_set_icp_arith)

// stand-alone steps (C-ABI parity with icpStep / computeRgbResidual / rgbStep / so3Step) run the same
// kernels on a scratch OdomDev prepared by cabi.cpp.
// ICP kernel arguments (by value, in the kernarg segment): per-model pointers + shared geometry
constexpr int kMaxBatch = 16;  // trackers per lock-step launch (grid.y): the models of one frame, or of several sequences' frames.
                               // Bounded by the 4 KB kernel-argument segment (IcpArgs + RgbArgs by value: 3.4 KB at 16)
struct IcpModelArgs {
    const float* vc; const float* nc;   // current-frame vertex / normal planes of this level
    const float* vp; const float* np;   // model prediction planes (global frame)
    const OdomDev* st;                  // device-resident pose + flags
    unsigned long long* acc;            // [kGroups][32] grouped accumulators
    float* err;                         // nullable ICP error surface [rows*cols]
    const unsigned char* occ;           // nullable: occupancy map of the model maps (model_maps_kernel), 1 byte per 4x4 level-0 block
    int row_begin, row_end;             // row band of THIS model's reduction (row_end == 0: all rows): its share when the model's
                                        // reduction is split over GPUs
    int cull;                           // workgroups / waves outside st->stats.cull_box leave before they load anything
    const float2* zr;                   // nullable: depth interval of every 64-pixel run of the current frame's level (FrameMapsArgs::zrange)
    int box_blocks;                     // > 0 (culled models): workgroups of this model in the launch, dealt the runs of its screen box
                                        // (box_blocks_for; the launcher drops it where the mapping does not apply); 0: one workgroup
                                        // per run of the whole image
};
constexpr int kNoBoxHint = 0x7fffffff;
// The 64-pixel runs of a level inside a level-0 screen box (OdomDev::stats.cull_box), for the host (sizing the launch) and the kernel
// (dealing them to waves) alike.  Image width a multiple of 64: the rectangle in units of runs -- run r is row y0 + r / nrx, columns
// (x0 + r % nrx) * 64 ...; otherwise (nrx == 0) the flat runs y0 + r that hold the rows of the box.
struct CullRuns { int x0, y0, nrx, total; };
__host__ __device__ inline CullRuns cull_runs(const int box[4], int L, int cols, int rows)
{
    int bx0 = (box[0] >> L) - 1, by0 = (box[1] >> L) - 1, bx1 = (box[2] >> L) + 1, by1 = (box[3] >> L) + 1;  // (as the per-wave test)
    bx0 = bx0 < 0 ? 0 : bx0; by0 = by0 < 0 ? 0 : by0; bx1 = bx1 > cols - 1 ? cols - 1 : bx1; by1 = by1 > rows - 1 ? rows - 1 : by1;
    if (bx0 > bx1 || by0 > by1) return CullRuns{0, 0, 0, 0};
    if ((cols & 63) == 0) { const int nrx = (bx1 >> 6) - (bx0 >> 6) + 1; return CullRuns{bx0 >> 6, by0, nrx, nrx * (by1 - by0 + 1)}; }
    const int q0 = (by0 * cols) >> 6, q1 = ((by1 + 1) * cols - 1) >> 6;
    return CullRuns{0, q0, 0, q1 - q0 + 1};
}
// Workgroups for a culled model at level L: HALF the runs of the level-0 screen box the model ended its previous tracking call with
// (box_hint; [0] == kNoBoxHint: none known -> 0 = the whole image's), + 25 % + two rows of runs -- every wave walks two runs, and on
// if the box has grown beyond that -- in multiples of 8 (slots start on XCD 0), at least 8.  Two runs per wave since the end of round 5:
// the culled trackers' workgroups finish long before the background's, so fewer of them (a shorter dispatch ramp for everybody) is
// worth their longer chains: 12.5 -> 12.1 us for the five-tracker launch; four runs per wave make them the tail (13.1 us).
inline int box_blocks_for(const int box_hint[4], int L, int cols, int rows, int threads)
{
    if (box_hint[0] == kNoBoxHint) return 0;
    const CullRuns cr = cull_runs(box_hint, L, cols, rows);
    const int wpb = threads / 64;
    const int runs = cr.total + cr.total / 4 + 2 * (cr.nrx > 0 ? cr.nrx : (cols + 63) / 64);
    int want = (((runs + 2 * wpb - 1) / (2 * wpb) + 7) / 8) * 8;
    return want < 8 ? 8 : want;
}
// Residual workgroups for a culled tracker at one level: the record slots between the first and the last RGB candidate of the previous
// tracking call (OdomDev::res_seen; < 0: unknown -> 0 = one workgroup per slot of the level) + 25 % + 2 -- the workgroups walk on by
// their number if this call's range is longer than that.
inline int residual_blocks_for(int seen) { return seen < 0 ? 0 : seen + seen / 4 + 2; }
// solve-kernel arguments (by value)
struct GnArgs {
    OdomDev* od[kMaxBatch];
    unsigned long long* icp_acc[kMaxBatch];
    unsigned long long* rgb_acc[kMaxBatch];
    OdomDev* od_host[kMaxBatch];   // nullable: pinned host copies of the states; the LAST solve of a schedule publishes its result there
    int slot_px;                   // pixels per record slot of the residual pass (RgbArgs::slot_px): unit of OdomDev::res_seen
    int icp_gram;                  // rounding specification of the ICP sums (IcpLaunch::gram): selects the scales of the unpack
};
// RGB residual / RGB step arguments (by value): everything but the pose-dependent state arrives in the kernarg
struct RgbModelArgs {
    OdomDev* st;
    const uint8_t* cand; const float* nextDepth; const float* lastDepth;
    const uint8_t* lastImage; const uint8_t* nextImage;
    cf_dataterm* corres; const float* cloud; const int16_t* dIdx; const int16_t* dIdy;
    unsigned long long* icp_acc; unsigned long long* rgb_acc;
    int no_counts;                      // the residual pass does not add its correspondence count / sigma to the accumulator: another
                                        // rank does, and the accumulators are summed over the ranks (split reduction)
    uint2* recs;                        // record slots of the device-resident loop (aliases corres: N x 8 B)
    unsigned* slot_counts;              // records per slot (behind the records in the same buffer)
    const unsigned* res_range;          // nullable (culled trackers): this level's pair of OdomDev::res_range -- only the record slots
                                        // between the first and the last candidate chunk are visited (the others hold no record)
    int res_blocks;                     // residual workgroups of this tracker in the launch (0: one per slot of the level); fewer than the
                                        // slots in range: the workgroups walk on by res_blocks
};
struct RgbArgs {
    RgbModelArgs m[kMaxBatch];
    int cols, rows;
    cf_cam il;                          // intrinsics of this level
    float sobelScale, maxDepthDelta;
    int compact;                        // residual pass writes record slots (recs) instead of the DataTerm image
    int slot_px;                        // pixels (= record capacity) per slot: 4 x the producer's workgroup size
    IDiv cdiv;                          // make_idiv(cols), set by the launchers
};
inline RgbModelArgs rgb_model_args(const OdomDev* h /* host mirror */, OdomDev* d_state, int level)
{
    return RgbModelArgs{d_state, h->cand[level], h->nextDepth[level], h->lastDepth[level], h->lastImage[level], h->nextImage[level],
                        h->corres[level], h->cloud[level], h->dIdx[level], h->dIdy[level], h->icp_acc, h->rgb_acc, 0,
                        reinterpret_cast<uint2*>(h->corres[level]),
                        reinterpret_cast<unsigned*>(reinterpret_cast<uint2*>(h->corres[level]) + (size_t)(h->width >> level) * (h->height >> level)),
                        h->res_range ? h->res_range + 2 * level : nullptr, 0};
}
constexpr int kMaxSlots = 2 * kMaxBatch;        // an ICP and a residual slot per model
struct IcpArgs {
    IcpModelArgs m[kMaxBatch];
    int cols, rows;
    cf_cam intr;                        // already divided by 2^level
    float distThres, angleThres;
    float angleSqLt, distSqLe;          // exact radicand bounds of the two gates (sqrt_gate_lt / sqrt_gate_le)
    int occ_w, occ_shift;               // occupancy row length (level-0 cols / 4) and 2 - level
    int flags;                          // bit0: write the error surface
    int row_begin, row_end;             // row band to reduce; row_end == 0: all rows
    IDiv cdiv;                          // make_idiv(cols), set by the launchers
    // layout of the one-dimensional grid (set by the launchers, icp_reduce_kernel): running totals of the workgroups per slot -- a slot
    // is the ICP reduction or the residual pass of one model -- and what every slot is; unused slots end at INT_MAX
    int slot_end[kMaxSlots];
    unsigned char slot_desc[kMaxSlots];       // model | kResidualSlot
    int slots_used;
};
constexpr unsigned char kResidualSlot = 0x80;   // the model's RGB residual pass
void launch_icp_level(hipStream_t s, IcpLaunch cfg, const IcpArgs& args, int n, int level, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
void launch_rgb_residual(hipStream_t s, const RgbArgs& ra, int n);
void launch_rgb_step(hipStream_t s, const RgbArgs& ra, int n);
// algorithmic bytes per pixel and model of the RGB residual pass: candidate mask 1 + next depth 4 + next
// intensity 1 + gathered last depth 4 + last intensity 1 + DataTerm record 16
constexpr uint64_t kRgbResidualBytes = 27;
// the compact-list flavour writes 8 B per VALID correspondence instead of 16 B per pixel; only its reads are counted
constexpr uint64_t kRgbResidualBytesCompact = 11;
void launch_acc_total(hipStream_t s, const unsigned long long* acc, unsigned long long* out);
void launch_rgb_cand(hipStream_t s, const int16_t* dIdx, const int16_t* dIdy, const float* next_depth,
                     const uint8_t* next_image, float min_scale, int cols, int rows, uint8_t* cand);
void launch_so3_step(hipStream_t s, const uint8_t* last_image, const uint8_t* next_image, const float basis[9],
                     const float kinv[9], const float krlr[9], int cols, int rows, unsigned long long* out16);

struct ProfSink {  // hipEvent pairs recorded around every ICP-reduce launch when enabled
    hipEvent_t* events; int capacity; int used; uint64_t bytes; uint64_t launches; int enabled;
};

// device-resident Gauss-Newton loop over `n` models (lock-step; blockIdx.y = model)
// cross-workgroup state of the SO3 pre-alignment (zero between launches): per-iteration totals + arrival counters
constexpr int kSo3Blocks = 16;
constexpr int kStepSubs = 16;   // first-level ticket counters of the RGB step (mode 2), one 128-byte line each
struct So3Sync { unsigned long long acc[10][16]; unsigned arrive, depart;
                 unsigned step_top, pad;                  // second-level ticket: first-level counters whose workgroups have all committed
                 unsigned step_sub[kStepSubs][32]; };     // [j][0]: committed workgroups among the slots = j (mod kStepSubs)
// mode 0: {ICP || residual -> DataTerm image} + rgb_step over the image + solve;
// mode 1: {ICP || residual -> per-workgroup record slots} + rgb step over the slots + solve
// mode 2: {ICP || residual -> record slots} + {rgb step over the slots, the tracker's workgroups on ONE XCD; the last to commit solves}
// hook: called after every {ICP || residual} launch for each model whose reduction is split over GPUs (split[m] != 0) with that
// model's ICP sums FOLDED to 32 words (acc_fold_kernel: group 0 of its accumulators): the caller's in-place SUM all-reduce over the ranks, enqueued on `s`
struct GnHook { int (*fn)(void* user, int op, void* dev_buf, uint64_t words, void* stream); void* user; int split[kMaxBatch]; };
// the trackers of a lock-step schedule: device states and the pinned host copies the first launch uploads them from
struct TrackerStates { OdomDev* dev[kMaxBatch]; const OdomDev* host[kMaxBatch]; };
bool launch_gn_track(hipStream_t s, IcpLaunch cfg, const TrackerStates& states, So3Sync* so3_syncs /* [n] */,
                     const GnHook* hook, const IcpArgs icp_args[3], const RgbArgs rgb_args[3], int n, int width, int height, bool so3,
                     bool pyramid, bool fast_odom, bool rgb, bool icp, int mode, ProfSink* prof,
                     OdomDev* const* h_states = nullptr /* [n] pinned host copies the last solve publishes to */,
                     const RgbPrepBatch* prep = nullptr /* n <= kPrepBatch models' RGB preparation (levels filled: rgb_prep_levels), run in the
                                                           first launch beside the SO3 pre-alignment */);
void rgb_prep_levels(RgbPrepBatch& b, int n, int W, int H);
float replay_icp_level0(hipStream_t s, IcpLaunch cfg, const IcpArgs& a0, const RgbArgs& r0, int n, int slots, int ablate, int reps, hipEvent_t e0, hipEvent_t e1);
#ifdef CF_ABLATE
void trace_so3_dump(hipStream_t s);
void trace_solve_begin();
void trace_solve_end(hipStream_t s, const char* path);
void trace_step_solve(hipStream_t s, IcpLaunch cfg, const IcpArgs& a0, const RgbArgs& r0, So3Sync* syncs, int n, const char* path);
void trace_icp_level0(hipStream_t s, IcpLaunch cfg, const IcpArgs& a0, const RgbArgs& r0, int n, int slots, const char* path);
#endif
bool probe_xcd_round_robin(hipStream_t s);   // does hardware workgroup b run on XCD b mod 8 (what cf_set_gn_mode 2 / the one-XCD meetings rely on)?
float sqrt_gate_lt(float T);  // smallest x with sqrtf(x) >= T
float sqrt_gate_le(float T);  // largest x with sqrtf(x) <= T

// ---- surfel launchers (surfel.hip) ----
// Every pass exists as a LOCK-STEP BATCH: one launch covers the models of a frame (or of several sequences' frames), kSurfBatch at a
// time -- the workgroups of the launch are dealt to the models (surfel.hip: BatchHdr).  The single-model launchers are batches of one.
constexpr int kSurfBatch = 16;   // models per launch (their arguments travel in the 4 KB kernel-argument segment)
struct IndexPassArgs {   // Model::predictIndices of one model: rasterise surfels [id_begin, id_end) into the z-keys, resolve
    const float* surfels; const unsigned* count; unsigned id_begin, id_end; float t_inv[16]; float maxDepth; int time, timeDelta;
    unsigned long long* keys; unsigned* index; float* vertConf; float* colorTime; float* normRad;
    const float* t_inv_dev;   // nullable: the inverse pose in device memory (OdomDev::pose_inv) instead of t_inv -- an index pass enqueued
                              // before the host has seen the tracked pose (cf_models_preindex)
    float* clean_rec;         // nullable: [rows * cols][8] -- what the clean pass stages per texel, packed by the resolve pass that feeds it:
    const float* clean_depth; // vertConf.xyzw | colorTime.z, colorTime.w, index, this filtered depth (SurfelCleanArgs::rec)
};
// pose.inverse() of a rigid transform: linear part by cofactors (the statement of the oracle, orc_surfel.c); host and device
__host__ __device__ inline void inv44f(const float a[16], float o[16])
{
    const float c00 = a[5] * a[10] - a[6] * a[9];
    const float c01 = a[6] * a[8] - a[4] * a[10];
    const float c02 = a[4] * a[9] - a[5] * a[8];
    const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const float id = 1.0f / det;
    float Li[9];
    Li[0] = c00 * id; Li[1] = (a[2] * a[9] - a[1] * a[10]) * id; Li[2] = (a[1] * a[6] - a[2] * a[5]) * id;
    Li[3] = c01 * id; Li[4] = (a[0] * a[10] - a[2] * a[8]) * id; Li[5] = (a[2] * a[4] - a[0] * a[6]) * id;
    Li[6] = c02 * id; Li[7] = (a[1] * a[8] - a[0] * a[9]) * id; Li[8] = (a[0] * a[5] - a[1] * a[4]) * id;
    for (int i = 0; i < 3; i++) {
        o[i * 4 + 0] = Li[i * 3 + 0]; o[i * 4 + 1] = Li[i * 3 + 1]; o[i * 4 + 2] = Li[i * 3 + 2];
        o[i * 4 + 3] = -(Li[i * 3 + 0] * a[3] + Li[i * 3 + 1] * a[7] + Li[i * 3 + 2] * a[11]);
    }
    o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}
// the tracked poses of n trackers (their device states, as the last solve left them), inverted into out[k][16]: what the index pass of a
// model needs of its pose, without the host
struct SplatPassArgs {   // ModelProjection::combinedPredict of one model
    const float* surfels; const unsigned* count; unsigned count_bound; float t_inv[16]; float maxDepth, confThreshold; int time, maxTime, timeDelta;
    const float* rays; unsigned long long* keys; uint8_t* image; float* vertexConf; float* normalRad; uint16_t* time16;
};
struct UpdatePassArgs { const float* in; const unsigned* count; unsigned count_bound; unsigned* owner; const float* records; int time; float* out; };
struct SurfelFuseArgs {
    const unsigned* index; const float* vertConf; const float* normRad;
    const uint8_t* rgba; const float* depth_raw; const float* depth_filt; const uint8_t* mask;
    const float* tcx; const float* tcy;
    float pose[16]; cf_cam cam; float inv_fx, inv_fy;
    int cols, rows, time; float weighting; int maskID; float maxDepth;
    float* records; unsigned* new_flags; unsigned* owner;
    int flags_clean = 0;  // 1: new_flags is known to hold zeros (left so by the last compaction): launch_associate_batch does not clear it
};
struct SurfelCleanArgs {
    const unsigned* index; const float* vertConf; const float* colorTime;
    const float* depth_filt; const uint8_t* mask;
    const float* rec;   // nullable: the packed records of the index pass in front of this call (IndexPassArgs::clean_rec, written with depth_filt)
    float t_inv[16]; cf_cam cam; int cols, rows, time; float confThreshold, outlierCoeff; int timeDelta, maskID;
};
struct CleanPassArgs {   // Model::clean of one model
    SurfelCleanArgs h; const float* surfels; const unsigned* count; const float* fresh; const unsigned* n_fresh; unsigned total_bound;
    float* staged; unsigned* flags;
};
struct ScanPassArgs {    // one ordered compaction (transform feedback): the flagged 48 B records of rec [n] to out, their number (+ add) to *total
    const float* rec; const unsigned* flags; long long n; unsigned* block_sums; unsigned* total; unsigned add_to_total; float* out; unsigned* total_host;
    int zero_flags = 0;   // 1: the pass leaves the flags it has read cleared (the association's new-surfel flags: the next frame's pass finds zeros, no fill launch)
};
void launch_index_keys_batch(hipStream_t s, const IndexPassArgs* items, int n, cf_cam cam, int cols, int rows);
bool launch_update_compaction_index_keys(hipStream_t s, const UpdatePassArgs* up, const ScanPassArgs* sc, const IndexPassArgs* ix, int n, cf_cam cam, int cols,
                                         int rows);   // the three stages as two launches (surfel.hip); false = not enqueued, use the separate launches
void launch_index_resolve_batch(hipStream_t s, const IndexPassArgs* items, int n, cf_cam cam, int cols, int rows);
void launch_combined_predict_batch(hipStream_t s, const SplatPassArgs* items, int n, cf_cam cam, int cols, int rows);
void launch_associate_batch(hipStream_t s, const SurfelFuseArgs* items, int n);   // zeroes the new_flags of every model that is not flags_clean first
void launch_update_batch(hipStream_t s, const UpdatePassArgs* items, int n);
void launch_clean_batch(hipStream_t s, const CleanPassArgs* items, int n);
void launch_scan_scatter_batch(hipStream_t s, const ScanPassArgs* items, int n);
void launch_bilateral(hipStream_t s, const float* depth, int cols, int rows, float maxD, float* out);
void launch_scan_scatter(hipStream_t s, const float* rec, const unsigned* flags, long long n, unsigned* block_sums, unsigned* total,
                         unsigned add_to_total, float* out, unsigned* total_host = nullptr /* pinned mirror of *total */);
void launch_feedback(hipStream_t s, const uint8_t* rgba, const float* depth, int cols, int rows, cf_cam cam, float inv_fx, float inv_fy,
                     const float* tcx, const float* tcy, int time, float maxDepth, float* rec, unsigned* flags);
void launch_init(hipStream_t s, const float* raw, const float* filt, const unsigned* raw_count, long long max_n, float* out);
void launch_index_keys(hipStream_t s, const float* surfels, const unsigned* count, unsigned id_begin, unsigned id_end, const float t_inv[16],
                       cf_cam cam, int cols, int rows, float maxDepth, int time, int timeDelta, unsigned long long* keys);
void launch_index_resolve(hipStream_t s, const float* surfels, const float t_inv[16], int cols, int rows, unsigned long long* keys,
                          unsigned* index, float* vertConf, float* colorTime, float* normRad);
void launch_predict_indices(hipStream_t s, const float* surfels, const unsigned* count, unsigned count_bound, const float t_inv[16], cf_cam cam,
                            int cols, int rows, float maxDepth, int time, int timeDelta, unsigned long long* keys, unsigned* index,
                            float* vertConf, float* colorTime, float* normRad);
void launch_splat_rays(hipStream_t s, cf_cam cam, int cols, int rows, float* rays /* [rows*cols*4] */);
void launch_combined_predict(hipStream_t s, const float* surfels, const unsigned* count, unsigned count_bound, const float t_inv[16], cf_cam cam,
                             int cols, int rows, float maxDepth, float confThreshold, int time, int maxTime, int timeDelta,
                             const float* rays, unsigned long long* keys, uint8_t* image, float* vertexConf, float* normalRad, uint16_t* time16);
void launch_fill_in(hipStream_t s, const float* pv, const float* pn, const uint8_t* pimg, const float* depth, const uint8_t* rgba, int cols,
                    int rows, cf_cam cam, float inv_fx, float inv_fy, int pass_geom, int pass_rgb, float* ov, float* on, uint8_t* oi);
void launch_fill_ratio(hipStream_t s, const uint8_t* pimg, int cols, int rows, unsigned* out2, unsigned* out2_host = nullptr /* pinned */);
void launch_associate(hipStream_t s, const SurfelFuseArgs& h);
void launch_update(hipStream_t s, const float* in, const unsigned* count, unsigned count_bound, unsigned* owner, const float* records, int time,
                   float* out);
void launch_clean(hipStream_t s, const float* surfels, const unsigned* count, const float* fresh, const unsigned* n_fresh, unsigned total_bound,
                  const SurfelCleanArgs& h, float* staged, unsigned* flags);
void launch_set_count(hipStream_t s, unsigned* out, unsigned v);

}  // namespace cf
