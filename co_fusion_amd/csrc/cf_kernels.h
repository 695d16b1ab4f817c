// cf_kernels.h -- internal launcher declarations + device-resident tracker state.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cofusion_hip.h"

namespace cf {

constexpr int kGroups = 64;  // accumulation groups for the grouped integer atomics

// Device-resident mirror of RGBDOdometry's members (Core/Utils/RGBDOdometry.h:78-137) plus the
// Gauss-Newton state that the reference keeps in host locals (RGBDOdometry.cpp:217-477).
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
    // Gauss-Newton state
    float Rprev[9], tprev[3], Rprev_inv[9], Rcurr[9], tcurr[3];
    double resultRt[16];
    float krkInv[9], kt[3];
    float lastRGBError;
    int level_done;
    float residual[2];
    // outputs
    cf_track_stats stats;
};

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
void launch_sobel(hipStream_t s, const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy);
void launch_cloud(hipStream_t s, const float* depth, int cols, int rows, cf_cam il, float* cloud3);

// ---- reduction launchers (track_reduce.hip) ----
struct IcpLaunch { int threads; int ppt; };  // threads per workgroup, pixels per thread

// stand-alone steps (C-ABI parity with icpStep / computeRgbResidual / rgbStep / so3Step) run the same
// kernels on a scratch OdomDev prepared by cabi.cpp.
void launch_icp_models(hipStream_t s, IcpLaunch cfg, OdomDev* const* d_models, int n, int width, int height, int level,
                       int write_err);
void launch_rgb_residual_models(hipStream_t s, OdomDev* const* d_models, int n, int width, int height, int level);
void launch_rgb_step_models(hipStream_t s, OdomDev* const* d_models, int n, int width, int height, int level);
void launch_acc_total(hipStream_t s, const unsigned long long* acc, unsigned long long* out);
void launch_rgb_cand(hipStream_t s, const int16_t* dIdx, const int16_t* dIdy, const float* next_depth,
                     const uint8_t* next_image, float min_scale, int cols, int rows, uint8_t* cand);
void launch_so3_step(hipStream_t s, const uint8_t* last_image, const uint8_t* next_image, const float basis[9],
                     const float kinv[9], const float krlr[9], int cols, int rows, unsigned long long* out16);

struct ProfSink {  // hipEvent pairs recorded around every ICP-reduce launch when enabled
    hipEvent_t* events; int capacity; int used; uint64_t bytes; uint64_t launches; int enabled;
};

// device-resident Gauss-Newton loop over `n` models (lock-step; blockIdx.y = model)
void launch_gn_track(hipStream_t s, IcpLaunch cfg, OdomDev* const* d_models /* device array of n pointers */, int n,
                     int width, int height, bool so3, bool pyramid, bool fast_odom, bool rgb, bool icp, ProfSink* prof);

}  // namespace cf
