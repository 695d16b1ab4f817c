// cabi.hip -- C-ABI (include/cofusion_hip.h) over the gfx950 kernels: context, tracker objects,
// stand-alone reduction steps.  No allocation happens on the per-frame path: every device buffer
// is created with the ctx / odom / model object that owns it.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "cf_host.h"
#include "gn_ref_host.h"

using namespace cf;

#define HIPCHK(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->set_error(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return CF_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

#define LAUNCHCHK(ctx) HIPCHK(ctx, hipGetLastError())

void cf_ctx::set_error(const std::string& m)
{
    std::lock_guard<std::mutex> lk(error_mutex);
    last_error = m;
}

template <typename T>
static int dmalloc(cf_ctx* ctx, T** p, size_t count)
{
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    HIPCHK(ctx, hipMemsetAsync(*p, 0, count * sizeof(T), ctx->stream));
    return CF_OK;
}

extern "C" {

int cf_create(const cf_config* cfg, cf_ctx** out)
{
    if (!cfg || !out || cfg->width <= 0 || cfg->height <= 0 || (cfg->width % 16) || (cfg->height % 4)) return CF_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CF_EHIP;  // fail loudly: no CPU fallback
    cf_ctx* ctx = new cf_ctx();
    ctx->cfg = *cfg;
    if (ctx->cfg.max_models <= 0) ctx->cfg.max_models = 1;
    *out = ctx;
    HIPCHK(ctx, hipSetDevice(cfg->device));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    ctx->icp_launch = IcpLaunch{256, 0, 0};   // (0 pixels per lane: launch_icp_rgbres chooses by level)
    if (int r = dmalloc(ctx, &ctx->d_acc_a, (size_t)kGroups * 32)) return r;
    if (int r = dmalloc(ctx, &ctx->d_acc_b, (size_t)kGroups * 32)) return r;
    if (int r = dmalloc(ctx, &ctx->d_out, 64)) return r;
    if (int r = dmalloc(ctx, &ctx->d_scratch_state, 1)) return r;
    if (int r = dmalloc(ctx, &ctx->d_state_pool, (size_t)cf_ctx::kStateSlots)) return r;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_state_pool), sizeof(OdomDev) * cf_ctx::kStateSlots, hipHostMallocCoherent));  // the last solve stores into it
    memset(ctx->h_state_pool, 0, sizeof(OdomDev) * cf_ctx::kStateSlots);
    if (int r = dmalloc(ctx, &ctx->d_so3_sync, (size_t)ctx->cfg.max_models + 1)) return r;
    // the ONE environment switch of the library: CF_ICP_ARITH = "product" (default) / "gram" / "reference", the rounding specification of the tracker's sums
    // for every context of the process (cf_set_icp_arith sets it per context; launch shape and data path have setters only)
    if (const char* e = getenv("CF_ICP_ARITH")) {
        if (cf_set_icp_arith(ctx, (!strcmp(e, "gram") || !strcmp(e, "1")) ? CF_ICP_ARITH_GRAM : (!strcmp(e, "reference") || !strcmp(e, "2")) ? CF_ICP_ARITH_REFERENCE
                                  : (!strcmp(e, "product") || !strcmp(e, "0")) ? CF_ICP_ARITH_PRODUCT : -1) != CF_OK) {
            ctx->set_error("CF_ICP_ARITH: product | gram | reference"); return CF_EINVAL;
        }
    }
    if (int r = dmalloc(ctx, &ctx->d_cand_scratch, (size_t)cfg->width * cfg->height)) return r;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_scratch_state), sizeof(OdomDev)));
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_out), sizeof(unsigned long long) * 64));
    ctx->prof.capacity = 8192;
    ctx->prof.events = new hipEvent_t[ctx->prof.capacity];
    for (int i = 0; i < ctx->prof.capacity; i++) HIPCHK(ctx, hipEventCreate(&ctx->prof.events[i]));
    ctx->prof.used = 0; ctx->prof.enabled = 0; ctx->prof.bytes = 0; ctx->prof.launches = 0;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

void cf_destroy(cf_ctx* ctx)
{
    if (ctx && ctx->batch_event) (void)hipEventDestroy(ctx->batch_event);
    if (ctx) for (int i = 0; i < cf_ctx::kSurfEvents; i++) if (ctx->surf_events[i]) (void)hipEventDestroy(ctx->surf_events[i]);
    if (!ctx) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)cf_rccl_destroy(ctx);
    (void)hipFree(ctx->d_acc_a); (void)hipFree(ctx->d_acc_b); (void)hipFree(ctx->d_out);
    (void)hipFree(ctx->d_scratch_state); (void)hipFree(ctx->d_so3_sync); (void)hipFree(ctx->d_cand_scratch);
    (void)hipFree(ctx->d_state_pool); (void)hipHostFree(ctx->h_state_pool);
    (void)hipHostFree(ctx->h_scratch_state); (void)hipHostFree(ctx->h_out);
    if (ctx->d_ref) (void)hipFree(ctx->d_ref);
    if (ctx->h_ref) (void)hipHostFree(ctx->h_ref);
    if (ctx->prof.events) {
        for (int i = 0; i < ctx->prof.capacity; i++) (void)hipEventDestroy(ctx->prof.events[i]);
        delete[] ctx->prof.events;
    }
    for (int i = 0; i < cf_ctx::kLanes; i++) {
        if (ctx->lanes[i]) (void)hipStreamDestroy(ctx->lanes[i]);
        if (ctx->lane_done[i]) (void)hipEventDestroy(ctx->lane_done[i]);
    }
    if (ctx->fork_point) (void)hipEventDestroy(ctx->fork_point);
    for (int i = 0; i < cf_ctx::kMarks; i++) if (ctx->marks[i]) (void)hipEventDestroy(ctx->marks[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* cf_last_error(const cf_ctx* ctx)
{
    if (!ctx) return "null ctx";
    // helper threads bound with cf_thread_lane may fail (and rewrite last_error) at the same time: hand out a per-thread copy taken
    // under the lock, valid until the calling thread's next cf_last_error
    static thread_local std::string copy;
    {
        std::lock_guard<std::mutex> lk(const_cast<cf_ctx*>(ctx)->error_mutex);
        copy = ctx->last_error;
    }
    return copy.c_str();
}
// Switching streams drains the old one first: allocations zero-fill asynchronously on the stream that was current when they
// were made, and nothing else would order that before the first use on the new stream.
int cf_set_stream(cf_ctx* ctx, void* s)  // NULL = the legacy default stream
{
    if (!ctx || ctx->forked) return CF_EINVAL;
    if (ctx->stream != (hipStream_t)s) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)s;
    return CF_OK;
}
int cf_use_own_stream(cf_ctx* ctx)
{
    if (!ctx || ctx->forked) return CF_EINVAL;
    if (ctx->stream != ctx->own_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return CF_OK;
}

// Independent pieces of one frame (the surfel passes of different models) may overlap on the GPU: cf_fork(lane) routes
// the following calls to auxiliary stream `lane`, ordered after everything enqueued on the context's stream at the first
// fork; cf_join returns to that stream and orders it after all lanes used since.
extern "C++" { thread_local cf_thread_binding cf_tls_binding; }

int cf_fork(cf_ctx* ctx, int lane)
{
    if (!ctx || lane < 0) return CF_EINVAL;
    lane %= cf_ctx::kLanes;
    if (!ctx->lanes[lane]) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->lanes[lane], hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->lane_done[lane], hipEventDisableTiming));
    }
    if (!ctx->fork_point) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->fork_point, hipEventDisableTiming));
    if (!ctx->forked) {  // leaving the main stream: everything enqueued on it so far precedes the lane's work
        ctx->forked_from = ctx->stream;
        HIPCHK(ctx, hipEventRecord(ctx->fork_point, ctx->forked_from));
        ctx->forked = true;
    }
    HIPCHK(ctx, hipStreamWaitEvent(ctx->lanes[lane], ctx->fork_point, 0));
    ctx->lanes_used |= 1u << lane;
    ctx->stream = ctx->lanes[lane];
    return CF_OK;
}
// cf_mark(slot) remembers the current point of the stream; cf_fork_after(lane, slot) routes the following calls to `lane`
// ordered after that point ONLY (slot < 0: after nothing) -- for work that does not depend on what the main stream still has
// queued, e.g. filtering the next frame while the previous frame's fusion passes run.  cf_join orders the main stream after it.
int cf_mark(cf_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= cf_ctx::kMarks) return CF_EINVAL;
    if (!ctx->marks[slot]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->marks[slot], hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->marks[slot], ctx->stream));
    ctx->mark_set[slot] = true;
    return CF_OK;
}
// host wait for a mark (e.g. before re-using a pinned staging buffer whose transfer was enqueued before the mark)
int cf_event_wait_host(cf_ctx* ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= cf_ctx::kMarks) return CF_EINVAL;
    if (ctx->mark_set[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->marks[slot]));
    return CF_OK;
}
int cf_fork_after(cf_ctx* ctx, int lane, int slot)
{
    if (!ctx || lane < 0 || slot >= cf_ctx::kMarks || ctx->forked) return CF_EINVAL;
    lane %= cf_ctx::kLanes;
    if (!ctx->lanes[lane]) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->lanes[lane], hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->lane_done[lane], hipEventDisableTiming));
    }
    if (slot >= 0 && ctx->mark_set[slot]) HIPCHK(ctx, hipStreamWaitEvent(ctx->lanes[lane], ctx->marks[slot], 0));
    ctx->forked_from = ctx->stream;
    ctx->forked = true;
    ctx->lanes_used |= 1u << lane;
    ctx->stream = ctx->lanes[lane];
    return CF_OK;
}
// back to the main stream WITHOUT waiting for the lanes (their work keeps running beside what follows); cf_join waits
int cf_main(cf_ctx* ctx)
{
    if (!ctx) return CF_EINVAL;
    if (ctx->forked) { ctx->stream = ctx->forked_from; ctx->forked = false; }
    return CF_OK;
}
int cf_join(cf_ctx* ctx)
{
    if (!ctx) return CF_EINVAL;
    if (ctx->forked) { ctx->stream = ctx->forked_from; ctx->forked = false; }
    for (int lane = 0; lane < cf_ctx::kLanes; lane++)
        if (ctx->lanes_used & (1u << lane)) {
            HIPCHK(ctx, hipEventRecord(ctx->lane_done[lane], ctx->lanes[lane]));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_done[lane], 0));
        }
    ctx->lanes_used = 0;
    return CF_OK;
}
// ... for one lane only: the main stream waits for what that lane holds, the other lanes keep running beside it
int cf_join_lane(cf_ctx* ctx, int lane)
{
    if (!ctx || lane < 0) return CF_EINVAL;
    lane %= cf_ctx::kLanes;
    if (ctx->forked) { ctx->stream = ctx->forked_from; ctx->forked = false; }
    if (ctx->lanes_used & (1u << lane)) {
        HIPCHK(ctx, hipEventRecord(ctx->lane_done[lane], ctx->lanes[lane]));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->lane_done[lane], 0));
        ctx->lanes_used &= ~(1u << lane);
    }
    return CF_OK;
}
// Bind the calling thread to lane `lane` of the context (lane < 0: unbind).  The owning thread forks the lane first (cf_fork, then
// cf_main), which orders the lane after the stream and books it for the next cf_join; the bound thread's model calls then go to
// the lane without touching the context's current stream.
int cf_thread_lane(cf_ctx* ctx, int lane)
{
    if (!ctx) return CF_EINVAL;
    if (lane < 0) { cf_tls_binding = cf_thread_binding{}; return CF_OK; }
    lane %= cf_ctx::kLanes;
    if (!ctx->lanes[lane] || !(ctx->lanes_used & (1u << lane))) { ctx->set_error("cf_thread_lane: lane was not forked"); return CF_ESTATE; }
    HIPCHK(ctx, hipSetDevice(ctx->cfg.device));
    cf_tls_binding.ctx = ctx;
    cf_tls_binding.stream = ctx->lanes[lane];
    return CF_OK;
}
void* cf_get_stream(cf_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int cf_synchronize(cf_ctx* ctx) { if (!ctx) return CF_EINVAL; HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return CF_OK; }
int cf_malloc(cf_ctx* ctx, uint64_t bytes, void** dptr)
{
    if (!ctx || !dptr) return CF_EINVAL;
    HIPCHK(ctx, hipMalloc(dptr, bytes));
    HIPCHK(ctx, hipMemsetAsync(*dptr, 0, bytes, ctx->stream));
    return CF_OK;
}
int cf_free(cf_ctx* ctx, void* dptr) { if (!ctx) return CF_EINVAL; HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(dptr)); return CF_OK; }
int cf_memcpy_h2d(cf_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx) return CF_EINVAL;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}
int cf_memcpy_d2h(cf_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx) return CF_EINVAL;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

// pinned host staging + asynchronous upload: the host-input path of the facade copies a frame into pinned memory and
// enqueues the transfer without waiting for it
int cf_malloc_host(cf_ctx* ctx, uint64_t bytes, void** hptr)
{
    if (!ctx || !hptr) return CF_EINVAL;
    HIPCHK(ctx, hipHostMalloc(hptr, bytes));
    return CF_OK;
}
int cf_free_host(cf_ctx* ctx, void* hptr) { if (!ctx) return CF_EINVAL; HIPCHK(ctx, hipHostFree(hptr)); return CF_OK; }
int cf_memcpy_h2d_async(cf_ctx* ctx, void* dst, const void* src_pinned, uint64_t bytes)
{
    if (!ctx) return CF_EINVAL;
    HIPCHK(ctx, hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CF_OK;
}
int cf_memcpy_d2h_async(cf_ctx* ctx, void* dst_pinned, const void* src, uint64_t bytes)
{
    if (!ctx) return CF_EINVAL;
    HIPCHK(ctx, hipMemcpyAsync(dst_pinned, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return CF_OK;
}
int cf_memcpy_d2d_async(cf_ctx* ctx, void* dst, const void* src, uint64_t bytes)
{
    if (!ctx) return CF_EINVAL;
    HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CF_OK;
}
int cf_rgb_to_rgba(cf_ctx* ctx, const uint8_t* rgb_dev, int cols, int rows, uint8_t* rgba_dev)
{
    if (!ctx || !rgb_dev || !rgba_dev || cols <= 0 || rows <= 0) return CF_EINVAL;
    launch_rgb_expand(ctx->stream, rgb_dev, cols * rows, rgba_dev);
    LAUNCHCHK(ctx);
    return CF_OK;
}

int cf_set_icp_launch(cf_ctx* ctx, int threads, int ppt)
{
    if (!ctx || (threads != 64 && threads != 128 && threads != 256 && threads != 512 && threads != 1024) ||
        (ppt != 0 && ppt != 1 && ppt != 2 && ppt != 4))
        return CF_EINVAL;
    ctx->icp_launch = IcpLaunch{threads, ppt, ctx->icp_launch.gram};
    return CF_OK;
}

int cf_set_icp_arith(cf_ctx* ctx, int mode)
{
    if (!ctx || (mode != CF_ICP_ARITH_PRODUCT && mode != CF_ICP_ARITH_GRAM && mode != CF_ICP_ARITH_REFERENCE)) return CF_EINVAL;
    ctx->icp_arith = mode;
    ctx->icp_launch.gram = mode == CF_ICP_ARITH_GRAM ? 1 : 0;
    return CF_OK;
}
int cf_get_icp_arith(cf_ctx* ctx) { return ctx ? ctx->icp_arith : CF_EINVAL; }

int cf_profile_enable(cf_ctx* ctx, int on) { if (!ctx || on < 0) return CF_EINVAL; ctx->prof.enabled = on; ctx->prof_calls = 0; return CF_OK; }
int cf_profile_read(cf_ctx* ctx, cf_profile* out, int reset)
{
    if (!ctx || !out) return CF_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double ms = ctx->prof_ms_accum;
    for (int i = 0; i + 1 < ctx->prof.used; i += 2) {
        float t = 0;
        HIPCHK(ctx, hipEventElapsedTime(&t, ctx->prof.events[i], ctx->prof.events[i + 1]));
        ms += t;
    }
    ctx->prof_ms_accum = ms;
    ctx->prof.used = 0;
    out->icp_ms_total = ms; out->icp_launches = ctx->prof.launches; out->icp_bytes = ctx->prof.bytes;
    for (int i = 0; i + 1 < ctx->surf_used; i += 2) {
        float t = 0;
        HIPCHK(ctx, hipEventElapsedTime(&t, ctx->surf_events[i], ctx->surf_events[i + 1]));
        ctx->surf_ms_accum += t;
    }
    ctx->surf_used = 0;
    out->surfel_ms_total = ctx->surf_ms_accum; out->surfel_calls = ctx->surf_calls; out->surfel_bytes = ctx->surf_bytes;
    if (reset) { ctx->prof_ms_accum = 0; ctx->prof.launches = 0; ctx->prof.bytes = 0; ctx->surf_ms_accum = 0; ctx->surf_calls = 0; ctx->surf_bytes = 0; }
    return CF_OK;
}

// ---------------------------------------------------------------- map preparation ----
#define PREP_PROLOGUE if (!ctx) return CF_EINVAL
int cf_create_vmap(cf_ctx* ctx, const float* depth, int cols, int rows, cf_cam intr, float cutoff, float* vmap)
{ PREP_PROLOGUE; launch_vmap(ctx->stream, depth, cols, rows, intr, cutoff, vmap); LAUNCHCHK(ctx); return CF_OK; }
int cf_create_nmap(cf_ctx* ctx, const float* vmap, int cols, int rows, float* nmap)
{ PREP_PROLOGUE; launch_nmap(ctx->stream, vmap, cols, rows, nmap); LAUNCHCHK(ctx); return CF_OK; }
int cf_copy_maps(cf_ctx* ctx, const float* v4, const float* n4, int cols, int rows, float* vmap, float* nmap)
{ PREP_PROLOGUE; launch_copy_maps(ctx->stream, v4, n4, cols, rows, vmap, nmap); LAUNCHCHK(ctx); return CF_OK; }
int cf_resize_map(cf_ctx* ctx, const float* in, int in_cols, int in_rows, float* out, int normalize)
{ PREP_PROLOGUE; launch_resize_map(ctx->stream, in, in_cols, in_rows, out, normalize != 0); LAUNCHCHK(ctx); return CF_OK; }
int cf_transform_maps(cf_ctx* ctx, float* vmap, float* nmap, int cols, int rows, const float R[9], const float t[3])
{ PREP_PROLOGUE; launch_transform_maps(ctx->stream, vmap, nmap, cols, rows, R, t); LAUNCHCHK(ctx); return CF_OK; }
int cf_vertices_to_depth(cf_ctx* ctx, const float* v4, int cols, int rows, float cutoff, float* depth)
{ PREP_PROLOGUE; launch_vertices_to_depth(ctx->stream, v4, cols, rows, cutoff, depth); LAUNCHCHK(ctx); return CF_OK; }
int cf_pyrdown_gauss_f32(cf_ctx* ctx, const float* src, int scols, int srows, float* dst)
{ PREP_PROLOGUE; launch_pyrdown_f32(ctx->stream, src, scols, srows, dst); LAUNCHCHK(ctx); return CF_OK; }
int cf_pyrdown_gauss_u8(cf_ctx* ctx, const uint8_t* src, int scols, int srows, uint8_t* dst)
{ PREP_PROLOGUE; launch_pyrdown_u8(ctx->stream, src, scols, srows, dst); LAUNCHCHK(ctx); return CF_OK; }
int cf_rgba_to_intensity(cf_ctx* ctx, const uint8_t* rgba, int cols, int rows, uint8_t* dst)
{ PREP_PROLOGUE; launch_intensity(ctx->stream, rgba, cols, rows, dst); LAUNCHCHK(ctx); return CF_OK; }
int cf_sobel(cf_ctx* ctx, const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy)
{ PREP_PROLOGUE; launch_sobel(ctx->stream, src, cols, rows, dx, dy); LAUNCHCHK(ctx); return CF_OK; }
int cf_project_cloud(cf_ctx* ctx, const float* depth, int cols, int rows, cf_cam il, float* cloud3)
{ PREP_PROLOGUE; launch_cloud(ctx->stream, depth, cols, rows, il, cloud3); LAUNCHCHK(ctx); return CF_OK; }
int cf_depth_pyramid(cf_ctx* ctx, const float* depth_filtered, int cols, int rows, float* l1, float* l2)
{
    PREP_PROLOGUE;
    launch_pyrdown_f32(ctx->stream, depth_filtered, cols, rows, l1);
    launch_pyrdown_f32(ctx->stream, l1, cols / 2, rows / 2, l2);
    LAUNCHCHK(ctx);
    return CF_OK;
}

// ------------------------------------------------------------ stand-alone reductions ----
static void se3_unpack_host(const unsigned long long* t, int F, float* A, float* b, float* residual)
{  // reduce.cu:481-498;  F < 0: the Gram form of the ICP sums, word (i, j) carries kGramBits[i] + kGramBits[j] fraction bits
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = (float)ldexp((double)(long long)t[shift++], -(F >= 0 ? F : kGramBits[i] + kGramBits[j]));
            if (j == 6) { if (b) b[i] = value; }
            else if (A) A[j * 6 + i] = A[i * 6 + j] = value;
        }
    if (residual) { residual[0] = (float)ldexp((double)(long long)t[27], -(F >= 0 ? F : 2 * kGramBits[6])); residual[1] = (float)(long long)t[28]; }
}

// Runs one kernel family on the ctx scratch state (single model, level 0 geometry = cols x rows).
static int scratch_begin(cf_ctx* ctx, int cols, int rows)
{
    OdomDev* h = ctx->h_scratch_state;
    memset(h, 0, sizeof(*h));
    h->width = cols; h->height = rows;
    h->icp_acc = ctx->d_acc_a; h->rgb_acc = ctx->d_acc_b;
    h->icp = 1; h->rgb = 1;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_acc_a, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_acc_b, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
    return CF_OK;
}
static int scratch_commit(cf_ctx* ctx)
{
    refresh_hot(ctx->h_scratch_state);   // the kernels read pose / flags through the hot block (cf_kernels.h: GnHot)
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_scratch_state, ctx->h_scratch_state, sizeof(OdomDev), hipMemcpyHostToDevice, ctx->stream));
    return CF_OK;
}
static int fetch_totals(cf_ctx* ctx, const unsigned long long* acc, int words)
{
    launch_acc_total(ctx->stream, acc, ctx->d_out);
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

int cf_icp_step(cf_ctx* ctx, const float Rcurr[9], const float tcurr[3], const float* vmap_curr, const float* nmap_curr,
                const float Rprev_inv[9], const float tprev[3], cf_cam intr, const float* vmap_g_prev,
                const float* nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows, float* A_host,
                float* b_host, float* residual_host, int64_t* sums_host, float* err_surface)
{
    return cf_icp_step_band(ctx, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres,
                            cols, rows, 0, rows, A_host, b_host, residual_host, sums_host, err_surface);
}

int cf_icp_step_band(cf_ctx* ctx, const float Rcurr[9], const float tcurr[3], const float* vmap_curr, const float* nmap_curr,
                     const float Rprev_inv[9], const float tprev[3], cf_cam intr, const float* vmap_g_prev,
                     const float* nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows, int row_begin, int row_end,
                     float* A_host, float* b_host, float* residual_host, int64_t* sums_host, float* err_surface)
{
    if (!ctx || !vmap_curr || !nmap_curr || !vmap_g_prev || !nmap_g_prev || (cols % 4)) return CF_EINVAL;
    if (row_begin < 0 || row_end > rows || row_begin >= row_end) return CF_EINVAL;
    if (ctx->icp_arith == CF_ICP_ARITH_REFERENCE) {   // the reference's own f32 tree (track_ref.hip): no fixed-point sums, no row bands
        if (row_begin != 0 || row_end != rows) { ctx->set_error("cf_icp_step_band: the reference-order arithmetic has no row bands (a band is another launch shape)"); return CF_EINVAL; }
        if (sums_host) memset(sums_host, 0, sizeof(int64_t) * 32);   // (no fixed-point sums in this mode)
        float out29[29];
        if (int r = ref_icp_step(ctx, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres, cols, rows, err_surface, out29)) return r;
        float A[36], b[6], res[2];
        refhost::unpack29(out29, A, b, res);
        if (A_host) memcpy(A_host, A, sizeof(A));
        if (b_host) memcpy(b_host, b, sizeof(b));
        if (residual_host) memcpy(residual_host, res, sizeof(res));
        return CF_OK;
    }
    if (int r = scratch_begin(ctx, cols, rows)) return r;
    OdomDev* h = ctx->h_scratch_state;
    memcpy(h->Rcurr, Rcurr, 36); memcpy(h->tcurr, tcurr, 12); memcpy(h->Rprev_inv, Rprev_inv, 36); memcpy(h->tprev, tprev, 12);
    h->intr = intr; h->vmap_curr[0] = vmap_curr; h->nmap_curr[0] = nmap_curr; h->vmap_g_prev[0] = vmap_g_prev;
    h->nmap_g_prev[0] = nmap_g_prev; h->distThres = dist_thres; h->angleThres = angle_thres; h->err_surface = err_surface;
    if (int r = scratch_commit(ctx)) return r;
    IcpArgs a{};
    a.m[0] = IcpModelArgs{vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev, ctx->d_scratch_state, ctx->d_acc_a, err_surface, nullptr, 0, 0};
    a.cols = cols; a.rows = rows; a.intr = intr; a.distThres = dist_thres; a.angleThres = angle_thres;
    a.angleSqLt = sqrt_gate_lt(angle_thres); a.distSqLe = sqrt_gate_le(dist_thres);
    a.flags = err_surface ? 1 : 0;
    a.row_begin = row_begin; a.row_end = row_end;
    launch_icp_level(ctx->stream, ctx->icp_launch, a, 1, 0);
    LAUNCHCHK(ctx);
    if (int r = fetch_totals(ctx, ctx->d_acc_a, 32)) return r;
    se3_unpack_host(ctx->h_out, ctx->icp_launch.gram ? -1 : CF_FIX_ICP, A_host, b_host, residual_host);
    if (sums_host) memcpy(sums_host, ctx->h_out, sizeof(int64_t) * 32);
    return CF_OK;
}

int cf_rgb_residual(cf_ctx* ctx, float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* last_depth,
                    const float* next_depth, const uint8_t* last_image, const uint8_t* next_image, cf_dataterm* corres,
                    float max_depth_delta, const float kt[3], const float krkinv[9], int cols, int rows,
                    int* sigma_sum_host, int* count_host)
{
    if (!ctx || !corres || (size_t)cols * rows > (size_t)ctx->cfg.width * ctx->cfg.height) return CF_EINVAL;
    if (int r = scratch_begin(ctx, cols, rows)) return r;
    launch_rgb_cand(ctx->stream, dIdx, dIdy, next_depth, next_image, min_scale, cols, rows, ctx->d_cand_scratch);
    OdomDev* h = ctx->h_scratch_state;
    h->cand[0] = ctx->d_cand_scratch; h->lastDepth[0] = last_depth; h->nextDepth[0] = next_depth;
    h->lastImage[0] = last_image; h->nextImage[0] = next_image; h->corres[0] = corres;
    h->maxDepthDeltaRGB = max_depth_delta; memcpy(h->kt, kt, 12); memcpy(h->krkInv, krkinv, 36);
    if (int r = scratch_commit(ctx)) return r;
    RgbArgs ra{};
    ra.m[0] = rgb_model_args(h, ctx->d_scratch_state, 0);
    ra.cols = cols; ra.rows = rows; ra.maxDepthDelta = max_depth_delta;
    launch_rgb_residual(ctx->stream, ra, 1);
    LAUNCHCHK(ctx);
    if (int r = fetch_totals(ctx, ctx->d_acc_a, 32)) return r;
    if (count_host) *count_host = (int)ctx->h_out[29];
    if (sigma_sum_host) *sigma_sum_host = (int)ctx->h_out[30];
    return CF_OK;
}

// fraction bits of the RGB sums for a given sigma: same integer rule as cf::rgb_fix_bits (cf_device.h)
static int rgb_fix_bits_host(float sigma)
{
    if (sigma == -1.0f || !(sigma >= 2.0f)) return 8;
    unsigned u; memcpy(&u, &sigma, 4);
    const int F = 8 + 2 * ((int)((u >> 23) & 255u) - 127);
    return F > 32 ? 32 : F;
}

int cf_rgb_step(cf_ctx* ctx, const cf_dataterm* corres, float sigma, const float* cloud3, float fx, float fy,
                const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows, float* A_host,
                float* b_host, int64_t* sums_host)
{
    if (!ctx || !corres) return CF_EINVAL;
    if (ctx->icp_arith == CF_ICP_ARITH_REFERENCE) {
        if (sums_host) memset(sums_host, 0, sizeof(int64_t) * 32);
        float out29[29];
        if (int r = ref_rgb_step(ctx, corres, sigma, cloud3, fx, fy, dIdx, dIdy, sobel_scale, cols, rows, out29)) return r;
        float A[36], b[6];
        refhost::unpack29(out29, A, b, nullptr);
        if (A_host) memcpy(A_host, A, sizeof(A));
        if (b_host) memcpy(b_host, b, sizeof(b));
        return CF_OK;
    }
    if (int r = scratch_begin(ctx, cols, rows)) return r;
    OdomDev* h = ctx->h_scratch_state;
    h->corres[0] = const_cast<cf_dataterm*>(corres); h->cloud[0] = cloud3; h->dIdx[0] = dIdx; h->dIdy[0] = dIdy;
    h->sobelScale = sobel_scale; h->intr = cf_cam{fx, fy, 0, 0};
    // rgb_step_kernel derives sigma from words 29/30 of the ICP accumulator (count, sum diff^2);
    // encode the caller's sigma so that the same rule reproduces it: sigma == -1 -> rgbOnly,
    // sigma == 1 -> (count 1, sigma 0) [tmpError == 0 branch], otherwise count = sigma.
    unsigned long long seed[2] = {0, 0};
    if (sigma == -1.f) h->rgbOnly = 1;
    else if (sigma == 1.f) { seed[0] = 1; seed[1] = 0; }
    else { seed[0] = (unsigned long long)(long long)(int)sigma; seed[1] = 1; }
    if ((float)(int)sigma != sigma) return CF_EINVAL;  // the reference only ever passes -1, 1 or an integer count
    if (int r = scratch_commit(ctx)) return r;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_acc_a + 29, seed, sizeof(seed), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // seed is on the host stack
    RgbArgs ra{};
    ra.m[0] = rgb_model_args(h, ctx->d_scratch_state, 0);
    ra.cols = cols; ra.rows = rows; ra.il = cf_cam{fx, fy, 0, 0}; ra.sobelScale = sobel_scale;
    launch_rgb_step(ctx->stream, ra, 1);
    LAUNCHCHK(ctx);
    if (int r = fetch_totals(ctx, ctx->d_acc_b, 32)) return r;
    se3_unpack_host(ctx->h_out, rgb_fix_bits_host(sigma), A_host, b_host, nullptr);
    if (sums_host) memcpy(sums_host, ctx->h_out, sizeof(int64_t) * 32);
    return CF_OK;
}

int cf_so3_step(cf_ctx* ctx, const uint8_t* last_image, const uint8_t* next_image, const float image_basis[9],
                const float kinv[9], const float krlr[9], int cols, int rows, float* A_host, float* b_host,
                float* residual_host, int64_t* sums_host)
{
    if (!ctx) return CF_EINVAL;
    if (ctx->icp_arith == CF_ICP_ARITH_REFERENCE) {
        if (sums_host) memset(sums_host, 0, sizeof(int64_t) * 16);
        float o[11];
        if (int r = ref_so3_step(ctx, last_image, next_image, image_basis, kinv, krlr, cols, rows, o)) return r;
        int shift = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = i; j < 4; ++j) {
                const float value = o[shift++];
                if (j == 3) { if (b_host) b_host[i] = value; }
                else if (A_host) A_host[j * 3 + i] = A_host[i * 3 + j] = value;
            }
        if (residual_host) { residual_host[0] = o[9]; residual_host[1] = o[10]; }
        return CF_OK;
    }
    launch_so3_step(ctx->stream, last_image, next_image, image_basis, kinv, krlr, cols, rows, ctx->d_out);
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long* t = ctx->h_out;
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {  // reduce.cu:1161-1172
            const float value = (float)ldexp((double)(long long)t[shift++], -CF_FIX_SO3);
            if (j == 3) { if (b_host) b_host[i] = value; }
            else if (A_host) A_host[j * 3 + i] = A_host[i * 3 + j] = value;
        }
    if (residual_host) { residual_host[0] = (float)ldexp((double)(long long)t[9], -CF_FIX_SO3); residual_host[1] = (float)(long long)t[10]; }
    if (sums_host) memcpy(sums_host, t, sizeof(int64_t) * 16);
    return CF_OK;
}

// ----------------------------------------------------------------------- RGBDOdometry ----
int cf_odom_create(cf_ctx* ctx, cf_odom** out)
{
    if (!ctx || !out) return CF_EINVAL;
    if ((size_t)ctx->cfg.width * ctx->cfg.height > (1u << 22)) { ctx->set_error("tracker: frames above 2^22 pixels are not supported (22-bit pixel index in the correspondence records)"); return CF_EINVAL; }
    cf_odom* od = new cf_odom();
    od->ctx = ctx;
    *out = od;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const size_t n0 = (size_t)W * H;
    if (int r = dmalloc(ctx, &od->vmaps_tmp, n0 * 4)) return r;
    if (int r = dmalloc(ctx, &od->nmaps_tmp, n0 * 4)) return r;
    for (int i = 0; i < CF_NUM_PYRS; i++) {
        const size_t n = (size_t)(W >> i) * (H >> i);
        if (int r = dmalloc(ctx, &od->vmap_g_prev[i], n * 3)) return r;
        if (int r = dmalloc(ctx, &od->nmap_g_prev[i], n * 3)) return r;
        if (int r = dmalloc(ctx, &od->vmap_curr[i], n * 3)) return r;
        if (int r = dmalloc(ctx, &od->nmap_curr[i], n * 3)) return r;
        if (int r = dmalloc(ctx, &od->lastDepth[i], n)) return r;
        if (int r = dmalloc(ctx, &od->nextDepth[i], n)) return r;
        if (int r = dmalloc(ctx, &od->lastImage[i], n)) return r;
        if (int r = dmalloc(ctx, &od->nextImage[i], n)) return r;
        if (int r = dmalloc(ctx, &od->lastNextImage[i], n)) return r;
        if (int r = dmalloc(ctx, &od->dIdx[i], n)) return r;
        if (int r = dmalloc(ctx, &od->dIdy[i], n)) return r;
        if (int r = dmalloc(ctx, &od->cloud[i], n * 3)) return r;
        if (int r = dmalloc(ctx, &od->corres[i], n)) return r;
        if (int r = dmalloc(ctx, &od->cand[i], n)) return r;
        if (int r = dmalloc(ctx, &od->zrange[i], (n + 63) / 64)) return r;
        od->ext_vmap_curr[i] = nullptr; od->ext_nmap_curr[i] = nullptr; od->ext_zrange[i] = nullptr;
    }
    if (int r = dmalloc(ctx, &od->icp_acc, (size_t)kGroups * 32)) return r;
    {   // 64 groups of grouped atomics, or (cf_set_gn_mode 2) one row of totals per workgroup of the RGB step: a workgroup per 2 record slots
        const size_t slots0 = (n0 + 255) / 256, quads0 = (slots0 + 1) / 2;   // (slots of >= 256 pixels: cf_set_icp_launch allows 64-thread workgroups)
        od->rgb_acc_words = (quads0 > (size_t)kGroups ? quads0 : (size_t)kGroups) * 32;
        if (int r = dmalloc(ctx, &od->rgb_acc, od->rgb_acc_words)) return r;
    }
    if (int r = dmalloc(ctx, &od->occ, ((size_t)(W >> 2) * (H >> 2) + 3) / 4 * 4)) return r;
    if (int r = dmalloc(ctx, &od->aabb, 8)) return r;
    if (int r = dmalloc(ctx, &od->res_range, 8)) return r;
    for (int k = 0; k < cf_ctx::kStateSlots && od->slot < 0; k++)
        if (!ctx->slot_used[k]) { ctx->slot_used[k] = true; od->slot = k; }
    if (od->slot >= 0) {
        od->d_state = ctx->d_state_pool + od->slot; od->h_state = ctx->h_state_pool + od->slot;
        HIPCHK(ctx, hipMemsetAsync(od->d_state, 0, sizeof(OdomDev), ctx->stream));
    } else {
        if (int r = dmalloc(ctx, &od->d_state, 1)) return r;
        HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&od->h_state), sizeof(OdomDev), hipHostMallocCoherent));
    }
    memset(od->h_state, 0, sizeof(OdomDev));
    // RGBDOdometry ctor defaults: RGBDOdometry.h:45-46, RGBDOdometry.cpp:31-36,103-105
    od->distThres = 0.10f;
    od->distSqLe = sqrt_gate_le(od->distThres);
    od->angleThres = (float)sin(20.f * 3.14159254f / 180.f);
    od->angleSqLt = sqrt_gate_lt(od->angleThres);
    od->sobelScale = (float)(1.0 / pow(2.0, 3));
    od->maxDepthDeltaRGB = 0.07f; od->maxDepthRGB = 6.0f;
    od->minGrad[0] = 5; od->minGrad[1] = 3; od->minGrad[2] = 1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

void cf_odom_destroy(cf_odom* od)
{
    if (!od) return;
    (void)hipStreamSynchronize(od->ctx->stream);
    (void)hipFree(od->vmaps_tmp); (void)hipFree(od->nmaps_tmp);
    for (int i = 0; i < CF_NUM_PYRS; i++) {
        (void)hipFree(od->vmap_g_prev[i]); (void)hipFree(od->nmap_g_prev[i]); (void)hipFree(od->vmap_curr[i]);
        (void)hipFree(od->nmap_curr[i]); (void)hipFree(od->lastDepth[i]); (void)hipFree(od->nextDepth[i]);
        (void)hipFree(od->lastImage[i]); (void)hipFree(od->nextImage[i]); (void)hipFree(od->lastNextImage[i]);
        (void)hipFree(od->dIdx[i]); (void)hipFree(od->dIdy[i]); (void)hipFree(od->cloud[i]); (void)hipFree(od->corres[i]);
        (void)hipFree(od->cand[i]); (void)hipFree(od->zrange[i]);
    }
    (void)hipFree(od->icp_acc); (void)hipFree(od->rgb_acc); (void)hipFree(od->occ); (void)hipFree(od->aabb); (void)hipFree(od->res_range);
    if (od->slot >= 0) od->ctx->slot_used[od->slot] = false;
    else { (void)hipFree(od->d_state); (void)hipHostFree(od->h_state); }
    delete od;
}

// RGBDOdometry::initICPModel, RGBDOdometry.cpp:143-175.  The reference copies the GL textures into
// vmaps_tmp/nmaps_tmp first (two 4.9 MB interop copies); vmaps_tmp is kept because initRGBModel /
// initRGB read depth from it afterwards (:179).
static ModelMapsArgs model_maps_args(cf_odom* od, const float* pred_v4, const float* pred_n4, const float pose[16])
{
    ModelMapsArgs a{};
    a.pred_v4 = pred_v4; a.pred_n4 = pred_n4; a.snapshot = od->vmaps_tmp; a.cols = od->ctx->cfg.width; a.rows = od->ctx->cfg.height;
    for (int i = 0; i < CF_NUM_PYRS; ++i) { a.vmap[i] = od->vmap_g_prev[i]; a.nmap[i] = od->nmap_g_prev[i]; }
    const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    const float t[3] = {pose[3], pose[7], pose[11]};
    memcpy(a.R, R, sizeof(R)); memcpy(a.t, t, sizeof(t));
    memcpy(od->map_pose, R, sizeof(R)); memcpy(od->map_pose + 9, t, sizeof(t));
    a.occ = od->occ; od->occ_valid = true;
    // the bounding box only pays for models that are culled (a model that fills the image would add atomics for nothing)
    const bool tiled = a.cols % 16 == 0 && a.rows % 4 == 0;  // launch_model_maps: the pass that reduces the box
    a.aabb = (od->use_occ && tiled) ? od->aabb : nullptr; od->box_valid = a.aabb != nullptr;
    return a;
}
static RgbdChain rgbd_chain(cf_odom* od, const uint8_t* rgba, float* const* depths, uint8_t* const* images)
{
    RgbdChain c{};
    c.v4 = od->vmaps_tmp; c.rgba = rgba;
    for (int i = 0; i < CF_NUM_PYRS; ++i) { c.depth[i] = depths[i]; c.image[i] = images[i]; }
    return c;
}

int cf_odom_init_icp_model(cf_odom* od, const float* pred_v4, const float* pred_n4, const float pose[16])
{
    if (!od || !pred_v4 || !pred_n4 || !pose) return CF_EINVAL;
    cf_ctx* ctx = od->ctx; hipStream_t s = ctx->stream;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    if (W % 4 == 0 && H % 4 == 0) {  // one launch: snapshot + copyMaps + resize chain + transform of every level
        ModelMapsBatch b{};
        b.m[0] = model_maps_args(od, pred_v4, pred_n4, pose);
        launch_model_maps(s, b, 1);
    } else {
        const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
        const float t[3] = {pose[3], pose[7], pose[11]};
        od->occ_valid = false; od->box_valid = false;
        HIPCHK(ctx, hipMemcpyAsync(od->vmaps_tmp, pred_v4, (size_t)W * H * 16, hipMemcpyDeviceToDevice, s));
        launch_copy_maps(s, od->vmaps_tmp, pred_n4, W, H, od->vmap_g_prev[0], od->nmap_g_prev[0]);
        for (int i = 1; i < CF_NUM_PYRS; ++i) {
            launch_resize_map(s, od->vmap_g_prev[i - 1], W >> (i - 1), H >> (i - 1), od->vmap_g_prev[i], false);
            launch_resize_map(s, od->nmap_g_prev[i - 1], W >> (i - 1), H >> (i - 1), od->nmap_g_prev[i], true);
        }
        for (int i = 0; i < CF_NUM_PYRS; ++i) launch_transform_maps(s, od->vmap_g_prev[i], od->nmap_g_prev[i], W >> i, H >> i, R, t);
    }
    LAUNCHCHK(ctx);
    return CF_OK;
}

// RGBDOdometry::populateRGBDData, RGBDOdometry.cpp:177-194
static int populate_rgbd(cf_odom* od, const uint8_t* rgba, float* const* depths, uint8_t* const* images)
{
    cf_ctx* ctx = od->ctx; hipStream_t s = ctx->stream;
    // verticesToDepth + imageBGRToIntensity, then both Gaussian pyramids: three launches for the two chains
    RgbdBatch b{};
    b.c[0] = rgbd_chain(od, rgba, depths, images);
    launch_rgbd_pyramids(s, b, 1, ctx->cfg.width, ctx->cfg.height, od->maxDepthRGB);
    LAUNCHCHK(ctx);
    return CF_OK;
}
int cf_odom_init_rgb_model(cf_odom* od, const uint8_t* pred_rgba) { if (!od || !pred_rgba) return CF_EINVAL; return populate_rgbd(od, pred_rgba, od->lastDepth, od->lastImage); }
int cf_odom_init_rgb(cf_odom* od, const uint8_t* rgba) { if (!od || !rgba) return CF_EINVAL; od->next_depth_is_last = false; return populate_rgbd(od, rgba, od->nextDepth, od->nextImage); }

// initICPModel + initRGBModel + initRGB of `n` trackers in four launches (one grid row per tracker / chain) instead of
// seven per tracker: what CoFusion::trackModels issues for every active model of a frame.
int cf_odom_init_models_batch(cf_ctx* ctx, cf_odom* const* ods, int n, const float* const* pred_v4, const float* const* pred_n4,
                              const uint8_t* const* pred_rgba, const float* const* poses, const uint8_t* frame_rgba)
{
    if (!ctx || n <= 0 || !frame_rgba) return CF_EINVAL;
    std::vector<const uint8_t*> frames((size_t)n, frame_rgba);
    return cf_odom_init_models_batch_frames(ctx, ods, n, pred_v4, pred_n4, pred_rgba, poses, frames.data());
}

// ... with one frame image per tracker: the trackers of SEVERAL sequences (each tracking its own frame) prepared by the same launches
int cf_odom_init_models_batch_frames(cf_ctx* ctx, cf_odom* const* ods, int n, const float* const* pred_v4, const float* const* pred_n4,
                                     const uint8_t* const* pred_rgba, const float* const* poses, const uint8_t* const* frame_rgba)
{
    return cf_odom_init_models_batch_select(ctx, ods, n, pred_v4, pred_n4, pred_rgba, nullptr, nullptr, nullptr, nullptr, 0.f, poses, frame_rgba);
}

// ... and with the fill-in decision of every tracker taken by the kernels (see the header): no host wait for the previous frame
int cf_odom_init_models_batch_select(cf_ctx* ctx, cf_odom* const* ods, int n, const float* const* pred_v4, const float* const* pred_n4,
                                     const uint8_t* const* pred_rgba, const float* const* alt_v4, const float* const* alt_n4,
                                     const uint8_t* const* alt_rgba, const uint32_t* const* fill_counts, float ratio,
                                     const float* const* poses, const uint8_t* const* frame_rgba)
{
    if (!ctx || !ods || n <= 0 || !pred_v4 || !pred_n4 || !pred_rgba || !poses || !frame_rgba) return CF_EINVAL;
    const bool choose = alt_v4 && alt_n4 && alt_rgba && fill_counts;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    if (W % 4 || H % 4) {
        if (choose)
            for (int k = 0; k < n; k++)
                if (fill_counts[k]) { ctx->set_error("cf_odom_init_models_batch_select: the device-side choice needs width and height to be multiples of 4"); return CF_EINVAL; }
        for (int k = 0; k < n; k++) {
            if (int r = cf_odom_init_icp_model(ods[k], pred_v4[k], pred_n4[k], poses[k])) return r;
            if (int r = cf_odom_init_rgb_model(ods[k], pred_rgba[k])) return r;
            if (int r = cf_odom_init_rgb(ods[k], frame_rgba[k])) return r;
        }
        return CF_OK;
    }
    hipStream_t s = ctx->stream;
    for (int base = 0; base < n; base += kPrepBatch) {
        const int nb = n - base < kPrepBatch ? n - base : kPrepBatch;
        ModelMapsBatch mb{};
        RgbdBatch rb{};
        for (int k = 0; k < nb; k++) {
            cf_odom* od = ods[base + k];
            if (!od || !pred_v4[base + k] || !pred_n4[base + k] || !pred_rgba[base + k] || !poses[base + k] || !frame_rgba[base + k]) return CF_EINVAL;
            mb.m[k] = model_maps_args(od, pred_v4[base + k], pred_n4[base + k], poses[base + k]);
            rb.c[k] = rgbd_chain(od, pred_rgba[base + k], od->lastDepth, od->lastImage);   // initRGBModel
            rb.c[k].v4 = pred_v4[base + k];  // (the snapshot vmaps_tmp is written by the same launch: read the prediction it copies)
            if (choose && fill_counts[base + k]) {
                if (!alt_v4[base + k] || !alt_n4[base + k] || !alt_rgba[base + k]) return CF_EINVAL;
                mb.m[k].alt_v4 = alt_v4[base + k]; mb.m[k].alt_n4 = alt_n4[base + k]; mb.m[k].sel = fill_counts[base + k]; mb.m[k].sel_ratio = ratio;
                rb.c[k].alt_v4 = alt_v4[base + k]; rb.c[k].alt_rgba = alt_rgba[base + k]; rb.c[k].sel = fill_counts[base + k]; rb.c[k].sel_ratio = ratio;
            }
            od->next_depth_is_last = true;
        }
        // initRGB: the intensity pyramid of the tracker's frame (its depth pyramid would be a second copy of the first chain's -- same source,
        // same cutoff --: intensity only).  The trackers of one sequence track the same frame: ONE chain per distinct frame, which stores every
        // level into the pyramids of all trackers of that frame (RgbdBatch::fan_owner) -- five models were five identical full-resolution grid rows
        int n_chains = nb;
        for (int k = 0; k < nb; k++) {
            cf_odom* od = ods[base + k];
            int owner = -1;
            for (int j = nb; j < n_chains; j++)
                if (rb.c[j].rgba == frame_rgba[base + k]) { owner = j; break; }
            if (owner < 0) {
                owner = n_chains++;
                rb.c[owner] = rgbd_chain(od, frame_rgba[base + k], od->nextDepth, od->nextImage);
                for (int i = 0; i < CF_NUM_PYRS; ++i) rb.c[owner].depth[i] = nullptr;
            }
            rb.fan_owner[k] = (signed char)(owner + 1);
            for (int i = 0; i < CF_NUM_PYRS; ++i) rb.fan_image[i][k] = od->nextImage[i];
        }
        launch_model_maps_and_pyramids(s, mb, nb, rb, n_chains, W, H, ods[base]->maxDepthRGB);
    }
    LAUNCHCHK(ctx);
    return CF_OK;
}

int cf_odom_init_first_rgb(cf_odom* od, const uint8_t* rgba)
{  // RGBDOdometry.cpp:206-215
    if (!od || !rgba) return CF_EINVAL;
    cf_ctx* ctx = od->ctx; hipStream_t s = ctx->stream;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    launch_intensity(s, rgba, W, H, od->lastNextImage[0]);
    for (int i = 0; i + 1 < CF_NUM_PYRS; i++) launch_pyrdown_u8(s, od->lastNextImage[i], W >> i, H >> i, od->lastNextImage[i + 1]);
    LAUNCHCHK(ctx);
    return CF_OK;
}

int cf_odom_init_icp(cf_odom* od, const float* const depth_pyr[CF_NUM_PYRS], float depth_cutoff)
{  // RGBDOdometry.cpp:110-118
    if (!od || !depth_pyr) return CF_EINVAL;
    cf_ctx* ctx = od->ctx; hipStream_t s = ctx->stream;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const cf_cam intr = {ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy};
    FrameMapsArgs a{};
    a.cutoff = depth_cutoff;
    for (int i = 0; i < CF_NUM_PYRS; ++i) {
        const int div = 1 << i;
        const cf_cam il = {intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div};
        a.depth[i] = depth_pyr[i]; a.vmap[i] = od->vmap_curr[i]; a.nmap[i] = od->nmap_curr[i]; a.zrange[i] = od->zrange[i];
        a.fx_inv[i] = 1.f / il.fx; a.fy_inv[i] = 1.f / il.fy; a.cx[i] = il.cx; a.cy[i] = il.cy;
        od->ext_vmap_curr[i] = nullptr; od->ext_nmap_curr[i] = nullptr; od->ext_zrange[i] = nullptr;
    }
    launch_frame_maps(s, a, W, H);  // createVMap + createNMap of the three levels in one launch
    od->zrange_valid = true;
    LAUNCHCHK(ctx);
    return CF_OK;
}

int cf_odom_set_culling(cf_odom* od, int on)
{
    if (!od) return CF_EINVAL;
    od->use_occ = on != 0;
    return CF_OK;
}

// One model's reductions split over GPUs: this rank reduces the image rows [row_begin, row_end) (level-0 rows, multiples of 4) inside the
// device-resident Gauss-Newton loop; after every {ICP || residual} launch the registered collective (cf_set_collective, op 0) sums
// the model's accumulators over the ranks, so every rank solves the same system and holds the same pose.  add_counts: exactly one
// rank of the split passes 1 (the RGB residual pass, which every rank runs in full, adds its count / sigma only there).
int cf_odom_set_band(cf_odom* od, int row_begin, int row_end, int add_counts)
{
    if (!od || row_begin < 0 || row_end < row_begin || row_end > od->ctx->cfg.height || (row_begin & 3) || (row_end & 3)) return CF_EINVAL;
    od->band_begin = row_begin; od->band_end = row_end; od->band_counts = add_counts != 0;
    return CF_OK;
}
int cf_set_collective(cf_ctx* ctx, int (*fn)(void*, int, void*, uint64_t, void*), void* user)
{
    if (!ctx) return CF_EINVAL;
    ctx->collective = fn; ctx->collective_user = user;
    return CF_OK;
}

int cf_odom_bind_frame_maps(cf_odom* od, const float* const vmaps[CF_NUM_PYRS], const float* const nmaps[CF_NUM_PYRS])
{
    if (!od) return CF_EINVAL;
    for (int i = 0; i < CF_NUM_PYRS; i++) { od->ext_vmap_curr[i] = vmaps ? vmaps[i] : nullptr; od->ext_nmap_curr[i] = nmaps ? nmaps[i] : nullptr; od->ext_zrange[i] = nullptr; }
    return CF_OK;
}
// the frame maps `owner` computed with cf_odom_init_icp (all models of a frame track the same frame), incl. their per-run depth
// intervals -- what lets the culled ICP reduction of `od` skip runs at other depths
int cf_odom_share_frame_maps(cf_odom* od, cf_odom* owner)
{
    if (!od || !owner || od->ctx != owner->ctx) return CF_EINVAL;
    for (int i = 0; i < CF_NUM_PYRS; i++) {
        od->ext_vmap_curr[i] = owner->ext_vmap_curr[i] ? owner->ext_vmap_curr[i] : owner->vmap_curr[i];
        od->ext_nmap_curr[i] = owner->ext_nmap_curr[i] ? owner->ext_nmap_curr[i] : owner->nmap_curr[i];
        od->ext_zrange[i] = owner->ext_vmap_curr[i] ? owner->ext_zrange[i] : (owner->zrange_valid ? owner->zrange[i] : nullptr);
    }
    return CF_OK;
}

static void inv33f_host(const float a[9], float o[9])
{  // Rprev.inverse() (RGBDOdometry.cpp:316), cofactor form
    float c00 = a[4] * a[8] - a[5] * a[7];
    float c01 = a[5] * a[6] - a[3] * a[8];
    float c02 = a[3] * a[7] - a[4] * a[6];
    float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    float id = 1.0f / det;
    o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// enqueue everything one model needs before the lock-step GN loop
static int odom_prepare(cf_odom* od, const float pose[16], const cf_track_opts* opts, float* err_surface, RgbPrepArgs* prep)
{
    cf_ctx* ctx = od->ctx;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const bool icp = !opts->rgb_only && opts->icp_weight > 0;
    const bool rgb = opts->rgb_only || opts->icp_weight < 100;
    const cf_cam intr = {ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy};
    if (rgb) {
        RgbPrepArgs a{};  // computeDerivativeImages (RGBDOdometry.cpp:231-235) + candidate mask + projectToPointCloud (:333)
        for (int i = 0; i < CF_NUM_PYRS; i++) {
            const int div = 1 << i;
            const cf_cam il = {intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div};
            a.nextImage[i] = od->nextImage[i]; a.nextDepth[i] = od->next_depth(i); a.lastDepth[i] = od->lastDepth[i];
            a.dIdx[i] = od->dIdx[i]; a.dIdy[i] = od->dIdy[i]; a.cand[i] = od->cand[i]; a.cloud[i] = od->cloud[i];
            a.minScale[i] = (float)(pow((double)od->minGrad[i], 2.0) / pow((double)od->sobelScale, 2.0));
            a.fx_inv[i] = 1.0f / il.fx; a.fy_inv[i] = 1.0f / il.fy; a.cx[i] = il.cx; a.cy[i] = il.cy;
        }
        *prep = a;  // launched once for all models of the batch by the caller
    }
    OdomDev* h = od->h_state;
    for (int i = 0; i < CF_NUM_PYRS; i++) {
        h->vmap_curr[i] = od->ext_vmap_curr[i] ? od->ext_vmap_curr[i] : od->vmap_curr[i];
        h->nmap_curr[i] = od->ext_nmap_curr[i] ? od->ext_nmap_curr[i] : od->nmap_curr[i];
        h->vmap_g_prev[i] = od->vmap_g_prev[i]; h->nmap_g_prev[i] = od->nmap_g_prev[i];
        h->lastDepth[i] = od->lastDepth[i]; h->nextDepth[i] = od->next_depth(i);
        h->lastImage[i] = od->lastImage[i]; h->nextImage[i] = od->nextImage[i]; h->lastNextImage[i] = od->lastNextImage[i];
        h->dIdx[i] = od->dIdx[i]; h->dIdy[i] = od->dIdy[i]; h->cloud[i] = od->cloud[i]; h->corres[i] = od->corres[i];
        h->cand[i] = od->cand[i];
        h->minGrad[i] = od->minGrad[i];
    }
    h->icp_acc = od->icp_acc; h->rgb_acc = od->rgb_acc; h->err_surface = err_surface;
    h->intr = intr; h->width = W; h->height = H;
    h->distThres = od->distThres; h->angleThres = od->angleThres; h->sobelScale = od->sobelScale;
    h->maxDepthDeltaRGB = od->maxDepthDeltaRGB; h->icpWeight = opts->icp_weight;
    h->icp = icp; h->rgb = rgb; h->rgbOnly = opts->rgb_only;
#ifdef CF_ABLATE
    static const bool no_box = getenv("CF_NO_SCREEN_BOX") != nullptr;  // diagnostics build: A/B of the screen-box culling on one box
#else
    constexpr bool no_box = false;
#endif
    // the screen box the previous tracking call ended with sizes this call's launches (launch_icp_kernel_arith); the pinned host state
    // holds it once that call's results were fetched
    if (h->cull && !od->result_pending) memcpy(od->box_hint, h->stats.cull_box, sizeof(od->box_hint));
    else od->box_hint[0] = kNoBoxHint;
    // ... and likewise the record slots its residual passes needed (a culled tracker's candidate mask is empty outside its prediction)
    for (int i = 0; i < CF_NUM_PYRS; i++) od->res_hint[i] = (h->cull && h->res_range && !od->result_pending) ? h->res_seen[i] : -1;
    // The screen box is the projection of a frustum piece expressed in the camera the model maps were prepared with (map_pose), while the
    // reduction projects with the pose of THIS call: the culling is conservative only if the two are the same pose -- which
    // Model::performTracking guarantees (both are the model's pose) and a C-ABI caller tracking from another initial guess does not.
    const float Rt_call[12] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10], pose[3], pose[7], pose[11]};
    const bool same_pose = memcmp(Rt_call, od->map_pose, sizeof(Rt_call)) == 0;
    h->aabb_acc = od->aabb; h->cull = (od->use_occ && od->box_valid && od->band_end == 0 && !no_box && same_pose) ? 1 : 0;
    memcpy(h->box_R, od->map_pose, 36); memcpy(h->box_t, od->map_pose + 9, 12);
    h->res_range = (h->cull && rgb) ? od->res_range : nullptr;
    h->host_twin = od->h_state;
    for (int i = 0; i < CF_NUM_PYRS; i++) h->res_seen[i] = -1;
    if (rgb) prep->res_range = h->res_range;
    // the first launch of a tracking call latches the accumulator and zeroes it (so3_prealign_kernel): a second tracking call on the
    // same preparation would read an empty box and cull every workgroup.  It falls back to the whole image instead.
    od->box_valid = false;
    const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    const float t[3] = {pose[3], pose[7], pose[11]};
    memcpy(h->Rprev, R, 36); memcpy(h->tprev, t, 12); memcpy(h->Rcurr, R, 36); memcpy(h->tcurr, t, 12);
    inv33f_host(R, h->Rprev_inv);
    memset(&h->stats, 0, sizeof(h->stats));
    h->solves = 0;
    {   // the Gauss-Newton launches this call will enqueue (launch_gn_track): every one of them runs one solve per tracker
        const int its = (opts->fast_odom ? 3 : 10) + (opts->pyramid ? 9 : 0);
        od->expected_solves = its;
    }
    // the accumulators are zero here: dmalloc clears them and every solve leaves them cleared
    od->pending_so3_swap = opts->so3 != 0;
    LAUNCHCHK(ctx);
    return CF_OK;
}

static void fill_rgb_args(cf_ctx* ctx, cf_odom* const* ods, int n, RgbArgs out[3])
{
    const cf_cam intr{ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy};
    for (int l = 0; l < 3; l++) {
        RgbArgs& a = out[l];
        memset(&a, 0, sizeof(a));
        const int div = 1 << l;
        a.cols = ctx->cfg.width >> l; a.rows = ctx->cfg.height >> l;
        a.il = cf_cam{intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div};
        a.sobelScale = ods[0]->sobelScale; a.maxDepthDelta = ods[0]->maxDepthDeltaRGB;
        for (int m = 0; m < n; m++) {
            a.m[m] = rgb_model_args(ods[m]->h_state, ods[m]->d_state, l);
            a.m[m].no_counts = (ods[m]->band_end > 0 && !ods[m]->band_counts) ? 1 : 0;
            a.m[m].res_blocks = a.m[m].res_range ? residual_blocks_for(ods[m]->res_hint[l]) : 0;
        }
    }
}

static void fill_icp_args(cf_ctx* ctx, cf_odom* const* ods, int n, IcpArgs out[3])
{
#ifdef CF_ABLATE
    static const bool no_zcull = getenv("CF_NO_ZCULL") != nullptr;  // diagnostics build: A/B of the depth-interval culling on one box
    static const bool no_occ = getenv("CF_NO_OCC") != nullptr;      // ... and of the occupancy look-up
#else
    constexpr bool no_zcull = false, no_occ = false;
#endif
    const cf_cam intr = {ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy};
    for (int l = 0; l < CF_NUM_PYRS; l++) {
        IcpArgs& a = out[l];
        memset(&a, 0, sizeof(a));
        const int div = 1 << l;
        a.cols = ctx->cfg.width >> l; a.rows = ctx->cfg.height >> l;
        a.intr = cf_cam{intr.fx / div, intr.fy / div, intr.cx / div, intr.cy / div};
        a.distThres = ods[0]->distThres; a.angleThres = ods[0]->angleThres;
        a.angleSqLt = ods[0]->angleSqLt; a.distSqLe = ods[0]->distSqLe;
        a.occ_w = ctx->cfg.width >> 2; a.occ_shift = 2 - l;
        for (int m = 0; m < n; m++) {
            cf_odom* od = ods[m];
            a.m[m] = IcpModelArgs{od->ext_vmap_curr[l] ? od->ext_vmap_curr[l] : od->vmap_curr[l],
                                  od->ext_nmap_curr[l] ? od->ext_nmap_curr[l] : od->nmap_curr[l],
                                  od->vmap_g_prev[l], od->nmap_g_prev[l], od->d_state, od->icp_acc,
                                  od->h_state->err_surface, (od->use_occ && od->occ_valid && !no_occ) ? od->occ : nullptr,
                                  od->band_end > 0 ? (od->band_begin >> l) : 0, od->band_end > 0 ? (od->band_end >> l) : 0,
                                  od->h_state->cull,
                                  no_zcull ? nullptr : (od->ext_vmap_curr[l] ? od->ext_zrange[l] : (od->zrange_valid ? od->zrange[l] : nullptr)),
                                  od->h_state->cull ? box_blocks_for(od->box_hint, l, a.cols, a.rows, ctx->icp_launch.threads) : 0};
        }
    }
}

int cf_odom_track_batch_async(cf_ctx* ctx, cf_odom* const* ods, int n, const float* const* poses_in,
                              const cf_track_opts* opts, float* const* err_surfaces)
{
    if (!ctx || !ods || n <= 0 || n > ctx->cfg.max_models || n > kMaxBatch || !poses_in || !opts) return CF_EINVAL;
    const bool want_rgb = opts->rgb_only || opts->icp_weight < 100;
    // The pinned host state of a tracker is rewritten below and read / written by the launches of its tracking call: a tracker whose
    // PREVIOUS call has not been fetched yet must drain first.  Per tracker (round 6): the chunks of a frame with more than kMaxBatch
    // trackers hold different trackers, so chunk k + 1 is prepared and enqueued while chunk k runs -- until round 5 a context-wide flag
    // drained the GPU between the chunks.
    {
        bool drain = false;
        for (int m = 0; m < n; m++) drain = drain || (ods[m] && ods[m]->result_pending);
        if (drain) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            for (int m = 0; m < n; m++) if (ods[m]) ods[m]->result_pending = false;   // (landed: what the launches wrote is in the pinned states)
        }
    }
    if (ctx->icp_arith == CF_ICP_ARITH_REFERENCE) {
        // THE REFERENCE'S OWN ORDER (track_ref.hip): per tracker the reference's host loop -- kernel, second-stage kernel, read-back, host
        // solve -- over the same prepared pyramids; nothing is culled (every thread's partial sum is part of the result), nothing is
        // split over GPUs (a rank's band would be another launch shape).  Synchronous; the results are where the default tracker's last
        // solve leaves them, so cf_odom_fetch_result and everything behind it are unchanged.
        for (int m = 0; m < n; m++)
            if (ods[m]->band_end > 0) { ctx->set_error("the reference-order arithmetic (cf_set_icp_arith 2) does not split a tracker's reductions over GPUs"); return CF_ESTATE; }
        RgbPrepBatch rp{};
        for (int base = 0; base < n; base += kPrepBatch) {
            const int nb = n - base < kPrepBatch ? n - base : kPrepBatch;
            rp = RgbPrepBatch{};
            for (int m = base; m < base + nb; m++) {
                if (int r = odom_prepare(ods[m], poses_in[m], opts, err_surfaces ? err_surfaces[m] : nullptr, &rp.m[m - base])) return r;
                ods[m]->h_state->cull = 0; ods[m]->h_state->res_range = nullptr; rp.m[m - base].res_range = nullptr;
                ods[m]->box_hint[0] = kNoBoxHint;
                HIPCHK(ctx, hipMemsetAsync(ods[m]->aabb, 0, 8 * sizeof(unsigned), ctx->stream));   // (the default tracker's first launch latches and clears it)
            }
            if (want_rgb) launch_rgb_prep(ctx->stream, rp, nb, ctx->cfg.width, ctx->cfg.height);
        }
        LAUNCHCHK(ctx);
        for (int m = 0; m < n; m++)
            if (int r = ref_track(ctx, ods[m], poses_in[m], opts, err_surfaces ? err_surfaces[m] : nullptr)) return r;
        for (int m = 0; m < n; m++) ods[m]->result_pending = true;
        return CF_OK;
    }
    for (int m = 0; m < n; m++)   // (modes 0 / 1 expect zeroed sums where mode 2 left per-workgroup rows, and the other way round)
        if (ods[m]->gn_epoch_seen != ctx->gn_mode_epoch) {
            HIPCHK(ctx, hipMemsetAsync(ods[m]->rgb_acc, 0, sizeof(unsigned long long) * ods[m]->rgb_acc_words, ctx->stream));
            ods[m]->gn_epoch_seen = ctx->gn_mode_epoch;
        }
    // a batch of <= kPrepBatch trackers with the SO3 pre-alignment: the RGB preparation rides in the pre-alignment's launch
    const bool prep_fused = want_rgb && opts->so3 != 0 && n <= kPrepBatch;
    RgbPrepBatch prep{};
    for (int base = 0; base < n; base += kPrepBatch) {  // the preparation launches hold kPrepBatch trackers each
        const int nb = n - base < kPrepBatch ? n - base : kPrepBatch;
        if (base) prep = RgbPrepBatch{};
        for (int m = base; m < base + nb; m++) {
            if (int r = odom_prepare(ods[m], poses_in[m], opts, err_surfaces ? err_surfaces[m] : nullptr, &prep.m[m - base])) return r;
        }
        if (want_rgb && !prep_fused) launch_rgb_prep(ctx->stream, prep, nb, ctx->cfg.width, ctx->cfg.height);
    }
    if (prep_fused) rgb_prep_levels(prep, n, ctx->cfg.width, ctx->cfg.height);
    // (no state upload here: the first launch of the schedule reads the pinned host states itself -- so3_prealign_kernel)
    TrackerStates states{};
    for (int m = 0; m < n; m++) { states.dev[m] = ods[m]->d_state; states.host[m] = ods[m]->h_state; }
    const bool so3_here = opts->so3 != 0;
    const bool icp = !opts->rgb_only && opts->icp_weight > 0;
    const bool rgb = opts->rgb_only || opts->icp_weight < 100;
    IcpArgs icp_args[3];
    fill_icp_args(ctx, ods, n, icp_args);
    RgbArgs rgb_args[3];
    fill_rgb_args(ctx, ods, n, rgb_args);
    OdomDev* h_states[kMaxBatch];
    for (int m = 0; m < n; m++) h_states[m] = ods[m]->h_state;
    GnHook hook{};
    hook.fn = ctx->collective; hook.user = ctx->collective_user;
    bool any_split = false;
    for (int m = 0; m < n; m++) { hook.split[m] = ods[m]->band_end > 0 ? 1 : 0; any_split = any_split || hook.split[m]; }
    if (any_split && !ctx->collective) { ctx->set_error("a tracker has a row band (cf_odom_set_band) but no collective is registered (cf_set_collective)"); return CF_ESTATE; }
    if (any_split && ctx->gn_mode == 0) { ctx->set_error("split reductions need the record-slot data path (cf_set_gn_mode 1)"); return CF_ESTATE; }
    // timing events on the level-0 launches of every prof.enabled-th tracking call (cf_profile_enable(ctx, N)): the event pairs cost
    // host time (static 640x480: 1275 frames/s without, 1210 with events on every call), sampling keeps the figure and the cost apart
    cf::ProfSink* prof = nullptr;
    if (ctx->prof.enabled > 0 && (ctx->prof_calls++ % (unsigned)ctx->prof.enabled) == 0) prof = &ctx->prof;
#ifdef CF_ABLATE
    static const int solve_trace_call = getenv("CF_SOLVE_TRACE") ? atoi(getenv("CF_SOLVE_TRACE")) : -1;   // phase stamps of that call's solves (tracker 0)
    static int solve_trace_seen = 0;
    const bool solve_trace = solve_trace_call >= 0 && solve_trace_seen++ == solve_trace_call;
    if (solve_trace) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); trace_solve_begin(); }
#endif
    if (!launch_gn_track(ctx->stream, ctx->icp_launch, states, ctx->d_so3_sync, any_split ? &hook : nullptr, icp_args, rgb_args, n,
                         ctx->cfg.width, ctx->cfg.height, so3_here, opts->pyramid != 0, opts->fast_odom != 0, rgb, icp, ctx->gn_mode, prof, h_states,
                         prep_fused ? &prep : nullptr)) {
        ctx->set_error("tracking: the registered collective failed inside the Gauss-Newton loop");
        return CF_ESTATE;
    }
    LAUNCHCHK(ctx);
#ifdef CF_ABLATE
    if (solve_trace) trace_solve_end(ctx->stream, getenv("CF_ICP_TRACE_OUT") ? getenv("CF_ICP_TRACE_OUT") : "solve_trace.txt");
    {
        static const int so3_trace_call = getenv("CF_SO3_TRACE") ? atoi(getenv("CF_SO3_TRACE")) : -1;   // stamps of that call's SO(3) pre-alignment (tracker 0)
        static int so3_seen = 0;
        if (so3_trace_call >= 0 && so3_seen++ == so3_trace_call) trace_so3_dump(ctx->stream);
    }
    // diagnostics: CF_ICP_REPLAY=<call> re-launches the level-0 {ICP || residual} launch of that tracking call (its converged state) back
    // to back under a list of ablation masks and prints the durations -- the decomposition of the launch quoted in DESIGN.md 4.1
    static const int replay_call = getenv("CF_ICP_REPLAY") ? atoi(getenv("CF_ICP_REPLAY")) : -1;
    static const int trace_call = getenv("CF_ICP_TRACE") ? atoi(getenv("CF_ICP_TRACE")) : -1;   // per-workgroup stamps of that call's level-0 launch
    static const int step_trace_call = getenv("CF_STEP_TRACE") ? atoi(getenv("CF_STEP_TRACE")) : -1;   // phases of that call's fused RGB step + solve launch
    static int step_trace_seen = 0;
    if (step_trace_call >= 0 && step_trace_seen++ == step_trace_call) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        trace_step_solve(ctx->stream, ctx->icp_launch, icp_args[0], rgb_args[0], ctx->d_so3_sync, n, getenv("CF_ICP_TRACE_OUT") ? getenv("CF_ICP_TRACE_OUT") : "step_trace.txt");
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    static int calls_seen = 0, trace_calls_seen = 0;
    if (trace_call >= 0 && trace_calls_seen++ == trace_call) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        trace_icp_level0(ctx->stream, ctx->icp_launch, icp_args[0], rgb_args[0], n, ctx->gn_mode, getenv("CF_ICP_TRACE_OUT") ? getenv("CF_ICP_TRACE_OUT") : "icp_trace.txt");
        for (int m = 0; m < n; m++) {
            HIPCHK(ctx, hipMemsetAsync(ods[m]->icp_acc, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ods[m]->rgb_acc, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (replay_call >= 0 && calls_seen++ == replay_call) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        const int masks[] = {0, 8, 16, 32, 8 | 32, 256, 256 | 32, 256 | 16 | 8, 8 | 256 | 32, 256 | 32 | 512, 256 | 32 | 1024, 1024};
        const char* names[] = {"full", "culled models: no ICP", "culled models: no residual", "no residual at all", "unculled ICP only",
                               "unculled models: no ICP", "culled ICP only", "unculled residual + culled nothing", "nothing (empty workgroups)",
                               "culled ICP: the cull test only", "culled ICP only, XCD bands", "full, XCD bands for culled models"};
        for (int k = 0; k < 12; k++) {
            const float us = replay_icp_level0(ctx->stream, ctx->icp_launch, icp_args[0], rgb_args[0], n, ctx->gn_mode, masks[k], 200,
                                               ctx->prof.events[ctx->prof.capacity - 2], ctx->prof.events[ctx->prof.capacity - 1]);
            fprintf(stderr, "[icp replay] %d trackers, mask %3d  %-40s %7.2f us\n", n, masks[k], names[k], us);
        }
        for (int m = 0; m < n; m++) {
            HIPCHK(ctx, hipMemsetAsync(ods[m]->icp_acc, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
            HIPCHK(ctx, hipMemsetAsync(ods[m]->rgb_acc, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
#endif
    // no read-back copy: the last solve of the schedule wrote every tracker's result into its pinned host state (h_states)
    for (int m = 0; m < n; m++) ods[m]->result_pending = true;
    return CF_OK;
}

// The frame's host wait (a polling variant on hipEventQuery was measured in round 3 and bought nothing: DESIGN-NOTES.md).
int cf_wait_stream(cf_ctx* ctx)
{
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

int cf_odom_fetch_result(cf_odom* od, float trans[3], float rot[9], cf_track_stats* stats)
{
    if (!od) return CF_EINVAL;
    cf_ctx* ctx = od->ctx;
    if (int r = cf_wait_stream(ctx)) return r;
    od->result_pending = false;
    if (trans) memcpy(trans, od->h_state->tcurr, 12);
    if (rot) memcpy(rot, od->h_state->Rcurr, 36);
    if (stats) *stats = od->h_state->stats;
    bool fault = od->h_state->stats.fault != 0;
    if (ctx->icp_arith != CF_ICP_ARITH_REFERENCE && od->expected_solves >= 0 && od->h_state->solves != od->expected_solves) {
        // a launch of the schedule ended without a solve: under cf_set_gn_mode 2 the tracker's workgroups did not meet in one L2 (nobody
        // drew the last ticket); the tickets they left behind would corrupt the next call -- cleared here
        (void)hipMemsetAsync(ctx->d_so3_sync, 0, sizeof(So3Sync) * ((size_t)ctx->cfg.max_models + 1), ctx->stream);
        fault = true;
    }
    od->expected_solves = -1;
    if (od->pending_so3_swap) {  // RGBDOdometry.cpp:469-473
        for (int i = 0; i < CF_NUM_PYRS; i++) std::swap(od->lastNextImage[i], od->nextImage[i]);
        od->pending_so3_swap = false;
    }
    if (fault) {  // a device-side wait between co-resident workgroups expired: the pose was computed from partial sums
        ctx->set_error("tracking: a bounded device-side wait expired (workgroups of one launch were not co-resident)");
        return CF_ESTATE;
    }
    return CF_OK;
}

// RGBDOdometry::getCovariance, RGBDOdometry.cpp:479.  Eigen's lu() is PartialPivLU: per column the row with the largest |entry| at or
// below the diagonal is the pivot (first maximum wins), the column below it is divided by the pivot, rank-1 update of the trailing
// block; inverse() = P e_j through the unit-lower and the upper triangle, column by column.  Host arithmetic in f64, as the reference's.
int cf_odom_get_covariance(const cf_track_stats* stats, double cov[36])
{
    if (!stats || !cov) return CF_EINVAL;
    constexpr int N = 6;
    double lu[N * N];
    int perm[N];
    memcpy(lu, stats->lastA, sizeof(lu));
    for (int i = 0; i < N; i++) perm[i] = i;
    for (int k = 0; k < N; k++) {
        int piv = k;
        double big = fabs(lu[k * N + k]);
        for (int r = k + 1; r < N; r++) { const double a = fabs(lu[r * N + k]); if (a > big) { big = a; piv = r; } }
        if (piv != k) {
            for (int c = 0; c < N; c++) std::swap(lu[k * N + c], lu[piv * N + c]);
            std::swap(perm[k], perm[piv]);
        }
        if (big != 0.0)
            for (int r = k + 1; r < N; r++) lu[r * N + k] /= lu[k * N + k];
        for (int r = k + 1; r < N; r++)
            for (int c = k + 1; c < N; c++) lu[r * N + c] -= lu[r * N + k] * lu[k * N + c];
    }
    for (int j = 0; j < N; j++) {
        double x[N];
        for (int i = 0; i < N; i++) x[i] = (perm[i] == j) ? 1.0 : 0.0;
        for (int i = 0; i < N; i++)
            for (int c = 0; c < i; c++) x[i] -= lu[i * N + c] * x[c];
        for (int i = N - 1; i >= 0; i--) {
            for (int c = i + 1; c < N; c++) x[i] -= lu[i * N + c] * x[c];
            x[i] /= lu[i * N + i];
        }
        for (int i = 0; i < N; i++) cov[i * N + j] = x[i];
    }
    return CF_OK;
}

int cf_set_gn_mode(cf_ctx* ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return CF_EINVAL;
    if (mode == 2) {
        // mode 2 is only correct where hardware workgroup b runs on XCD b mod 8 (rgb_step_solve_kernel: the tracker's workgroups meet in ONE
        // L2).  Checked once per context on the device at hand; refused -- the mode stays what it was -- where it does not hold (another
        // XCD count, a partition mode, CU masking).  ADVICE r5.
        if (ctx->xcd_round_robin < 0) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); ctx->xcd_round_robin = probe_xcd_round_robin(ctx->stream) ? 1 : 0; }
        if (!ctx->xcd_round_robin) { ctx->set_error("cf_set_gn_mode 2: workgroups of a launch are not dealt round-robin over 8 XCDs on this device; the mode needs that"); return CF_ESTATE; }
    }
    if (mode != ctx->gn_mode) {
        // the RGB accumulators mean different things under mode 2 (one row per step workgroup, overwritten) and modes 0 / 1 (kGroups rows of
        // zeroed sums): a switch starts from clean ones.  Trackers created later start clean anyway.
        ctx->gn_mode_epoch++;
    }
    ctx->gn_mode = mode;
    return CF_OK;
}

int cf_odom_get_incremental_transformation(cf_odom* od, float trans[3], float rot[9], const cf_track_opts* opts,
                                           float* icp_err_surface, cf_track_stats* stats)
{
    if (!od || !trans || !rot || !opts) return CF_EINVAL;
    const float pose[16] = {rot[0], rot[1], rot[2], trans[0], rot[3], rot[4], rot[5], trans[1],
                            rot[6], rot[7], rot[8], trans[2], 0, 0, 0, 1};
    const float* poses[1] = {pose};
    float* errs[1] = {icp_err_surface};
    cf_odom* ods[1] = {od};
    if (int r = cf_odom_track_batch_async(od->ctx, ods, 1, poses, opts, errs)) return r;
    return cf_odom_fetch_result(od, trans, rot, stats);
}

// Micro-benchmark: `iters` back-to-back ICP-reduce launches at `level` on the state left by the last
// tracking call (same maps, same pose); returns the average microseconds per launch (hipEvents on the
// launch stream, one pair around the whole batch so event overhead does not pollute the figure).
int cf_odom_bench_icp(cf_odom* od, int level, int iters, float* avg_us)
{
    if (!od || level < 0 || level >= CF_NUM_PYRS || iters <= 0 || !avg_us) return CF_EINVAL;
    cf_ctx* ctx = od->ctx;
#ifdef CF_ABLATE
    const char* ab = getenv("CF_ICP_ABLATE");   // diagnostics build: ablation bits of the launch (track_reduce.hip: ABL)
#else
    const char* ab = nullptr;
#endif
    IcpArgs all[3];
    cf_odom* ods[1] = {od};
    fill_icp_args(ctx, ods, 1, all);
    IcpArgs a = all[level];
    a.flags = ab ? (atoi(ab) << 8) : 0;
    for (int i = 0; i < 3; i++) launch_icp_level(ctx->stream, ctx->icp_launch, a, 1, level);
    HIPCHK(ctx, hipEventRecord(ctx->prof.events[ctx->prof.capacity - 2], ctx->stream));
    for (int i = 0; i < iters; i++) launch_icp_level(ctx->stream, ctx->icp_launch, a, 1, level);
    HIPCHK(ctx, hipEventRecord(ctx->prof.events[ctx->prof.capacity - 1], ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->prof.events[ctx->prof.capacity - 2], ctx->prof.events[ctx->prof.capacity - 1]));
    *avg_us = ms * 1000.f / iters;
    HIPCHK(ctx, hipMemsetAsync(od->icp_acc, 0, sizeof(unsigned long long) * kGroups * 32, ctx->stream));
    return CF_OK;
}

// What the level-0 {ICP || residual} launch of the LAST fetched tracking call visited for this tracker, in pixels: the 64-pixel runs inside
// its final screen box (the whole image when it was not culled) and the record slots between its first and last RGB candidate x the
// pixels per slot (the whole image likewise).  The physical byte count of bench.py's roofline is made of these.
int cf_odom_level0_visited(cf_odom* od, uint64_t* icp_pixels, uint64_t* residual_pixels)
{
    if (!od || !icp_pixels || !residual_pixels) return CF_EINVAL;
    const cf_ctx* ctx = od->ctx;
    const OdomDev* h = od->h_state;
    const uint64_t N = (uint64_t)ctx->cfg.width * ctx->cfg.height;
    *icp_pixels = N; *residual_pixels = h->rgb ? N : 0;
    if (!h->icp) *icp_pixels = 0;
    if (h->cull && ctx->icp_arith != CF_ICP_ARITH_REFERENCE) {
        if (h->icp) { const CullRuns cr = cull_runs(h->stats.cull_box, 0, ctx->cfg.width, ctx->cfg.height); *icp_pixels = (uint64_t)cr.total * 64 < N ? (uint64_t)cr.total * 64 : N; }
        if (h->rgb && h->res_range && h->res_seen[0] >= 0) { const uint64_t px = (uint64_t)h->res_seen[0] * (uint64_t)(ctx->icp_launch.threads * 4); *residual_pixels = px < N ? px : N; }
    }
    return CF_OK;
}

int cf_odom_buffer(cf_odom* od, int which, int level, void** dptr, uint64_t* bytes)
{
    if (!od || level < 0 || level >= CF_NUM_PYRS || !dptr) return CF_EINVAL;
    const size_t n = (size_t)(od->ctx->cfg.width >> level) * (od->ctx->cfg.height >> level);
    void* p = nullptr; size_t b = 0;
    // a caller that takes the pointer of the current vertex map may write it: the depth intervals cf_odom_init_icp stored beside the
    // map no longer describe it (the culled reduction would skip runs that now hold matching depths)
    if (which == 0 && !od->ext_vmap_curr[level]) od->zrange_valid = false;
    switch (which) {
        case 0: p = od->ext_vmap_curr[level] ? (void*)od->ext_vmap_curr[level] : od->vmap_curr[level]; b = n * 12; break;
        case 1: p = od->ext_nmap_curr[level] ? (void*)od->ext_nmap_curr[level] : od->nmap_curr[level]; b = n * 12; break;
        case 2: p = od->vmap_g_prev[level]; b = n * 12; break;
        case 3: p = od->nmap_g_prev[level]; b = n * 12; break;
        case 4: p = od->lastDepth[level]; b = n * 4; break;
        case 5: p = od->next_depth(level); b = n * 4; break;
        case 6: p = od->lastImage[level]; b = n; break;
        case 7: p = od->nextImage[level]; b = n; break;
        case 8: p = od->lastNextImage[level]; b = n; break;
        case 9: p = od->dIdx[level]; b = n * 2; break;
        case 10: p = od->dIdy[level]; b = n * 2; break;
        case 11: p = od->cloud[level]; b = n * 12; break;
        case 12: p = od->corres[level]; b = n * 16; break;
        default: return CF_EINVAL;
    }
    *dptr = p; if (bytes) *bytes = b;
    return CF_OK;
}

}  // extern "C"

