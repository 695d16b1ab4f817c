// cabi_model.hip -- C-ABI of the surfel model (Core/Model/Model.h:117-235, Core/Model/ModelProjection.h)
// and of the frame-level pre-processing (CoFusion::filterDepth).  All buffers are allocated when the
// model is created; nothing is allocated per frame.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "cf_host.h"

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

struct cf_model {
    cf_ctx* ctx = nullptr;
    uint32_t max_surfels = 0;
    uint32_t count_host = 0;          // exact, or (while count_pending) an upper bound used for launch sizes
    // The exact count after clean() travels back asynchronously (pinned copy + event): the frame loop never waits
    // for it -- every kernel guards on the device-side count, so launches only need an upper bound -- and entry
    // points that expose the number to the host (cf_model_count, download, buffer 11) resolve it on demand.
    bool count_pending = false;
    hipEvent_t count_event = nullptr;
    hipEvent_t count_wait = nullptr;  // the event that marks the pending count: count_event, or the context's batch event (cf_models_frame_passes)
    // same for the fill-in ratio of the latest splat prediction (computed right after combinedPredict)
    bool ratio_valid = false;
    hipEvent_t ratio_event = nullptr;
    float* buf[2] = {nullptr, nullptr};  // ping-pong surfel buffers (Model::vbos[2])
    int target = 0;
    float* staged = nullptr;          // clean staging [max_surfels + N/4]
    unsigned* flags = nullptr;        // [max_surfels + N/4]
    unsigned* block_sums = nullptr;
    unsigned* d_count = nullptr;      // device surfel count
    unsigned* d_nfresh = nullptr;     // appended new-unstable count
    unsigned* d_tmp2 = nullptr;       // [4] scratch counters
    unsigned* h_counts = nullptr;     // pinned [4]
    // per-pixel rank-ordered fusion records
    float* records = nullptr;         // [N*12]
    float* fresh = nullptr;           // new unstable surfels [N/4 * 12] (Model::newUnstableBuffer)
    unsigned* new_flags = nullptr;    // [N]
    bool new_flags_clean = true;      // all zeros (allocated so; the association's compaction clears what it reads): no fill launch in front of the next association
    unsigned* owner = nullptr;        // [max_surfels]
    float* fb_rec = nullptr;          // feedback records (raw), [N*12]
    float* fb_raw = nullptr;          // compacted raw feedback [N*12]
    float* fb_filt = nullptr;         // compacted filtered feedback [N*12]
    // index map (ModelProjection sparse* textures) and splat prediction
    unsigned long long* keys = nullptr;
    unsigned* index = nullptr;
    float *vertConf = nullptr, *colorTime = nullptr, *normRad = nullptr;
    float* clean_rec = nullptr;   // [H*W][8]: vertConf | colorTime.zw, index, filtered depth per texel, packed by the index pass in front of the clean stage (cf_models_frame_passes)
    uint8_t* splat_image = nullptr;
    float *splat_vertex = nullptr, *splat_normal = nullptr;
    uint16_t* splat_time = nullptr;
    // FillIn textures
    float *fill_vertex = nullptr, *fill_normal = nullptr;
    uint8_t* fill_image = nullptr;
    float *tcx = nullptr, *tcy = nullptr;
    // cf_models_preindex: the inverse of the tracked pose in device memory, the tracker it came from, and whether the index map of the
    // current frame was already rasterised with it (cf_models_frame_passes then skips its first index pass -- if the pose it is given is
    // the tracker's, bit for bit)
    float* t_inv_dev = nullptr;
    const cf_odom* preindex_od = nullptr;
    int preindex_time = -1;
    float preindex_cutoff = 0.f; int preindex_delta = 0; uint32_t preindex_nb = 0;   // ... and the other arguments that pass was rasterised with
    void drop_preindex() { preindex_od = nullptr; preindex_time = -1; }             // every entry that rewrites the surfels or the index maps calls it
    float* rays = nullptr;  // per-pixel view rays of the splat fragment stage, float4 [H*W]
    float inv_fx = 0, inv_fy = 0;
};

template <typename T>
static int dmalloc(cf_ctx* ctx, T** p, size_t count)
{
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    HIPCHK(ctx, hipMemsetAsync(*p, 0, count * sizeof(T), ctx->cur()));
    return CF_OK;
}

using cf::inv44f;   // (cf_kernels.h: shared with the device)

static inline cf_cam ctx_cam(const cf_ctx* ctx) { return cf_cam{ctx->cfg.fx, ctx->cfg.fy, ctx->cfg.cx, ctx->cfg.cy}; }

extern "C" {

// CoFusion::filterDepth (CoFusion.cpp:567-574)
int cf_bilateral(cf_ctx* ctx, const float* depth, int cols, int rows, float maxD, float* out)
{
    if (!ctx || !depth || !out) return CF_EINVAL;
    launch_bilateral(ctx->cur(), depth, cols, rows, maxD, out);
    LAUNCHCHK(ctx);
    return CF_OK;
}

int cf_model_create(cf_ctx* ctx, int max_surfels, cf_model** out)
{
    if (!ctx || !out || max_surfels <= 0) return CF_EINVAL;
    cf_model* m = new cf_model();
    m->ctx = ctx; m->max_surfels = (uint32_t)max_surfels;
    *out = m;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const size_t N = (size_t)W * H, M = (size_t)max_surfels, Q = N / 4 + 64;
    if (int r = dmalloc(ctx, &m->buf[0], M * 12)) return r;
    if (int r = dmalloc(ctx, &m->buf[1], M * 12)) return r;
    if (int r = dmalloc(ctx, &m->staged, (M + Q) * 12)) return r;
    if (int r = dmalloc(ctx, &m->flags, M + Q)) return r;
    if (int r = dmalloc(ctx, &m->block_sums, (M + Q) / 2048 + N / 2048 + 16)) return r;
    if (int r = dmalloc(ctx, &m->d_count, 1)) return r;
    if (int r = dmalloc(ctx, &m->d_nfresh, 1)) return r;
    if (int r = dmalloc(ctx, &m->d_tmp2, 4)) return r;
    if (int r = dmalloc(ctx, &m->records, N * 12)) return r;
    if (int r = dmalloc(ctx, &m->fresh, Q * 12)) return r;
    if (int r = dmalloc(ctx, &m->new_flags, N)) return r;
    if (int r = dmalloc(ctx, &m->owner, M)) return r;
    HIPCHK(ctx, hipMemsetAsync(m->owner, 0xFF, M * sizeof(unsigned), ctx->cur()));
    if (int r = dmalloc(ctx, &m->fb_rec, N * 12)) return r;
    if (int r = dmalloc(ctx, &m->fb_raw, N * 12)) return r;
    if (int r = dmalloc(ctx, &m->fb_filt, N * 12)) return r;
    if (int r = dmalloc(ctx, &m->keys, N)) return r;
    HIPCHK(ctx, hipMemsetAsync(m->keys, 0xFF, N * sizeof(unsigned long long), ctx->cur()));  // empty z-buffer; every resolve pass re-clears it
    if (int r = dmalloc(ctx, &m->index, N)) return r;
    if (int r = dmalloc(ctx, &m->vertConf, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->colorTime, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->normRad, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->clean_rec, N * 8)) return r;
    if (int r = dmalloc(ctx, &m->splat_image, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->splat_vertex, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->splat_normal, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->splat_time, N)) return r;
    if (int r = dmalloc(ctx, &m->fill_vertex, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->fill_normal, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->fill_image, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->tcx, (size_t)W)) return r;
    if (int r = dmalloc(ctx, &m->tcy, (size_t)H)) return r;
    if (int r = dmalloc(ctx, &m->rays, N * 4)) return r;
    if (int r = dmalloc(ctx, &m->t_inv_dev, 16)) return r;
    launch_splat_rays(ctx->cur(), ctx_cam(ctx), W, H, m->rays);
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&m->h_counts), sizeof(unsigned) * 4, hipHostMallocCoherent));  // kernels store the counts here
    HIPCHK(ctx, hipEventCreateWithFlags(&m->count_event, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&m->ratio_event, hipEventDisableTiming));
    // texcoords exactly as the reference builds its uv buffer (Model.cpp:166-170)
    std::vector<float> tx(W), ty(H);
    for (int i = 0; i < W; i++) tx[i] = (float)((double)((float)i / (float)W) + 1.0 / (2.0 * (double)(float)W));
    for (int j = 0; j < H; j++) ty[j] = (float)((double)((float)j / (float)H) + 1.0 / (2.0 * (double)(float)H));
    HIPCHK(ctx, hipMemcpyAsync(m->tcx, tx.data(), sizeof(float) * W, hipMemcpyHostToDevice, ctx->cur()));
    HIPCHK(ctx, hipMemcpyAsync(m->tcy, ty.data(), sizeof(float) * H, hipMemcpyHostToDevice, ctx->cur()));
    m->inv_fx = (float)(1.0 / (double)ctx->cfg.fx); m->inv_fy = (float)(1.0 / (double)ctx->cfg.fy);
    HIPCHK(ctx, hipStreamSynchronize(ctx->cur()));
    return CF_OK;
}

void cf_model_destroy(cf_model* m)
{
    if (!m) return;
    (void)hipStreamSynchronize(m->ctx->cur());
    void* ptrs[] = {m->buf[0], m->buf[1], m->staged, m->flags, m->block_sums, m->d_count, m->d_nfresh, m->d_tmp2, m->records,
                    m->fresh, m->new_flags, m->owner, m->fb_rec, m->fb_raw, m->fb_filt, m->keys, m->index, m->vertConf,
                    m->colorTime, m->normRad, m->splat_image, m->splat_vertex, m->splat_normal, m->splat_time, m->fill_vertex,
                    m->fill_normal, m->fill_image, m->tcx, m->tcy, m->rays, m->t_inv_dev, m->clean_rec};
    for (void* p : ptrs) (void)hipFree(p);
    (void)hipHostFree(m->h_counts);
    if (m->count_event) (void)hipEventDestroy(m->count_event);
    if (m->ratio_event) (void)hipEventDestroy(m->ratio_event);
    delete m;
}

static int adopt_count(cf_model* m)
{
    m->count_host = m->h_counts[0];
    m->count_pending = false;
    if (m->count_host > m->max_surfels) { m->ctx->set_error("surfel buffer overflow"); return CF_ENOMEM; }
    return CF_OK;
}
static int sync_count(cf_model* m)
{
    cf_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipMemcpyAsync(m->h_counts, m->d_count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->cur()));
    HIPCHK(ctx, hipStreamSynchronize(ctx->cur()));
    return adopt_count(m);
}
// the compaction that was just enqueued writes the new count into pinned host memory itself (h_counts[0]); the event marks when
// it has landed.  Until then count_host holds `upper_bound`
static int post_count(cf_model* m, uint32_t upper_bound)
{
    cf_ctx* ctx = m->ctx;
    HIPCHK(ctx, hipEventRecord(m->count_event, ctx->cur()));
    m->count_wait = m->count_event;
    m->count_host = upper_bound;
    m->count_pending = true;
    return CF_OK;
}
// launch bound: exact when the read-back has landed, otherwise the upper bound
static int count_bound(cf_model* m, uint32_t* out)
{
    if (m->count_pending) {
        if (!m->count_wait) { if (int r = sync_count(m)) return r; *out = m->count_host; return CF_OK; }   // (a batch that failed before recording its event)
        if (hipEventQuery(m->count_wait) == hipSuccess) { if (int r = adopt_count(m)) return r; }
        else (void)hipGetLastError();  // hipErrorNotReady is not an error here
    }
    *out = m->count_host;
    return CF_OK;
}
static int exact_count(cf_model* m, uint32_t* out)
{
    if (m->count_pending) {
        if (!m->count_wait) { if (int r = sync_count(m)) return r; *out = m->count_host; return CF_OK; }
        HIPCHK(m->ctx, hipEventSynchronize(m->count_wait));
        if (int r = adopt_count(m)) return r;
    }
    *out = m->count_host;
    return CF_OK;
}

// CoFusion::computeFeedbackBuffers (CoFusion.cpp:157-169) + Model::initialise (Model.cpp:227-272)
int cf_model_initialise(cf_model* m, const uint8_t* rgba, const float* depth_raw, const float* depth_filt, int time, float maxDepth)
{
    if (!m || !rgba || !depth_raw || !depth_filt) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx; hipStream_t s = ctx->cur();
    const int W = ctx->cfg.width, H = ctx->cfg.height; const long long N = (long long)W * H;
    if ((uint32_t)N > m->max_surfels) return CF_ENOMEM;
    const cf_cam cam = ctx_cam(ctx);
    // raw feedback
    launch_feedback(s, rgba, depth_raw, W, H, cam, m->inv_fx, m->inv_fy, m->tcx, m->tcy, time, maxDepth, m->fb_rec, m->new_flags);
    HIPCHK(ctx, hipMemsetAsync(m->fb_raw, 0, sizeof(float) * 12 * N, s));
    launch_scan_scatter(s, m->fb_rec, m->new_flags, N, m->block_sums, m->d_count, 0, m->fb_raw);
    // filtered feedback (zero-filled past its own count, like the reference's zero-initialised VBO)
    launch_feedback(s, rgba, depth_filt, W, H, cam, m->inv_fx, m->inv_fy, m->tcx, m->tcy, time, maxDepth, m->fb_rec, m->new_flags);
    HIPCHK(ctx, hipMemsetAsync(m->fb_filt, 0, sizeof(float) * 12 * N, s));
    launch_scan_scatter(s, m->fb_rec, m->new_flags, N, m->block_sums, m->d_tmp2, 0, m->fb_filt);
    m->new_flags_clean = false;   // (the feedback passes wrote every flag)
    launch_init(s, m->fb_raw, m->fb_filt, m->d_count, N, m->buf[m->target]);
    LAUNCHCHK(ctx);
    return sync_count(m);
}

int cf_model_count(cf_model* m, uint32_t* count) { if (!m || !count) return CF_EINVAL; return exact_count(m, count); }

int cf_model_predict_indices(cf_model* m, const float pose[16], int time, float maxDepth, int timeDelta)
{
    if (!m || !pose) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx;
    float t_inv[16];
    inv44f(pose, t_inv);
    uint32_t nb = 0;
    if (int r = count_bound(m, &nb)) return r;
    launch_predict_indices(ctx->cur(), m->buf[m->target], m->d_count, nb, t_inv, ctx_cam(ctx), ctx->cfg.width, ctx->cfg.height,
                           maxDepth, time, timeDelta, m->keys, m->index, m->vertConf, m->colorTime, m->normRad);
    LAUNCHCHK(ctx);
    return CF_OK;
}

// The two halves of predictIndices, for a surfel map sharded over several GPUs: every rank rasterises its surfel range
// into a key map (depth-ordered 64-bit keys, empty = all ones), the ranks MIN-all-reduce the key maps (u64 order), and
// each resolves the reduced map -- bit-identical to cf_model_predict_indices on one GPU, because the per-pixel z-test
// is a minimum over surfels.
int cf_model_index_keys(cf_model* m, const float pose[16], int time, float maxDepth, int timeDelta, uint32_t surfel_begin, uint32_t surfel_end,
                        uint64_t* keys_dev)
{
    if (!m || !pose || !keys_dev || surfel_begin > surfel_end) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    float t_inv[16];
    inv44f(pose, t_inv);
    uint32_t nb = 0;
    if (int r = count_bound(m, &nb)) return r;
    HIPCHK(ctx, hipMemsetAsync(keys_dev, 0xFF, sizeof(uint64_t) * (size_t)W * H, ctx->cur()));
    launch_index_keys(ctx->cur(), m->buf[m->target], m->d_count, surfel_begin, surfel_end < nb ? surfel_end : nb, t_inv, ctx_cam(ctx), W, H, maxDepth,
                      time, timeDelta, reinterpret_cast<unsigned long long*>(keys_dev));
    LAUNCHCHK(ctx);
    return CF_OK;
}
int cf_model_index_resolve(cf_model* m, const float pose[16], uint64_t* keys_dev)
{
    if (!m || !pose || !keys_dev) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx;
    float t_inv[16];
    inv44f(pose, t_inv);
    launch_index_resolve(ctx->cur(), m->buf[m->target], t_inv, ctx->cfg.width, ctx->cfg.height, reinterpret_cast<unsigned long long*>(keys_dev),
                         m->index, m->vertConf, m->colorTime, m->normRad);
    LAUNCHCHK(ctx);
    return CF_OK;
}

// Model::predictIndices with the rasterisation split over `nshards` ranks by surfel range (the model map is replicated): this rank
// rasterises its range of the EXACT surfel count into the 64-bit z-keys, the registered collective (cf_set_collective, op 1: MIN of
// unsigned 64-bit words) composites the key maps of all ranks, and every rank resolves the same winners -- the index map of
// cf_model_predict_indices bit for bit (the z-test is a minimum over surfels).
int cf_model_predict_indices_sharded(cf_model* m, const float pose[16], int time, float maxDepth, int timeDelta, int shard, int nshards)
{
    if (!m || !pose || nshards < 1 || shard < 0 || shard >= nshards) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx;
    if (!ctx->collective) { ctx->set_error("cf_model_predict_indices_sharded: no collective registered (cf_set_collective)"); return CF_ESTATE; }
    uint32_t n = 0;
    if (int r = exact_count(m, &n)) return r;   // identical on every replica, unlike the asynchronous upper bound
    const uint32_t b = (uint32_t)(((uint64_t)n * (uint64_t)shard) / (uint64_t)nshards), e = (uint32_t)(((uint64_t)n * (uint64_t)(shard + 1)) / (uint64_t)nshards);
    if (int r = cf_model_index_keys(m, pose, time, maxDepth, timeDelta, b, e, reinterpret_cast<uint64_t*>(m->keys))) return r;
    if (ctx->collective(ctx->collective_user, 1, m->keys, (uint64_t)ctx->cfg.width * ctx->cfg.height, (void*)ctx->cur()) != 0) {
        ctx->set_error("cf_model_predict_indices_sharded: the collective failed");
        return CF_ESTATE;
    }
    return cf_model_index_resolve(m, pose, reinterpret_cast<uint64_t*>(m->keys));
}

int cf_model_combined_predict(cf_model* m, const float pose[16], float maxDepth, float confThreshold, int time, int maxTime, int timeDelta)
{
    if (!m || !pose) return CF_EINVAL;
    cf_ctx* ctx = m->ctx;
    float t_inv[16];
    inv44f(pose, t_inv);
    uint32_t nb = 0;
    if (int r = count_bound(m, &nb)) return r;
    launch_combined_predict(ctx->cur(), m->buf[m->target], m->d_count, nb, t_inv, ctx_cam(ctx), ctx->cfg.width, ctx->cfg.height,
                            maxDepth, confThreshold, time, maxTime, timeDelta, m->rays, m->keys, m->splat_image, m->splat_vertex, m->splat_normal,
                            m->splat_time);
    LAUNCHCHK(ctx);
    m->ratio_valid = false;  // a new prediction: any prefetched fill-in ratio is stale
    return CF_OK;
}

// CoFusion::requiresFillIn looks at the frame's LAST prediction at the start of the next frame.  Calling this right
// after that prediction counts its covered pixels and reads the two numbers back asynchronously, so that the question
// never stalls the frame loop (cf_model_requires_fill_in falls back to a blocking count otherwise).
int cf_model_prefetch_fill_ratio(cf_model* m)
{
    if (!m) return CF_EINVAL;
    cf_ctx* ctx = m->ctx;
    launch_fill_ratio(ctx->cur(), m->splat_image, ctx->cfg.width, ctx->cfg.height, m->d_tmp2 + 2, m->h_counts + 2);  // the kernel publishes to pinned memory
    LAUNCHCHK(ctx);
    HIPCHK(ctx, hipEventRecord(m->ratio_event, ctx->cur()));
    m->ratio_valid = true;
    return CF_OK;
}

// Model::performFillIn (Model.cpp:901-909); depth is the FILTERED depth (CoFusion.cpp:541)
int cf_model_perform_fill_in(cf_model* m, const uint8_t* rgba, const float* depth_filt, int passthrough_geom, int passthrough_rgb)
{
    if (!m || !rgba || !depth_filt) return CF_EINVAL;
    cf_ctx* ctx = m->ctx;
    launch_fill_in(ctx->cur(), m->splat_vertex, m->splat_normal, m->splat_image, depth_filt, rgba, ctx->cfg.width, ctx->cfg.height, ctx_cam(ctx),
                   m->inv_fx, m->inv_fy, passthrough_geom, passthrough_rgb, m->fill_vertex, m->fill_normal, m->fill_image);
    LAUNCHCHK(ctx);
    return CF_OK;
}

// CoFusion::requiresFillIn (CoFusion.cpp:547-565)
int cf_model_requires_fill_in(cf_model* m, float ratio, int* out)
{
    if (!m || !out) return CF_EINVAL;
    cf_ctx* ctx = m->ctx;
    if (m->ratio_valid) {
        HIPCHK(ctx, hipEventSynchronize(m->ratio_event));
    } else {
        launch_fill_ratio(ctx->cur(), m->splat_image, ctx->cfg.width, ctx->cfg.height, m->d_tmp2 + 2, m->h_counts + 2);
        LAUNCHCHK(ctx);
        HIPCHK(ctx, hipStreamSynchronize(ctx->cur()));
    }
    *out = ((float)m->h_counts[2] / (float)m->h_counts[3] < ratio) ? 1 : 0;
    return CF_OK;
}

int cf_model_fill_ratio_device(cf_model* m, const uint32_t** counts_dev)
{
    if (!m || !counts_dev) return CF_EINVAL;
    *counts_dev = m->ratio_valid ? m->d_tmp2 + 2 : nullptr;
    return CF_OK;
}

// Model::fuse (Model.cpp:408-563).  Needs cf_model_predict_indices for the same pose first.
int cf_model_fuse(cf_model* m, const float pose[16], int time, const uint8_t* rgba, const uint8_t* mask, const float* depth_raw,
                  const float* depth_filt, float maxDepth, float weighting, int maskID)
{
    if (!m || !pose || !rgba || !mask || !depth_raw || !depth_filt) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx; hipStream_t s = ctx->cur();
    const int W = ctx->cfg.width, H = ctx->cfg.height; const long long N = (long long)W * H;
    SurfelFuseArgs a;
    a.index = m->index; a.vertConf = m->vertConf; a.normRad = m->normRad; a.rgba = rgba; a.depth_raw = depth_raw; a.depth_filt = depth_filt;
    a.mask = mask; a.tcx = m->tcx; a.tcy = m->tcy; memcpy(a.pose, pose, sizeof(a.pose)); a.cam = ctx_cam(ctx); a.inv_fx = m->inv_fx;
    a.inv_fy = m->inv_fy; a.cols = W; a.rows = H; a.time = time; a.weighting = weighting; a.maskID = maskID; a.maxDepth = maxDepth;
    a.records = m->records; a.new_flags = m->new_flags; a.owner = m->owner; a.flags_clean = m->new_flags_clean ? 1 : 0;
    launch_associate(s, a);
    // append the new unstable vertices in column-major draw order (transform feedback of data.geom); the flags are left cleared
    {
        ScanPassArgs sp{m->records, m->new_flags, N, m->block_sums, m->d_nfresh, 0, m->fresh, nullptr};
        sp.zero_flags = 1;
        launch_scan_scatter_batch(s, &sp, 1);
        m->new_flags_clean = true;
    }
    // update.vert over all surfels into the other buffer, then swap (Model.cpp:559)
    uint32_t nb = 0;
    if (int r = count_bound(m, &nb)) return r;
    launch_update(s, m->buf[m->target], m->d_count, nb, m->owner, m->records, time, m->buf[1 - m->target]);
    m->target = 1 - m->target;
    LAUNCHCHK(ctx);
    return CF_OK;
}

// Model::clean (Model.cpp:565-697).  Needs cf_model_predict_indices (after fuse) for the same pose first.
int cf_model_clean(cf_model* m, const float pose[16], int time, float confThreshold, float outlierCoeff, int timeDelta, const float* depth_filt,
                   const uint8_t* mask, int maskID, uint32_t* count_out)
{
    if (!m || !pose || !depth_filt || !mask) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx; hipStream_t s = ctx->cur();
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    uint32_t nb = 0;
    if (int r = count_bound(m, &nb)) return r;
    const unsigned bound = nb + (unsigned)((W / 2) * (H / 2));
    if (bound > m->max_surfels + (unsigned)(W * H / 4 + 64)) return CF_ENOMEM;
    SurfelCleanArgs a;
    a.index = m->index; a.vertConf = m->vertConf; a.colorTime = m->colorTime; a.depth_filt = depth_filt; a.mask = mask; a.rec = nullptr;
    inv44f(pose, a.t_inv); a.cam = ctx_cam(ctx); a.cols = W; a.rows = H; a.time = time; a.confThreshold = confThreshold;
    a.outlierCoeff = outlierCoeff; a.timeDelta = timeDelta; a.maskID = maskID;
    launch_clean(s, m->buf[m->target], m->d_count, m->fresh, m->d_nfresh, bound, a, m->staged, m->flags);
    launch_scan_scatter(s, m->staged, m->flags, bound, m->block_sums, m->d_count, 0, m->buf[1 - m->target], m->h_counts);  // the kept total IS the new count
    m->target = 1 - m->target;
    LAUNCHCHK(ctx);
    const uint32_t upper = bound < m->max_surfels ? bound : m->max_surfels;
    if (int r = post_count(m, upper)) return r;
    if (count_out) return exact_count(m, count_out);  // the GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN query: waits
    return CF_OK;
}

// The FIRST index pass of the frame's surfel chain, enqueued before the host has seen the tracked poses (see the header): the trackers'
// device states give the poses.  cf_models_frame_passes skips its own first index pass for a model prepared here -- after checking that
// the pose it is handed is the tracker's result bit for bit (an overridden pose simply rasterises again).
int cf_models_preindex(cf_ctx* ctx, const cf_model_preindex* items, int n, float depth_cutoff, int time_delta)
{
    if (!ctx || !items || n <= 0) return CF_EINVAL;
    hipStream_t s = ctx->cur();
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    std::vector<IndexPassArgs> a((size_t)n);
    for (int k = 0; k < n; k++) {
        cf_model* m = items[k].model; const cf_odom* od = items[k].odom;
        if (!m || !od || m->ctx != ctx || od->ctx != ctx) return CF_EINVAL;
        uint32_t nb = 0;
        if (int r = count_bound(m, &nb)) return r;
        const float* pinv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(od->d_state) + offsetof(cf::OdomDev, pose_inv));
        IndexPassArgs& p = a[k];
        p.surfels = m->buf[m->target]; p.count = m->d_count; p.id_begin = 0; p.id_end = nb; p.maxDepth = depth_cutoff; p.time = items[k].time;
        p.timeDelta = time_delta; p.keys = m->keys; p.index = m->index; p.vertConf = m->vertConf; p.colorTime = m->colorTime; p.normRad = m->normRad;
        p.t_inv_dev = pinv;   // (the inverse of the tracked pose, left in the tracker's state by the last solve of its schedule)
    }
    launch_index_keys_batch(s, a.data(), n, ctx_cam(ctx), W, H);
    launch_index_resolve_batch(s, a.data(), n, ctx_cam(ctx), W, H);
    LAUNCHCHK(ctx);
    for (int k = 0; k < n; k++) {
        cf_model* m = items[k].model;
        m->preindex_od = items[k].odom; m->preindex_time = items[k].time; m->preindex_cutoff = depth_cutoff; m->preindex_delta = time_delta; m->preindex_nb = a[k].id_end;
    }
    return CF_OK;
}
// is the index map of `m` already the one the first index pass of this chain would produce?
static bool preindexed_with(const cf_model* m, const float pose[16], int time, float depth_cutoff, int time_delta, uint32_t nb)
{
    if (!m->preindex_od || m->preindex_time != time) return false;
    if (m->preindex_cutoff != depth_cutoff || m->preindex_delta != time_delta || m->preindex_nb != nb) return false;   // (every argument of the pass, not the pose alone)
    const cf::OdomDev* h = m->preindex_od->h_state;   // (fetched: the caller has the pose from there)
    for (int r = 0; r < 3; r++) {
        if (memcmp(&pose[r * 4], &h->Rcurr[r * 3], 12) != 0 || memcmp(&pose[r * 4 + 3], &h->tcurr[r], 4) != 0) return false;
    }
    return pose[12] == 0 && pose[13] == 0 && pose[14] == 0 && pose[15] == 1;
}

// The second half of a frame for several models in lock-step (see the header): the statements of cf_model_predict_indices / _fuse /
// _clean / _combined_predict, every stage one batched launch.
int cf_models_frame_passes(cf_ctx* ctx, const cf_model_pass* items, int n, float depth_cutoff, float outlier_coeff, int time_delta)
{
    if (!ctx || !items || n <= 0) return CF_EINVAL;
    hipStream_t s = ctx->cur();
    const int W = ctx->cfg.width, H = ctx->cfg.height; const long long N = (long long)W * H;
    const cf_cam cam = ctx_cam(ctx);
    std::vector<int> fusing;
    for (int k = 0; k < n; k++) {
        const cf_model_pass& it = items[k];
        if (!it.model || it.model->ctx != ctx || !it.pose || !it.rgba || !it.depth_filtered) return CF_EINVAL;
        if (it.do_fuse) { if (!it.mask || !it.depth_raw) return CF_EINVAL; fusing.push_back(k); }
    }
    // everything that can fail is checked BEFORE the first launch: a return from the middle of the chain would leave every
    // fusing model half-fused with its buffers flipped and no count posted.  The clean stage's staging bound only depends on the count
    // the model enters the chain with (the update pass rewrites the surfels in place, new ones wait in `fresh`).
    for (int k : fusing) {
        cf_model* m = items[k].model;
        uint32_t nb = 0;
        if (int r = count_bound(m, &nb)) return r;
        if (nb + (unsigned)((W / 2) * (H / 2)) > m->max_surfels + (unsigned)(W * H / 4 + 64)) {
            ctx->set_error("cf_models_frame_passes: a model's surfel buffer cannot hold the clean stage's staging area (max_surfels too small)");
            return CF_ENOMEM;
        }
    }
    for (int k = 0; k < n; k++) { uint32_t nb = 0; if (int r = count_bound(items[k].model, &nb)) return r; }
    // profiling (cf_profile_enable N): an event pair around every N-th chain + its algorithmic bytes
    int surf_ev = -1;
    if (ctx->prof.enabled > 0 && (ctx->surf_calls_seen++ % (unsigned)ctx->prof.enabled) == 0 && ctx->surf_used + 2 <= cf_ctx::kSurfEvents) {
        for (int e = ctx->surf_used; e < ctx->surf_used + 2; e++)
            if (!ctx->surf_events[e]) HIPCHK(ctx, hipEventCreate(&ctx->surf_events[e]));
        surf_ev = ctx->surf_used; ctx->surf_used += 2;
        HIPCHK(ctx, hipEventRecord(ctx->surf_events[surf_ev], s));
        for (int k = 0; k < n; k++) ctx->surf_bytes += (uint64_t)items[k].model->count_host * 48u * (items[k].do_fuse ? 8u : 2u);
        ctx->surf_calls++;
    }
    static const bool clean_rec_on = getenv("CF_NO_CLEAN_REC") == nullptr;   // (diagnostic: the clean stage stages from the three index-map arrays as until round 6)
    auto index_args_of = [&](const std::vector<int>& which, bool feeds_clean, std::vector<IndexPassArgs>& a) -> int {
        a.assign(which.size(), IndexPassArgs{});
        for (size_t q = 0; q < which.size(); q++) {
            const cf_model_pass& it = items[which[q]]; cf_model* m = it.model;
            uint32_t nb = 0;
            if (int r = count_bound(m, &nb)) return r;
            IndexPassArgs& p = a[q];
            p.surfels = m->buf[m->target]; p.count = m->d_count; p.id_begin = 0; p.id_end = nb; inv44f(it.pose, p.t_inv); p.maxDepth = depth_cutoff;
            p.time = it.time; p.timeDelta = time_delta; p.keys = m->keys; p.index = m->index; p.vertConf = m->vertConf; p.colorTime = m->colorTime;
            p.normRad = m->normRad;
            // the pass in front of the clean stage also packs what that stage reads per texel -- with this frame's filtered depth -- into
            // one 32-byte record (clean_kernel stages a 4x4 neighbourhood per surfel: one array and 16-byte pieces instead of three arrays)
            if (feeds_clean && clean_rec_on) { p.clean_rec = m->clean_rec; p.clean_depth = it.depth_filtered; }
        }
        return CF_OK;
    };
    auto index_pass = [&](const std::vector<int>& which) -> int {
        std::vector<IndexPassArgs> a;
        if (int r = index_args_of(which, false, a)) return r;
        launch_index_keys_batch(s, a.data(), (int)a.size(), cam, W, H);
        launch_index_resolve_batch(s, a.data(), (int)a.size(), cam, W, H);
        return CF_OK;
    };
    if (!fusing.empty()) {
        const int nf = (int)fusing.size();
        {   // (models whose index map cf_models_preindex already rasterised with this very pose skip the pass)
            std::vector<int> todo;
            for (int k : fusing) {
                uint32_t nb = 0;
                if (int r = count_bound(items[k].model, &nb)) return r;
                if (!preindexed_with(items[k].model, items[k].pose, items[k].time, depth_cutoff, time_delta, nb)) todo.push_back(k);
            }
            if (!todo.empty()) { if (int r = index_pass(todo)) return r; }
        }
        // Model::fuse (Model.cpp:408-563)
        std::vector<SurfelFuseArgs> fa(nf);
        std::vector<ScanPassArgs> sa(nf);
        std::vector<UpdatePassArgs> ua(nf);
        for (int q = 0; q < nf; q++) {
            const cf_model_pass& it = items[fusing[q]]; cf_model* m = it.model;
            SurfelFuseArgs& a = fa[q];
            a.index = m->index; a.vertConf = m->vertConf; a.normRad = m->normRad; a.rgba = it.rgba; a.depth_raw = it.depth_raw; a.depth_filt = it.depth_filtered;
            a.mask = it.mask; a.tcx = m->tcx; a.tcy = m->tcy; memcpy(a.pose, it.pose, sizeof(a.pose)); a.cam = cam; a.inv_fx = m->inv_fx;
            a.inv_fy = m->inv_fy; a.cols = W; a.rows = H; a.time = it.time; a.weighting = it.weighting; a.maskID = it.mask_id; a.maxDepth = it.fuse_max_depth;
            a.records = m->records; a.new_flags = m->new_flags; a.owner = m->owner; a.flags_clean = m->new_flags_clean ? 1 : 0;
            sa[q] = ScanPassArgs{m->records, m->new_flags, N, m->block_sums, m->d_nfresh, 0, m->fresh, nullptr};
            sa[q].zero_flags = 1;   // (the compaction leaves the flags it read cleared: the next frame's association starts from zeros without a fill launch)
            m->new_flags_clean = true;
            uint32_t nb = 0;
            if (int r = count_bound(m, &nb)) return r;
            ua[q] = UpdatePassArgs{m->buf[m->target], m->d_count, nb, m->owner, m->records, it.time, m->buf[1 - m->target]};
        }
        launch_associate_batch(s, fa.data(), nf);
        // the new unstable vertices in column-major draw order (transform feedback of data.geom) || update.vert over all surfels into the
        // other buffer, then swap (Model.cpp:559) || the rasterisation of the index pass in front of the clean stage: three stages that do
        // not read each other's outputs except update -> rasterisation, as two launches (launch_update_compaction_index_keys)
        for (int q = 0; q < nf; q++) items[fusing[q]].model->target = 1 - items[fusing[q]].model->target;
        {
            std::vector<IndexPassArgs> ia;
            if (int r = index_args_of(fusing, true, ia)) return r;   // (of the swapped buffers: what the update writes)
            static const bool side_by_side = getenv("CF_NO_SIDE_BY_SIDE") == nullptr;   // (diagnostic: the four separate launches)
            if (!(side_by_side && launch_update_compaction_index_keys(s, ua.data(), sa.data(), ia.data(), nf, cam, W, H))) {
                launch_scan_scatter_batch(s, sa.data(), nf);
                launch_update_batch(s, ua.data(), nf);
                launch_index_keys_batch(s, ia.data(), nf, cam, W, H);
            }
            launch_index_resolve_batch(s, ia.data(), nf, cam, W, H);
        }
        // Model::clean (Model.cpp:565-697)
        std::vector<CleanPassArgs> ca(nf);
        std::vector<uint32_t> upper(nf);
        for (int q = 0; q < nf; q++) {
            const cf_model_pass& it = items[fusing[q]]; cf_model* m = it.model;
            uint32_t nb = 0;
            if (int r = count_bound(m, &nb)) return r;
            const unsigned bound = nb + (unsigned)((W / 2) * (H / 2));
            if (bound > m->max_surfels + (unsigned)(W * H / 4 + 64)) return CF_ENOMEM;
            CleanPassArgs& c = ca[q];
            c.h.index = m->index; c.h.vertConf = m->vertConf; c.h.colorTime = m->colorTime; c.h.depth_filt = it.depth_filtered; c.h.mask = it.mask;
            c.h.rec = clean_rec_on ? m->clean_rec : nullptr;
            inv44f(it.pose, c.h.t_inv); c.h.cam = cam; c.h.cols = W; c.h.rows = H; c.h.time = it.time; c.h.confThreshold = it.conf_threshold;
            c.h.outlierCoeff = outlier_coeff; c.h.timeDelta = time_delta; c.h.maskID = it.mask_id;
            c.surfels = m->buf[m->target]; c.count = m->d_count; c.fresh = m->fresh; c.n_fresh = m->d_nfresh; c.total_bound = bound; c.staged = m->staged;
            c.flags = m->flags;
            sa[q] = ScanPassArgs{m->staged, m->flags, (long long)bound, m->block_sums, m->d_count, 0, m->buf[1 - m->target], m->h_counts};  // the kept total IS the new count
            upper[q] = bound < m->max_surfels ? bound : m->max_surfels;
        }
        launch_clean_batch(s, ca.data(), nf);
        launch_scan_scatter_batch(s, sa.data(), nf);
        // the compactions write the new counts into pinned host memory themselves; ONE event (recorded below, behind the prediction
        // launches: the host does not keep the GPU waiting for five event records here) marks them all.  Until it has completed the
        // models carry the upper bound.
        if (!ctx->batch_event) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->batch_event, hipEventDisableTiming));
        for (int q = 0; q < nf; q++) {
            cf_model* m = items[fusing[q]].model;
            m->target = 1 - m->target;
            m->count_host = upper[q]; m->count_pending = true; m->count_wait = nullptr;   // (nullptr until the event below is recorded)
        }
    }
    // combinedPredict(maxDepthProcessed, time, time, timeDelta) of every model (CoFusion.cpp:533-545)
    std::vector<SplatPassArgs> pa(n);
    for (int k = 0; k < n; k++) {
        const cf_model_pass& it = items[k]; cf_model* m = it.model;
        uint32_t nb = m->count_host;   // (exact, or the upper bound the clean stage has just set)
        if (!(m->count_pending && !m->count_wait)) { if (int r = count_bound(m, &nb)) return r; }
        SplatPassArgs& p = pa[k];
        p.surfels = m->buf[m->target]; p.count = m->d_count; p.count_bound = nb; inv44f(it.pose, p.t_inv); p.maxDepth = depth_cutoff;
        p.confThreshold = it.conf_threshold; p.time = it.time; p.maxTime = it.time; p.timeDelta = time_delta; p.rays = m->rays; p.keys = m->keys;
        p.image = m->splat_image; p.vertexConf = m->splat_vertex; p.normalRad = m->splat_normal; p.time16 = m->splat_time;
        m->ratio_valid = false;  // a new prediction: any prefetched fill-in ratio is stale
    }
    launch_combined_predict_batch(s, pa.data(), n, cam, W, H);
    LAUNCHCHK(ctx);
    if (surf_ev >= 0) HIPCHK(ctx, hipEventRecord(ctx->surf_events[surf_ev + 1], s));
    if (!fusing.empty()) {
        HIPCHK(ctx, hipEventRecord(ctx->batch_event, s));
        for (int k : fusing) items[k].model->count_wait = ctx->batch_event;
    }
    for (int k = 0; k < n; k++) items[k].model->drop_preindex();
    return CF_OK;
}

// Model::downloadMap (Model.cpp:867-899)
int cf_model_download_map(cf_model* m, float* host_surfels, uint32_t capacity, uint32_t* count)
{
    if (!m || !count) return CF_EINVAL;
    cf_ctx* ctx = m->ctx;
    if (int r = exact_count(m, count)) return r;
    if (host_surfels) {
        const uint32_t n = m->count_host < capacity ? m->count_host : capacity;
        HIPCHK(ctx, hipMemcpyAsync(host_surfels, m->buf[m->target], (size_t)n * 48, hipMemcpyDeviceToHost, ctx->cur()));
        HIPCHK(ctx, hipStreamSynchronize(ctx->cur()));
    }
    return CF_OK;
}

int cf_model_upload_map(cf_model* m, const float* host_surfels, uint32_t count)
{
    if (!m || (!host_surfels && count) || count > m->max_surfels) return CF_EINVAL;
    m->drop_preindex();   // (the index map / surfels cf_models_preindex saw are about to change)
    cf_ctx* ctx = m->ctx;
    if (count) HIPCHK(ctx, hipMemcpyAsync(m->buf[m->target], host_surfels, (size_t)count * 48, hipMemcpyHostToDevice, ctx->cur()));
    launch_set_count(ctx->cur(), m->d_count, count);
    HIPCHK(ctx, hipStreamSynchronize(ctx->cur()));
    m->count_host = count; m->count_pending = false;
    return CF_OK;
}

// which: 0 index(u32) 1 vertConf 2 colorTime 3 normRad (f32x4) | 4 splat image (rgba8) 5 splat vertexConf 6 splat normalRad
// (f32x4) 7 splat time (u16) | 8 fill vertex 9 fill normal (f32x4) 10 fill image (rgba8) | 11 surfels (f32x12, count entries)
int cf_model_buffer(cf_model* m, int which, void** dptr, uint64_t* bytes)
{
    if (!m || !dptr) return CF_EINVAL;
    const size_t N = (size_t)m->ctx->cfg.width * m->ctx->cfg.height;
    void* p = nullptr; size_t b = 0;
    switch (which) {
        case 0: p = m->index; b = N * 4; break;
        case 1: p = m->vertConf; b = N * 16; break;
        case 2: p = m->colorTime; b = N * 16; break;
        case 3: p = m->normRad; b = N * 16; break;
        case 4: p = m->splat_image; b = N * 4; break;
        case 5: p = m->splat_vertex; b = N * 16; break;
        case 6: p = m->splat_normal; b = N * 16; break;
        case 7: p = m->splat_time; b = N * 2; break;
        case 8: p = m->fill_vertex; b = N * 16; break;
        case 9: p = m->fill_normal; b = N * 16; break;
        case 10: p = m->fill_image; b = N * 4; break;
        case 11: { uint32_t c = 0; if (int r = exact_count(m, &c)) return r; p = m->buf[m->target]; b = (size_t)c * 48; break; }
        default: return CF_EINVAL;
    }
    *dptr = p; if (bytes) *bytes = b;
    return CF_OK;
}

// Model::computeFusionWeight (Model.cpp:391-406) + Model::rodrigues2 (Model.cpp:817-865); host math.
// The JacobiSVD re-orthonormalisation of rodrigues2 is the identity for rotation products.
float cf_fusion_weight(const float pose[16], const float lastPose[16], float weightMultiplier)
{
    float pinv[16], diff[16];
    inv44f(pose, pinv);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += pinv[i * 4 + k] * lastPose[k * 4 + j];
            diff[i * 4 + j] = s;
        }
    const float tn = sqrtf(diff[3] * diff[3] + diff[7] * diff[7] + diff[11] * diff[11]);
    const float* R = diff;
    double rx = R[2 * 4 + 1] - R[1 * 4 + 2], ry = R[0 * 4 + 2] - R[2 * 4 + 0], rz = R[1 * 4 + 0] - R[0 * 4 + 1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (double)((R[0] + R[5] + R[10]) - 1.0f) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        double t;
        if (c > 0) rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0.0 ? t : 0.0);
            t = (R[5] + 1) * 0.5; ry = sqrt(t > 0.0 ? t : 0.0) * (R[0 * 4 + 1] < 0 ? -1.0 : 1.0);
            t = (R[10] + 1) * 0.5; rz = sqrt(t > 0.0 ? t : 0.0) * (R[0 * 4 + 2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[1 * 4 + 2] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    const float rv[3] = {(float)rx, (float)ry, (float)rz};
    const float rn = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    float weighting = tn > rn ? tn : rn;
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    const float w = 1.0f - (weighting / largest);
    weighting = (w > minWeight ? w : minWeight) * weightMultiplier;
    return weighting;
}

}  // extern "C"
