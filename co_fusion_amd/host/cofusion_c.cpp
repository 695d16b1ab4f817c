// cofusion_c.cpp -- flat C wrapper (include/cofusion.h) around the C++ facade.
#include "../../include/cofusion.h"

#include <cstring>
#include <exception>
#include <iterator>
#include <string>
#include <vector>


#include "CoFusion.h"
#include "KlgIO.h"

using namespace cofusion;

struct cofusion_handle { CoFusion* cf; bool borrowed = false; };  // borrowed: a sequence of a lock-step group (owned and stepped by the group)
struct cofusion_group { CoFusionGroup* g; std::vector<cofusion_handle> handles; };  // handles: borrowed views of the sequences
static thread_local std::string g_err;

#define GUARD(expr)                                  \
    try { expr; }                                    \
    catch (const std::exception& e) { g_err = e.what(); return -1; } \
    catch (...) { g_err = "unknown exception"; return -1; }

extern "C" {

void cofusion_default_config(cofusion_config* c)
{
    const CoFusion::Config d;
    c->width = d.width; c->height = d.height; c->fx = d.fx; c->fy = d.fy; c->cx = d.cx; c->cy = d.cy; c->device = d.device;
    c->max_surfels = d.maxSurfels; c->max_models = d.maxModels; c->conf_global_init = d.confGlobalInit;
    c->conf_object_init = d.confObjectInit; c->depth_cutoff = d.depthCutoff; c->icp_weight = d.icpWeight;
    c->outlier_coefficient = d.outlierCoefficient; c->fast_odom = d.fastOdom; c->so3 = d.so3; c->frame_to_frame_rgb = d.frameToFrameRGB;
    c->pyramid = d.pyramid; c->rgb_only = d.rgbOnly; c->model_spawn_offset = d.modelSpawnOffset;
    c->enable_multiple_models = d.enableMultipleModels;
    c->enable_pose_logging = d.enablePoseLogging;
    c->rank = d.rank; c->world = d.world;
    c->device_frames_complete = d.deviceFramesComplete;
    c->mid_frame_predict = d.midFramePredict;
    c->shard_background = d.shardBackground;
    c->enqueue_threads = d.enqueueThreads;
    c->colocate_background = d.colocateBackground;
    c->reloc = d.reloc;
    c->early_index_maps = d.earlyIndexMaps;
}

static CoFusion::Config to_config(const cofusion_config* c)
{
    CoFusion::Config d;
    d.width = c->width; d.height = c->height; d.fx = c->fx; d.fy = c->fy; d.cx = c->cx; d.cy = c->cy; d.device = c->device;
    d.maxSurfels = c->max_surfels; d.maxModels = c->max_models; d.confGlobalInit = c->conf_global_init;
    d.confObjectInit = c->conf_object_init; d.depthCutoff = c->depth_cutoff; d.icpWeight = c->icp_weight;
    d.outlierCoefficient = c->outlier_coefficient; d.fastOdom = c->fast_odom; d.so3 = c->so3; d.frameToFrameRGB = c->frame_to_frame_rgb;
    d.pyramid = c->pyramid; d.rgbOnly = c->rgb_only; d.modelSpawnOffset = c->model_spawn_offset;
    d.enableMultipleModels = c->enable_multiple_models;
    d.enablePoseLogging = c->enable_pose_logging != 0;
    d.rank = c->rank; d.world = c->world < 1 ? 1 : c->world;
    d.deviceFramesComplete = c->device_frames_complete != 0;
    d.midFramePredict = c->mid_frame_predict != 0;
    d.shardBackground = c->shard_background != 0;
    d.enqueueThreads = c->enqueue_threads < 0 ? 0 : c->enqueue_threads;
    d.colocateBackground = c->colocate_background != 0;
    d.reloc = c->reloc != 0;
    d.earlyIndexMaps = c->early_index_maps != 0;
    return d;
}

int cofusion_create(const cofusion_config* c, cofusion_handle** out)
{
    if (!c || !out) { g_err = "null argument"; return -1; }
    const CoFusion::Config d = to_config(c);
    GUARD(*out = new cofusion_handle{new CoFusion(d)});
    return 0;
}
void cofusion_destroy(cofusion_handle* h)
{
    if (!h || h->borrowed) return;  // a group's sequence belongs to the group (cofusion_group_destroy)
    delete h->cf; delete h;
}
const char* cofusion_last_error(void) { return g_err.c_str(); }
int cofusion_set_stream(cofusion_handle* h, void* s) { return cf_set_stream(h->cf->context(), s); }

static int run_frame(cofusion_handle* h, const FrameData& f, const float* in_pose)
{
    if (!h) { g_err = "null handle"; return -1; }
    if (h->borrowed) { g_err = "this handle is a sequence of a lock-step group: step it with cofusion_group_process_frames"; return -1; }
    Mat4f p;
    if (in_pose) for (int i = 0; i < 16; i++) p.m[i] = in_pose[i];
    GUARD(h->cf->processFrame(f, in_pose ? &p : nullptr));
    return 0;
}
int cofusion_process_frame(cofusion_handle* h, int64_t ts, const uint8_t* rgb, const float* depth, const uint8_t* mask, const float* in_pose)
{
    FrameData f; f.timestamp = ts; f.rgb = rgb; f.depth = depth; f.mask = mask;
    return run_frame(h, f, in_pose);
}
int cofusion_process_frame_device(cofusion_handle* h, int64_t ts, const float* depth_dev, const uint8_t* rgba_dev, const float* in_pose)
{
    FrameData f; f.timestamp = ts; f.depth_dev = depth_dev; f.rgba_dev = rgba_dev;
    return run_frame(h, f, in_pose);
}
int cofusion_num_models(cofusion_handle* h) { return (int)h->cf->getModels().size(); }
int cofusion_tick(cofusion_handle* h) { return h->cf->getTick(); }
int cofusion_is_lost(cofusion_handle* h) { return h && h->cf->getLost() ? 1 : 0; }

static Model* model_at(cofusion_handle* h, int index)
{
    auto& l = h->cf->getModels();
    if (index < 0 || index >= (int)l.size()) return nullptr;
    auto it = l.begin();
    std::advance(it, index);
    return it->get();
}
int cofusion_model_info(cofusion_handle* h, int index, unsigned* id, unsigned* count, float pose[16], float* conf)
{
    Model* m = model_at(h, index);
    if (!m) { g_err = "model index out of range"; return -1; }
    if (id) *id = m->getID();
    if (count) *count = m->lastCount();
    if (pose) for (int i = 0; i < 16; i++) pose[i] = m->getPose().m[i];
    if (conf) *conf = m->getConfidenceThreshold();
    return 0;
}
int cofusion_model_download(cofusion_handle* h, int index, float* surfels, uint32_t capacity, uint32_t* count)
{
    Model* m = model_at(h, index);
    if (!m) { g_err = "model index out of range"; return -1; }
    if (!m->isOwned()) { if (count) *count = 0; return 0; }
    return cf_model_download_map(m->handle(), surfels, capacity, count);
}
int cofusion_model_icp_stats(cofusion_handle* h, int index, float* err, float* cnt)
{
    Model* m = model_at(h, index);
    if (!m) { g_err = "model index out of range"; return -1; }
    if (err) *err = m->lastStats.last_icp_error;
    if (cnt) *cnt = m->lastStats.last_icp_count;
    return 0;
}
int cofusion_model_cull_box(cofusion_handle* h, int index, int box[4])
{
    Model* m = model_at(h, index);
    if (!m || !box) { g_err = "model index out of range"; return -1; }
    for (int k = 0; k < 4; k++) box[k] = m->lastStats.cull_box[k];
    return 0;
}
int cofusion_model_level0_visited(cofusion_handle* h, int index, uint64_t* icp_pixels, uint64_t* residual_pixels)
{
    Model* m = model_at(h, index);
    if (!m || !m->getFrameOdometry()) { g_err = "model index out of range"; return -1; }
    if (cf_odom_level0_visited(m->getFrameOdometry(), icp_pixels, residual_pixels)) { g_err = "cf_odom_level0_visited"; return -1; }
    return 0;
}
int cofusion_model_tracking_inputs(cofusion_handle* h, int index, float* vertex4, float* normal4, uint8_t* image_rgba)
{
    Model* m = model_at(h, index);
    if (!m) { g_err = "model index out of range"; return -1; }
    if (!m->isOwned()) { g_err = "model is a shadow on this rank"; return -1; }
    const float* v; const float* n; const uint8_t* img;
    GUARD(m->trackingInputs(m->requiresFillIn(), h->cf->cfg.frameToFrameRGB, v, n, img));
    cf_ctx* ctx = h->cf->context();
    const uint64_t N = (uint64_t)h->cf->cfg.width * h->cf->cfg.height;
    if (vertex4 && cf_memcpy_d2h(ctx, vertex4, v, N * 16)) { g_err = cf_last_error(ctx); return -1; }
    if (normal4 && cf_memcpy_d2h(ctx, normal4, n, N * 16)) { g_err = cf_last_error(ctx); return -1; }
    if (image_rgba && cf_memcpy_d2h(ctx, image_rgba, img, N * 4)) { g_err = cf_last_error(ctx); return -1; }
    return 0;
}
const uint8_t* cofusion_mask_device(cofusion_handle* h) { return h->cf->maskDevice(); }
void* cofusion_context(cofusion_handle* h) { return h->cf->context(); }
int cofusion_set_crf(cofusion_handle* h, float uwe, float uke, float thn, float wa, float ws, float srgb, float sdepth, float spos, float minr,
                     float maxr, unsigned its)
{
    Segmentation& s = h->cf->segmentation();
    s.unaryWeightError = uwe; s.unaryKError = uke; s.unaryThresholdNew = thn; s.weightAppearance = wa; s.weightSmoothness = ws;
    s.scaleFeaturesRGB = 1.0f / srgb; s.scaleFeaturesDepth = 1.0f / sdepth; s.scaleFeaturesPos = 1.0f / spos;
    s.minRelSizeNew = minr; s.maxRelSizeNew = maxr; s.crfIterations = its;
    return 0;
}

int cofusion_set_allreduce(cofusion_handle* h, cofusion_allreduce_i64_fn fn, void* user)
{
    h->cf->setAllreduce(fn, user);
    return 0;
}
int cofusion_set_allreduce_device(cofusion_handle* h, cofusion_allreduce_dev_fn fn, void* user)
{
    h->cf->setAllreduceDevice(fn, user);
    return 0;
}
// ---- lock-step group of sequences on one GPU (CoFusionGroup) ----
int cofusion_group_create(const cofusion_config* c, int sequences, cofusion_group** out)
{
    if (!c || !out) { g_err = "null argument"; return -1; }
    const CoFusion::Config d = to_config(c);
    try {
        cofusion_group* g = new cofusion_group{new CoFusionGroup(d, sequences), std::vector<cofusion_handle>()};
        for (int s = 0; s < sequences; s++) g->handles.push_back(cofusion_handle{&g->g->sequence(s), true});
        *out = g;
    }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
    catch (...) { g_err = "unknown exception"; return -1; }
    return 0;
}
void cofusion_group_destroy(cofusion_group* g) { if (g) { delete g->g; delete g; } }
int cofusion_group_size(cofusion_group* g) { return g ? g->g->size() : 0; }
cofusion_handle* cofusion_group_sequence(cofusion_group* g, int s)
{
    if (!g || s < 0 || s >= g->g->size()) { g_err = "sequence index out of range"; return nullptr; }
    return &g->handles[(size_t)s];
}
int cofusion_group_set_stream(cofusion_group* g, void* s) { return g ? cf_set_stream(g->g->context(), s) : -1; }
int cofusion_group_process_frames(cofusion_group* g, const int64_t* ts, const uint8_t* const* rgb, const float* const* depth, const uint8_t* const* mask)
{
    if (!g || !rgb || !depth) { g_err = "null argument"; return -1; }
    std::vector<FrameData> f((size_t)g->g->size());
    for (size_t s = 0; s < f.size(); s++) { f[s].timestamp = ts ? ts[s] : 0; f[s].rgb = rgb[s]; f[s].depth = depth[s]; f[s].mask = mask ? mask[s] : nullptr; }
    GUARD(g->g->processFrames(f.data()));
    return 0;
}
int cofusion_group_process_frames_device(cofusion_group* g, const int64_t* ts, const float* const* depth_dev, const uint8_t* const* rgba_dev)
{
    if (!g || !rgba_dev || !depth_dev) { g_err = "null argument"; return -1; }
    std::vector<FrameData> f((size_t)g->g->size());
    for (size_t s = 0; s < f.size(); s++) { f[s].timestamp = ts ? ts[s] : 0; f[s].depth_dev = depth_dev[s]; f[s].rgba_dev = rgba_dev[s]; }
    GUARD(g->g->processFrames(f.data()));
    return 0;
}

int cofusion_rccl_unique_id(void* id128)
{
    // (creating the id needs no context: rank 0 calls this before any instance exists.  Forwarded to the C-ABI library, the only one
    // of the two that links RCCL)
    if (!id128) { g_err = "null id buffer"; return -1; }
    if (cf_rccl_unique_id(id128) != CF_OK) { g_err = "cf_rccl_unique_id (ncclGetUniqueId) failed"; return -1; }
    return 0;
}
int cofusion_init_rccl(cofusion_handle* h, const void* id128)
{
    if (!h || !id128) { g_err = "null argument"; return -1; }
    GUARD(h->cf->initRccl(id128));
    return 0;
}
int cofusion_broadcast(cofusion_handle* h, void* dev_buf, uint64_t bytes, int root)
{
    if (!h || !dev_buf) { g_err = "null argument"; return -1; }
    GUARD(h->cf->broadcast(dev_buf, bytes, root));
    return 0;
}
int cofusion_model_owned(cofusion_handle* h, int index)
{
    Model* m = model_at(h, index);
    if (!m) { g_err = "model index out of range"; return -1; }
    return m->isOwned() ? 1 : 0;
}
/* diagnostics: host wall-clock per processFrame phase of the calling thread (PhaseTimes order), frames counted */
int cofusion_debug_phase_ms(double* out, int n, long* frames, int reset)
{
    PhaseTimes& t = phaseTimes();
    for (int i = 0; i < n && i < PhaseTimes::Count; i++) out[i] = t.ms[i];
    if (frames) *frames = t.frames;
    if (reset) t = PhaseTimes();
    return PhaseTimes::Count;
}
int cofusion_set_export_segmentation(cofusion_handle* h, const char* prefix)
{
    h->cf->setExportSegmentation(prefix ? prefix : "");
    return 0;
}
int cofusion_save_ply(cofusion_handle* h, const char* prefix)
{
    try { const int n = h->cf->savePly(prefix ? prefix : ""); if (n < 0) g_err = "savePly: cannot write"; return n; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int cofusion_export_poses(cofusion_handle* h, const char* prefix)
{
    try { const int n = h->cf->exportPoses(prefix ? prefix : ""); if (n < 0) g_err = "exportPoses: cannot write"; return n; }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}

struct cofusion_klg_reader { KlgLogReader r; cofusion_klg_reader(const char* f, int w, int hh, bool fl) : r(f, w, hh, fl) {} };
struct cofusion_klg_writer { KlgLogWriter w; cofusion_klg_writer(const char* f, int ww, int hh, bool c) : w(f, ww, hh, c) {} };
int cofusion_klg_open(const char* file, int width, int height, int flip, cofusion_klg_reader** out, int* num_frames)
{
    if (!file || !out || width <= 0 || height <= 0) { g_err = "cofusion_klg_open: bad arguments"; return -1; }
    auto* r = new cofusion_klg_reader(file, width, height, flip != 0);
    if (!r->r.ok()) { g_err = r->r.error(); delete r; return -1; }
    if (num_frames) *num_frames = r->r.getNumFrames();
    *out = r;
    return 0;
}
int cofusion_klg_next(cofusion_klg_reader* r, int64_t* ts, float* depth_m, uint8_t* rgb)
{
    if (!r) return -1;
    if (!r->r.hasMore()) return 1;  // end of log
    if (!r->r.getNext()) { g_err = r->r.error(); return -1; }
    if (ts) *ts = r->r.timestamp;
    if (depth_m) memcpy(depth_m, r->r.depth.data(), r->r.depth.size() * sizeof(float));
    if (rgb) memcpy(rgb, r->r.rgb.data(), r->r.rgb.size());
    return 0;
}
int cofusion_klg_set_reference_compatible(cofusion_klg_reader* r, int on) { if (!r) return -1; r->r.referenceCompatible = on != 0; return 0; }
void cofusion_klg_close(cofusion_klg_reader* r) { delete r; }
int cofusion_klg_create(const char* file, int width, int height, int compress_depth, cofusion_klg_writer** out)
{
    if (!file || !out || width <= 0 || height <= 0) { g_err = "cofusion_klg_create: bad arguments"; return -1; }
    auto* w = new cofusion_klg_writer(file, width, height, compress_depth != 0);
    if (!w->w.ok()) { g_err = std::string("cannot create ") + file; delete w; return -1; }
    *out = w;
    return 0;
}
int cofusion_klg_write(cofusion_klg_writer* w, int64_t ts, const float* depth_m, const uint8_t* rgb)
{
    if (!w || !depth_m) return -1;
    if (!w->w.write(ts, depth_m, rgb)) { g_err = "klg write failed"; return -1; }
    return 0;
}
int cofusion_klg_finish(cofusion_klg_writer* w) { if (!w) return -1; w->w.close(); delete w; return 0; }

}  // extern "C"
