// CoFusion.cpp -- host logic of the facade (see CoFusion.h).  Restates Core/CoFusion.cpp:171-644,
// Core/Model/Model.cpp:319-406 and Core/Segmentation/Segmentation.cpp:59-706 on top of the C-ABI.
// Compiled with -ffp-contract=off: the host arithmetic here is mirrored by the CPU oracle.
#include "CoFusion.h"

#include <algorithm>
#include <chrono>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace cofusion {

PhaseTimes& phaseTimes() { static thread_local PhaseTimes t; return t; }
namespace {
struct PhaseTimer {
    int id; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseTimer(int i) : id(i) {}
    ~PhaseTimer() { phaseTimes().ms[id] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

static void check(cf_ctx* ctx, int rc, const char* what)
{
    if (rc != CF_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (ctx ? cf_last_error(ctx) : ""));
}

// ------------------------------------------------------------------------------- Mat4f ----
Mat4f Mat4f::identity()
{
    Mat4f r;
    for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.f : 0.f;
    return r;
}
Mat4f Mat4f::inverse() const
{
    const float* a = m;
    Mat4f r;
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
        r.m[i * 4 + 0] = Li[i * 3 + 0]; r.m[i * 4 + 1] = Li[i * 3 + 1]; r.m[i * 4 + 2] = Li[i * 3 + 2];
        r.m[i * 4 + 3] = -(Li[i * 3 + 0] * a[3] + Li[i * 3 + 1] * a[7] + Li[i * 3 + 2] * a[11]);
    }
    r.m[12] = 0; r.m[13] = 0; r.m[14] = 0; r.m[15] = 1;
    return r;
}
Mat4f Mat4f::operator*(const Mat4f& o) const
{
    Mat4f r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += m[i * 4 + k] * o.m[k * 4 + j];
            r.m[i * 4 + j] = s;
        }
    return r;
}

void Distributed::sum(int64_t* buf, uint64_t n) const
{
    if (!active()) return;
    if (!allreduce_i64) throw std::runtime_error("model-parallel CoFusion: no all-reduce callback registered (cofusion_set_allreduce)");
    if (allreduce_i64(buf, n, user) != 0) throw std::runtime_error("model-parallel CoFusion: the all-reduce callback failed");
}

// in-place SUM all-reduce of a device buffer: through the device collective when one is registered (RCCL on the context's stream,
// no host visit), otherwise staged through the host callback
void Distributed::sumDevice(cf_ctx* ctx, int64_t* dev, uint64_t n) const
{
    if (!active()) return;
    if (allreduce_dev) {
        if (allreduce_dev(dev, n, cf_get_stream(ctx), user_dev) != 0) throw std::runtime_error("model-parallel CoFusion: the device all-reduce callback failed");
        return;
    }
    std::vector<int64_t> h(n);
    check(ctx, cf_memcpy_d2h(ctx, h.data(), dev, n * 8), "sums read-back");
    sum(h.data(), n);
    check(ctx, cf_memcpy_h2d(ctx, dev, h.data(), n * 8), "sums upload");
}

// ---- RCCL inside the library (csrc/rccl_comm.hip) as this instance's collective ----
namespace {
int rccl_sum_device(int64_t* dev, uint64_t n, void* stream, void* user)
{
    return cf_rccl_allreduce(static_cast<cf_ctx*>(user), dev, n, 0, stream) == CF_OK ? 0 : -1;
}
int rccl_sum_host(int64_t* buf, uint64_t n, void* user)
{   // the few host-side exchanges (surfel counts at retirement, the pose exchange of frames without the device segmentation)
    CoFusion* cf = static_cast<CoFusion*>(user);
    return cf->rcclSumHost(buf, n);
}
}  // namespace

int CoFusion::rcclSumHost(int64_t* buf, uint64_t n)
{
    if (n > rcclStageWords) {
        if (rcclStage) cf_free(ctx, rcclStage);
        rcclStage = nullptr; rcclStageWords = 0;
        void* p = nullptr;
        const uint64_t words = n < 4096 ? 4096 : n;
        if (cf_malloc(ctx, words * 8, &p) != CF_OK) return -1;
        rcclStage = static_cast<int64_t*>(p); rcclStageWords = words;
    }
    if (cf_memcpy_h2d(ctx, rcclStage, buf, n * 8) != CF_OK) return -1;
    if (cf_rccl_allreduce(ctx, rcclStage, n, 0, cf_get_stream(ctx)) != CF_OK) return -1;
    return cf_memcpy_d2h(ctx, buf, rcclStage, n * 8) == CF_OK ? 0 : -1;
}

void CoFusion::initRccl(const void* id128)
{
    check(ctx, cf_rccl_init(ctx, id128, dist.rank, dist.world), "cf_rccl_init");
    dist.allreduce_dev = rccl_sum_device; dist.user_dev = ctx;
    dist.allreduce_i64 = rccl_sum_host; dist.user = this;
}

void CoFusion::broadcast(void* dev_buf, uint64_t bytes, int root)
{
    check(ctx, cf_rccl_broadcast(ctx, dev_buf, bytes, root, cf_get_stream(ctx)), "cf_rccl_broadcast");
}

// ------------------------------------------------------------------------------- Model ----
Model::Model(cf_ctx* c, unsigned char id_, float confidenceThresh, bool enableFillIn, int maxSurfels, float maxDepth_, bool owned_)
    : ctx(c), pose(Mat4f::identity()), lastPose(Mat4f::identity()), confidenceThreshold(confidenceThresh), maxDepth(maxDepth_), id(id_),
      fillIn(enableFillIn), owned(owned_)
{
    if (!owned) return;  // shadow of a model owned by another rank: replicated state only
    check(ctx, cf_model_create(ctx, maxSurfels, &model), "cf_model_create");
    check(ctx, cf_odom_create(ctx, &odom), "cf_odom_create");
    // object models cover a small part of the image: their ICP launches skip the gathers into empty blocks of the prediction
    if (!enableFillIn) check(ctx, cf_odom_set_culling(odom, 1), "cf_odom_set_culling");
    // icpError texture (Model.cpp:112-117), f32 [H*W]; zero-initialised like the reference's upload (GPUTexture.cpp:48-53)
    void* p = nullptr;
    uint64_t bytes = 0;
    void* indexMap = nullptr;
    check(ctx, cf_model_buffer(model, 0, &indexMap, &bytes), "cf_model_buffer");  // the u32 index map has the same N*4 bytes
    check(ctx, cf_malloc(ctx, bytes, &p), "cf_malloc");
    icpError = static_cast<float*>(p);
}
Model::~Model()
{
    if (icpError) cf_free(ctx, icpError);
    if (odom) cf_odom_destroy(odom);
    if (model) cf_model_destroy(model);
}
unsigned Model::lastCount() const
{
    uint32_t c = 0;
    if (owned) cf_model_count(model, &c);
    return c;
}
void Model::initialise(const uint8_t* rgba, const float* depthRaw, const float* depthFiltered, int time, float maxD)
{
    if (!owned) return;
    check(ctx, cf_model_initialise(model, rgba, depthRaw, depthFiltered, time, maxD), "cf_model_initialise");
}
static const void* mbuf(cf_ctx* ctx, cf_model* m, int which)
{
    void* p = nullptr;
    check(ctx, cf_model_buffer(m, which, &p, nullptr), "cf_model_buffer");
    return p;
}
const float* Model::vertexConfProjection() const { return owned ? static_cast<const float*>(mbuf(ctx, model, 5)) : nullptr; }

void Model::trackingInputs(bool doFillIn, bool frameToFrameRGB, const float*& v, const float*& n, const uint8_t*& img) const
{  // Model.cpp:350-367: which prediction feeds initICPModel / initRGBModel
    if (doFillIn) {
        v = static_cast<const float*>(mbuf(ctx, model, 8)); n = static_cast<const float*>(mbuf(ctx, model, 9));
        img = static_cast<const uint8_t*>(mbuf(ctx, model, 10));
    } else {
        v = static_cast<const float*>(mbuf(ctx, model, 5)); n = static_cast<const float*>(mbuf(ctx, model, 6));
        img = static_cast<const uint8_t*>(mbuf(ctx, model, (frameToFrameRGB && allowsFillIn()) ? 10 : 4));
    }
}
void Model::bindFrameMaps(const float* const depthPyr[3], float depthCutoff, Model* frameOwner)
{
    if (frameOwner == this) {
        // the current-frame vertex/normal pyramids do not depend on the model (the mask test is commented out
        // in createVMap, cudafuncs.cu:119): computed once, shared with every other model of this frame
        check(ctx, cf_odom_init_icp(odom, depthPyr, depthCutoff), "initICP");
    } else {
        check(ctx, cf_odom_share_frame_maps(odom, frameOwner->odom), "share_frame_maps");
    }
}
void Model::initICP(bool doFillIn, bool frameToFrameRGB, const float* const depthPyr[3], float depthCutoff, const uint8_t* rgba,
                    Model* frameOwner)
{  // Model.cpp:350-367.  WARNING initICP* must be called before initRGB* (they share vmaps_tmp)
    if (!owned) return;
    const float* v; const float* n; const uint8_t* img;
    trackingInputs(doFillIn, frameToFrameRGB, v, n, img);
    check(ctx, cf_odom_init_icp_model(odom, v, n, pose.m), "initICPModel");
    check(ctx, cf_odom_init_rgb_model(odom, img), "initRGBModel");
    bindFrameMaps(depthPyr, depthCutoff, frameOwner);
    check(ctx, cf_odom_init_rgb(odom, rgba), "initRGB");
}
float Model::computeFusionWeight(float weightMultiplier) const { return cf_fusion_weight(pose.m, lastPose.m, weightMultiplier); }

void Model::fuse(int time, const uint8_t* rgba, const uint8_t* mask, const float* depthRaw, const float* depthFiltered, float depthCutoff,
                 float weightMultiplier)
{
    if (!owned) return;
    const float md = depthCutoff < maxDepth ? depthCutoff : maxDepth;  // std::min(depthCutoff, maxDepth), Model.cpp:443
    check(ctx, cf_model_fuse(model, pose.m, time, rgba, mask, depthRaw, depthFiltered, md, computeFusionWeight(weightMultiplier), (int)id),
          "cf_model_fuse");
}
void Model::clean(int time, int timeDelta, float /*depthCutoff*/, const float* depthFiltered, const uint8_t* mask, float outlierCoeff)
{
    if (!owned) return;
    // the surfel count is read back asynchronously (cf_model_count resolves it on demand): no host wait here
    check(ctx, cf_model_clean(model, pose.m, time, confidenceThreshold, outlierCoeff, timeDelta, depthFiltered, mask, (int)id, nullptr), "cf_model_clean");
}
void Model::predictIndices(int time, float depthCutoff, int timeDelta)
{
    if (!owned) return;
    if (shards > 1) check(ctx, cf_model_predict_indices_sharded(model, pose.m, time, depthCutoff, timeDelta, shard, shards), "predictIndices (sharded)");
    else check(ctx, cf_model_predict_indices(model, pose.m, time, depthCutoff, timeDelta), "predictIndices");
}
void Model::combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta)
{
    if (!owned) return;
    check(ctx, cf_model_combined_predict(model, pose.m, depthCutoff, confidenceThreshold, time, maxTime, timeDelta), "combinedPredict");
}
void Model::performFillIn(const uint8_t* rgba, const float* depthFiltered, bool frameToFrameRGB, bool lost)
{
    if (fillIn && owned) check(ctx, cf_model_perform_fill_in(model, rgba, depthFiltered, lost ? 1 : 0, (lost || frameToFrameRGB) ? 1 : 0), "performFillIn");
}
void Model::prefetchFillRatio()
{
    if (owned && allowsFillIn()) check(ctx, cf_model_prefetch_fill_ratio(model), "prefetch_fill_ratio");
}
bool Model::requiresFillIn(float ratio)
{
    if (!allowsFillIn() || !owned) return false;
    int out = 0;
    check(ctx, cf_model_requires_fill_in(model, ratio, &out), "requiresFillIn");
    return out != 0;
}
const uint32_t* Model::fillRatioDevice() const
{
    if (!allowsFillIn() || !owned) return nullptr;
    const uint32_t* counts = nullptr;
    check(ctx, cf_model_fill_ratio_device(model, &counts), "fill_ratio_device");
    return counts;
}
std::vector<float> Model::downloadMap() const
{
    if (!owned) return {};
    const unsigned n = lastCount();
    std::vector<float> out((size_t)n * 12);
    uint32_t c = 0;
    check(ctx, cf_model_download_map(model, out.data(), n, &c), "downloadMap");
    return out;
}

// ------------------------------------------------------------------------ Segmentation ----
static const int SPIX = 16;

Segmentation::Segmentation(cf_ctx* c, int w, int h, const Distributed* d) : ctx(c), width(w), height(h), dist(d)
{
    check(ctx, cf_seg_create(ctx, &seg), "cf_seg_create");
    memset(gtMapping, 0, sizeof(gtMapping));
    if (dist && dist->active()) {
        void* p = nullptr;
        check(ctx, cf_malloc(ctx, (uint64_t)w * h * 16, &p), "cf_malloc");  // cf_malloc returns zeroed memory
        zeroImage = static_cast<float*>(p);
    }
}
void Segmentation::startSlic(const uint8_t* rgba_dev)
{
    check(ctx, cf_seg_slic(seg, rgba_dev), "cf_seg_slic");
    slicStarted = true;
}
Segmentation::~Segmentation()
{
    if (zeroImage) cf_free(ctx, zeroImage);
    cf_seg_destroy(seg);
}

SegmentationResult Segmentation::performSegmentation(ModelList& models, const FrameData& frame, const float* depth_dev, const uint8_t* rgba_dev,
                                                     const uint8_t* rgba_first_rows, unsigned char nextModelID, bool allowNew,
                                                     uint8_t* full_dev)
{
    if (frame.mask) return performSegmentationGT(models, frame, nextModelID, allowNew, full_dev);
    return performSegmentationCRF(models, depth_dev, rgba_dev, rgba_first_rows, nextModelID, allowNew, full_dev);
}

namespace {
struct CompData { unsigned char label; int top, right, bottom, left, size; };

// ConnectedLabels.hpp:50-172
int connectedLabels(const uint8_t* in, int cols, int rows, std::vector<int>& comp, std::vector<CompData>& stats)
{
    std::vector<int> roots;
    auto newComponent = [&roots]() { int r = (int)roots.size(); roots.push_back(r); return r; };
    auto findRoot = [&roots](int i) { while (i != roots[i]) i = roots[i]; return i; };
    comp.assign((size_t)cols * rows, 0);
    comp[0] = newComponent();
    for (int c = 1; c < cols; c++) comp[c] = (in[c] == in[c - 1]) ? comp[c - 1] : newComponent();
    for (int r = 1; r < rows; r++) {
        const uint8_t *row = in + (size_t)r * cols, *last = in + (size_t)(r - 1) * cols;
        int *cr = comp.data() + (size_t)r * cols, *lc = comp.data() + (size_t)(r - 1) * cols;
        cr[0] = (row[0] == last[0]) ? lc[0] : newComponent();
        for (int c = 1; c < cols; c++) {
            if (row[c] == row[c - 1]) {
                const int cLeft = cr[c - 1], cTop = lc[c];
                if (row[c] == last[c] && cLeft != cTop) {
                    const int r1 = findRoot(cTop), r2 = findRoot(cLeft);
                    if (r1 < r2) { roots[r2] = r1; cr[c] = r1; } else { roots[r1] = r2; cr[c] = r2; }
                } else cr[c] = cLeft;
            } else if (row[c] == last[c]) cr[c] = lc[c];
            else cr[c] = newComponent();
        }
    }
    std::vector<int> mapping(roots.size());
    int rootCnt = 0;
    for (int id = 0; id < (int)roots.size(); id++) {
        const int root = findRoot(id);
        if (root == id) mapping[root] = rootCnt++;
        else roots[id] = root;
    }
    for (auto& c : roots) c = mapping[c];
    stats.assign(rootCnt, CompData{0, 2147483647, 0, 0, 2147483647, 0});
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const int c = roots[comp[(size_t)y * cols + x]];
            comp[(size_t)y * cols + x] = c;
            CompData& d = stats[c];
            d.size++; d.label = in[(size_t)y * cols + x];
            if (y < d.top) d.top = y;
            if (y > d.bottom) d.bottom = y;
            if (x < d.left) d.left = x;
            if (x > d.right) d.right = x;
        }
    return rootCnt;
}

// Slic::downsample<float> normalisation incl. the empty-superpixel fallback (Slic.h:63-76, 192-206)
void finishMean(const int64_t* sumq, const uint32_t* cntOwn, const uint32_t* spixelCounts, const int32_t* resample, int K, float* out)
{
    for (int k = 0; k < K; k++) out[k] = (float)std::ldexp((double)sumq[k], -32);
    for (int k = 0; k < K; k++) {
        int cnt = (int)cntOwn[k], read = k;
        if (cnt == 0) { read = resample[k]; cnt = (int)spixelCounts[read]; }
        out[k] = out[read] / (float)cnt;
    }
}
}  // namespace

// Device-resident flavour (default): sums, unaries, mean field, component analysis and up-sampling are enqueued without a host wait;
// finishCRF() collects the decisions after the frame's one synchronisation.
void Segmentation::enqueueCRF(ModelList& models, const float* depth_dev, const uint8_t* rgba_dev, unsigned char nextModelID, bool allowNew,
                              uint8_t* full_dev)
{
    PhaseTimer pt(PhaseTimes::SegSlicAccumulate);
    const int n_models = (int)models.size();
    if (!slicStarted) check(ctx, cf_seg_slic(seg, rgba_dev), "cf_seg_slic");  // otherwise enqueued by startSlic() beside the tracking
    slicStarted = false;
    std::vector<const float*> icpPtr(n_models), vcPtr(n_models);
    std::vector<uint32_t> ids(n_models);
    {
        int m = 0;
        for (auto& mdl : models) {
            // a shadow contributes zeros here; its owner's sums arrive through the all-reduce below
            const bool mine = mdl->isOwned() && (!dist || dist->contributes(mdl->getID()));
            icpPtr[m] = mine ? mdl->icpErrorSurface() : zeroImage;
            vcPtr[m] = mine ? mdl->vertexConfProjection() : zeroImage;
            ids[m] = mdl->getID();
            m++;
        }
    }
    int64_t* sums_dev = nullptr; uint64_t words = 0;
    check(ctx, cf_seg_sums(seg, depth_dev, n_models, icpPtr.data(), vcPtr.data(), &sums_dev, &words), "cf_seg_sums");
    posesPublished = false;
    if (dist && dist->active()) {
        // the tracking launches are queued ahead of this point: the poses they leave in the trackers' device state travel in the tail of
        // the same block, so the frame needs ONE collective and no separate (blocking) pose exchange
        std::vector<cf_odom*> trackers(n_models, nullptr);
        int m = 0;
        for (auto& mdl : models) { if (mdl->isOwned() && dist->contributes(mdl->getID())) trackers[m] = mdl->getFrameOdometry(); m++; }
        check(ctx, cf_seg_publish_poses(seg, n_models, trackers.data()), "cf_seg_publish_poses");
        posesPublished = true;
        dist->sumDevice(ctx, sums_dev, words);  // exact: integer sums, every word has exactly one contributor
    }
    const cf_seg_params P = deviceParams();
    check(ctx, cf_seg_infer(seg, &P, rgba_dev, n_models, ids.data(), nextModelID, allowNew ? 1 : 0, full_dev), "cf_seg_infer");
    pendingModels = n_models;
}

cf_seg_params Segmentation::deviceParams() const
{
    cf_seg_params P{};
    P.unaryWeightError = unaryWeightError; P.unaryKError = unaryKError; P.unaryThresholdNew = unaryThresholdNew;
    P.weightAppearance = weightAppearance; P.weightSmoothness = weightSmoothness;
    P.scaleFeaturesRGB = scaleFeaturesRGB; P.scaleFeaturesDepth = scaleFeaturesDepth; P.scaleFeaturesPos = scaleFeaturesPos;
    P.minRelSizeNew = minRelSizeNew; P.maxRelSizeNew = maxRelSizeNew; P.crfIterations = (int)crfIterations;
    return P;
}

void Segmentation::collectCRF(ModelList& models, const float* depth_dev, const uint8_t* rgba_dev, unsigned char nextModelID, bool allowNew,
                              uint8_t* full_dev, cf_seg_job& job)
{
    if (dist && dist->active()) throw std::runtime_error("Segmentation::collectCRF: batched segmentation is for single-process sequences");
    const int n_models = (int)models.size();
    if (!slicStarted) check(ctx, cf_seg_slic(seg, rgba_dev), "cf_seg_slic");  // otherwise enqueued by startSlic() beside the tracking
    slicStarted = false;
    jobIcp.resize(n_models); jobConf.resize(n_models); jobIds.resize(n_models);
    int m = 0;
    for (auto& mdl : models) {
        jobIcp[m] = mdl->icpErrorSurface(); jobConf[m] = mdl->vertexConfProjection(); jobIds[m] = mdl->getID();
        m++;
    }
    posesPublished = false;
    job = cf_seg_job{};
    job.seg = seg; job.depth = depth_dev; job.n_models = n_models; job.icp_err = jobIcp.data(); job.vertconf4 = jobConf.data();
    job.rgba = rgba_dev; job.model_ids = jobIds.data(); job.next_model_id = nextModelID; job.allow_new = allowNew ? 1 : 0; job.full_dev = full_dev;
    pendingModels = n_models;
}

void Segmentation::runBatch(cf_ctx* ctx, const Segmentation& params, const std::vector<cf_seg_job>& jobs)
{
    if (jobs.empty()) return;
    PhaseTimer pt(PhaseTimes::SegSlicAccumulate);
    const cf_seg_params P = params.deviceParams();
    check(ctx, cf_seg_run_batch(ctx, &P, jobs.data(), (int)jobs.size()), "cf_seg_run_batch");
}

bool Segmentation::fetchPublishedPoses(size_t nModels, std::vector<int64_t>& words)
{
    if (!posesPublished || (int)nModels != pendingModels) return false;
    words.resize(nModels * 18);
    check(ctx, cf_seg_fetch_poses(seg, (int)nModels, words.data()), "cf_seg_fetch_poses");
    posesPublished = false;
    return true;
}

SegmentationResult Segmentation::finishCRF()
{
    PhaseTimer pt(PhaseTimes::SegPost);
    SegmentationResult result;
    cf_seg_result r{};
    const int K = (width / SPIX) * (height / SPIX);
    result.lowMap.resize(K);
    check(ctx, cf_seg_fetch(seg, &r, result.lowMap.data()), "cf_seg_fetch");
    result.hasNewLabel = r.has_new_label != 0;
    result.depthRange = r.depth_range;
    for (int i = 0; i < r.n_models; i++) {
        SegmentationResult::ModelData md;
        md.id = r.model[i].id; md.modelIndex = i < pendingModels ? i : -1;
        md.superPixelCount = r.model[i].superPixelCount; md.avgConfidence = r.model[i].avgConfidence;
        md.depthMean = r.model[i].depthMean; md.depthStd = r.model[i].depthStd;
        md.top = r.model[i].top; md.right = r.model[i].right; md.bottom = r.model[i].bottom; md.left = r.model[i].left;
        result.modelData.push_back(md);
    }
    return result;
}

// Host flavour (CF_SEG_HOST=1, diagnostics): the reference's host logic verbatim around the GPU SLIC / sums / mean field, with a
// host wait after each of them.  Same results as the device flavour.
SegmentationResult Segmentation::performSegmentationCRF(ModelList& models, const float* depth_dev, const uint8_t* rgba_dev,
                                                        const uint8_t* rgba_first_rows, unsigned char nextModelID, bool allowNew,
                                                        uint8_t* full_dev)
{
    SegmentationResult result;
    const int n_models = (int)models.size();
    const int numLabels = allowNew ? n_models + 1 : n_models;
    const float MAX_DEPTH = 100;
    const int gx = width / SPIX, gy = height / SPIX, K = gx * gy;

    std::unique_ptr<PhaseTimer> pt(new PhaseTimer(PhaseTimes::SegSlicAccumulate));
    if (!slicStarted) check(ctx, cf_seg_slic(seg, rgba_dev), "cf_seg_slic");  // otherwise enqueued by startSlic() beside the tracking
    slicStarted = false;
    std::vector<uint32_t> spc(K), dcnt(K);
    std::vector<int64_t> dsum(K), icpSum((size_t)K * n_models), confSum((size_t)K * n_models);
    std::vector<int32_t> resample(K);
    std::vector<const float*> icpPtr(n_models), vcPtr(n_models);
    {
        int m = 0;
        for (auto& mdl : models) {
            // a shadow contributes zeros here; its owner's sums arrive through the all-reduce below
            const bool mine = mdl->isOwned() && (!dist || dist->contributes(mdl->getID()));
            icpPtr[m] = mine ? mdl->icpErrorSurface() : zeroImage;
            vcPtr[m] = mine ? mdl->vertexConfProjection() : zeroImage;
            m++;
        }
    }
    check(ctx, cf_seg_accumulate(seg, depth_dev, n_models, icpPtr.data(), vcPtr.data(), spc.data(), dcnt.data(), dsum.data(), icpSum.data(),
                                 confSum.data(), resample.data()),
          "cf_seg_accumulate");
    if (dist && dist->active()) {  // exact: integer sums, every model has exactly one owner
        dist->sum(icpSum.data(), icpSum.size());
        dist->sum(confSum.data(), confSum.size());
    }
    pt.reset(new PhaseTimer(PhaseTimes::SegUnary));
    std::vector<float> lowDepth(K);
    finishMean(dsum.data(), dcnt.data(), spc.data(), resample.data(), K, lowDepth.data());
    float depthMin = 3.402823466e+38f, depthMax = 0;
    for (int i = 0; i < K; i++) {
        const float d = lowDepth[i];
        if (d > MAX_DEPTH || d < 0 || !std::isfinite(d)) continue;
        if (depthMax < d) depthMax = d;
        if (depthMin > d) depthMin = d;
    }
    result.depthRange = depthMax - depthMin;
    const float depthRange = result.depthRange;

    std::vector<std::vector<float>> lowICP(n_models, std::vector<float>(K)), lowConf(n_models, std::vector<float>(K));
    int modelIdToIndex[256];
    for (int i = 0; i < 256; i++) modelIdToIndex[i] = 0;
    {
        int m = 0;
        for (auto& mdl : models) {
            finishMean(icpSum.data() + (size_t)m * K, spc.data(), spc.data(), resample.data(), K, lowICP[m].data());
            finishMean(confSum.data() + (size_t)m * K, spc.data(), spc.data(), resample.data(), K, lowConf[m].data());
            SegmentationResult::ModelData md;
            md.id = mdl->getID(); md.modelIndex = m;
            modelIdToIndex[md.id & 255] = m;
            float avg = 0;
            for (int j = 0; j < K; j++) {
                float& c = lowConf[m][j];
                if (!std::isfinite(c)) { c = 0; continue; }
                avg += c;
            }
            md.avgConfidence = avg / (float)K;
            result.modelData.push_back(md);
            m++;
        }
    }
    if (allowNew) {
        modelIdToIndex[nextModelID] = n_models;
        SegmentationResult::ModelData md;
        md.id = nextModelID; md.modelIndex = -1;
        result.modelData.push_back(md);
    }
    int n_md = (int)result.modelData.size();

    const int L = numLabels;
    std::vector<float> unary((size_t)K * L);
    for (int k = 0; k < K; k++) {  // Segmentation.cpp:237-298
        if ((double)lowConf[0][k] < 0.3) lowICP[0][k] = (float)((double)depthRange * 0.01);
        for (int i = 1; i < n_models; i++)
            if ((double)lowConf[i][k] <= 0.4) lowICP[i][k] = depthRange * unaryKError;
        float lowestError = lowICP[0][k] / depthRange;
        for (int i = 0; i < n_models; i++) {
            float error = lowICP[i][k];
            error /= depthRange;
            if (error < lowestError) lowestError = error;
            unary[(size_t)k * L + i] = unaryWeightError * error;
        }
        if (allowNew) unary[(size_t)k * L + n_models] = std::fmax(unaryThresholdNew - unaryWeightError * lowestError, 0.01f);
    }
    std::vector<float> f1((size_t)K * 2), f2((size_t)K * 6);
    for (int j = 0; j < gy; j++)
        for (int i = 0; i < gx; i++) {
            const int index = j * gx + i;
            f1[index * 2 + 0] = (float)i / 2.0f; f1[index * 2 + 1] = (float)j / 2.0f;
            f2[index * 6 + 0] = (float)i * scaleFeaturesPos;
            f2[index * 6 + 1] = (float)j * scaleFeaturesPos;
            // colour features index the FULL-resolution image with the LOW-resolution index (Segmentation.cpp:445-447)
            f2[index * 6 + 2] = (float)rgba_first_rows[(size_t)index * 4 + 0] * scaleFeaturesRGB;
            f2[index * 6 + 3] = (float)rgba_first_rows[(size_t)index * 4 + 1] * scaleFeaturesRGB;
            f2[index * 6 + 4] = (float)rgba_first_rows[(size_t)index * 4 + 2] * scaleFeaturesRGB;
            f2[index * 6 + 5] = std::fmin(lowDepth[index] * scaleFeaturesDepth, 100.0f);
        }
    for (auto& u : unary) if (u <= 1e-5f) u = 1e-5f;
    std::vector<float> Q((size_t)K * L);
    pt.reset(new PhaseTimer(PhaseTimes::SegCrf));
    check(ctx, cf_seg_crf(seg, unary.data(), L, f1.data(), f2.data(), weightSmoothness, weightAppearance, (int)crfIterations, Q.data()), "cf_seg_crf");
    pt.reset(new PhaseTimer(PhaseTimes::SegPost));
    std::vector<uint8_t> map(K);
    for (int i = 0; i < K; i++) {
        int m = 0; float best = Q[(size_t)i * L];
        for (int l = 1; l < L; l++) if (Q[(size_t)i * L + l] > best) { best = Q[(size_t)i * L + l]; m = l; }
        map[i] = (uint8_t)result.modelData[m].id;
    }

    std::vector<int> comp; std::vector<CompData> cc;
    const int ncc = connectedLabels(map.data(), gx, gy, comp, cc);
    {  // onlyKeepLargest (Segmentation.cpp:496-517): every label but the smallest key keeps its largest component
        int minLabel = 256;
        for (int i = 0; i < ncc; i++) if (cc[i].label < minLabel) minLabel = cc[i].label;
        for (int lab2 = 0; lab2 < 256; lab2++) {
            if (lab2 == minLabel) continue;
            int keep = -1;
            for (int i = 0; i < ncc; i++) {
                if (cc[i].label != lab2) continue;
                if (keep < 0) { keep = i; continue; }
                if (cc[keep].size < cc[i].size) { cc[keep].label = 255; keep = i; } else cc[i].label = 255;
            }
        }
    }
    if (allowNew) {  // :521-530
        const int minSize = (int)((float)K * minRelSizeNew), maxSize = (int)((float)K * maxRelSizeNew);
        for (int i = 0; i < ncc; i++)
            if (cc[i].label == nextModelID && (cc[i].size < minSize || cc[i].size > maxSize)) cc[i].label = 255;
    }
    for (auto& md : result.modelData) {  // :532-547
        for (int i = 0; i < ncc; i++) {
            if (cc[i].label != (md.id & 255)) continue;
            if (cc[i].left < md.left) md.left = cc[i].left;
            if (cc[i].top < md.top) md.top = cc[i].top;
            if (cc[i].right > md.right) md.right = cc[i].right;
            if (cc[i].bottom > md.bottom) md.bottom = cc[i].bottom;
        }
        md.left = (unsigned short)(int)(md.left * SPIX + SPIX * 0.5); md.top = (unsigned short)(int)(md.top * SPIX + SPIX * 0.5);
        md.right = (unsigned short)(int)(md.right * SPIX + SPIX * 0.5); md.bottom = (unsigned short)(int)(md.bottom * SPIX + SPIX * 0.5);
    }
    {
        const unsigned borderSize = 20, fullHeight = (unsigned)height, fullWidth = (unsigned)width;  // :549-563
        for (auto& md : result.modelData) {
            if (md.id == 0) continue;
            const unsigned top = (unsigned)md.top, bottom = (unsigned)md.bottom, left = (unsigned)md.left, right = (unsigned)md.right;
            if ((top < borderSize && bottom < borderSize) || (left < borderSize && right < borderSize) ||
                (top > fullHeight - borderSize && bottom > fullHeight - borderSize) || (left > fullWidth - borderSize && right > fullWidth - borderSize))
                for (int i = 0; i < ncc; i++) if (cc[i].label == (md.id & 255)) cc[i].label = 255;
        }
    }
    for (int i = 0; i < K; i++) map[i] = cc[comp[i]].label;
    {  // depth statistics with one trimming pass (:570-621)
        std::vector<float> sumsDepth(n_md, 0.f), sumsDev(n_md, 0.f);
        std::vector<unsigned> cnts(n_md, 0);
        for (int i = 0; i < K; i++) { if (map[i] == 255) continue; const int ix = modelIdToIndex[map[i]]; sumsDepth[ix] += lowDepth[i]; cnts[ix]++; }
        for (int m = 0; m < n_md; m++) result.modelData[m].depthMean = cnts[m] ? sumsDepth[m] / (float)cnts[m] : 0;
        for (int i = 0; i < K; i++) { if (map[i] == 255) continue; const int ix = modelIdToIndex[map[i]]; sumsDev[ix] += std::fabs(result.modelData[ix].depthMean - lowDepth[i]); }
        for (int m = 0; m < n_md; m++) result.modelData[m].depthStd = cnts[m] ? sumsDev[m] / (float)cnts[m] : 0;
        for (int i = 0; i < K; i++) {
            if (map[i] == 255) continue;
            const int ix = modelIdToIndex[map[i]];
            if (ix != 0) {
                const float d = lowDepth[i];
                if ((double)d > 1.1 * (double)result.modelData[ix].depthStd + (double)result.modelData[ix].depthMean) {
                    sumsDepth[ix] -= d; sumsDev[ix] -= std::fabs(result.modelData[ix].depthMean - d); cnts[ix]--;
                }
            }
        }
        for (int m = 0; m < n_md; m++) {
            result.modelData[m].depthMean = cnts[m] ? sumsDepth[m] / (float)cnts[m] : 0;
            result.modelData[m].depthStd = cnts[m] ? sumsDev[m] / (float)cnts[m] : 0;
        }
    }
    for (int k = 0; k < K; k++) { if (map[k] == 255) continue; result.modelData[modelIdToIndex[map[k]]].superPixelCount++; }
    if (allowNew) {
        if (result.modelData.back().superPixelCount > 0) result.hasNewLabel = true;
        else result.modelData.pop_back();
    }
    check(ctx, cf_seg_upsample(seg, map.data(), full_dev), "cf_seg_upsample");
    result.lowMap = map;
    return result;
}

SegmentationResult Segmentation::performSegmentationGT(ModelList& models, const FrameData& frame, unsigned char nextModelID, bool allowNew,
                                                       uint8_t* full_dev)
{  // Segmentation.cpp:59-119 (host O(N); frame.mask / frame.depth are host buffers)
    SegmentationResult result;
    const size_t N = (size_t)width * height;
    std::vector<uint8_t> full(N, 0);
    unsigned outIds[256];
    int modelIdToIndex[256];
    memset(outIds, 0, sizeof(outIds));
    for (int i = 0; i < 256; i++) modelIdToIndex[i] = 0;
    int mIndex = 0;
    for (auto& m : models) modelIdToIndex[m->getID() & 255] = mIndex++;
    modelIdToIndex[nextModelID] = mIndex;
    for (size_t i = 0; i < N; i++) {
        const uint8_t vIn = frame.mask[i];
        if (vIn) {
            if (gtMapping[vIn] != 0) { full[i] = gtMapping[vIn]; outIds[full[i]]++; }
            else if (allowNew && !result.hasNewLabel) { full[i] = nextModelID; gtMapping[vIn] = nextModelID; result.hasNewLabel = true; outIds[full[i]]++; }
        } else outIds[0]++;
    }
    int idx = 0;
    for (auto& m : models) {
        SegmentationResult::ModelData md;
        md.id = m->getID(); md.modelIndex = idx++; md.superPixelCount = outIds[m->getID() & 255] / (16 * 16); md.avgConfidence = 0.4f;
        result.modelData.push_back(md);
    }
    if (result.hasNewLabel) {
        SegmentationResult::ModelData md;
        md.id = nextModelID; md.modelIndex = -1; md.avgConfidence = 0.4f;
        const float c = (float)(outIds[nextModelID] / (16 * 16));
        md.superPixelCount = (unsigned)(c > 1.0f ? c : 1.0f);
        result.modelData.push_back(md);
    }
    const int n_md = (int)result.modelData.size();
    std::vector<unsigned> cnts(n_md + 1, 0);
    for (size_t i = 0; i < N; i++) { const int ix = modelIdToIndex[full[i]]; result.modelData[ix].depthMean += frame.depth[i]; cnts[ix]++; }
    for (int m = 0; m < n_md; m++) result.modelData[m].depthMean /= cnts[m] ? (float)cnts[m] : 1.0f;
    for (size_t i = 0; i < N; i++) { const int ix = modelIdToIndex[full[i]]; result.modelData[ix].depthStd += std::fabs(result.modelData[ix].depthMean - frame.depth[i]); }
    for (int m = 0; m < n_md; m++) result.modelData[m].depthStd /= cnts[m] ? (float)cnts[m] : 1.0f;
    check(ctx, cf_memcpy_h2d(ctx, full_dev, full.data(), N), "mask upload");
    return result;
}

// ----------------------------------------------------------------------------- CoFusion ----
static cf_ctx* make_ctx(const CoFusion::Config& c)
{
    cf_config cc{};
    cc.width = c.width; cc.height = c.height; cc.fx = c.fx; cc.fy = c.fy; cc.cx = c.cx; cc.cy = c.cy; cc.device = c.device;
    cc.max_models = c.maxModels; cc.max_surfels = c.maxSurfels;
    cf_ctx* ctx = nullptr;
    const int rc = cf_create(&cc, &ctx);
    if (rc != CF_OK) {
        std::string msg = ctx ? cf_last_error(ctx) : "no HIP device (the Co-Fusion hot path has no CPU fallback)";
        if (ctx) cf_destroy(ctx);
        throw std::runtime_error("cf_create failed: " + msg);
    }
    return ctx;
}

// Helper threads for the per-model launch chains.  run(n, fn) calls fn(0..n-1), each index once, from the helpers and from the
// calling thread, and returns when all are done; the first exception is re-thrown on the caller.
class EnqueuePool {
public:
    explicit EnqueuePool(int helpers)
    {
        for (int i = 0; i < helpers; i++) workers.emplace_back([this] { loop(); });
    }
    ~EnqueuePool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        wake.notify_all();
        for (auto& w : workers) w.join();
    }
    int helpers() const { return (int)workers.size(); }
    void run(int count, const std::function<void(int)>& fn)
    {
        uint64_t g;
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &fn; total = count; next = 0; done = 0; error = nullptr; g = ++generation;
        }
        wake.notify_all();
        work(g);
        std::unique_lock<std::mutex> lk(mu);
        finished.wait(lk, [&] { return done == total; });
        job = nullptr;
        if (error) { std::exception_ptr e = error; error = nullptr; lk.unlock(); std::rethrow_exception(e); }
    }

private:
    void work(uint64_t g)
    {
        for (;;) {
            int i;
            const std::function<void(int)>* fn;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (g != generation || next >= total) return;
                i = next++; fn = job;
            }
            std::exception_ptr e;
            try { (*fn)(i); } catch (...) { e = std::current_exception(); }
            std::lock_guard<std::mutex> lk(mu);
            if (e && !error) error = e;
            if (++done == total) finished.notify_all();
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                wake.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
            }
            work(seen);
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable wake, finished;
    const std::function<void(int)>* job = nullptr;
    std::exception_ptr error;
    uint64_t generation = 0;
    int total = 0, next = 0, done = 0;
    bool stop = false;
};

CoFusion::CoFusion(const Config& c) : CoFusion(c, nullptr, 0) {}

CoFusion::CoFusion(const Config& c, cf_ctx* shared, int sequenceIndex)
    : cfg(c), ownsCtx(shared == nullptr), markBase(4 * sequenceIndex), ctx(shared ? shared : make_ctx(c))
{
    dist.rank = cfg.rank; dist.world = cfg.world < 1 ? 1 : cfg.world;
    dist.shardBackground = cfg.shardBackground && dist.world > 1;
    dist.colocate = cfg.colocateBackground;
    if (cfg.reloc && dist.world > 1) throw std::runtime_error("CoFusion: reloc (failure detection) is a single-GPU option (the background's normal matrix is not exchanged between ranks)");
    labelGenerator.reset(new Segmentation(ctx, cfg.width, cfg.height, &dist));
    const size_t N = (size_t)cfg.width * cfg.height;
    void* p = nullptr;
    check(ctx, cf_malloc(ctx, N * 4, &p), "cf_malloc"); depth_dev = static_cast<float*>(p);
    for (int b = 0; b < 2; b++) {
        check(ctx, cf_malloc(ctx, N * 4, &p), "cf_malloc"); depthFilteredBuf[b] = static_cast<float*>(p);
        check(ctx, cf_malloc(ctx, N, &p), "cf_malloc"); depthPyr1Buf[b] = static_cast<float*>(p);
        check(ctx, cf_malloc(ctx, N / 4, &p), "cf_malloc"); depthPyr2Buf[b] = static_cast<float*>(p);
    }
    depthFiltered_dev = depthFilteredBuf[0]; depthPyr1 = depthPyr1Buf[0]; depthPyr2 = depthPyr2Buf[0];
    check(ctx, cf_malloc(ctx, N * 4, &p), "cf_malloc"); rgba_dev = static_cast<uint8_t*>(p);
    check(ctx, cf_malloc(ctx, N * 3, &p), "cf_malloc"); rgb_dev = static_cast<uint8_t*>(p);
    check(ctx, cf_malloc(ctx, N, &p), "cf_malloc"); mask_dev = static_cast<uint8_t*>(p);  // zero-filled: the -static mask
    // host-input path: two pinned staging sets (depth f32 + rgb u8x3), so that copying frame t+1 into one does not wait
    // for the transfer of frame t out of the other
    for (int b = 0; b < 2; b++) {
        check(ctx, cf_malloc_host(ctx, N * 4 + N * 3, &p), "cf_malloc_host"); stage[b] = static_cast<uint8_t*>(p);
    }
    globalModel = std::make_shared<Model>(ctx, getNextModelID(true), cfg.confGlobalInit, true, cfg.maxSurfels, 3.402823466e+38f,
                                          dist.ownsHere(0));
    if (dist.shardBackground) {
        globalModel->shard = dist.rank; globalModel->shards = dist.world;
        // image rows of the ICP reduction: multiples of 4 (three pyramid levels)
        const int rows4 = cfg.height / 4, b = (rows4 * dist.rank / dist.world) * 4, e = (rows4 * (dist.rank + 1) / dist.world) * 4;
        if (globalModel->isOwned()) check(ctx, cf_odom_set_band(globalModel->getFrameOdometry(), b, e, dist.rank == 0 ? 1 : 0), "cf_odom_set_band");
    }
    globalModel->loggingPoses = cfg.enablePoseLogging;
    models.push_back(globalModel);
    int helpers = cfg.enqueueThreads;
    // (the helper threads only ever enqueue the per-model chains of the model-parallel modes: single-process operation runs all models'
    // passes as one chain of batched launches -- passesBatched -- and would never use them)
    if (helpers > 0 && useLanes && dist.active()) pool = std::make_shared<EnqueuePool>(helpers > 7 ? 7 : helpers);
}

CoFusion::~CoFusion()
{
    pool.reset();
    models.clear(); inactiveModels.clear(); newModel.reset(); globalModel.reset();
    cf_free(ctx, depth_dev);
    for (int b = 0; b < 2; b++) { cf_free(ctx, depthFilteredBuf[b]); cf_free(ctx, depthPyr1Buf[b]); cf_free(ctx, depthPyr2Buf[b]); }
    cf_free(ctx, rgba_dev); cf_free(ctx, rgb_dev); cf_free(ctx, mask_dev);
    for (int b = 0; b < 2; b++) if (stage[b]) cf_free_host(ctx, stage[b]);
    labelGenerator.reset();
    if (rcclStage) cf_free(ctx, rcclStage);
    if (ownsCtx) cf_destroy(ctx);
}

unsigned char CoFusion::getNextModelID(bool assign)
{  // CoFusion.cpp:628-644
    unsigned char next = nextID;
    if (assign) {
        if (models.size() == 256) throw std::range_error("getNextModelID(): Maximum amount of models is already in use (256).");
        while (true) {
            nextID++;
            bool isOccupied = false;
            for (auto& m : models) if (nextID == m->getID()) isOccupied = true;
            if (!isOccupied) break;
        }
    }
    return next;
}

void CoFusion::spawnObjectModel()
{  // CoFusion.cpp:588-598
    const unsigned char nid = getNextModelID(true);
    newModel = std::make_shared<Model>(ctx, nid, cfg.confObjectInit, false, cfg.maxSurfels, 3.402823466e+38f, dist.owner(nid) == dist.rank);
    newModel->loggingPoses = cfg.enablePoseLogging;
    if (newModel->isOwned()) check(ctx, cf_odom_init_first_rgb(newModel->getFrameOdometry(), curRgba), "initFirstRGB");
}
void CoFusion::moveNewModelToList()
{
    if (newModel) { models.push_back(newModel); newModel.reset(); }
}
ModelList::iterator CoFusion::inactivateModel(ModelList::iterator it)
{  // CoFusion.cpp:611-626
    ModelPointer m = *it;
    int64_t count = (int64_t)m->lastCount();  // known to the owner only: every rank needs it for the same decision
    dist.sum(&count, 1);
    if (!enableSmartModelDelete || ((unsigned)count >= modelKeepMinSurfels && m->getConfidenceThreshold() > modelKeepConfThreshold))
        inactiveModels.push_back(m);
    return --models.erase(it);
}

void CoFusion::predict(bool lastOfFrame)
{  // CoFusion.cpp:533-545
    // the models' predictions are independent of each other: one auxiliary stream per model lets them overlap
    const bool overlap = models.size() > 1 && useLanes;
    int lane = 0;
    for (auto& model : models) {
        if (overlap) check(ctx, cf_fork(ctx, lane++), "cf_fork");
        model->combinedPredict(maxDepthProcessed, tick, tick, cfg.timeDelta);
        if (lastOfFrame) model->prefetchFillRatio();  // requiresFillIn() of the next frame asks about THIS prediction
        model->performFillIn(curRgba, depthFiltered_dev, cfg.frameToFrameRGB, lost);
    }
    if (overlap) check(ctx, cf_join(ctx), "cf_join");
}

// One model's share of the frame's second half: CoFusion.cpp:316-330 (index map, fuse, index map, clean) and :350 + :533-545
// (end-of-frame prediction and fill-in).  Nothing here reads another model's buffers.
void CoFusion::modelPasses(Model& model, bool fuse, float weightMultiplier, bool lost)
{
    if (fuse) {
        // (the first index map of a tracked model was enqueued right behind its tracking, beside the segmentation: processFrame)
        model.predictIndices(tick, maxDepthProcessed, cfg.timeDelta);
        model.fuse(tick, curRgba, mask_dev, curDepth, depthFiltered_dev, maxDepthProcessed, weightMultiplier);
        model.predictIndices(tick, maxDepthProcessed, cfg.timeDelta);
        model.clean(tick, cfg.timeDelta, maxDepthProcessed, depthFiltered_dev, mask_dev, cfg.outlierCoefficient);
    }
    model.combinedPredict(maxDepthProcessed, tick, tick, cfg.timeDelta);
    model.prefetchFillRatio();  // requiresFillIn() of the next frame asks about THIS prediction
    model.performFillIn(curRgba, depthFiltered_dev, cfg.frameToFrameRGB, lost);
}
// The reference runs these passes in six loops over the models (CoFusion.cpp:316-330, :533-545).  The passes of different models
// touch disjoint buffers (shared inputs: frame, mask), so each model's whole chain goes to its own stream, without a barrier
// between fusion and prediction; with Config::enqueueThreads the chains are also ENQUEUED by different host threads (~19 launches
// per model: worth it when the host's launch rate is the limit).
// LOCK-STEP (round 4, single-process operation): the passes of all models as one chain of batched launches (cf_models_frame_passes) --
// every stage one launch whose workgroups are dealt to the models -- instead of one chain of ~16 launch-floor kernels per model on
// per-model streams (which share a few hardware queues: the five chains of configs[2] ran mostly one after another, 450 us of a
// 1.5 ms frame).  The model-parallel modes keep the per-model path: a rank's passes differ from the other ranks' and a split
// background talks to them from inside its index pass.
void CoFusion::frameFuseCollect(std::vector<cf_model_pass>& items)
{
    for (auto& m : models) {
        cf_model_pass it{};
        it.model = m->model; it.pose = m->pose.m; it.rgba = curRgba; it.mask = mask_dev; it.depth_raw = curDepth; it.depth_filtered = depthFiltered_dev;
        it.do_fuse = st.fuseNow ? 1 : 0; it.time = tick;
        it.fuse_max_depth = maxDepthProcessed < m->getMaxDepth() ? maxDepthProcessed : m->getMaxDepth();  // std::min(depthCutoff, maxDepth), Model.cpp:443
        it.weighting = m->computeFusionWeight(st.weightMultiplier); it.mask_id = (int)m->getID(); it.conf_threshold = m->getConfidenceThreshold();
        items.push_back(it);
    }
}
void CoFusion::frameFuseFinish()
{
    for (auto& m : models) {
        m->prefetchFillRatio();  // requiresFillIn() of the next frame asks about THIS prediction
        m->performFillIn(curRgba, depthFiltered_dev, cfg.frameToFrameRGB, lost);
    }
}

void CoFusion::fuseAndPredict(bool fuse, float weightMultiplier, bool lost, bool join, int laneOffset)
{
    if (passesBatched() && fuse == st.fuseNow && weightMultiplier == st.weightMultiplier) {
        std::vector<cf_model_pass> items;
        frameFuseCollect(items);
        check(ctx, cf_models_frame_passes(ctx, items.data(), (int)items.size(), maxDepthProcessed, cfg.outlierCoefficient, cfg.timeDelta), "cf_models_frame_passes");
        frameFuseFinish();
        return;
    }
    // (a sequence of a lock-step group puts even a single model's chain on a lane -- the chains of the OTHER sequences run beside it --
    // and leaves the join to the group)
    const bool overlap = (models.size() > 1 || !join) && useLanes;
    if (!overlap) {
        check(ctx, cf_join(ctx), "cf_join");  // (an index map enqueued on a lane before a deactivation left one model: order the stream after it)
        for (auto& model : models) modelPasses(*model, fuse, weightMultiplier, lost);
        return;
    }
    std::vector<Model*> list;
    for (auto& model : models) if (model->isOwned()) list.push_back(model.get());
    const int n = (int)list.size();
    // a split background talks to the other ranks from inside its passes (the caller's collective): keep that on this thread
    const bool threaded = pool && n > 1 && !dist.shardBackground && join;
    const int lanes = 6;  // lanes 6 and 7 belong to the frame head and the superpixel pass
    if (!threaded) {
        for (int i = 0; i < n; i++) {
            // a model whose first index map is already queued on a lane continues on that lane (stream order is the dependency)
            check(ctx, cf_fork(ctx, (laneOffset + i) % lanes), "cf_fork");
            modelPasses(*list[i], fuse, weightMultiplier, lost);
        }
        if (join) check(ctx, cf_join(ctx), "cf_join");
        else check(ctx, cf_main(ctx), "cf_main");
        return;
    }
    for (int i = 0; i < n && i < lanes; i++) check(ctx, cf_fork(ctx, i), "cf_fork");
    check(ctx, cf_main(ctx), "cf_main");
    struct Unbind { cf_ctx* c; ~Unbind() { cf_thread_lane(c, -1); } };
    std::exception_ptr failed;
    try {
        pool->run(n, [&](int i) {
            Unbind u{ctx};
            check(ctx, cf_thread_lane(ctx, i % lanes), "cf_thread_lane");
            modelPasses(*list[i], fuse, weightMultiplier, lost);
        });
    } catch (...) { failed = std::current_exception(); }
    check(ctx, cf_join(ctx), "cf_join");
    if (failed) std::rethrow_exception(failed);
}

// CoFusion.cpp:213-217 + Model::performTracking (Model.cpp:369-389), in two halves so that the trackers of SEVERAL sequences (a lock-step
// group, CoFusionGroup) can advance through the same launches: every sequence adds its owned models to a batch (trackCollect), the batch
// is prepared and tracked at once (trackLaunch), each sequence collects its own poses later (fetchTracking).
void CoFusion::trackCollect(TrackBatch& batch, const float* const depthPyr[3])
{
    Model* owner = nullptr;  // computes the frame-wide vertex / normal pyramids the other models of this sequence (and rank) share
    trackPending.clear();
    for (auto& m : models) {
        m->lastPose = m->pose;
        if (!m->isOwned()) continue;
        if (!owner) owner = m.get();
        TrackBatch::Item it;
        it.model = m.get(); it.owner = owner; it.frameRgba = curRgba; it.maxDepth = maxDepthProcessed;
        for (int l = 0; l < 3; l++) it.depthPyr[l] = depthPyr[l];
        // CoFusion::requiresFillIn (CoFusion.cpp:547-565) asks about the previous frame's last prediction.  Answering it here would make
        // the host wait for that frame to drain before this one can be enqueued; with the counts prefetched the preparation kernels
        // take the decision themselves and get both sets of inputs.
        it.altV = nullptr; it.altN = nullptr; it.altImg = nullptr;
        it.fillCounts = (cfg.width % 4 == 0 && cfg.height % 4 == 0) ? m->fillRatioDevice() : nullptr;
        if (it.fillCounts) {
            m->trackingInputs(false, cfg.frameToFrameRGB, it.predV, it.predN, it.predImg);
            m->trackingInputs(true, cfg.frameToFrameRGB, it.altV, it.altN, it.altImg);
        } else m->trackingInputs(m->requiresFillIn(), cfg.frameToFrameRGB, it.predV, it.predN, it.predImg);
        batch.items.push_back(it);
        trackPending.push_back(m.get());
    }
}

void CoFusion::trackLaunch(cf_ctx* ctx, TrackBatch& batch, const Config& cfg)
{
    const size_t n = batch.items.size();
    if (n == 0) return;
    // Model::initICP of every model (initICPModel + initRGBModel + initICP + initRGB), batched: one launch per preparation kernel for
    // all models of the batch
    std::vector<cf_odom*> ods; std::vector<const float*> pv, pn, pp, av, an; std::vector<const uint8_t*> pi, fr, ai; std::vector<const uint32_t*> fc;
    for (auto& it : batch.items) {
        ods.push_back(it.model->odom); pv.push_back(it.predV); pn.push_back(it.predN); pi.push_back(it.predImg); pp.push_back(it.model->pose.m);
        fr.push_back(it.frameRgba);
        av.push_back(it.altV); an.push_back(it.altN); ai.push_back(it.altImg); fc.push_back(it.fillCounts);
    }
    check(ctx, cf_odom_init_models_batch_select(ctx, ods.data(), (int)n, pv.data(), pn.data(), pi.data(), av.data(), an.data(), ai.data(), fc.data(),
                                                0.75f /* Model::requiresFillIn's default ratio */, pp.data(), fr.data()), "init_models_batch");
    for (auto& it : batch.items) it.model->bindFrameMaps(it.depthPyr, it.maxDepth, it.owner);
    cf_track_opts opts{};
    opts.rgb_only = cfg.rgbOnly; opts.pyramid = cfg.pyramid; opts.fast_odom = cfg.fastOdom; opts.so3 = cfg.so3; opts.icp_weight = cfg.icpWeight;
    // lock-step launches of at most 16 trackers (kMaxBatch, csrc/cf_kernels.h); the chunks hold different trackers, so chunk k + 1 is
    // prepared and enqueued while chunk k runs (round 6: the drain between chunks is per tracker, not per context); everything is left
    // in flight -- fetchTracking collects it after whatever the caller enqueues behind it
    const size_t B = 16;
    for (size_t base = 0; base < n; base += B) {
        const int k_n = (int)std::min(B, n - base);
        cf_odom* o[B]; const float* poses[B]; float* errs[B];
        for (int k = 0; k < k_n; k++) { Model* m = batch.items[base + k].model; o[k] = m->odom; poses[k] = m->pose.m; errs[k] = m->icpError; }
        check(ctx, cf_odom_track_batch_async(ctx, o, k_n, poses, &opts, errs), "track_batch");
    }
}

void CoFusion::trackModels(const float* const depthPyr[3])
{
    TrackBatch batch;
    trackCollect(batch, depthPyr);
    trackLaunch(ctx, batch, cfg);
}

void CoFusion::fetchTracking(bool exchange)
{
    for (Model* m : trackPending) {
        float t[3], R[9];
        check(ctx, cf_odom_fetch_result(m->odom, t, R, &m->lastStats), "fetch_result");
        for (int r = 0; r < 3; r++) { m->pose.m[r * 4 + 0] = R[r * 3 + 0]; m->pose.m[r * 4 + 1] = R[r * 3 + 1]; m->pose.m[r * 4 + 2] = R[r * 3 + 2]; m->pose.m[r * 4 + 3] = t[r]; }
    }
    // the image swap of a tracked frame (RGBDOdometry.cpp:469-473) -- on every rank, also one that owns no tracker this frame: a model
    // it is given later starts from the last TRACKED frame's image like everybody else's
    trackPending.clear();
    if (exchange) exchangeTracking();
}

void CoFusion::exchangeTracking()
{   // owners publish pose + ICP statistics as bit patterns (one 64-bit slot per float), shadows contribute zeros
    if (!dist.active()) return;
    const int R = 16 + 2;
    std::vector<int64_t> buf;
    size_t k = 0;
    // with the device segmentation in flight the poses came along with its sums (Segmentation::enqueueCRF); otherwise (ground-truth
    // masks, single-model mode, host segmentation) a collective of its own
    if (!labelGenerator->fetchPublishedPoses(models.size(), buf)) {
        buf.assign(models.size() * R, 0);
        for (auto& m : models) {
            if (m->isOwned() && dist.contributes(m->getID())) {
                for (int i = 0; i < 16; i++) { uint32_t b; memcpy(&b, &m->pose.m[i], 4); buf[k * R + i] = (int64_t)b; }
                uint32_t b; memcpy(&b, &m->lastStats.last_icp_error, 4); buf[k * R + 16] = (int64_t)b;
                memcpy(&b, &m->lastStats.last_icp_count, 4); buf[k * R + 17] = (int64_t)b;
            }
            k++;
        }
        dist.sum(buf.data(), buf.size());
    }
    k = 0;
    for (auto& m : models) {
        if (!m->isOwned()) {
            for (int i = 0; i < 16; i++) { const uint32_t b = (uint32_t)buf[k * R + i]; memcpy(&m->pose.m[i], &b, 4); }
            uint32_t b = (uint32_t)buf[k * R + 16]; memcpy(&m->lastStats.last_icp_error, &b, 4);
            b = (uint32_t)buf[k * R + 17]; memcpy(&m->lastStats.last_icp_count, &b, 4);
        }
        k++;
    }
}

// CoFusion::processFrame (Core/CoFusion.cpp:171-524) as a sequence of stages.  One sequence runs them back to back (processFrame);
// a lock-step group of sequences (CoFusionGroup) runs each stage for all its sequences before the next, with ONE set of tracking
// launches for the trackers of all of them.
//   frameBegin   upload, depth filter + pyramid, first-frame initialisation, superpixels started aside
//   trackCollect / trackLaunch   (above)
//   frameMiddle  segmentation enqueued, the frame's host wait (poses + segmentation decisions), model bookkeeping
//   frameFuse    per-model fusion / clean-up / prediction chains
//   frameEnd     clock, pose log
void CoFusion::frameBegin(const FrameData& frame, const Mat4f* inPose, float weightMultiplier, bool bootstrap)
{
    st = FrameStage{};
    st.frame = &frame; st.inPose = inPose; st.weightMultiplier = weightMultiplier; st.bootstrap = bootstrap;
    const size_t N = (size_t)cfg.width * cfg.height;
    if (ownsCtx) check(ctx, cf_join(ctx), "cf_join");  // a previous call that threw inside a forked region must not leave the context on a lane
    // upload (CoFusion.cpp:179-184); RGB -> RGBA like the GL_RGBA texture upload
    if (frame.rgba_dev && frame.depth_dev) { curRgba = frame.rgba_dev; curDepth = frame.depth_dev; }
    else {
        // one memcpy into pinned staging, transfers only enqueued (no host wait), RGB -> RGBA on the device.  The staging set
        // used two frames ago is free again once the stream has passed that frame's transfer (stageMark).
        const unsigned sb = uploads & 1u;
        if (uploads >= 2) check(ctx, cf_event_wait_host(ctx, markBase + 2 + (int)sb), "staging wait");
        uint8_t* sp = stage[sb];
        memcpy(sp, frame.depth, N * 4);
        memcpy(sp + N * 4, frame.rgb, N * 3);
        check(ctx, cf_memcpy_h2d_async(ctx, depth_dev, sp, N * 4), "depth upload");
        check(ctx, cf_memcpy_h2d_async(ctx, rgb_dev, sp + N * 4, N * 3), "rgb upload");
        check(ctx, cf_mark(ctx, markBase + 2 + (int)sb), "cf_mark");
        check(ctx, cf_rgb_to_rgba(ctx, rgb_dev, cfg.width, cfg.height, rgba_dev), "rgb expand");
        uploads++;
        curRgba = rgba_dev; curDepth = depth_dev;
    }
    // filterDepth + the depth pyramid only read the new frame.  With a device-resident frame they go to an auxiliary stream that
    // does not wait for the fusion passes of the previous frame still queued on the main stream (those read the OTHER buffer set);
    // mark[b] = end of the last frame that used buffer set b, which is all this lane has to wait for.
    const bool willTrack = tick > 1 && (bootstrap || !inPose);
    const unsigned b = frameParity & 1u;
    st.willTrack = willTrack; st.b = b;
    frameParity++;
    depthFiltered_dev = depthFilteredBuf[b]; depthPyr1 = depthPyr1Buf[b]; depthPyr2 = depthPyr2Buf[b];
    const bool headAside = useLanes && cfg.deviceFramesComplete && frame.depth_dev != nullptr;
    if (headAside) check(ctx, cf_fork_after(ctx, 6, markBase + (int)b), "cf_fork_after");
    check(ctx, cf_bilateral(ctx, curDepth, cfg.width, cfg.height, cfg.depthCutoff, depthFiltered_dev), "filterDepth");
    if (willTrack) check(ctx, cf_depth_pyramid(ctx, depthFiltered_dev, cfg.width, cfg.height, depthPyr1, depthPyr2), "generateCUDATextures");
    if (headAside) check(ctx, cf_join_lane(ctx, 6), "cf_join_lane");

    st.pyr[0] = depthFiltered_dev; st.pyr[1] = depthPyr1; st.pyr[2] = depthPyr2;
    if (tick == 1) {
        globalModel->initialise(curRgba, curDepth, depthFiltered_dev, tick, maxDepthProcessed);
        if (globalModel->isOwned()) check(ctx, cf_odom_init_first_rgb(globalModel->getFrameOdometry(), curRgba), "initFirstRGB");
    } else if (willTrack) {
        // the superpixels only depend on the colour image: SLIC runs on an auxiliary stream beside the (latency-bound)
        // tracking launches and is joined before the segmentation needs it
        st.slicAside = cfg.enableMultipleModels && !frame.mask && useLanes;
        if (st.slicAside) {
            check(ctx, cf_fork(ctx, 7), "cf_fork");
            labelGenerator->startSlic(curRgba);
            check(ctx, cf_main(ctx), "cf_main");
        }
    }
}

void CoFusion::frameSegment(std::vector<cf_seg_job>* jobs)
{
    const FrameData& frame = *st.frame;
    const Mat4f* inPose = st.inPose;
    const bool bootstrap = st.bootstrap;
    st.fuseNow = false;
    if (tick == 1) return;
    if (bootstrap || !inPose) {
        if (st.slicAside) check(ctx, cf_join_lane(ctx, 7), "cf_join_lane");
        // a new label needs a free model slot: the context (trackers' staging, the segmenter's label dimension) was sized for
        // cfg.maxModels models, at most 255 -- model ids are 8 bits and 255 marks a rejected superpixel (CoFusion.cpp:631-634)
        bool allowNew = false;
        const bool segOnDevice = cfg.enableMultipleModels && !frame.mask;
        if (cfg.enableMultipleModels) {
            if (spawnOffset < cfg.modelSpawnOffset) spawnOffset++;
            const size_t modelCap = (size_t)std::min(cfg.maxModels, 255);
            allowNew = spawnOffset >= cfg.modelSpawnOffset && models.size() < modelCap;
            if (spawnOffset >= cfg.modelSpawnOffset && !allowNew && !capReported) {  // say so once: the reference would go on to 256 ids
                fprintf(stderr, "[cofusion] %zu active models: the model cap (min(max_models, 255)) suppresses further spawns\n", models.size());
                capReported = true;
            }
        }
        // the motion segmentation reads device data only (ICP error surfaces, predictions): enqueued right behind the tracking
        // launches, so that poses AND segmentation decisions are collected by ONE host wait
        // (a sequence of a lock-step group only describes its chain: the group issues the chains of all sequences as ONE)
        if (segOnDevice && jobs) {
            jobs->emplace_back();
            labelGenerator->collectCRF(models, curDepth, curRgba, getNextModelID(), allowNew, mask_dev, jobs->back());
        } else if (segOnDevice) labelGenerator->enqueueCRF(models, curDepth, curRgba, getNextModelID(), allowNew, mask_dev);
        st.allowNew = allowNew; st.segOnDevice = segOnDevice;
    }
}

void CoFusion::frameCollect()
{
    const FrameData& frame = *st.frame;
    const Mat4f* inPose = st.inPose;
    const bool bootstrap = st.bootstrap;
    const size_t N = (size_t)cfg.width * cfg.height;
    st.fuseNow = false;
    if (tick == 1) return;
    bool trackingOk = true;
    if (bootstrap || !inPose) {
        const bool allowNew = st.allowNew, segOnDevice = st.segOnDevice;
        { PhaseTimer t(PhaseTimes::Track); fetchTracking(true); }
        if (bootstrap) globalModel->overridePose(globalModel->getPose() * (*inPose));
        if (cfg.reloc) {  // CoFusion.cpp:225 and 301-338 (nothing in between reads trackingCount / lost: evaluated together)
            const cf_track_stats& gs = globalModel->lastStats;
            trackingOk = (double)gs.last_icp_error < 1e-04;
            if (!lost) {
                double cov[36];
                check(ctx, cf_odom_get_covariance(&gs, cov), "cf_odom_get_covariance");
                for (int i = 0; i < 6; i++)
                    if (cov[i * 6 + i] > 1e-04) { trackingOk = false; break; }
                if (!trackingOk) {
                    if (++trackingCount > 10) lost = true;
                } else trackingCount = 0;
            }
            // (lastFrameRecovery, :321-337, is only ever set by the fern database, :365-367 -- closeLoops is off in Co-Fusion and the
            // database out of scope: a lost camera stays lost)
        }

        if (cfg.enableMultipleModels) {
            auto getMaxDepth = [](const SegmentationResult::ModelData& d) -> float { return d.depthMean + d.depthStd * 1.2; };
            SegmentationResult seg;
            if (segOnDevice) seg = labelGenerator->finishCRF();
            else {
                // the colour features of the CRF read the first K pixels of the full-resolution image
                const int K = (cfg.width / 16) * (cfg.height / 16);
                std::vector<uint8_t> firstRows((size_t)K * 4);
                if (frame.rgba_dev) check(ctx, cf_memcpy_d2h(ctx, firstRows.data(), curRgba, (size_t)K * 4), "rgb readback");
                else for (int i = 0; i < K; i++) { firstRows[i * 4] = frame.rgb[i * 3]; firstRows[i * 4 + 1] = frame.rgb[i * 3 + 1]; firstRows[i * 4 + 2] = frame.rgb[i * 3 + 2]; firstRows[i * 4 + 3] = 255; }
                seg = labelGenerator->performSegmentation(models, frame, curDepth, curRgba, firstRows.data(), getNextModelID(), allowNew, mask_dev);
            }
            if (!exportSegmentationPrefix.empty()) {  // CoFusion.cpp:235-240: labels > 254 (rejected) are written as 0
                std::vector<uint8_t> labels(N);
                check(ctx, cf_memcpy_d2h(ctx, labels.data(), mask_dev, N), "mask readback");
                for (auto& v : labels) if (v > 254) v = 0;
                writePngGray8(exportSegmentationPrefix + "Segmentation" + std::to_string(tick) + ".png", labels.data(), cfg.width, cfg.height);
            }
            if (seg.hasNewLabel) {
                spawnObjectModel();
                spawnOffset = 0;
                newModel->setMaxDepth(getMaxDepth(seg.modelData.back()));
            }
            {
                auto it = models.begin();
                for (unsigned i = 1; i < models.size(); i++) (*++it)->setMaxDepth(getMaxDepth(seg.modelData[i]));
            }
            if (seg.hasNewLabel) {
                newModel->predictIndices(tick, maxDepthProcessed, cfg.timeDelta);
                newModel->fuse(tick, curRgba, mask_dev, curDepth, depthFiltered_dev, maxDepthProcessed, 100);
                newModel->clean(tick, cfg.timeDelta, maxDepthProcessed, depthFiltered_dev, mask_dev, cfg.outlierCoefficient);
                moveNewModelToList();
            }
            for (auto& md : seg.modelData) {  // CoFusion.cpp:284-291
                if (md.superPixelCount <= 0) {
                    auto it = models.begin();
                    std::advance(it, md.modelIndex);
                    if ((*it)->incrementUnseenCount() > 0 && md.id != 0) {
                        inactivateModel(it);
                        // later entries referred to list positions that have now shifted by one
                        for (auto& o : seg.modelData) if (o.modelIndex > md.modelIndex) o.modelIndex--;
                    }
                }
            }
            {
                auto it = models.begin();
                for (unsigned i = 1; i < models.size(); i++) {  // :294-298 (indices into modelData are NOT re-aligned, as in the reference)
                    const float oldConf = (*++it)->getConfidenceThreshold();
                    (*it)->setConfidenceThreshold(std::fmin(std::fmax(oldConf, seg.modelData[i].avgConfidence), 9.0f));
                }
            }
        }
    } else {
        globalModel->overridePose(*inPose);
    }
    // CoFusion.cpp:346 predicts every model here, between tracking and fusion.  Nothing in the frame loop reads that prediction: fuse
    // and clean work on the index maps, the segmentation read the PREVIOUS prediction before this point, and the prediction at the
    // end of the frame overwrites all of it -- in the reference it only feeds the GUI.  Off by default (Config::midFramePredict).
    if (cfg.midFramePredict) { PhaseTimer t(PhaseTimes::Predict); predict(); }
    st.fuseNow = !cfg.rgbOnly && trackingOk && !lost;
}

void CoFusion::frameFuse(bool join, int laneOffset)
{
    PhaseTimer t(st.fuseNow ? PhaseTimes::Fuse : PhaseTimes::Predict);
    fuseAndPredict(st.fuseNow, st.weightMultiplier, lost, join, laneOffset);
}

void CoFusion::frameEnd()
{
    const FrameData& frame = *st.frame;
    check(ctx, cf_mark(ctx, markBase + (int)st.b), "cf_mark");  // everything that reads this frame's filtered depth is enqueued
    phaseTimes().frames++;
    if (!lost) tick++;
    moveNewModelToList();

    bool first = true;
    for (auto& model : models) {  // pose log, CoFusion.cpp:502-520
        if (!model->isLoggingPoses()) { first = false; continue; }
        const Mat4f p = first ? globalModel->getPose() : globalModel->getPose() * model->getPose().inverse();
        Model::PoseLogItem item;
        item.ts = frame.timestamp;
        item.p[0] = p.m[3]; item.p[1] = p.m[7]; item.p[2] = p.m[11];
        // quaternion of the rotation block (Eigen::Quaternionf(rotObject)), Shepperd's method
        const float m00 = p.m[0], m11 = p.m[5], m22 = p.m[10];
        float t = m00 + m11 + m22, qx, qy, qz, qw;
        if (t > 0) { t = std::sqrt(t + 1.0f); qw = 0.5f * t; t = 0.5f / t; qx = (p.m[9] - p.m[6]) * t; qy = (p.m[2] - p.m[8]) * t; qz = (p.m[4] - p.m[1]) * t; }
        else {
            int i = 0;
            if (m11 > m00) i = 1;
            if (m22 > p.m[i * 5]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(p.m[i * 5] - p.m[j * 5] - p.m[k * 5] + 1.0f);
            float q[3];
            q[i] = 0.5f * t; t = 0.5f / t;
            qw = (p.m[k * 4 + j] - p.m[j * 4 + k]) * t;
            q[j] = (p.m[j * 4 + i] + p.m[i * 4 + j]) * t;
            q[k] = (p.m[k * 4 + i] + p.m[i * 4 + k]) * t;
            qx = q[0]; qy = q[1]; qz = q[2];
        }
        item.p[3] = qx; item.p[4] = qy; item.p[5] = qz; item.p[6] = qw;
        model->poseLog.push_back(item);
        first = false;
    }
}

bool CoFusion::processFrame(const FrameData& frame, const Mat4f* inPose, float weightMultiplier, bool bootstrap)
{
    frameBegin(frame, inPose, weightMultiplier, bootstrap);
    if (st.willTrack) { PhaseTimer t(PhaseTimes::Track); trackModels(st.pyr); }
    frameSegment(nullptr);
    framePreIndex();
    frameCollect();
    frameFuse(true, 0);
    frameEnd();
    return false;
}

// The first index maps of the surfel chain (Model::predictIndices before Model::fuse, CoFusion.cpp:316-318) enqueued BEHIND the
// segmentation and BEFORE the frame's host wait, with the poses the trackers left on the device: the GPU rasterises them while the host
// collects poses and decisions and prepares the rest of the chain (cf_models_preindex).  Single-process operation only (the chain of
// batched launches); a model that is not fused afterwards has had its index map overwritten early -- nothing reads it in between.
void CoFusion::framePreIndex()
{
    if (!cfg.earlyIndexMaps || !passesBatched() || trackPending.empty() || cfg.rgbOnly || lost) return;
    std::vector<cf_model_preindex> items;
    for (Model* m : trackPending) items.push_back(cf_model_preindex{m->model, m->odom, (int)tick});
    check(ctx, cf_models_preindex(ctx, items.data(), (int)items.size(), maxDepthProcessed, cfg.timeDelta), "cf_models_preindex");
}

// ------------------------------------------------------------------------ CoFusionGroup ----
// Several independent RGB-D sequences on ONE GPU, advancing in lock-step.  Most kernels of the hot path run at their launch floor
// (a 640x480 frame is a few MB: ~5 us of launch latency per pass against 1-2 us of streaming), so a second sequence in a context of
// its own buys little (the contexts compete for the hardware queues: 1.45x with 2-4 contexts).  Here the sequences share
// ONE context and every set of tracking launches -- map preparation, SO(3) pre-alignment, the 57 launches of the Gauss-Newton loop
// -- carries the trackers of ALL sequences (grid.y = tracker, as for the models of one frame), the per-model surfel chains of all
// sequences share the context's lanes, and each sequence keeps its own maps, segmentation and clock.  Results per sequence are those
// of a CoFusion of its own, bit for bit (the reductions are exact integer sums, independent of what else is in the launch).
CoFusionGroup::CoFusionGroup(const CoFusion::Config& c, int sequences) : cfg(c)
{
    if (sequences < 1 || sequences > 16) throw std::runtime_error("CoFusionGroup: 1..16 sequences");
    if (c.world > 1) throw std::runtime_error("CoFusionGroup: a group lives on one GPU (world == 1)");
    CoFusion::Config shared = c;
    shared.maxModels = c.maxModels * sequences;  // trackers of all sequences in the context's staging
    ctx = make_ctx(shared);
    try {
        for (int s = 0; s < sequences; s++) seqs.emplace_back(new CoFusion(c, ctx, s));
    } catch (...) { seqs.clear(); cf_destroy(ctx); throw; }
}

CoFusionGroup::~CoFusionGroup()
{
    seqs.clear();
    cf_destroy(ctx);
}

void CoFusionGroup::processFrames(const FrameData* frames, const Mat4f* const* inPoses)
{
    // A stage that throws leaves the sequences at different points of the frame (some tracked but not fused, clocks apart, tracking
    // results pending): the "bit-identical to separate instances" guarantee is gone and cannot be restored from here, so the group
    // refuses further frames instead of silently continuing.
    if (failed) throw std::runtime_error("CoFusionGroup: a previous processFrames failed part-way; the sequences are out of step (destroy the group)");
    try { stepAll(frames, inPoses); }
    catch (...) {
        failed = true;
        (void)cf_join(ctx);  // leave no forked lane behind
        throw;
    }
}

void CoFusionGroup::stepAll(const FrameData* frames, const Mat4f* const* inPoses)
{
    const int S = (int)seqs.size();
    for (int s = 0; s < S; s++) seqs[s]->frameBegin(frames[s], inPoses ? inPoses[s] : nullptr, 1.f, false);
    {   // ONE set of tracking launches for the trackers of every sequence that tracks this frame
        PhaseTimer t(PhaseTimes::Track);
        CoFusion::TrackBatch batch;
        for (int s = 0; s < S; s++) if (seqs[s]->frameTracks()) seqs[s]->trackCollect(batch);
        CoFusion::trackLaunch(ctx, batch, cfg);
    }
    // the segmentation chains of the multi-object sequences as ONE chain of batched launches (each chain is ~30 launch-floor kernels:
    // alone it leaves the GPU idle, and on lanes side by side the chains only stretched each other), then ONE host wait for the poses
    // and the decisions of all sequences
    {
        std::vector<cf_seg_job> jobs;
        std::vector<int> owner;   // the sequence every job belongs to
        jobs.reserve((size_t)S);
        for (int s = 0; s < S; s++) {
            const size_t before = jobs.size();
            seqs[s]->frameSegment(&jobs);
            if (jobs.size() > before) owner.push_back(s);
        }
        // cf_seg_run_batch takes ONE parameter set per chain: sequences whose CRF settings differ (cofusion_set_crf on a borrowed handle)
        // get chains of their own, so that every sequence still equals a separate instance bit for bit
        std::vector<char> done(jobs.size(), 0);
        for (size_t a = 0; a < jobs.size(); a++) {
            if (done[a]) continue;
            const cf_seg_params Pa = seqs[owner[a]]->segmentation().deviceParams();
            std::vector<cf_seg_job> chain;
            for (size_t b = a; b < jobs.size(); b++) {
                if (done[b]) continue;
                const cf_seg_params Pb = seqs[owner[b]]->segmentation().deviceParams();
                if (memcmp(&Pa, &Pb, sizeof(Pa)) == 0) { chain.push_back(jobs[b]); done[b] = 1; }
            }
            Segmentation::runBatch(ctx, seqs[owner[a]]->segmentation(), chain);
        }
    }
    for (int s = 0; s < S; s++) seqs[s]->frameCollect();
    {   // the surfel passes of ALL sequences' models in one chain of batched launches
        PhaseTimer t(PhaseTimes::Fuse);
        std::vector<cf_model_pass> items;
        for (int s = 0; s < S; s++) seqs[s]->frameFuseCollect(items);
        check(ctx, cf_models_frame_passes(ctx, items.data(), (int)items.size(), seqs[0]->depthLimit(), cfg.outlierCoefficient, cfg.timeDelta),
              "cf_models_frame_passes");
        for (int s = 0; s < S; s++) seqs[s]->frameFuseFinish();
    }
    for (int s = 0; s < S; s++) seqs[s]->frameEnd();
}

}  // namespace cofusion
