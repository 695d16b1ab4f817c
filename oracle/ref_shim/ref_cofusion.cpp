// ref_cofusion.cpp -- the rest of the translation unit that holds the reference's own frame loop (see stub/CoFusionPin.h),
// TEST INFRASTRUCTURE ONLY: the members of the stand-in classes that are not the reference's text, and flat C entry points.
// Every image-space / surfel / tracking pass is the CPU oracle's (oracle/orc.h); Model::performTracking, Model::fuse's weighting
// and depth limit, Model::clean's arguments restate Core/Model/Model.cpp:345-389, 408-420, 565-580 (an OpenGL class that cannot be
// compiled here), the constructor restates CoFusion.cpp:21-77 and the GUI defaults MainController applies every frame.
extern "C" {
#include "orc.h"
}
#include <math.h>
#include <string.h>

// the reference's own RGBDOdometry class (ref_odo.cpp, another translation unit of this library: CUDA kernels under the emulator, f32 tree
// reductions): optional tracker of the frame loop, for the trajectory-level comparison with the exact-integer arithmetic of the oracle
extern "C" {
void* ref_odo_create(int w, int h, float cx, float cy, float fx, float fy);
void ref_odo_destroy(void* p);
void ref_odo_init_first_rgb(void* p, const unsigned char* rgba);
void ref_odo_init_icp_model(void* p, const float* v4, const float* n4, float cutoff, const float* pose_row_major);
void ref_odo_init_rgb_model(void* p, const unsigned char* rgba);
void ref_odo_init_icp(void* p, const float* const* depth_pyr, float cutoff);
void ref_odo_init_rgb(void* p, const unsigned char* rgba);
void ref_odo_track(void* p, float* trans, float* rot_row_major, int rgb_only, float icp_weight, int pyramid, int fast_odom, int so3,
                   float* icp_err_surface, float* stats6, double* lastA36, double* lastb6);
}

namespace {
bool g_reference_tracker = false;             // ref_cf_use_reference_tracker
bool g_reloc = false;                         // ref_cf_set_reloc: the constructor's `reloc` argument (CoFusion.h:47) of the next instance
orc_cam g_cam;
int g_w = 0, g_h = 0;
float g_outlier = 3.0f;                       // GUI default of the outlier coefficient (GUI.h:213), set through Model::GPUSetup
std::vector<float> g_depth_pyr[3];            // Model::GPUSetup::depth_tmp (generateCUDATextures)

void to_row_major(const Eigen::Matrix4f& m, float out[16]) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = m(i, j); }
}  // namespace

struct ModelImpl {
    int count = 0;
    std::vector<float> surfels, scratch, new_unstable;
    int n_new = 0;
    std::vector<unsigned char> img;            // combinedPredict: image RGBA8, vertexConf, normalRad, time
    std::vector<float> vc, nr;
    std::vector<uint16_t> tm;
    std::vector<uint32_t> idx;                 // predictIndices
    std::vector<float> ivc, ict, inr;
    std::vector<float> fv, fn;                 // fill-in textures
    std::vector<unsigned char> fi;
    std::vector<float> icp_error;
    std::unique_ptr<GPUTexture> rgbProjection;
    std::unique_ptr<PinOdometry> odom;
    ModelImpl()
    {
        const size_t N = (size_t)g_w * g_h;
        img.assign(N * 4, 0); vc.assign(N * 4, 0.f); nr.assign(N * 4, 0.f); tm.assign(N, 0);
        idx.assign(N, 0); ivc.assign(N * 4, 0.f); ict.assign(N * 4, 0.f); inr.assign(N * 4, 0.f);
        fv.assign(N * 4, 0.f); fn.assign(N * 4, 0.f); fi.assign(N * 4, 0);
        icp_error.assign(N, 0.f);
        new_unstable.assign((N / 4 + 16) * 12, 0.f);
        rgbProjection.reset(new GPUTexture(img.data(), g_w, g_h));
        odom.reset(new PinOdometry(g_w, g_h, g_cam.cx, g_cam.cy, g_cam.fx, g_cam.fy));
    }
};

PinOdometry::PinOdometry(int width, int height, float cx, float cy, float fx, float fy)
{
    orc = orc_odom_create(width, height, cx, cy, fx, fy);
    if (g_reference_tracker) ref = ref_odo_create(width, height, cx, cy, fx, fy);
}
PinOdometry::~PinOdometry() { orc_odom_destroy((orc_odometry*)orc); if (ref) ref_odo_destroy(ref); }
Eigen::MatrixXd PinOdometry::getCovariance()
{   // RGBDOdometry.cpp:479
    double cov[36];
    orc_covariance(lastA, cov);
    Eigen::MatrixXd m(6, 6);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) m(i, j) = cov[i * 6 + j];
    return m;
}
void PinOdometry::initFirstRGB(GPUTexture* rgb)
{
    orc_odom_init_first_rgb((orc_odometry*)orc, rgb->data<uint8_t>());
    if (ref) ref_odo_init_first_rgb(ref, rgb->data<uint8_t>());
}

Model::Model(unsigned char id, float confidenceThresh, bool enableFillIn, bool, bool enablePoseLogging, MatchingType, float maxDepth_)
    : impl(new ModelImpl()), pose(Eigen::Matrix4f::Identity()), lastPose(Eigen::Matrix4f::Identity()), confidenceThreshold(confidenceThresh),
      maxDepth(maxDepth_), id_(id), fillIn_(enableFillIn)
{
    if (enablePoseLogging) poseLog.reserve(1000);  // Model.cpp:132
}
Model::~Model() { delete impl; }
cv::Mat Model::downloadVertexConfTexture() { return impl ? cv::Mat(g_h, g_w, CV_32FC4, impl->vc.data()) : vc_; }
cv::Mat Model::downloadICPErrorTexture() { return impl ? cv::Mat(g_h, g_w, CV_32FC1, impl->icp_error.data()) : icp_; }
unsigned int Model::lastCount() { return (unsigned)impl->count; }
PinOdometry& Model::getFrameOdometry() { return *impl->odom; }
GPUTexture* Model::getRGBProjection() { return impl->rgbProjection.get(); }

void Model::initialise(const FeedbackBuffer& raw, const FeedbackBuffer& filtered)
{   // Model.cpp:227-272
    impl->surfels.assign((size_t)std::max(raw.count, 1) * 12, 0.f);
    impl->count = orc_model_initialise(raw.data.data(), raw.count, filtered.data.data(), impl->surfels.data());
}
void Model::generateCUDATextures(GPUTexture* depth, GPUTexture*)
{   // Model.cpp:319-343: the filtered depth and its two coarser levels (the mask pyramid is not read downstream)
    const size_t N = (size_t)g_w * g_h;
    g_depth_pyr[0].assign(depth->data<float>(), depth->data<float>() + N);
    g_depth_pyr[1].assign(N / 4, 0.f); g_depth_pyr[2].assign(N / 16, 0.f);
    orc_depth_pyramid(g_depth_pyr[0].data(), g_w, g_h, g_depth_pyr[1].data(), g_depth_pyr[2].data());
}
void Model::performTracking(bool frameToFrameRGB, bool rgbOnly, float icpWeight, bool pyramid, bool fastOdom, bool so3, float maxDepthProcessed,
                            GPUTexture* rgb, int64_t, bool doFillIn)
{   // Model.cpp:345-389
    assert(fillIn_ || !doFillIn);
    lastPose = pose;
    orc_odometry* od = (orc_odometry*)impl->odom->orc;
    float p[16];
    to_row_major(pose, p);
    if (doFillIn) {
        orc_odom_init_icp_model(od, impl->fv.data(), impl->fn.data(), p);
        orc_odom_init_rgb_model(od, impl->fi.data());
    } else {
        orc_odom_init_icp_model(od, impl->vc.data(), impl->nr.data(), p);
        orc_odom_init_rgb_model(od, (frameToFrameRGB && allowsFillIn()) ? impl->fi.data() : impl->img.data());
    }
    const float* pyr[3] = {g_depth_pyr[0].data(), g_depth_pyr[1].data(), g_depth_pyr[2].data()};
    if (impl->odom->ref) {  // the same five initialisers and the same arguments on the reference's class (Model.cpp:352-378)
        void* r = impl->odom->ref;
        if (doFillIn) { ref_odo_init_icp_model(r, impl->fv.data(), impl->fn.data(), maxDepthProcessed, p); ref_odo_init_rgb_model(r, impl->fi.data()); }
        else {
            ref_odo_init_icp_model(r, impl->vc.data(), impl->nr.data(), maxDepthProcessed, p);
            ref_odo_init_rgb_model(r, (frameToFrameRGB && allowsFillIn()) ? impl->fi.data() : impl->img.data());
        }
        ref_odo_init_icp(r, pyr, maxDepthProcessed);
        ref_odo_init_rgb(r, rgb->data<uint8_t>());
        float trans[3] = {pose(0, 3), pose(1, 3), pose(2, 3)}, rot[9], stats[6];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rot[i * 3 + j] = pose(i, j);
        double lastb[6];
        ref_odo_track(r, trans, rot, rgbOnly ? 1 : 0, icpWeight, pyramid ? 1 : 0, fastOdom ? 1 : 0, so3 ? 1 : 0, impl->icp_error.data(), stats, impl->odom->lastA, lastb);
        impl->odom->lastICPError = stats[0]; impl->odom->lastICPCount = stats[1];
        for (int i = 0; i < 3; i++) { pose(i, 3) = trans[i]; for (int j = 0; j < 3; j++) pose(i, j) = rot[i * 3 + j]; }
        return;
    }
    orc_odom_init_icp(od, pyr, maxDepthProcessed);
    orc_odom_init_rgb(od, rgb->data<uint8_t>());
    float trans[3] = {pose(0, 3), pose(1, 3), pose(2, 3)}, rot[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rot[i * 3 + j] = pose(i, j);
    orc_track_opts o{rgbOnly ? 1 : 0, pyramid ? 1 : 0, fastOdom ? 1 : 0, so3 ? 1 : 0, icpWeight};
    orc_track_stats st;
    orc_odom_get_incremental_transformation(od, trans, rot, &o, impl->icp_error.data(), &st);
    impl->odom->lastICPError = st.last_icp_error; impl->odom->lastICPCount = st.last_icp_count;
    memcpy(impl->odom->lastA, st.lastA, sizeof(st.lastA));
    for (int i = 0; i < 3; i++) { pose(i, 3) = trans[i]; for (int j = 0; j < 3; j++) pose(i, j) = rot[i * 3 + j]; }
}
void Model::predictIndices(int time, float depthCutoff, int timeDelta)
{
    float p[16];
    to_row_major(pose, p);
    orc_predict_indices(impl->surfels.data(), impl->count, p, g_cam, g_w, g_h, depthCutoff, time, timeDelta, impl->idx.data(), impl->ivc.data(),
                        impl->ict.data(), impl->inr.data());
}
void Model::combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta, ModelProjection::Prediction)
{
    float p[16];
    to_row_major(pose, p);
    orc_combined_predict(impl->surfels.data(), impl->count, p, g_cam, g_w, g_h, depthCutoff, confidenceThreshold, time, maxTime, timeDelta,
                         impl->img.data(), impl->vc.data(), impl->nr.data(), impl->tm.data());
}
void Model::performFillIn(GPUTexture* rawRGB, GPUTexture* rawDepth, bool frameToFrameRGB, bool lost)
{   // Model.cpp:699-712 -> FillIn::vertex / normal / image with the pass-through flags
    if (!fillIn_) return;
    orc_fill_in(impl->vc.data(), impl->nr.data(), impl->img.data(), rawDepth->data<float>(), rawRGB->data<uint8_t>(), g_w, g_h, g_cam, lost ? 1 : 0,
                (lost || frameToFrameRGB) ? 1 : 0, impl->fv.data(), impl->fn.data(), impl->fi.data());
}
void Model::fuse(const int& time, GPUTexture* rgb, GPUTexture* mask, GPUTexture* depthRaw, GPUTexture* depthFiltered, const float depthCutoff,
                 const float weightMultiplier)
{   // Model.cpp:408-563; weighting: computeFusionWeight (:391-406); depth limit: min(depthCutoff, maxDepth) (:437)
    float p[16], lp[16];
    to_row_major(pose, p); to_row_major(lastPose, lp);
    const float weighting = orc_fusion_weight(p, lp, weightMultiplier);
    impl->scratch.assign((size_t)std::max(impl->count, 1) * 12, 0.f);
    orc_fuse(impl->surfels.data(), impl->count, impl->idx.data(), impl->ivc.data(), impl->inr.data(), rgb->data<uint8_t>(), depthRaw->data<float>(),
             depthFiltered->data<float>(), mask->data<uint8_t>(), p, g_cam, g_w, g_h, time, weighting, (int)id_, std::min(depthCutoff, maxDepth),
             impl->scratch.data(), impl->new_unstable.data(), &impl->n_new);
    impl->surfels.swap(impl->scratch);
}
void Model::clean(const int& time, std::vector<float>&, const int timeDelta, const float, const bool, GPUTexture* depthFiltered, GPUTexture* mask)
{   // Model.cpp:565-697
    float p[16];
    to_row_major(pose, p);
    impl->scratch.assign((size_t)(impl->count + impl->n_new + 1) * 12, 0.f);
    impl->count = orc_clean(impl->surfels.data(), impl->count, impl->new_unstable.data(), impl->n_new, impl->idx.data(), impl->ivc.data(),
                            impl->ict.data(), depthFiltered->data<float>(), mask->data<uint8_t>(), p, g_cam, g_w, g_h, time, confidenceThreshold,
                            g_outlier, timeDelta, (int)id_, impl->scratch.data());
    impl->surfels.swap(impl->scratch);
    impl->n_new = 0;
}

void GPUResize::image(GPUTexture* source, Img<Eigen::Matrix<unsigned char, 3, 1>>& dest)
{
    const int cols = source->width(), rows = source->height();
    const unsigned char* src = source->data<unsigned char>();
    for (int j = 0; j < dest.rows; j++)
        for (int i = 0; i < dest.cols; i++) {
            int sx = (int)floorf((((float)i + 0.5f) / (float)dest.cols) * (float)cols), sy = (int)floorf((((float)j + 0.5f) / (float)dest.rows) * (float)rows);
            sx = sx < 0 ? 0 : (sx >= cols ? cols - 1 : sx); sy = sy < 0 ? 0 : (sy >= rows ? rows - 1 : sy);
            const unsigned char* p = src + ((size_t)sy * cols + sx) * 4;
            auto& d = dest.at<Eigen::Matrix<unsigned char, 3, 1>>(j, i);
            d(0) = p[0]; d(1) = p[1]; d(2) = p[2];
        }
}

CoFusion::CoFusion(int width, int height, float fx, float fy, float cx, float cy, float initConfidenceGlobal, float initConfidenceObject, float depthCut,
                   float icpThresh, bool so3_, unsigned modelSpawnOffset_, bool enableMultipleModels_)
    : modelMatchingType(Model::MatchingType::Drost), newModelListeners(0), inactiveModelListeners(0), modelToModel(width, height, cx, cy, fx, fy),
      tick(1), timeDelta(2147483647 / 2) /* openLoop, MainController.cpp:328 */, icpCountThresh(40000), icpErrThresh(5e-05f), covThresh(1e-05f),
      deforms(0), fernDeforms(0), consSample(20), imageBuff(height / 20, width / 20), consBuff(height / 20, width / 20),
      timesBuff(height / 20, width / 20), closeLoops(false), iclnuim(false), reloc(g_reloc), lost(false), lastFrameRecovery(false), trackingCount(0),
      maxDepthProcessed(20.0f), rgbOnly(false), icpWeight(icpThresh), pyramid(true), fastOdom(false), initConfThresGlobal(initConfidenceGlobal),
      initConfThresObject(initConfidenceObject), fernThresh(0.3095f), so3(so3_), frameToFrameRGB(false), depthCutoff(depthCut),
      modelSpawnOffset(modelSpawnOffset_), exportSegmentation(false)
{   // CoFusion.cpp:21-77 (createTextures: :115-137)
    enableMultipleModels = enableMultipleModels_;
    textures[GPUTexture::RGB] = new GPUTexture(width, height, 4);
    textures[GPUTexture::DEPTH_METRIC] = new GPUTexture(width, height, 4);
    textures[GPUTexture::DEPTH_METRIC_FILTERED] = new GPUTexture(width, height, 4);
    textures[GPUTexture::MASK] = new GPUTexture(width, height, 1);
    feedbackBuffers[FeedbackBuffer::RAW] = new FeedbackBuffer();
    feedbackBuffers[FeedbackBuffer::FILTERED] = new FeedbackBuffer();
    labelGenerator.init(width, height, Segmentation::METHOD::CONNECTED_COMPONENTS);
    // the values MainController pushes into the segmentation every frame (GUI.h:206-227)
    labelGenerator.setUnaryWeightError(75.f); labelGenerator.setUnaryKError(0.0375f); labelGenerator.setUnaryThresholdNew(5.5f);
    labelGenerator.setPairwiseWeightAppearance(7.f); labelGenerator.setPairwiseWeightSmoothness(2.f);
    labelGenerator.setPairwiseSigmaRGB(10.f); labelGenerator.setPairwiseSigmaDepth(0.9f); labelGenerator.setPairwiseSigmaPosition(1.8f);
    labelGenerator.setIterationsCRF(10); labelGenerator.setNewModelMinRelativeSize(0.015f); labelGenerator.setNewModelMaxRelativeSize(0.4f);
    globalModel = std::make_shared<Model>(getNextModelID(true), initConfidenceGlobal, true, true, enablePoseLogging);
    models.push_back(globalModel);
}
CoFusion::~CoFusion()
{
    for (auto& t : textures) delete t.second;
    for (auto& f : feedbackBuffers) delete f.second;
}
void CoFusion::filterDepth()
{   // CoFusion.cpp:567-574 -> depth_bilateral_metric.frag
    orc_bilateral(textures[GPUTexture::DEPTH_METRIC]->data<float>(), g_w, g_h, depthCutoff, textures[GPUTexture::DEPTH_METRIC_FILTERED]->data<float>());
}
void CoFusion::computeFeedbackBuffers()
{   // CoFusion.cpp:161-169 -> FeedbackBuffer::compute (vertex_feedback.*) on the raw and the filtered depth
    const char* keys[2] = {FeedbackBuffer::RAW, FeedbackBuffer::FILTERED};
    const char* depth[2] = {GPUTexture::DEPTH_METRIC, GPUTexture::DEPTH_METRIC_FILTERED};
    for (int k = 0; k < 2; k++) {
        FeedbackBuffer* fb = feedbackBuffers[keys[k]];
        fb->data.assign((size_t)g_w * g_h * 12, 0.f);
        fb->count = orc_vertex_feedback(textures[GPUTexture::RGB]->data<uint8_t>(), textures[depth[k]]->data<float>(), g_w, g_h, g_cam, tick, maxDepthProcessed,
                                        fb->data.data());
    }
}

extern "C" {

void* ref_cf_create(int w, int h, float fx, float fy, float cx, float cy, float conf_global, float conf_object, float depth_cut, float icp_weight,
                    int so3, unsigned model_spawn_offset, int enable_multiple_models)
{
    g_w = w; g_h = h; g_cam = orc_cam{fx, fy, cx, cy};
    Resolution::setResolution(w, h);
    return new CoFusion(w, h, fx, fy, cx, cy, conf_global, conf_object, depth_cut, icp_weight, so3 != 0, model_spawn_offset, enable_multiple_models != 0);
}
void ref_cf_destroy(void* p) { delete (CoFusion*)p; }
// 1: models created from now on track with the reference's own RGBDOdometry class instead of the oracle's restatement of it
void ref_cf_use_reference_tracker(int on) { g_reference_tracker = on != 0; }
void ref_cf_set_reloc(int on) { g_reloc = on != 0; }
int ref_cf_lost(void* p) { return ((CoFusion*)p)->isLost() ? 1 : 0; }
void ref_cf_set_tracking_options(void* p, int rgb_only, int pyramid, int fast_odom, int frame_to_frame_rgb)
{
    ((CoFusion*)p)->setTrackingOptions(rgb_only != 0, pyramid != 0, fast_odom != 0, frame_to_frame_rgb != 0);
}
// depth f32 [H*W] metres, rgb u8 [H*W*3]; gt_mask nullable u8 [H*W] (FrameData::mask); in_pose nullable row-major 4x4 (the ground-truth
// odometry of GUI/Tools/GroundTruthOdometry.cpp hands processFrame a pose instead of letting it track)
int ref_cf_process_frame_pose(void* p, const float* depth, const unsigned char* rgb3, const unsigned char* gt_mask, long long timestamp,
                              const float* in_pose_row_major)
{
    CoFusion* cf = (CoFusion*)p;
    FrameData frame;
    frame.timestamp = timestamp;
    frame.rgb = cv::Mat(g_h, g_w, CV_8UC3, (void*)rgb3);
    frame.depth = cv::Mat(g_h, g_w, CV_32FC1, (void*)depth);
    if (gt_mask) frame.mask = cv::Mat(g_h, g_w, CV_8UC1, (void*)gt_mask);
    Eigen::Matrix4f pose;
    if (in_pose_row_major) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) pose(i, j) = in_pose_row_major[i * 4 + j];
    return cf->processFrame(frame, in_pose_row_major ? &pose : nullptr) ? 1 : 0;
}
int ref_cf_process_frame(void* p, const float* depth, const unsigned char* rgb3, const unsigned char* gt_mask, long long timestamp)
{
    CoFusion* cf = (CoFusion*)p;
    FrameData frame;
    frame.timestamp = timestamp;
    frame.rgb = cv::Mat(g_h, g_w, CV_8UC3, (void*)rgb3);
    frame.depth = cv::Mat(g_h, g_w, CV_32FC1, (void*)depth);
    if (gt_mask) frame.mask = cv::Mat(g_h, g_w, CV_8UC1, (void*)gt_mask);
    return cf->processFrame(frame) ? 1 : 0;
}
int ref_cf_num_models(void* p) { return (int)((CoFusion*)p)->getModels().size(); }
int ref_cf_tick(void* p) { return ((CoFusion*)p)->getTick(); }
// pose row-major [16]; returns the surfel count
int ref_cf_model_info(void* p, int index, unsigned* id, float* pose16, float* conf_threshold, unsigned* unseen, int* pose_log_items)
{
    auto& ms = ((CoFusion*)p)->getModels();
    auto it = ms.begin();
    std::advance(it, index);
    Model& m = **it;
    *id = m.getID(); *conf_threshold = m.getConfidenceThreshold(); *unseen = m.unseenCount; *pose_log_items = (int)m.getPoseLog().size();
    to_row_major(m.getPose(), pose16);
    return (int)m.lastCount();
}
void ref_cf_model_surfels(void* p, int index, float* out)
{
    auto& ms = ((CoFusion*)p)->getModels();
    auto it = ms.begin();
    std::advance(it, index);
    memcpy(out, (*it)->impl->surfels.data(), (size_t)(*it)->impl->count * 12 * sizeof(float));
}
void ref_cf_mask(void* p, unsigned char* out) { memcpy(out, ((CoFusion*)p)->maskTexture()->data<unsigned char>(), (size_t)g_w * g_h); }
// last pose-log item of a model: [x y z qx qy qz qw]
void ref_cf_model_last_pose_log(void* p, int index, float* out7)
{
    auto& ms = ((CoFusion*)p)->getModels();
    auto it = ms.begin();
    std::advance(it, index);
    const auto& log = (*it)->getPoseLog();
    for (int k = 0; k < 7; k++) out7[k] = log.empty() ? 0.f : log.back().p(k);
}

}  // extern "C"
