// ref_cofusion.cpp -- the rest of the translation unit that holds the reference's own frame loop (see stub/CoFusionPin.h),
// TEST INFRASTRUCTURE ONLY: the members of the stand-in classes that are not the reference's text, and flat C entry points.
// Every image-space / surfel / tracking pass is the CPU oracle's (oracle/orc.h).  Since round 6 Model::initICP, performTracking, fuse and
// clean are the reference's own TEXT as well (Core/Model/Model.cpp:350-389, 408-697, pasted by build_ref.py): their OpenGL calls are
// recorded by stub/glpin.h and the draw handler below runs the oracle's pass with the recorded uniforms, textures and buffers -- the
// argument handling of those methods is no longer restated.  The constructor restates CoFusion.cpp:21-77 and the GUI defaults
// MainController applies every frame; computeFusionWeight is the oracle's (its text is pinned on its own: ref_weight.cpp).
extern "C" {
#include "orc.h"
}
#include <math.h>
#include <stdlib.h>
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
const float g_outlier = 3.0f;                 // GUI default of the outlier coefficient (GUI.h:213), set through Model::GPUSetup (ref_cf_create)

void to_row_major(const Eigen::Matrix4f& m, float out[16]) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = m(i, j); }
}  // namespace

struct ModelImpl {
    Model* owner = nullptr;
    int count = 0;
    std::vector<float> buf[2], new_unstable;   // the two surfel buffers (Model::vbos: indexed by target / renderSource) and newUnstableBuffer
    std::vector<float>& surfels() { return buf[owner->target]; }
    int n_new = 0;
    std::vector<unsigned char> img;            // combinedPredict: image RGBA8, vertexConf, normalRad, time
    std::vector<float> vc, nr;
    std::vector<uint16_t> tm;
    std::vector<uint32_t> idx;                 // predictIndices
    std::vector<float> ivc, ict, inr;
    std::vector<float> fv, fn;                 // fill-in textures
    std::vector<unsigned char> fi;
    std::vector<float> icp_error;
    std::unique_ptr<GPUTexture> rgbProjection, vcTex, nrTex, fvTex, fnTex, fiTex, idxTex, ivcTex, ictTex, inrTex;
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
        vcTex.reset(new GPUTexture(vc.data(), g_w, g_h)); nrTex.reset(new GPUTexture(nr.data(), g_w, g_h));
        fvTex.reset(new GPUTexture(fv.data(), g_w, g_h)); fnTex.reset(new GPUTexture(fn.data(), g_w, g_h)); fiTex.reset(new GPUTexture(fi.data(), g_w, g_h));
        idxTex.reset(new GPUTexture(idx.data(), g_w, g_h)); ivcTex.reset(new GPUTexture(ivc.data(), g_w, g_h));
        ictTex.reset(new GPUTexture(ict.data(), g_w, g_h)); inrTex.reset(new GPUTexture(inr.data(), g_w, g_h));
        odom.reset(new PinOdometry(g_w, g_h, g_cam.cx, g_cam.cy, g_cam.fx, g_cam.fy));
    }
};
// buffer name (Model::vbos[k].dataBuffer / .stateObject, newUnstableBuffer.*, uvo) -> (model, which): what a recorded glBindBuffer / draw resolves to
struct BufferName { Model* model; int which; };   // which: 0 / 1 = vbos[k], 2 = newUnstableBuffer, 3 = uvo
static std::map<unsigned, BufferName> g_buffers;

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

// ---- the tracker calls of Model::initICP / performTracking (now the reference's text): RGBDOdometry.h:42-64 on the oracle, and on the
// reference's own class when ref_cf_use_reference_tracker is set ----
void PinOdometry::initICPModel(GPUTexture* predictedVertices, GPUTexture* predictedNormals, const float depthCutoff, const Eigen::Matrix4f& modelPose)
{
    float p[16];
    to_row_major(modelPose, p);
    if (ref) ref_odo_init_icp_model(ref, predictedVertices->data<float>(), predictedNormals->data<float>(), depthCutoff, p);
    else orc_odom_init_icp_model((orc_odometry*)orc, predictedVertices->data<float>(), predictedNormals->data<float>(), p);
}
void PinOdometry::initRGBModel(GPUTexture* rgb)
{
    if (ref) ref_odo_init_rgb_model(ref, rgb->data<uint8_t>());
    else orc_odom_init_rgb_model((orc_odometry*)orc, rgb->data<uint8_t>());
}
void PinOdometry::initICP(const std::vector<std::vector<float>>& depthPyramid, const std::vector<std::vector<unsigned char>>&, const float depthCutoff)
{   // (the mask pyramid is dead data in the reference: createVMap ignores it, cudafuncs.cu:119)
    const float* pyr[3] = {depthPyramid[0].data(), depthPyramid[1].data(), depthPyramid[2].data()};
    if (ref) ref_odo_init_icp(ref, pyr, depthCutoff);
    else orc_odom_init_icp((orc_odometry*)orc, pyr, depthCutoff);
}
void PinOdometry::initRGB(GPUTexture* rgb)
{
    if (ref) ref_odo_init_rgb(ref, rgb->data<uint8_t>());
    else orc_odom_init_rgb((orc_odometry*)orc, rgb->data<uint8_t>());
}
void PinOdometry::getIncrementalTransformation(Eigen::Vector3f& trans, Eigen::Matrix<float, 3, 3, Eigen::RowMajor>& rot, const bool& rgbOnly,
                                               const float& icpWeight, const bool& pyramid, const bool& fastOdom, const bool& so3,
                                               const cudaSurfaceObject_t& icpErrorSurface, const cudaSurfaceObject_t&)
{
    if (!orc) return;   // (modelToModel of the dead loop-closure branch is never initialised with maps)
    float* err = icpErrorSurface ? ((GPUTexture*)(uintptr_t)icpErrorSurface)->data<float>() : nullptr;
    float t[3] = {trans(0), trans(1), trans(2)}, R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = rot(i, j);
    if (ref) {
        float stats[6]; double lastb[6];
        ref_odo_track(ref, t, R, rgbOnly ? 1 : 0, icpWeight, pyramid ? 1 : 0, fastOdom ? 1 : 0, so3 ? 1 : 0, err, stats, lastA, lastb);
        lastICPError = stats[0]; lastICPCount = stats[1];
    } else {
        orc_track_opts o{rgbOnly ? 1 : 0, pyramid ? 1 : 0, fastOdom ? 1 : 0, so3 ? 1 : 0, icpWeight};
        orc_track_stats st;
        orc_odom_get_incremental_transformation((orc_odometry*)orc, t, R, &o, err, &st);
        lastICPError = st.last_icp_error; lastICPCount = st.last_icp_count;
        memcpy(lastA, st.lastA, sizeof(st.lastA));
    }
    for (int i = 0; i < 3; i++) { trans(i) = t[i]; for (int j = 0; j < 3; j++) rot(i, j) = R[i * 3 + j]; }
}

GPUTexture Model::deformationNodes(1, 1, 4);

Model::Model(unsigned char id_arg, float confidenceThresh, bool enableFillIn, bool, bool enablePoseLogging, MatchingType, float maxDepth_)
    : impl(new ModelImpl()), pose(Eigen::Matrix4f::Identity()), lastPose(Eigen::Matrix4f::Identity()), confidenceThreshold(confidenceThresh),
      maxDepth(maxDepth_), id_(id_arg), fillIn_(enableFillIn)
{
    if (enablePoseLogging) poseLog.reserve(1000);  // Model.cpp:132
    impl->owner = this;
    id = id_arg;
    if (enableFillIn) fillIn.reset(new char(1));
    frameToModelPtr = impl->odom.get();
    // the GL names the pasted text passes around
    for (int k = 0; k < 2; k++) {
        vbos[k].dataBuffer = glpin::new_id(); vbos[k].stateObject = glpin::new_id();
        g_buffers[vbos[k].dataBuffer] = BufferName{this, k}; g_buffers[vbos[k].stateObject] = BufferName{this, k};
    }
    newUnstableBuffer.dataBuffer = glpin::new_id(); newUnstableBuffer.stateObject = glpin::new_id();
    g_buffers[newUnstableBuffer.dataBuffer] = BufferName{this, 2}; g_buffers[newUnstableBuffer.stateObject] = BufferName{this, 2};
    uvo = glpin::new_id(); g_buffers[uvo] = BufferName{this, 3};
    uvSize = g_w * g_h;   // one texture coordinate per pixel, column-major (Model.cpp:164-172); the text only hands the count to glDrawArrays
    countQuery = glpin::new_id();
    icpError.reset(new GPUTexture(impl->icp_error.data(), g_w, g_h));
    rgbError.reset(new GPUTexture(1, 1, 4));
    indexMap.sparseIndex = impl->idxTex.get(); indexMap.sparseVertConf = impl->ivcTex.get(); indexMap.sparseColorTime = impl->ictTex.get();
    indexMap.sparseNormalRad = impl->inrTex.get(); indexMap.depthTex = impl->ivcTex.get();   // (the index map's depth attachment: not sampled by the oracle's clean)
}
Model::~Model()
{
    for (auto it = g_buffers.begin(); it != g_buffers.end();) it = (it->second.model == this) ? g_buffers.erase(it) : std::next(it);
    delete impl;
}
cv::Mat Model::downloadVertexConfTexture() { return impl ? cv::Mat(g_h, g_w, CV_32FC4, impl->vc.data()) : vc_; }
cv::Mat Model::downloadICPErrorTexture() { return impl ? cv::Mat(g_h, g_w, CV_32FC1, impl->icp_error.data()) : icp_; }
unsigned int Model::lastCount() { return (unsigned)impl->count; }
PinOdometry& Model::getFrameOdometry() { return *impl->odom; }
GPUTexture* Model::getRGBProjection() { return impl->rgbProjection.get(); }
GPUTexture* Model::getVertexConfProjection() { return impl->vcTex.get(); }
GPUTexture* Model::getNormalProjection() { return impl->nrTex.get(); }
GPUTexture* Model::getFillInImageTexture() { return impl->fiTex.get(); }
GPUTexture* Model::getFillInNormalTexture() { return impl->fnTex.get(); }
GPUTexture* Model::getFillInVertexTexture() { return impl->fvTex.get(); }
float Model::computeFusionWeight(float weightMultiplier) const
{   // Model.cpp:391-406 -- the oracle's statement (held to the reference text by ref_weight.cpp / test_fusion_weight_against_the_reference_text)
    float p[16], lp[16];
    to_row_major(pose, p); to_row_major(lastPose, lp);
    return orc_fusion_weight(p, lp, weightMultiplier);
}

void Model::initialise(const FeedbackBuffer& raw, const FeedbackBuffer& filtered)
{   // Model.cpp:227-272
    impl->surfels().assign((size_t)std::max(raw.count, 1) * 12, 0.f);
    impl->count = orc_model_initialise(raw.data.data(), raw.count, filtered.data.data(), impl->surfels().data());
}
void Model::generateCUDATextures(GPUTexture* depth, GPUTexture*)
{   // Model.cpp:319-343: the filtered depth and its two coarser levels into GPUSetup::depth_tmp (the mask pyramid is not read downstream)
    auto& pyr = GPUSetup::getInstance().depth_tmp;
    const size_t N = (size_t)g_w * g_h;
    pyr[0].assign(depth->data<float>(), depth->data<float>() + N);
    pyr[1].assign(N / 4, 0.f); pyr[2].assign(N / 16, 0.f);
    orc_depth_pyramid(pyr[0].data(), g_w, g_h, pyr[1].data(), pyr[2].data());
}
// (Model::initICP and Model::performTracking: the reference's text, pasted by build_ref.py)
void Model::predictIndices(int time, float depthCutoff, int timeDelta)
{
    float p[16];
    to_row_major(pose, p);
    orc_predict_indices(impl->surfels().data(), impl->count, p, g_cam, g_w, g_h, depthCutoff, time, timeDelta, impl->idx.data(), impl->ivc.data(),
                        impl->ict.data(), impl->inr.data());
}
void Model::combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta, ModelProjection::Prediction)
{
    float p[16];
    to_row_major(pose, p);
    orc_combined_predict(impl->surfels().data(), impl->count, p, g_cam, g_w, g_h, depthCutoff, confidenceThreshold, time, maxTime, timeDelta,
                         impl->img.data(), impl->vc.data(), impl->nr.data(), impl->tm.data());
}
void Model::performFillIn(GPUTexture* rawRGB, GPUTexture* rawDepth, bool frameToFrameRGB, bool lost)
{   // Model.cpp:699-712 -> FillIn::vertex / normal / image with the pass-through flags
    if (!fillIn_) return;
    orc_fill_in(impl->vc.data(), impl->nr.data(), impl->img.data(), rawDepth->data<float>(), rawRGB->data<uint8_t>(), g_w, g_h, g_cam, lost ? 1 : 0,
                (lost || frameToFrameRGB) ? 1 : 0, impl->fv.data(), impl->fn.data(), impl->fi.data());
}

// ---- the draw handler: what a draw call of the pasted Model::fuse / Model::clean text means on the oracle --------------------------
// The recorded state says which program is bound, what its uniforms hold, which texture sits on which unit, which buffer feeds the
// attributes and which takes the transform feedback.  Sampler units are resolved through the program's OWN sampler uniforms (the text
// sets "drSampler" = 1 and binds a texture to unit 1: the handler asks for "drSampler"), so a texture bound to the wrong unit, a value
// handed to the wrong uniform or the wrong input / output buffer changes what the oracle computes -- and the fixtures no longer match.
namespace {
[[noreturn]] void pin_fail(const char* what) { fprintf(stderr, "glpin: %s\n", what); abort(); }
const Uniform& uni(const glpin::State& st, const char* name)
{
    auto it = st.uniforms.find(name);
    if (it == st.uniforms.end()) { fprintf(stderr, "glpin: program %s: uniform %s was not set\n", st.program.c_str(), name); abort(); }
    return it->second;
}
float uni_f(const glpin::State& st, const char* n) { const Uniform& u = uni(st, n); if (u.t != Uniform::FLOAT) pin_fail("uniform type: float expected"); return u.f; }
int uni_i(const glpin::State& st, const char* n) { const Uniform& u = uni(st, n); if (u.t != Uniform::INT) pin_fail("uniform type: int expected"); return u.i; }
unsigned uni_u(const glpin::State& st, const char* n) { const Uniform& u = uni(st, n); if (u.t != Uniform::UINT) pin_fail("uniform type: unsigned expected"); return u.ui; }
GPUTexture* sampler(const glpin::State& st, const char* name)
{
    const int unit = uni_i(st, name);
    GPUTexture* t = (unit >= 0 && unit < 16) ? GPUTexture::by_name(st.tex[unit]) : nullptr;
    if (!t) { fprintf(stderr, "glpin: program %s: no texture on the unit of sampler %s\n", st.program.c_str(), name); abort(); }
    return t;
}
BufferName buffer(unsigned name)
{
    auto it = g_buffers.find(name);
    if (it == g_buffers.end()) pin_fail("draw with an unknown buffer name");
    return it->second;
}
void check_cam(const Uniform& u, bool inverse_focal)
{   // data.vert takes (cx, cy, 1/fx, 1/fy), copy_unstable.vert (cx, cy, fx, fy) (Model.cpp:434-435, 603-604)
    if (u.t != Uniform::VEC4) pin_fail("cam: vec4 expected");
    const float fx = inverse_focal ? (float)(1.0 / g_cam.fx) : g_cam.fx, fy = inverse_focal ? (float)(1.0 / g_cam.fy) : g_cam.fy;
    if (u.v4(0) != g_cam.cx || u.v4(1) != g_cam.cy || u.v4(2) != fx || u.v4(3) != fy) pin_fail("cam uniform is not the camera of this run");
}
struct PendingData {   // what the data pass (first draw of Model::fuse) recorded, consumed by the update pass
    bool valid = false;
    Model* m = nullptr;
    GPUTexture *rgb, *depthRaw, *depthFiltered, *index, *vertConf, *normRad, *mask;
    float pose[16], time, weighting, maxDepth;
    unsigned maskID;
} g_data;
struct PendingClean { bool valid = false; Model* m = nullptr; } g_clean;

unsigned g_draws[3] = {0, 0, 0};   // draw calls handled per program (data, update, unstable): ref_cf_glpin_draws
// COFUSION_GLPIN_MUTATE=k corrupts the RECORDED state before the handler reads it, the way a plumbing mistake in the pasted text would:
// 1 = raw and filtered depth on each other's units in the data pass, 2 = the clean pass's input depth on the prediction's unit,
// 3 = the data pass's time one frame late.  tests/test_cpu_refpin.py shows each one breaks the reproduction of the committed fixture.
const int g_mutate = getenv("COFUSION_GLPIN_MUTATE") ? atoi(getenv("COFUSION_GLPIN_MUTATE")) : 0;

void on_draw(glpin::State& st, int kind, unsigned arg)
{
    g_draws[st.program == "data" ? 0 : st.program == "update" ? 1 : 2]++;
    if (g_mutate == 1 && st.program == "data") std::swap(st.tex[1], st.tex[2]);
    if (g_mutate == 2 && st.program == "unstable" && g_clean.valid) std::swap(st.tex[5], st.tex[6]);   // (the second draw is the one that is interpreted)
    if (g_mutate == 3 && st.program == "data") st.uniforms.at("time").f += 1.0f;
    if (st.program == "data") {   // Model::fuse, PROGRAM1 (Model.cpp:422-497): glDrawArrays over the uv buffer, feedback into newUnstableBuffer
        if (kind != 0 || !st.in_feedback) pin_fail("data pass: glDrawArrays inside transform feedback expected");
        const BufferName in = buffer(st.array_buffer), out = buffer(st.tf_buffer);
        if (in.which != 3 || out.which != 2 || in.model != out.model || (int)arg != in.model->uvSize) pin_fail("data pass: uv buffer in, newUnstableBuffer out");
        if (uni_f(st, "cols") != (float)g_w || uni_f(st, "rows") != (float)g_h || uni_f(st, "scale") != 1.0f || uni_f(st, "texDim") != (float)Model::TEXTURE_DIMENSION)
            pin_fail("data pass: cols / rows / scale / texDim");
        check_cam(uni(st, "cam"), true);
        PendingData d;
        d.valid = true; d.m = in.model;
        d.rgb = sampler(st, "cSampler"); d.depthRaw = sampler(st, "drSampler"); d.depthFiltered = sampler(st, "drfSampler"); d.index = sampler(st, "indexSampler");
        d.vertConf = sampler(st, "vertConfSampler"); (void)sampler(st, "colorTimeSampler"); d.normRad = sampler(st, "normRadSampler"); d.mask = sampler(st, "maskSampler");
        const Uniform& P = uni(st, "pose");
        if (P.t != Uniform::MAT4) pin_fail("pose: mat4 expected");
        to_row_major(P.m4, d.pose);
        d.time = uni_f(st, "time"); d.weighting = uni_f(st, "weighting"); d.maxDepth = uni_f(st, "maxDepth"); d.maskID = uni_u(st, "maskID");
        g_data = d;
        return;
    }
    if (st.program == "update") {   // Model::fuse, PROGRAM2 (Model.cpp:499-562): vbos[target] in, vbos[renderSource] out; the oracle runs both passes here
        if (kind != 1 || !st.in_feedback || !g_data.valid) pin_fail("update pass: glDrawTransformFeedback behind a data pass expected");
        const BufferName in = buffer(st.array_buffer), drawn = buffer(arg), out = buffer(st.tf_buffer);
        Model* m = g_data.m;
        if (in.model != m || drawn.model != m || out.model != m || in.which != m->target || drawn.which != m->target || out.which != m->renderSource)
            pin_fail("update pass: vbos[target] in, vbos[renderSource] out");
        if ((float)uni_i(st, "time") != g_data.time || uni_f(st, "texDim") != (float)Model::TEXTURE_DIMENSION) pin_fail("update pass: time / texDim");
        if (GPUTexture::by_name(st.tex[uni_i(st, "vertSamp")]) != &m->gpu.updateMapVertsConfs || GPUTexture::by_name(st.tex[uni_i(st, "colorSamp")]) != &m->gpu.updateMapColorsTime ||
            GPUTexture::by_name(st.tex[uni_i(st, "normSamp")]) != &m->gpu.updateMapNormsRadii) pin_fail("update pass: the three update maps");
        ModelImpl* I = m->impl;
        std::vector<float>& src = I->buf[in.which]; std::vector<float>& dst = I->buf[out.which];
        dst.assign((size_t)std::max(I->count, 1) * 12, 0.f);
        orc_fuse(src.data(), I->count, g_data.index->data<uint32_t>(), g_data.vertConf->data<float>(), g_data.normRad->data<float>(), g_data.rgb->data<uint8_t>(),
                 g_data.depthRaw->data<float>(), g_data.depthFiltered->data<float>(), g_data.mask->data<uint8_t>(), g_data.pose, g_cam, g_w, g_h, (int)g_data.time,
                 g_data.weighting, (int)g_data.maskID, g_data.maxDepth, dst.data(), I->new_unstable.data(), &I->n_new);
        g_data.valid = false;
        return;
    }
    if (st.program == "unstable") {   // Model::clean (Model.cpp:565-697): two feedback draws inside one query -- the map, then the new surfels
        if (kind != 1 || !st.in_feedback || !st.in_query) pin_fail("clean: glDrawTransformFeedback inside feedback and query expected");
        const BufferName in = buffer(st.array_buffer), drawn = buffer(arg), out = buffer(st.tf_buffer);
        if (!g_clean.valid) {   // first draw: the model's own buffer
            Model* m = in.model;
            if (drawn.model != m || out.model != m || in.which != m->target || drawn.which != m->target || out.which != m->renderSource) pin_fail("clean, first draw: vbos[target] in, vbos[renderSource] out");
            g_clean.valid = true; g_clean.m = m;
            return;
        }
        Model* m = g_clean.m;   // second draw: newUnstableBuffer appended behind it
        if (in.model != m || drawn.model != m || out.model != m || in.which != 2 || drawn.which != 2 || out.which != m->renderSource) pin_fail("clean, second draw: newUnstableBuffer in, vbos[renderSource] out");
        g_clean.valid = false;
        if (uni_f(st, "cols") != (float)g_w || uni_f(st, "rows") != (float)g_h || uni_f(st, "scale") != 1.0f || uni_f(st, "nodes") != 0.f || uni_i(st, "isFern") != 0)
            pin_fail("clean: cols / rows / scale / nodes / isFern");
        check_cam(uni(st, "cam"), false);
        const Uniform& Ti = uni(st, "t_inv");
        if (Ti.t != Uniform::MAT4) pin_fail("t_inv: mat4 expected");
        float p[16], pinv[16];
        to_row_major(m->pose, p);
        orc_inverse_pose(p, pinv);   // (the text takes Eigen's general inverse of the pose: the same matrix up to rounding; the oracle's pass inverts the pose itself)
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (fabsf(Ti.m4(i, j) - pinv[i * 4 + j]) > 1e-4f * (1.f + fabsf(pinv[i * 4 + j]))) pin_fail("t_inv is not the inverse of the model's pose");
        GPUTexture* index = sampler(st, "indexSampler"); GPUTexture* vertConf = sampler(st, "vertConfSampler"); GPUTexture* colorTime = sampler(st, "colorTimeSampler");
        (void)sampler(st, "normRadSampler"); (void)sampler(st, "nodeSampler"); (void)sampler(st, "depthSamplerPrediction");
        GPUTexture* depthInput = sampler(st, "depthSamplerInput"); GPUTexture* mask = sampler(st, "maskSampler");
        ModelImpl* I = m->impl;
        std::vector<float>& src = I->buf[m->target]; std::vector<float>& dst = I->buf[m->renderSource];
        dst.assign((size_t)(I->count + I->n_new + 1) * 12, 0.f);
        I->count = orc_clean(src.data(), I->count, I->new_unstable.data(), I->n_new, index->data<uint32_t>(), vertConf->data<float>(), colorTime->data<float>(),
                             depthInput->data<float>(), mask->data<uint8_t>(), p, g_cam, g_w, g_h, uni_i(st, "time"), uni_f(st, "confThreshold"), uni_f(st, "outlierCoeff"),
                             uni_i(st, "timeDelta"), (int)uni_u(st, "maskID"), dst.data());
        (void)uni_f(st, "maxDepth"); (void)uni_f(st, "nodeCols");   // (set by the text; the deformation branch they belong to is dead: nodes == 0)
        I->n_new = 0;
        st.query_result = (unsigned)I->count;
        return;
    }
    pin_fail("draw call of an unknown program");
}
struct InstallHandler { InstallHandler() { glpin::state().on_draw = on_draw; } } g_install_handler;
}  // namespace

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
    Intrinsics::setIntrinics(fx, fy, cx, cy);
    Model::GPUSetup::getInstance().outlierCoefficient = g_outlier;   // MainController pushes the GUI's value every frame
    return new CoFusion(w, h, fx, fy, cx, cy, conf_global, conf_object, depth_cut, icp_weight, so3 != 0, model_spawn_offset, enable_multiple_models != 0);
}
void ref_cf_destroy(void* p) { delete (CoFusion*)p; }
// draw calls of the pasted Model::fuse / Model::clean text the handler has served: [data, update, unstable]
void ref_cf_glpin_draws(unsigned* out3) { for (int k = 0; k < 3; k++) out3[k] = g_draws[k]; }
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
    memcpy(out, (*it)->impl->surfels().data(), (size_t)(*it)->impl->count * 12 * sizeof(float));
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
