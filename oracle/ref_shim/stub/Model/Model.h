// Stand-in for Core/Model/Model.h (TEST INFRASTRUCTURE ONLY).  The real class owns OpenGL buffers, textures and shaders.  Two users:
//  * Core/Segmentation/Segmentation.cpp calls getID / downloadVertexConfTexture / downloadICPErrorTexture on a model;
//  * Core/CoFusion.cpp's frame loop (processFrame and its helpers, compiled as text by build_ref.py together with
//    stub/CoFusionPin.h) drives the whole per-model interface.  Here every pass is the CPU oracle's (oracle/orc.h): what that pin
//    checks is the reference's ORCHESTRATION -- which pass runs when, on which model, with which arguments, who is spawned,
//    deactivated, re-thresholded -- not the passes themselves (those are pinned against the CUDA kernels / shaders separately).
// Members defined inline are restatements of Model.h's own inline members (file:line given); the others live in ref_cofusion.cpp.
#pragma once
#include <Eigen/Core>
#include <opencv2/imgproc/imgproc.hpp>
#include <stdint.h>

#include <limits>
#include <list>
#include <memory>
#include <vector>

class GPUTexture;
class FeedbackBuffer;
struct ModelImpl;

// ModelProjection: only the prediction kind (ModelProjection.h:41) and the texture getters the dead loop-closure branch names
class ModelProjection {
  public:
    enum Prediction { ACTIVE, INACTIVE };
    GPUTexture* getOldVertexTex() { return nullptr; }
    GPUTexture* getOldNormalTex() { return nullptr; }
    GPUTexture* getOldImageTex() { return nullptr; }
    GPUTexture* getOldTimeTex() { return nullptr; }
    GPUTexture* getSplatVertexConfTex() { return nullptr; }
    GPUTexture* getSplatNormalTex() { return nullptr; }
    GPUTexture* getSplatImageTex() { return nullptr; }
    template <class... A> void synthesizeDepth(A&&...) {}
};

// the tracker a model owns (RGBDOdometry): the frame loop itself only touches these members; tracking runs on the oracle
class PinOdometry {
  public:
    PinOdometry(int width, int height, float cx, float cy, float fx, float fy);
    ~PinOdometry();
    PinOdometry(const PinOdometry&) = delete;
    void initFirstRGB(GPUTexture* rgb);
    Eigen::MatrixXd getCovariance();                       // ref_cofusion.cpp: orc_covariance of the last tracking call's lastA
    template <class... A> void initICPModel(A&&...) {}   // (dead loop-closure branch of processFrame)
    template <class... A> void initRGBModel(A&&...) {}
    template <class... A> void initICP(A&&...) {}
    template <class... A> void initRGB(A&&...) {}
    template <class... A> void getIncrementalTransformation(A&&...) {}
    float lastICPError = 0, lastICPCount = 0;
    double lastA[36] = {0};
    void* orc = nullptr;  // orc_odometry*
    void* ref = nullptr;  // the reference's own RGBDOdometry behind ref_odo.cpp's C entry points (ref_cf_use_reference_tracker)
};

class Model {
  public:
    enum class MatchingType { Drost };
    // Segmentation pin (ref_seg.cpp): a model that only carries the two images the segmentation downloads
    Model(unsigned char id, cv::Mat vertConf /* CV_32FC4 */, cv::Mat icpError /* CV_32FC1 */) : id_(id), vc_(vertConf), icp_(icpError) {}
    // Model.h:117-119
    Model(unsigned char id, float confidenceThresh, bool enableFillIn = true, bool enableErrorRecording = true, bool enablePoseLogging = false,
          MatchingType matchingType = MatchingType::Drost, float maxDepth = std::numeric_limits<float>::max());
    ~Model();
    Model(const Model&) = delete;

    unsigned int getID() const { return id_; }
    cv::Mat downloadVertexConfTexture();
    cv::Mat downloadICPErrorTexture();

    unsigned int lastCount();
    void initialise(const FeedbackBuffer& rawFeedback, const FeedbackBuffer& filteredFeedback);
    static void generateCUDATextures(GPUTexture* depth, GPUTexture* mask);
    void performTracking(bool frameToFrameRGB, bool rgbOnly, float icpWeight, bool pyramid, bool fastOdom, bool so3, float maxDepthProcessed,
                         GPUTexture* rgb, int64_t logTimestamp, bool tryFillIn = false);
    void fuse(const int& time, GPUTexture* rgb, GPUTexture* mask, GPUTexture* depthRaw, GPUTexture* depthFiltered, const float depthCutoff,
              const float weightMultiplier);
    void clean(const int& time, std::vector<float>& graph, const int timeDelta, const float depthCutoff, const bool isFern,
               GPUTexture* depthFiltered, GPUTexture* mask);
    void eraseErrorGeometry(GPUTexture*) {}
    bool allowsFillIn() const { return fillIn_; }                                                                   // Model.h:160
    void performFillIn(GPUTexture* rawRGB, GPUTexture* rawDepth, bool frameToFrameRGB, bool lost);
    void combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta, ModelProjection::Prediction predictionType);
    void predictIndices(int time, float depthCutoff, int timeDelta);
    float getConfidenceThreshold() const { return confidenceThreshold; }                                            // Model.h:173-177
    void setConfidenceThreshold(float confThresh) { confidenceThreshold = confThresh; }
    void setMaxDepth(float d) { maxDepth = d; }
    GPUTexture* getRGBProjection();
    GPUTexture* getFillInImageTexture() { return nullptr; }   // (dead loop-closure branch)
    GPUTexture* getFillInNormalTexture() { return nullptr; }
    GPUTexture* getFillInVertexTexture() { return nullptr; }
    int getModel() { return 0; }
    const Eigen::Matrix4f& getPose() const { return pose; }                                                         // Model.h:215-222
    void overridePose(const Eigen::Matrix4f& p) { pose = p; lastPose = p; }
    PinOdometry& getFrameOdometry();
    ModelProjection& getIndexMap() { return indexMap; }
    unsigned incrementUnseenCount()                                                                                 // Model.h:232-235
    {
        if (unseenCount < std::numeric_limits<unsigned>::max()) return ++unseenCount;
        return unseenCount;
    }
    struct PoseLogItem { int64_t ts; Eigen::Matrix<float, 7, 1> p; };                                               // Model.h:237-243
    bool isLoggingPoses() const { return poseLog.capacity() > 0; }
    std::vector<PoseLogItem>& getPoseLog() { return poseLog; }

    // harness access
    ModelImpl* impl = nullptr;
    Eigen::Matrix4f pose, lastPose;
    float confidenceThreshold = 0, maxDepth = 0;
    unsigned unseenCount = 0;

  private:
    unsigned int id_;
    cv::Mat vc_, icp_;
    bool fillIn_ = false;
    std::vector<PoseLogItem> poseLog;
    ModelProjection indexMap;
};
typedef std::shared_ptr<Model> ModelPointer;
typedef std::list<ModelPointer> ModelList;
typedef ModelList::iterator ModelListIterator;
