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

#include "GPUTexture.h"
#include "glpin.h"               // the recording OpenGL behind the text of Model::fuse / Model::clean (round 6)
#include "Utils/Intrinsics.h"    // the reference's own (header-only singleton)
#include "Utils/Resolution.h"    // the reference's own

class FeedbackBuffer;
struct ModelImpl;
struct Vertex { static const int SIZE = sizeof(Eigen::Vector4f) * 3; };   // Shaders/Vertex.cpp:43

// ModelProjection: only the prediction kind (ModelProjection.h:41) and the texture getters the dead loop-closure branch names
class ModelProjection {
  public:
    enum Prediction { ACTIVE, INACTIVE };
    static const int FACTOR = 1;                                           // ModelProjection.cpp:22
    // the index-map textures Model::fuse / Model::clean bind (ModelProjection.h:60-70): views of the owning model's buffers (ref_cofusion.cpp)
    GPUTexture* getSparseIndexTex() { return sparseIndex; }
    GPUTexture* getSparseVertConfTex() { return sparseVertConf; }
    GPUTexture* getSparseColorTimeTex() { return sparseColorTime; }
    GPUTexture* getSparseNormalRadTex() { return sparseNormalRad; }
    GPUTexture* getDepthTex() { return depthTex; }
    GPUTexture *sparseIndex = nullptr, *sparseVertConf = nullptr, *sparseColorTime = nullptr, *sparseNormalRad = nullptr, *depthTex = nullptr;
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
    // the five calls of Model::initICP / performTracking (Model.cpp:350-389: that text drives them since round 6), RGBDOdometry.h:42-64
    void initICPModel(GPUTexture* predictedVertices, GPUTexture* predictedNormals, const float depthCutoff, const Eigen::Matrix4f& modelPose);
    void initRGBModel(GPUTexture* rgb);
    void initICP(const std::vector<std::vector<float>>& depthPyramid, const std::vector<std::vector<unsigned char>>& maskPyramid, const float depthCutoff);
    void initICP(GPUTexture*, GPUTexture*, const float) {}   // (the texture flavour: only the dead loop-closure branch of processFrame calls it)
    void initRGB(GPUTexture* rgb);
    void getIncrementalTransformation(Eigen::Vector3f& trans, Eigen::Matrix<float, 3, 3, Eigen::RowMajor>& rot, const bool& rgbOnly, const float& icpWeight,
                                      const bool& pyramid, const bool& fastOdom, const bool& so3, const cudaSurfaceObject_t& icpErrorSurface = 0,
                                      const cudaSurfaceObject_t& rgbErrorSurface = 0);
    float lastICPError = 0, lastICPCount = 0;
    double lastA[36] = {0};
    void* orc = nullptr;  // orc_odometry*
    void* ref = nullptr;  // the reference's own RGBDOdometry behind ref_odo.cpp's C entry points (ref_cf_use_reference_tracker)
};

class Model {
  public:
    enum class MatchingType { Drost };
    // ---- what the TEXT of Model::initICP / performTracking / fuse / clean uses (Model.h:52-330), round 6 ----
    struct OutputBuffer { GLuint dataBuffer = 0, stateObject = 0; };       // Shaders/Shaders.h
    struct GPUSetup {
        static GPUSetup& getInstance() { static GPUSetup g; return g; }
        std::shared_ptr<Shader> dataProgram{new Shader("data")}, updateProgram{new Shader("update")}, unstableProgram{new Shader("unstable")};
        struct { int width = 0, height = 0; } renderBuffer;
        struct { void Bind() const {} void Unbind() const {} } frameBuffer;
        GPUTexture updateMapVertsConfs{1, 1, 16}, updateMapColorsTime{1, 1, 16}, updateMapNormsRadii{1, 1, 16};
        std::vector<std::vector<float>> depth_tmp{3};
        std::vector<std::vector<unsigned char>> mask_tmp{3};
        float outlierCoefficient = 0.9;
    };
    static const int TEXTURE_DIMENSION = 3072, NODE_TEXTURE_DIMENSION = 16384, MAX_NODES = 16384 / 16;   // Model.cpp:95-100
    void initICP(bool doFillIn, bool frameToFrameRGB, float depthCutoff, GPUTexture* rgb);
    float computeFusionWeight(float weightMultiplier) const;   // ref_cofusion.cpp: the oracle's (its text is pinned on its own: ref_weight.cpp)
    GPUTexture* getVertexConfProjection();
    GPUTexture* getNormalProjection();
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
    GPUTexture* getFillInImageTexture();
    GPUTexture* getFillInNormalTexture();
    GPUTexture* getFillInVertexTexture();
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
    // members the pasted text names (Model.h:236-330)
    OutputBuffer vbos[2], newUnstableBuffer;
    int target = 0, renderSource = 1;      // swapped after FUSE and CLEAN
    GLuint countQuery = 0;
    unsigned int count = 0;
    static GPUTexture deformationNodes;
    GLuint uvo = 0;
    int uvSize = 0;
    unsigned int id = 0;
    std::unique_ptr<GPUTexture> icpError, rgbError;
    const GPUSetup& gpu = GPUSetup::getInstance();
    ModelProjection indexMap;
    PinOdometry* frameToModelPtr = nullptr;
    std::unique_ptr<char> fillIn;          // non-null: the model allows fill-in (the text only tests it)

  private:
    unsigned int id_;
    cv::Mat vc_, icp_;
    bool fillIn_ = false;
    std::vector<PoseLogItem> poseLog;
};
// `frameToModel` is a member object in the reference; the pasted text reaches the stand-in's tracker through this
#define frameToModel (*frameToModelPtr)
typedef std::shared_ptr<Model> ModelPointer;
typedef std::list<ModelPointer> ModelList;
typedef ModelList::iterator ModelListIterator;
