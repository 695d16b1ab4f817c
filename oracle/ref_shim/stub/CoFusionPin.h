// Declarations around the text of Core/CoFusion.cpp's frame loop (TEST INFRASTRUCTURE ONLY).
//
// build_ref.py pastes the reference's OWN function bodies -- CoFusion::processFrame, performSegmentation, predict, requiresFillIn,
// spawnObjectModel, moveNewModelToList, inactivateModel, getNextModelID (Core/CoFusion.cpp:111-113, 171-524, 533-565, 588-644),
// cut out of the file where it lies at build time -- behind this header.  The real CoFusion.h drags in OpenGL, Pangolin, the
// deformation graph, the fern database and every shader wrapper; this header declares the class with the members those functions
// use (names, types and defaults as in CoFusion.h:266-391 and the constructor's initialiser list, CoFusion.cpp:21-77) and
// compile-only stand-ins for the collaborators of the loop-closure branch, which is off in Co-Fusion (closeLoops == false; the
// relocalisation switch `reloc` is a constructor argument and pinned: ref_cf_set_reloc).  The passes a frame
// consists of run on the CPU oracle (stub/Model/Model.h, ref_cofusion.cpp): the pin is about the ORDER and CONDITIONS of the
// frame loop, i.e. SURVEY.md 8 row a17.
#pragma once
#include <Eigen/Geometry>
#include <stdint.h>

#include <algorithm>
#include <cassert>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "GPUTexture.h"
#include "Model/Model.h"
#include "Utils/Stopwatch.h"
#include "Utils/Resolution.h"       // the reference's own (header-only singleton)
#include "Utils/Img.h"              // the reference's own
#include "Callbacks.h"              // the reference's own
#include "FrameData.h"              // the reference's own
#include "Segmentation/Segmentation.h"  // the reference's own (compiled in another translation unit of this library)

namespace cv {
static const int THRESH_TOZERO_INV = 4;
inline void threshold(const Mat&, Mat&, double, double, int) {}   // (exportSegmentation is off)
inline bool imwrite(const std::string&, const Mat&) { return true; }
}  // namespace cv

// vertex_feedback.* output of one image: [count][12] floats (Shaders/FeedbackBuffer.h)
class FeedbackBuffer {
  public:
    static constexpr const char* RAW = "RAW";
    static constexpr const char* FILTERED = "FILTERED";
    std::vector<float> data;
    int count = 0;
};

// ---- collaborators of the loop-closure / relocalisation branches (closeLoops == false, reloc == false): compile-only -------------
class Ferns {
  public:
    struct SurfaceConstraint {
        SurfaceConstraint(const Eigen::Vector4f& s, const Eigen::Vector4f& t) : sourcePoint(s), targetPoint(t) {}
        Eigen::Vector4f sourcePoint, targetPoint;
    };
    struct Frame { int srcTime = 0; Eigen::Matrix4f pose; };
    template <class... A> Eigen::Matrix4f findFrame(A&&...) { return Eigen::Matrix4f::Identity(); }
    int lastClosest = -1;
    std::vector<Frame*> frames;
};
class Deformation {
  public:
    struct Constraint {};
    template <class... A> void addConstraint(A&&...) {}
    template <class... A> bool constrain(A&&...) { return false; }
};
struct PoseMatch {
    template <class... A> PoseMatch(A&&...) {}
};
class GPUResize {
  public:
    // Shaders/GPUResize + resize.frag: nearest texel of the source at the centre of every destination texel (pinned on its own:
    // test_requires_fill_in_matches_resize_shader)
    void image(GPUTexture* source, Img<Eigen::Matrix<unsigned char, 3, 1>>& dest);
    template <class... A> void vertex(A&&...) {}
    template <class... A> void time(A&&...) {}
};

class CoFusion {
  public:
    CoFusion(int width, int height, float fx, float fy, float cx, float cy, float initConfidenceGlobal, float initConfidenceObject, float depthCut,
             float icpThresh, bool so3, unsigned modelSpawnOffset, bool enableMultipleModels);
    ~CoFusion();

    SegmentationResult performSegmentation(const FrameData& frame);
    bool processFrame(const FrameData& frame, const Eigen::Matrix4f* inPose = 0, const float weightMultiplier = 1.f, const bool bootstrap = false);
    void predict();

    // (harness access)
    ModelList& getModels() { return models; }
    GPUTexture* maskTexture() { return textures[GPUTexture::MASK]; }
    int getTick() const { return tick; }
    bool isLost() const { return lost; }           // CoFusion::getLost (CoFusion.cpp:846-848)
    Segmentation& segmentation() { return labelGenerator; }
    void setTrackingOptions(bool rgbOnly_, bool pyramid_, bool fastOdom_, bool frameToFrameRGB_)   // CoFusion::setRgbOnly / setPyramid / setFastOdom / setFrameToFrameRGB
    { rgbOnly = rgbOnly_; pyramid = pyramid_; fastOdom = fastOdom_; frameToFrameRGB = frameToFrameRGB_; }

  private:
    void spawnObjectModel();
    void moveNewModelToList();
    ModelListIterator inactivateModel(const ModelListIterator& it);
    unsigned char getNextModelID(bool assign = false);
    void computeFeedbackBuffers();   // ref_cofusion.cpp (vertex_feedback passes on the oracle)
    void filterDepth();              // ref_cofusion.cpp (bilateral filter on the oracle)
    bool requiresFillIn(ModelPointer model, float ratio = 0.75f);

    // state the frame loop reads and writes (what CoFusion.h:266-391 declares for it), grouped by type; defaults as in that header
    ModelList models, inactiveModels, preallocatedModels;   // models.front() is the static environment
    ModelPointer newModel, globalModel;
    unsigned char nextID = 0;
    Segmentation labelGenerator;
    Model::MatchingType modelMatchingType;
    CallbackBuffer<std::shared_ptr<Model>> newModelListeners, inactiveModelListeners;
    PinOdometry modelToModel;
    Ferns ferns;
    Deformation localDeformation, globalDeformation;
    std::map<std::string, GPUTexture*> textures;
    std::map<std::string, FeedbackBuffer*> feedbackBuffers;
    GPUResize resize;
    std::vector<PoseMatch> poseMatches;
    std::vector<Deformation::Constraint> relativeCons;
    Img<Eigen::Matrix<unsigned char, 3, 1>> imageBuff;
    Img<Eigen::Vector4f> consBuff;
    Img<unsigned short> timesBuff;
    std::string exportDir;
    int tick, deforms, fernDeforms, trackingCount;
    const int timeDelta, icpCountThresh, consSample;
    const float icpErrThresh, covThresh, maxDepthProcessed;
    const bool closeLoops, iclnuim, reloc;
    bool lost, lastFrameRecovery, rgbOnly, pyramid, fastOdom, so3, frameToFrameRGB, exportSegmentation;
    bool enableMultipleModels = true, enableSmartModelDelete = true, enableRedetection = false, enableModelMerging = false,
         enableSpawnSubtraction = true, enablePoseLogging = true;
    float icpWeight, initConfThresGlobal, initConfThresObject, fernThresh, depthCutoff, modelKeepConfThreshold = 0.3;
    unsigned modelDeactivateCount = 10, modelKeepMinSurfels = 4000, modelSpawnOffset, spawnOffset = 0;
};
