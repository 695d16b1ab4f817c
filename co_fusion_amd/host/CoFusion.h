// CoFusion.h -- host-side C++ facade with the reference's class/method names over the C-ABI
// (include/cofusion_hip.h).  This is the layer a maintainer of martinruenz/co-fusion would keep:
//   CoFusion  <- Core/CoFusion.h:44-391      (frame orchestrator, model list, spawn/deactivate)
//   Model     <- Core/Model/Model.h:51-285   (per-model surfel map + tracker)
//   Segmentation <- Core/Segmentation/Segmentation.h:30-140
// Eigen/OpenCV/Pangolin types are replaced by plain structs: poses are ROW-major float[16] (Mat4f),
// images are {pointer,width,height} views.  Only host logic lives here; every per-pixel / per-surfel
// operation is a call into the HIP library.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <list>
#include <memory>
#include <string>
#include <vector>

#include "../../include/cofusion_hip.h"

namespace cofusion {

struct Mat4f {
    float m[16];
    static Mat4f identity();
    Mat4f inverse() const;             // affine inverse, linear part by cofactors (f32)
    Mat4f operator*(const Mat4f& o) const;
};

// Core/FrameData.h:25-42 (host views; rgb is 3 bytes/pixel R,G,B; depth metres f32; mask u8 or null)
struct FrameData {
    int64_t timestamp = 0;
    const uint8_t* rgb = nullptr;
    const float* depth = nullptr;
    const uint8_t* mask = nullptr;
    // optional: the same frame already resident in HBM (depth f32, rgba u8x4); skips the upload
    const float* depth_dev = nullptr;
    const uint8_t* rgba_dev = nullptr;
};

struct SegmentationResult {
    struct ModelData {
        unsigned id = 0;
        int modelIndex = -1;  // position in the model list when the segmentation ran (-1: new label)
        unsigned superPixelCount = 0;
        float avgConfidence = 0, depthMean = 0, depthStd = 0;
        int top = 65535, right = 0, bottom = 0, left = 65535;
    };
    bool hasNewLabel = false;
    float depthRange = 0;
    std::vector<ModelData> modelData;
    std::vector<uint8_t> lowMap;  // 40x30 labels after component analysis (255 = rejected)
};

class CoFusion;
class EnqueuePool;

// host wall-clock per phase of processFrame (diagnostics: where the host keeps the GPU waiting)
struct PhaseTimes {
    enum { Prepare, Track, SegSlicAccumulate, SegUnary, SegCrf, SegPost, ModelLogic, Fuse, Predict, Count };
    double ms[Count] = {0};
    long frames = 0;
};
PhaseTimes& phaseTimes();
// 8-bit greyscale PNG (zlib), Export.cpp
bool writePngGray8(const std::string& path, const uint8_t* data, int width, int height);

// Model-parallel operation over several GPUs (one process per GPU): every rank runs the same frame loop and takes the
// same decisions; a model's surfel map and tracker live only on its owner rank, the other ranks keep a data-less
// shadow.  What crosses ranks -- poses and tracking statistics after tracking, per-superpixel ICP-error / confidence
// sums before the CRF, surfel counts when a model is retired -- goes through ONE primitive: an in-place SUM
// all-reduce of 64-bit integers (owners contribute their bit patterns, everybody else zeros), which is exact.
struct Distributed {
    int rank = 0, world = 1;
    int (*allreduce_i64)(int64_t* buf, uint64_t n, void* user) = nullptr;  // 0 on success
    void* user = nullptr;
    bool active() const { return world > 1; }
    // the background map (by far the largest) alone on rank 0, objects round-robin over the other ranks; colocate (Config::
    // colocateBackground, BASELINE.json configs[3]: "8 object models sharded one-per-GPU"): objects round-robin over ALL ranks, the
    // background shares rank 0 with the objects that land there
    bool colocate = false;
    int owner(unsigned id) const
    {
        if (world <= 1 || id == 0) return 0;
        return colocate ? (int)((id - 1) % (unsigned)world) : 1 + (int)((id - 1) % (unsigned)(world - 1));
    }
    // Background split over the ranks (Config::shardBackground): every rank keeps a replica of the background map and takes a share of
    // its two reductions -- the surfel range it rasterises into the index map (MIN all-reduce of the z-keys) and the image rows it
    // reduces in the ICP step (SUM all-reduce of the normal-equation accumulators after every launch of the Gauss-Newton loop).
    bool shardBackground = false;
    bool ownsHere(unsigned id) const { return (id == 0 && shardBackground && active()) ? true : owner(id) == rank; }
    // for the one-owner exchanges (poses, segmentation sums): does THIS rank contribute model `id`?
    bool contributes(unsigned id) const { return (id == 0 && shardBackground && active()) ? rank == 0 : owner(id) == rank; }
    void sum(int64_t* buf, uint64_t n) const;  // throws if the collective is missing or fails
    // the same collective on a DEVICE buffer, enqueued on `stream` (RCCL): no host visit.  Optional: without it device buffers are
    // staged through the host callback.
    int (*allreduce_dev)(int64_t* dev_buf, uint64_t n, void* hip_stream, void* user) = nullptr;
    void* user_dev = nullptr;
    void sumDevice(cf_ctx* ctx, int64_t* dev_buf, uint64_t n) const;
};

class Model {
  public:
    Model(cf_ctx* ctx, unsigned char id, float confidenceThresh, bool enableFillIn, int maxSurfels,
          float maxDepth = 3.402823466e+38f, bool owned = true);
    ~Model();
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    unsigned lastCount() const;
    void initialise(const uint8_t* rgba, const float* depthRaw, const float* depthFiltered, int time, float maxDepth);
    // Model::initICP (Model.cpp:350-367); frame-wide current maps are owned by `frameOdom` (model 0)
    void initICP(bool doFillIn, bool frameToFrameRGB, const float* const depthPyr[3], float depthCutoff, const uint8_t* rgba,
                 Model* frameOwner);
    // the two halves of initICP, used by the batched preparation of CoFusion::trackModels
    void trackingInputs(bool doFillIn, bool frameToFrameRGB, const float*& v, const float*& n, const uint8_t*& img) const;
    void bindFrameMaps(const float* const depthPyr[3], float depthCutoff, Model* frameOwner);
    float computeFusionWeight(float weightMultiplier) const;
    void fuse(int time, const uint8_t* rgba, const uint8_t* mask, const float* depthRaw, const float* depthFiltered, float depthCutoff,
              float weightMultiplier);
    void clean(int time, int timeDelta, float depthCutoff, const float* depthFiltered, const uint8_t* mask, float outlierCoeff);
    void predictIndices(int time, float depthCutoff, int timeDelta);
    void combinedPredict(float depthCutoff, int time, int maxTime, int timeDelta);
    void performFillIn(const uint8_t* rgba, const float* depthFiltered, bool frameToFrameRGB, bool lost);
    bool requiresFillIn(float ratio = 0.75f);
    // the same question left to the kernels that prepare the tracking (cf_odom_init_models_batch_select): device address of the counts
    // the last prefetchFillRatio() left, nullptr when this model never fills in or nothing was prefetched.  Never waits.
    const uint32_t* fillRatioDevice() const;
    void prefetchFillRatio();
    bool allowsFillIn() const { return fillIn; }
    std::vector<float> downloadMap() const;  // count x 12 floats

    float getConfidenceThreshold() const { return confidenceThreshold; }
    void setConfidenceThreshold(float v) { confidenceThreshold = v; }
    void setMaxDepth(float d) { maxDepth = d; }
    float getMaxDepth() const { return maxDepth; }
    const Mat4f& getPose() const { return pose; }
    const Mat4f& getLastPose() const { return lastPose; }
    void overridePose(const Mat4f& p) { pose = p; lastPose = p; }
    unsigned getID() const { return id; }
    unsigned incrementUnseenCount() { if (unseenCount < 0xFFFFFFFFu) return ++unseenCount; return unseenCount; }
    void resetUnseenCount() { unseenCount = 0; }
    cf_odom* getFrameOdometry() { return odom; }
    cf_model* handle() { return model; }
    float* icpErrorSurface() { return icpError; }
    const float* vertexConfProjection() const;
    cf_track_stats lastStats{};
    bool isOwned() const { return owned; }
    // split over the ranks (background replica): shard index / count, 0 / 1 when the model lives on one rank
    int shard = 0, shards = 1;

    struct PoseLogItem { int64_t ts; float p[7]; };  // x,y,z, qx,qy,qz,qw (Model.h:230-233)
    std::vector<PoseLogItem> poseLog;
    bool loggingPoses = false;
    bool isLoggingPoses() const { return loggingPoses; }

  private:
    friend class CoFusion;
    cf_ctx* ctx;
    cf_model* model = nullptr;
    cf_odom* odom = nullptr;
    float* icpError = nullptr;  // device f32 [H*W] (Model::icpError texture)
    Mat4f pose, lastPose;
    float confidenceThreshold;
    float maxDepth;
    unsigned id;
    unsigned unseenCount = 0;
    bool fillIn;
    bool owned = true;   // false: shadow of a model that lives on another rank (no device objects, only the replicated state)
};
typedef std::shared_ptr<Model> ModelPointer;
typedef std::list<ModelPointer> ModelList;

class Segmentation {
  public:
    Segmentation(cf_ctx* ctx, int width, int height, const Distributed* dist = nullptr);
    ~Segmentation();
    Segmentation(const Segmentation&) = delete;
    Segmentation& operator=(const Segmentation&) = delete;
    // Segmentation::performSegmentation (Segmentation.cpp:59-119 GT branch, :124-706 CRF branch)
    SegmentationResult performSegmentation(ModelList& models, const FrameData& frame, const float* depth_dev, const uint8_t* rgba_dev,
                                           const uint8_t* rgba_host_first_rows, unsigned char nextModelID, bool allowNew,
                                           uint8_t* fullSegmentation_dev);
    // enqueue SLIC for this frame's image ahead of performSegmentation (on whatever stream the context currently uses)
    void startSlic(const uint8_t* rgba_dev);
    // performSegmentationCRF in two halves without a host wait in between: everything is enqueued (sums, unaries, mean field,
    // component analysis, up-sampling into fullSegmentation_dev), the decisions are collected later
    void enqueueCRF(ModelList& models, const float* depth_dev, const uint8_t* rgba_dev, unsigned char nextModelID, bool allowNew,
                    uint8_t* fullSegmentation_dev);
    SegmentationResult finishCRF();
    // ... of several sequences through shared launches (lock-step groups): collectCRF does what enqueueCRF does up to the launches and
    // describes them as a job (its arrays live in this object until the next call), runBatch enqueues the jobs of all sequences in one
    // chain (cf_seg_run_batch); finishCRF as usual.  Single-process sequences only.
    void collectCRF(ModelList& models, const float* depth_dev, const uint8_t* rgba_dev, unsigned char nextModelID, bool allowNew,
                    uint8_t* fullSegmentation_dev, cf_seg_job& job);
    static void runBatch(cf_ctx* ctx, const Segmentation& params, const std::vector<cf_seg_job>& jobs);
    // model-parallel operation: enqueueCRF put every owner's tracked pose behind the sums it all-reduces; after the frame's host wait
    // this hands out [models][18] words (pose row-major, ICP error, ICP inlier count as f32 bit patterns).  false: nothing was published
    bool fetchPublishedPoses(size_t nModels, std::vector<int64_t>& words);
    // setters (Segmentation.h:100-120); defaults are the GUI values the reference applies every frame (GUI.h:206-227)
    float unaryWeightError = 75.f, unaryKError = 0.0375f, unaryThresholdNew = 5.5f;
    float weightAppearance = 7.f, weightSmoothness = 2.f;
    float scaleFeaturesRGB = 1.0f / 10, scaleFeaturesDepth = 1.0f / 0.9f, scaleFeaturesPos = 1.0f / 1.8f;
    float minRelSizeNew = 0.015f, maxRelSizeNew = 0.4f;
    unsigned crfIterations = 10;

  private:
    SegmentationResult performSegmentationCRF(ModelList& models, const float* depth_dev, const uint8_t* rgba_dev,
                                              const uint8_t* rgba_first_rows, unsigned char nextModelID, bool allowNew, uint8_t* full_dev);
    SegmentationResult performSegmentationGT(ModelList& models, const FrameData& frame, unsigned char nextModelID, bool allowNew,
                                             uint8_t* full_dev);
    cf_ctx* ctx;
    cf_segmenter* seg = nullptr;
    int width, height;
    uint8_t gtMapping[256];
    const Distributed* dist = nullptr;
    bool slicStarted = false;
    int pendingModels = 0;        // models of the segmentation enqueueCRF left in flight
    bool posesPublished = false;  // ... which carries the owners' poses in its all-reduce
    float* zeroImage = nullptr;   // device zeros [H*W*4] standing in for the ICP error / confidence maps of shadow models
    std::vector<const float*> jobIcp, jobConf;  // collectCRF's arrays
    std::vector<uint32_t> jobIds;
  public:
    cf_seg_params deviceParams() const;   // (the group compares the sequences' settings: one batched chain per distinct set)
};

class CoFusion {
  public:
    struct Config {
        int width = 640, height = 480;
        float fx = 528, fy = 528, cx = 320, cy = 240;
        int device = 0;
        int maxSurfels = 3072 * 3072;       // Model::MAX_VERTICES default (Model.cpp:92-98)
        int maxModels = 16;
        int timeDelta = 2147483647 / 2;     // openLoop (MainController.cpp:328)
        float confGlobalInit = 10.0f, confObjectInit = 0.01f;  // MainController.cpp:174-175
        float depthCutoff = 5.0f, icpWeight = 10.0f;           // GUI.h:211-212
        float outlierCoefficient = 3.0f;                       // GUI.h:213
        bool fastOdom = false, so3 = true, frameToFrameRGB = false, pyramid = true, rgbOnly = false;
        unsigned modelSpawnOffset = 22;                        // GUI.h:219
        bool enableMultipleModels = true;
        bool enablePoseLogging = false;                        // CoFusion ctor argument (CoFusion.h:59)
        int rank = 0, world = 1;                               // model-parallel operation (see Distributed)
        // Device-resident frames (FrameData::depth_dev): true = the buffers are COMPLETE when processFrame is called (e.g. a ring
        // of frames uploaded ahead); the depth filter of the new frame then runs on an auxiliary stream that is NOT ordered after
        // the work still queued on the context's stream, i.e. beside the previous frame's fusion passes.  false (default) = they
        // may be produced by work queued on that stream just before the call, and are consumed in stream order.
        bool deviceFramesComplete = false;
        // run CoFusion::predict() between tracking and fusion as the reference does (CoFusion.cpp:346).  Its outputs are overwritten by
        // the end-of-frame prediction before anything in the frame loop reads them (they are what the reference's GUI shows), so the
        // default skips the pass; results are identical either way.
        bool midFramePredict = false;
        // model-parallel operation only: split the background's index-map rasterisation (by surfel range) and ICP reduction (by image
        // rows) over all ranks, each holding a replica of the background map (see Distributed::shardBackground)
        bool shardBackground = false;
        // model-parallel operation only: object models round-robin over ALL ranks, the background sharing rank 0 with the objects that
        // land there (BASELINE.json configs[3]: 8 object models one per GPU); default: the background alone on rank 0
        bool colocateBackground = false;
        // CoFusion's `reloc` constructor argument (CoFusion.h:47, -rl on the command line): the failure detection of the frame loop --
        // a frame whose background ICP error / pose covariance is out of bounds is not fused, and after ten such frames the camera is
        // `lost`: no fusion, the clock stops (CoFusion.cpp:225, 301-338, 463, 495).  The fern-based recovery is out of scope.
        bool reloc = false;
        // the index maps of the tracked models rasterised before the frame's host wait, with the poses on the device (framePreIndex)
        bool earlyIndexMaps = true;
        // host threads that enqueue the per-model surfel passes (fusion, clean-up, prediction) beside the calling thread, one model's
        // chain of launches each.  Pays when the host's launch rate is the limit (a profiler attached, a slow or busy host); on an idle
        // host the calling thread alone keeps the lanes fed (610 / 606 / 604 fps with 0 / 2 / 4 helpers), hence default
        // 0 = everything from the calling thread.  Results do not depend on it.
        int enqueueThreads = 0;
    };
    explicit CoFusion(const Config& cfg);
    // a sequence of a lock-step group (CoFusionGroup): the context is the group's, shared with the other sequences
    CoFusion(const Config& cfg, cf_ctx* sharedContext, int sequenceIndex);
    ~CoFusion();

    // CoFusion::processFrame (Core/CoFusion.cpp:171-524)
    bool processFrame(const FrameData& frame, const Mat4f* inPose = nullptr, float weightMultiplier = 1.f, bool bootstrap = false);
    void predict(bool lastOfFrame = false);          // CoFusion.cpp:533-545
    // CoFusion::savePly / exportPoses (CoFusion.cpp:646-783); exportDir is a prefix ("out/"); return files written or -1
    int savePly(const std::string& exportDir);
    int exportPoses(const std::string& exportDir);
    // exportSegmentation (CoFusion.cpp:235-240): when set, every segmented frame writes <prefix>Segmentation<tick>.png (8-bit labels)
    void setExportSegmentation(const std::string& prefix) { exportSegmentationPrefix = prefix; }
    // the collective of the model-parallel mode (cfg.world > 1); must be set before the first frame
    void setAllreduce(int (*fn)(int64_t*, uint64_t, void*), void* user) { dist.allreduce_i64 = fn; dist.user = user; }
    void setAllreduceDevice(int (*fn)(int64_t*, uint64_t, void*, void*), void* user) { dist.allreduce_dev = fn; dist.user_dev = user; }
    // the library's own RCCL communicator (cf_rccl_init) as the collective of this instance: device buffers are all-reduced in place by
    // ncclAllReduce on the context's stream, host buffers through a small device staging buffer; also registers the collective of a
    // split background.  `id128` is the ncclUniqueId rank 0 created (cofusion_rccl_unique_id).  Collective call: every rank of cfg.world.
    void initRccl(const void* id128);
    // depth + colour of a frame (any device buffer) from rank `root` to every rank: ncclBroadcast on the context's stream
    void broadcast(void* dev_buf, uint64_t bytes, int root);
    int rcclSumHost(int64_t* buf, uint64_t n);  // (the host-buffer flavour of the collective; 0 on success)
    const Distributed& distributed() const { return dist; }
    ModelList& getModels() { return models; }
    ModelPointer getBackgroundModel() { return globalModel; }
    const Mat4f& getCurrPose() const { return globalModel->getPose(); }
    int getTick() const { return tick; }
    bool getLost() const { return lost; }  // CoFusion::getLost (CoFusion.h:183-185): the camera is lost (Config::reloc)
    const uint8_t* maskDevice() const { return mask_dev; }
    Segmentation& segmentation() { return *labelGenerator; }
    cf_ctx* context() { return ctx; }
    Config cfg;

    // ---- the stages of processFrame (see CoFusion.cpp); CoFusionGroup runs them stage by stage over its sequences ----
    struct TrackBatch {   // the trackers of one set of lock-step launches: the owned models of one frame, or of several sequences' frames
        struct Item { Model* model; Model* owner; const float* depthPyr[3]; const uint8_t* frameRgba; float maxDepth;
                      const float* predV; const float* predN; const uint8_t* predImg;
                      // the fill-in flavour of the three and the device counts that decide between them (fillCounts == nullptr: no choice)
                      const float* altV; const float* altN; const uint8_t* altImg; const uint32_t* fillCounts; };
        std::vector<Item> items;
    };
    void frameBegin(const FrameData& frame, const Mat4f* inPose, float weightMultiplier, bool bootstrap);
    bool frameTracks() const { return st.willTrack; }
    void trackCollect(TrackBatch& batch) { trackCollect(batch, st.pyr); }
    static void trackLaunch(cf_ctx* ctx, TrackBatch& batch, const Config& cfg);
    // segmentation enqueued (jobs != nullptr: described as a job for the group's shared launches instead -- Segmentation::runBatch)
    void frameSegment(std::vector<cf_seg_job>* jobs);
    void framePreIndex();
    void frameCollect();           // the frame's host wait (poses + segmentation decisions), model bookkeeping
    void frameFuse(bool join, int laneOffset);
    // ... in two halves for a lock-step group: every sequence adds its models' passes to ONE batch (cf_models_frame_passes), the group
    // launches it, each sequence then runs its fill-in
    bool passesBatched() const { return !dist.active(); }
    float depthLimit() const { return maxDepthProcessed; }
    void frameFuseCollect(std::vector<cf_model_pass>& items);
    void frameFuseFinish();
    void frameEnd();

  private:
    struct FrameStage {   // what the stages of one frame hand to each other
        const FrameData* frame = nullptr; const Mat4f* inPose = nullptr; float weightMultiplier = 1.f; bool bootstrap = false;
        unsigned b = 0; bool willTrack = false, slicAside = false, fuseNow = false, allowNew = false, segOnDevice = false;
        const float* pyr[3] = {nullptr, nullptr, nullptr};
    } st;
    bool ownsCtx = true;
    int markBase = 0;   // this sequence's four event slots of the context (cf_mark)
    void trackCollect(TrackBatch& batch, const float* const depthPyr[3]);
    void spawnObjectModel();
    void moveNewModelToList();
    ModelList::iterator inactivateModel(ModelList::iterator it);
    unsigned char getNextModelID(bool assign = false);
    void trackModels(const float* const depthPyr[3]);   // enqueues; poses arrive with fetchTracking
    void fetchTracking(bool exchange);
    void exchangeTracking();
    std::vector<Model*> trackPending;
    Distributed dist;

    cf_ctx* ctx = nullptr;
    ModelList models, inactiveModels;
    ModelPointer newModel, globalModel;
    unsigned char nextID = 0;
    std::unique_ptr<Segmentation> labelGenerator;
    int tick = 1;
    float maxDepthProcessed = 20.0f;
    unsigned spawnOffset = 0;
    int64_t* rcclStage = nullptr;      // device staging buffer of the host-buffer all-reduce (initRccl)
    uint64_t rcclStageWords = 0;
    bool capReported = false;  // the model cap suppressed a spawn and said so
    bool lost = false;          // CoFusion.h:362 (reloc)
    int trackingCount = 0;      // CoFusion.h:364
    // device frame buffers (CoFusion::textures)
    // filtered depth + its pyramid are double buffered: the filter of frame t+1 runs on an auxiliary stream while the fusion
    // passes of frame t still read frame t's filtered depth (processFrame)
    float *depth_dev = nullptr, *depthFiltered_dev = nullptr, *depthPyr1 = nullptr, *depthPyr2 = nullptr;
    float *depthFilteredBuf[2] = {nullptr, nullptr}, *depthPyr1Buf[2] = {nullptr, nullptr}, *depthPyr2Buf[2] = {nullptr, nullptr};
    unsigned frameParity = 0;
    uint8_t *rgba_dev = nullptr, *rgb_dev = nullptr, *mask_dev = nullptr;
    uint8_t* stage[2] = {nullptr, nullptr};   // pinned staging of the host-input path (depth f32 | rgb u8x3)
    unsigned uploads = 0;
    const float* curDepth = nullptr;   // device pointers of the frame being processed
    const uint8_t* curRgba = nullptr;
    unsigned modelKeepMinSurfels = 4000;
    float modelKeepConfThreshold = 0.3f;
    bool enableSmartModelDelete = true;
    std::string exportSegmentationPrefix;
    bool useLanes = true;  // per-model auxiliary streams
    std::shared_ptr<EnqueuePool> pool;                      // Config::enqueueThreads helpers
    void modelPasses(Model& model, bool fuse, float weightMultiplier, bool lost);
    void fuseAndPredict(bool fuse, float weightMultiplier, bool lost, bool join = true, int laneOffset = 0);
};

// Several independent sequences on one GPU in lock-step: one context, one set of tracking launches for the trackers of all of them
// (CoFusion.cpp, "CoFusionGroup").  Same configuration (image size, intrinsics, options) for every sequence.
class CoFusionGroup {
  public:
    CoFusionGroup(const CoFusion::Config& cfg, int sequences);
    ~CoFusionGroup();
    CoFusionGroup(const CoFusionGroup&) = delete;
    CoFusionGroup& operator=(const CoFusionGroup&) = delete;
    int size() const { return (int)seqs.size(); }
    CoFusion& sequence(int s) { return *seqs.at((size_t)s); }
    // one frame of EVERY sequence: frames[s] goes to sequence s (inPoses: nullable, entries nullable)
    void processFrames(const FrameData* frames, const Mat4f* const* inPoses = nullptr);
    cf_ctx* context() { return ctx; }
    bool hasFailed() const { return failed; }

  private:
    void stepAll(const FrameData* frames, const Mat4f* const* inPoses);
    CoFusion::Config cfg;
    cf_ctx* ctx = nullptr;
    std::vector<std::unique_ptr<CoFusion>> seqs;
    bool failed = false;  // a stage threw: the sequences are out of step, further frames are refused
};

}  // namespace cofusion
