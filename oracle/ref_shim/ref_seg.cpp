// ref_seg.cpp -- flat C entry points over the reference's OWN Segmentation::performSegmentation / performSegmentationCRF
// (Core/Segmentation/Segmentation.cpp:59-706, Slic.h, Slic.cpp, ConnectedLabels.hpp), compiled from /root/reference by
// build_ref.py.  TEST INFRASTRUCTURE ONLY.  This file is appended to the generated translation unit that holds
// Segmentation.cpp's text, so Segmentation / SegmentationResult / Model (stand-in) / FrameData are in scope.
//
// What is NOT the reference here, and why: gSLICr and densecrf are third-party libraries absent from the reference tree; their
// stand-ins (include/gSLICr.h, include/densecrf.h) delegate to the oracle's SLIC and exact mean-field operations.  Everything
// else -- down/up-sampling incl. the empty-superpixel fallback, depth range, unary construction, the inference loop's arithmetic
// around the two library calls, arg-max, connected components, largest-component / size / border gates, bounding boxes, depth
// statistics with the trimming pass, super-pixel counts, the ground-truth-mask branch -- is the reference's own text.
#include <string.h>

struct ref_seg_params {  // orc_seg_params with the three pairwise SIGMAS in place of their reciprocals
    float unaryWeightError, unaryKError, unaryThresholdNew, weightAppearance, weightSmoothness;
    float sigmaRGB, sigmaDepth, sigmaPos, minRelSizeNew, maxRelSizeNew;
    int crfIterations;
};
struct ref_seg_model {  // == orc_seg_model
    unsigned id, superPixelCount;
    float avgConfidence, depthMean, depthStd;
    int top, right, bottom, left;
};

static void fill_models(const SegmentationResult& r, ref_seg_model* out, int* n_out)
{
    *n_out = (int)r.modelData.size();
    for (size_t i = 0; i < r.modelData.size(); i++) {
        const auto& m = r.modelData[i];
        out[i].id = m.id; out[i].superPixelCount = m.superPixelCount; out[i].avgConfidence = m.avgConfidence;
        out[i].depthMean = m.depthMean; out[i].depthStd = m.depthStd;
        out[i].top = m.top; out[i].right = m.right; out[i].bottom = m.bottom; out[i].left = m.left;
    }
}

extern "C" int ref_segment_crf(const ref_seg_params* P, int cols, int rows, const unsigned char* rgb3, const float* depth, int n_models,
                               const unsigned* model_ids, const float* const* icp_err, const float* const* vertconf4, unsigned nextModelID,
                               int allowNew, unsigned char* full_seg, ref_seg_model* out_models, int* n_out, int* hasNewLabel,
                               float* depthRange, float* lowDepth_out, float* lowICP_out /* [n_models][K] after the unary edits */,
                               float* lowConf_out)
{
    Segmentation seg;
    seg.init(cols, rows, Segmentation::METHOD::CONNECTED_COMPONENTS);
    seg.setUnaryWeightError(P->unaryWeightError); seg.setUnaryKError(P->unaryKError); seg.setUnaryThresholdNew(P->unaryThresholdNew);
    seg.setPairwiseWeightAppearance(P->weightAppearance); seg.setPairwiseWeightSmoothness(P->weightSmoothness);
    // the reference's setters take the sigmas and store 1.0f / sigma (Segmentation.h:100-102, GUI.h:216-218)
    seg.setPairwiseSigmaRGB(P->sigmaRGB); seg.setPairwiseSigmaDepth(P->sigmaDepth); seg.setPairwiseSigmaPosition(P->sigmaPos);
    seg.setIterationsCRF((unsigned)P->crfIterations);
    seg.setNewModelMinRelativeSize(P->minRelSizeNew); seg.setNewModelMaxRelativeSize(P->maxRelSizeNew);

    FrameData frame;
    frame.timestamp = 0;
    frame.rgb = cv::Mat(rows, cols, CV_8UC3, (void*)rgb3);
    frame.depth = cv::Mat(rows, cols, CV_32FC1, (void*)depth);
    std::list<std::shared_ptr<Model>> models;
    for (int m = 0; m < n_models; m++)
        models.push_back(std::make_shared<Model>((unsigned char)model_ids[m], cv::Mat(rows, cols, CV_32FC4, (void*)vertconf4[m]),
                                                 cv::Mat(rows, cols, CV_32FC1, (void*)icp_err[m])));
    SegmentationResult r = seg.performSegmentationCRF(models, frame, (unsigned char)nextModelID, allowNew != 0);
    memcpy(full_seg, r.fullSegmentation.data, (size_t)cols * rows);
    fill_models(r, out_models, n_out);
    *hasNewLabel = r.hasNewLabel ? 1 : 0;
    *depthRange = r.depthRange;
    const size_t K = r.lowDepth.total();
    if (lowDepth_out) memcpy(lowDepth_out, r.lowDepth.data, K * sizeof(float));
    for (int m = 0; m < n_models; m++) {
        if (lowICP_out) memcpy(lowICP_out + (size_t)m * K, r.modelData[m].lowICP.data, K * sizeof(float));
        if (lowConf_out) memcpy(lowConf_out + (size_t)m * K, r.modelData[m].lowConf.data, K * sizeof(float));
    }
    return 0;
}

// the ground-truth-mask branch (Segmentation.cpp:59-119).  Its label mapping is a function-local static of the reference: it
// persists across calls for the life of the process, exactly as in the reference.
extern "C" int ref_segment_gt(const unsigned char* gt_mask, const float* depth, int cols, int rows, int n_models, const unsigned* model_ids,
                              unsigned nextModelID, int allowNew, unsigned char* full_seg, ref_seg_model* out_models, int* n_out,
                              int* hasNewLabel)
{
    Segmentation seg;
    seg.init(cols, rows, Segmentation::METHOD::CONNECTED_COMPONENTS);
    FrameData frame;
    frame.timestamp = 0;
    frame.mask = cv::Mat(rows, cols, CV_8UC1, (void*)gt_mask);
    frame.depth = cv::Mat(rows, cols, CV_32FC1, (void*)depth);
    std::list<std::shared_ptr<Model>> models;
    for (int m = 0; m < n_models; m++) models.push_back(std::make_shared<Model>((unsigned char)model_ids[m], cv::Mat(), cv::Mat()));
    SegmentationResult r = seg.performSegmentation(models, frame, (unsigned char)nextModelID, allowNew != 0);
    memcpy(full_seg, r.fullSegmentation.data, (size_t)cols * rows);
    fill_models(r, out_models, n_out);
    *hasNewLabel = r.hasNewLabel ? 1 : 0;
    return 0;
}
