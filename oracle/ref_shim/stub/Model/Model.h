// Stand-in for Core/Model/Model.h (TEST INFRASTRUCTURE ONLY): the three members Core/Segmentation/Segmentation.cpp calls on a
// model.  The real class owns OpenGL textures; here the two downloads return host images supplied by the test harness.
#pragma once
#include <opencv2/imgproc/imgproc.hpp>
class Model {
  public:
    Model(unsigned char id, cv::Mat vertConf /* CV_32FC4 */, cv::Mat icpError /* CV_32FC1 */) : id_(id), vc_(vertConf), icp_(icpError) {}
    unsigned char getID() const { return id_; }
    cv::Mat downloadVertexConfTexture() { return vc_; }
    cv::Mat downloadICPErrorTexture() { return icp_; }
  private:
    unsigned char id_;
    cv::Mat vc_, icp_;
};
