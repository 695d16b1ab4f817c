// Minimal cv::Mat stand-in so that g++ can compile the reference's header-only Core/Segmentation/ConnectedLabels.hpp where it
// lies (TEST INFRASTRUCTURE ONLY, oracle/ref_shim).  Only what that header touches: a dense row-major 2-D buffer of uchar or int.
#pragma once
#include <assert.h>
#include <limits>
#include <list>
#include <memory>
#include <stddef.h>
#include <vector>

typedef unsigned char uchar;
#define CV_8UC1 0
#define CV_32SC1 4

namespace cv {
template <class T> struct DataType;
template <> struct DataType<int> { enum { type = CV_32SC1 }; };
template <> struct DataType<uchar> { enum { type = CV_8UC1 }; };

struct Mat {
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int t) : rows(r), cols(c), type_(t), own_(new uchar[(size_t)r * c * (t == CV_8UC1 ? 1 : 4)]()) { data = own_.get(); }
    Mat(int r, int c, int t, void* external) : rows(r), cols(c), data((uchar*)external), type_(t) {}
    int type() const { return type_; }
    size_t total() const { return (size_t)rows * cols; }
    template <class T> T* ptr(int r = 0) { return (T*)data + (size_t)r * cols; }
    template <class T> const T* ptr(int r = 0) const { return (const T*)data + (size_t)r * cols; }
private:
    int type_ = CV_8UC1;
    std::shared_ptr<uchar> own_;
};
}  // namespace cv
