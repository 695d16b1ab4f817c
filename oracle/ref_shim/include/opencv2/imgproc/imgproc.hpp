// Minimal cv::Mat stand-in so that g++ can compile the reference's segmentation sources where they lie
// (Core/Segmentation/{ConnectedLabels.hpp, Slic.h, Slic.cpp, Segmentation.h, Segmentation.cpp}, Core/FrameData.h).
// TEST INFRASTRUCTURE ONLY (oracle/ref_shim).  Only what those files touch: a dense, continuous, reference-counted row-major 2-D
// buffer with OpenCV's type codes, Vec3b / Vec3i / Point2i.
#pragma once
#include <assert.h>
#include <cmath>
#include <limits>
#include <list>
#include <memory>
#include <stddef.h>
#include <string.h>
#include <vector>

typedef unsigned char uchar;
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8U 0
#define CV_8S 1
#define CV_32S 4
#define CV_32F 5
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32SC3 CV_MAKETYPE(CV_32S, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)

namespace cv {
template <class T> struct DataType;
template <> struct DataType<int> { enum { type = CV_32SC1 }; };
template <> struct DataType<uchar> { enum { type = CV_8UC1 }; };
template <> struct DataType<char> { enum { type = CV_MAKETYPE(CV_8S, 1) }; };
template <> struct DataType<float> { enum { type = CV_32FC1 }; };

template <class T, int N> struct Vec {
    T v[N];
    Vec() { for (int i = 0; i < N; i++) v[i] = T(); }
    Vec(T a, T b, T c) { static_assert(N == 3, "Vec3 only"); v[0] = a; v[1] = b; v[2] = c; }
    T& operator[](int i) { return v[i]; }
    const T& operator[](int i) const { return v[i]; }
    Vec& operator+=(const Vec& o) { for (int i = 0; i < N; i++) v[i] += o.v[i]; return *this; }
};
typedef Vec<uchar, 3> Vec3b;
typedef Vec<int, 3> Vec3i;
template <> struct DataType<Vec3b> { enum { type = CV_8UC3 }; };
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<int> Point2i;

struct Mat {
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    Mat() {}
    Mat(int r, int c, int t) : rows(r), cols(c), type_(t), own_(new uchar[(size_t)r * c * elem(t)](), std::default_delete<uchar[]>()) { data = own_.get(); }
    Mat(int r, int c, int t, void* external) : rows(r), cols(c), data((uchar*)external), type_(t) {}
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }  // the owning constructor value-initialises
    int type() const { return type_; }
    int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
    bool isContinuous() const { return true; }
    size_t total() const { return (size_t)rows * cols; }
    template <class T> T* ptr(int r = 0) { return (T*)data + (size_t)r * cols; }
    template <class T> const T* ptr(int r = 0) const { return (const T*)data + (size_t)r * cols; }
    static size_t elem(int t) { const int d = t & 7; return (size_t)((t >> CV_CN_SHIFT) + 1) * (d == CV_8U || d == CV_8S ? 1 : 4); }
private:
    int type_ = CV_8UC1;
    std::shared_ptr<uchar> own_;
};
}  // namespace cv
