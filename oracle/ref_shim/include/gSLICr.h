// gSLICr stand-in (TEST INFRASTRUCTURE ONLY).  gSLICr is a third-party dependency that is NOT part of the reference tree
// (HEAD clone, Scripts/install.sh:85).  This header provides the few types Core/Segmentation/Slic.{h,cpp} touch; the engine's
// Perform_Segmentation is the ORACLE's SLIC (oracle/orc_segment.c: orc_slic, the published algorithm with the call-site settings of
// Slic.cpp:33-43), so that everything the reference itself does around the superpixels runs on the reference's own code.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
extern "C" void orc_slic(const uint8_t* rgba, int cols, int rows, int32_t* labels);
#define MEMORYDEVICE_CPU 0
namespace gSLICr {
enum COLOR_SPACE { CIELAB = 0, XYZ, RGB };
enum SEG_METHOD { GIVEN_NUM = 0, GIVEN_SIZE };
struct Vector2i { int x = 0, y = 0; int operator[](int i) const { return i == 0 ? x : y; } };
struct Vector4u { unsigned char r, g, b, a; };
template <class T> class Image {
  public:
    Vector2i noDims;
    size_t dataSize;
    Image(Vector2i size, bool, bool) : noDims(size), dataSize((size_t)size.x * size.y), buf(dataSize) {}
    T* GetData(int) { return buf.data(); }
    const T* GetData(int) const { return buf.data(); }
  private:
    std::vector<T> buf;
};
typedef Image<Vector4u> UChar4Image;
typedef Image<int> IntImage;
namespace objects {
struct settings {
    Vector2i img_size; int no_segs = 0, spixel_size = 0, no_iters = 0; float coh_weight = 0; bool do_enforce_connectivity = false;
    COLOR_SPACE color_space = RGB; SEG_METHOD seg_method = GIVEN_SIZE;
};
}
namespace engines {
class seg_engine_GPU {
  public:
    explicit seg_engine_GPU(const objects::settings& s) : st(s), mask(s.img_size, true, true) {}
    void Perform_Segmentation(UChar4Image* in)
    {
        const Vector4u* p = in->GetData(MEMORYDEVICE_CPU);
        std::vector<uint8_t> rgba(in->dataSize * 4);
        // Slic::setInputImage swaps red and blue on the way in (swapRedBlue defaults to true, Slic.cpp:70-77); the RGB-space
        // distance is symmetric in the channels, so the oracle is handed the channels in the order it sees them itself
        for (size_t i = 0; i < in->dataSize; i++) { rgba[4 * i] = p[i].b; rgba[4 * i + 1] = p[i].g; rgba[4 * i + 2] = p[i].r; rgba[4 * i + 3] = 255; }
        orc_slic(rgba.data(), st.img_size.x, st.img_size.y, mask.GetData(MEMORYDEVICE_CPU));
    }
    const IntImage* Get_Seg_Mask() const { return &mask; }
  private:
    objects::settings st;
    IntImage mask;
};
}
}  // namespace gSLICr
