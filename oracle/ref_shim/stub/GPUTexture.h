// Stand-in for Core/GPUTexture.h (a Pangolin OpenGL texture with CUDA interop), TEST INFRASTRUCTURE ONLY: host memory behind the
// members the reference's host code touches -- Core/Utils/RGBDOdometry.cpp (map / unmap / the cudaArray behind the texture) and
// Core/CoFusion.cpp (`textures[...]->texture->Upload(data, format, type)`, the texture names).
#pragma once
#include <cuda_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include <vector>

// the OpenGL enumerants at the Upload call sites of CoFusion.cpp (values as in GL/gl.h, only their identity matters)
#define GL_RGB 0x1907
#define GL_LUMINANCE 0x1909
#define GL_LUMINANCE_INTEGER_EXT 0x8D9C
#define GL_UNSIGNED_BYTE 0x1401
#define GL_FLOAT 0x1406

class GPUTexture {
  public:
    struct GlTexture {
        GPUTexture* owner;
        unsigned tid;   // the GL texture name the reference's glBindTexture calls pass around (stub/glpin.h records them)
        // glTexSubImage2D into the texture's storage: GL_RGB bytes land in an RGBA8 texture with alpha 255, luminance data as they are
        void Upload(const void* data, int format, int /*type*/)
        {
            const size_t n = (size_t)owner->arr.width * owner->arr.height;
            if (format == GL_RGB) {
                const unsigned char* s = (const unsigned char*)data; unsigned char* d = (unsigned char*)owner->arr.data;
                for (size_t i = 0; i < n; i++) { d[4 * i] = s[3 * i]; d[4 * i + 1] = s[3 * i + 1]; d[4 * i + 2] = s[3 * i + 2]; d[4 * i + 3] = 255; }
            } else memcpy(owner->arr.data, data, n * owner->texel_bytes);
        }
    };
    static constexpr const char* RGB = "RGB";
    static constexpr const char* DEPTH_METRIC = "DEPTH_METRIC";
    static constexpr const char* DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
    static constexpr const char* MASK = "MASKS";
    static constexpr const char* DEPTH_NORM = "DEPTH_NORM";
    static constexpr const char* MASK_COLOR = "MASKS_COLOR";

    GPUTexture(void* texels, int width, int height) : arr{texels, width, height}, texel_bytes(0), gl{this, enrol(this)}, texture(&gl) {}  // a view
    GPUTexture(int width, int height, int bytes_per_texel)
        : own((size_t)width * height * bytes_per_texel, 0), arr{own.data(), width, height}, texel_bytes(bytes_per_texel), gl{this, enrol(this)}, texture(&gl) {}
    ~GPUTexture() { registry()[gl.tid] = nullptr; }
    // texture name -> object (what a recorded glBindTexture resolves to); a view may be re-pointed at new storage of the same size
    static std::vector<GPUTexture*>& registry() { static std::vector<GPUTexture*> r(1, nullptr); return r; }
    static unsigned enrol(GPUTexture* t) { registry().push_back(t); return (unsigned)registry().size() - 1; }
    static GPUTexture* by_name(unsigned tid) { return tid < registry().size() ? registry()[tid] : nullptr; }
    void repoint(void* texels) { arr.data = texels; }
    // Model::performTracking hands the error textures to the tracker as CUDA surfaces (Model.cpp:383-384): here the object itself
    cudaSurfaceObject_t getCudaSurface() { return (cudaSurfaceObject_t)(uintptr_t)this; }
    GPUTexture(const GPUTexture&) = delete;
    GPUTexture& operator=(const GPUTexture&) = delete;
    void cudaMap() {}
    void cudaUnmap() {}
    cudaArray* getCudaArray() { return &arr; }
    template <class T> T* data() { return (T*)arr.data; }
    int width() const { return arr.width; }
    int height() const { return arr.height; }

  private:
    std::vector<unsigned char> own;
    cudaArray arr;
    int texel_bytes;
    GlTexture gl;

  public:
    GlTexture* texture;
};
