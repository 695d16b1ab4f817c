// Stand-in for Core/GPUTexture.h (a Pangolin OpenGL texture with CUDA interop), TEST INFRASTRUCTURE ONLY: what
// Core/Utils/RGBDOdometry.cpp uses of it -- map / unmap and the cudaArray behind the texture -- over host memory.
#pragma once
#include <cuda_runtime_api.h>
class GPUTexture {
  public:
    GPUTexture(void* texels, int width, int height) : arr{texels, width, height} {}
    void cudaMap() {}
    void cudaUnmap() {}
    cudaArray* getCudaArray() { return &arr; }
  private:
    cudaArray arr;
};
