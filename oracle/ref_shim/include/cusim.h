// cusim.h — a minimal CPU SIMT emulator so that g++ can compile and EXECUTE the reference's own CUDA sources
// (/root/reference/Core/Cuda/{reduce,cudafuncs}.cu, containers/device_memory.cpp) where they lie.
//
// TEST INFRASTRUCTURE ONLY (oracle/): it pins the C restatement in oracle/*.c against the reference's real kernels.
// Nothing here is reference code; it restates the small part of the CUDA programming model those files use:
//   * vector types / make_* / min / max / rsqrtf / __int_as_float / __float2int_rn
//   * cudaMalloc / cudaMallocPitch / cudaMemcpy / cudaMemcpy2D / cudaFree / cudaMemcpyToSymbol on host memory
//   * surface objects (surf2Dwrite) and a 2-D uchar4 texture (tex2D) over plain arrays
//   * kernel launches: the build script (build_ref.py) rewrites `k<<<g,b>>>(args);` into cusim::launch(...)
//     which runs the blocks one after another; threads of a block are ucontext fibers when the translation unit
//     uses __syncthreads() (round-robin, a barrier releases when every live thread of the block has arrived),
//     plain loops otherwise.
// Floating point: compiled with -ffp-contract=off -fno-fast-math, i.e. IEEE single precision without FMA
// contraction (nvcc would contract a*b+c; the oracle and the HIP kernels do not).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <stdint.h>
#include <limits>
#include <cmath>

// device translation units (.cu) see __CUDACC__ as under nvcc; a HOST translation unit of the reference (RGBDOdometry.cpp, compiled
// with -DCUSIM_HOST_TU) must not: Core/Cuda/types.cuh gives mat33 its Eigen constructor only to the host compiler
#if !defined(__CUDACC__) && !defined(CUSIM_HOST_TU)
#define __CUDACC__ 1
#endif
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__ static

// ---- vector types --------------------------------------------------------------------------------------------------
struct float1 { float x; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct short2 { short x, y; };
struct ushort2 { unsigned short x, y; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef unsigned short ushort;
typedef unsigned char uchar;

static inline float1 make_float1(float x) { return float1{x}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline short2 make_short2(short x, short y) { return short2{x, y}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }

// ---- device intrinsics ---------------------------------------------------------------------------------------------
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline int __float2int_rn(float f) { return (int)nearbyintf(f); }
static inline int __float2int_rd(float f) { return (int)floorf(f); }
static inline float __fdividef(float a, float b) { return a / b; }
using std::isnan;
using std::isfinite;
using std::isinf;

// ---- runtime API on host memory ------------------------------------------------------------------------------------
enum cudaError { cudaSuccess = 0, cudaErrorMemoryAllocation = 2 };
typedef cudaError cudaError_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
static inline const char* cudaGetErrorString(cudaError e) { return e == cudaSuccess ? "no error" : "cusim error"; }
static inline cudaError cudaGetLastError() { return cudaSuccess; }
static inline cudaError cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError cudaMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)1 << 34; return cudaSuccess; }
template <class T> static inline cudaError cudaMalloc(T** p, size_t n) { *p = (T*)calloc(n ? n : 1, 1); return *p ? cudaSuccess : cudaErrorMemoryAllocation; }
static inline cudaError cudaMallocPitch(void** p, size_t* pitch, size_t wbytes, size_t h)
{
    *pitch = (wbytes + 511) & ~(size_t)511;  // pitched like a real device allocation: rows are NOT contiguous
    *p = calloc(*pitch * (h ? h : 1), 1);
    return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
static inline cudaError cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
static inline cudaError cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, cudaMemcpyKind)
{
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, wbytes);
    return cudaSuccess;
}
struct cudaArray;
static inline cudaError cudaMemcpyFromArray(void* d, const cudaArray* a, size_t, size_t, size_t n, cudaMemcpyKind);  // defined below the type
struct cudaDeviceProp { char name[256]; };
static inline cudaError cudaGetDeviceProperties(cudaDeviceProp* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "cusim (CPU SIMT emulator)"); return cudaSuccess; }
template <class T> static inline cudaError cudaMemcpyToSymbol(T& sym, const void* s, size_t n) { memcpy((void*)&sym, s, n); return cudaSuccess; }

// ---- surfaces / textures ---------------------------------------------------------------------------------------------
typedef unsigned long long cudaSurfaceObject_t;  // 0 = none, else a cusim::Surface*
struct cudaArray { void* data; int width, height; };
static inline cudaError cudaMemcpyFromArray(void* d, const cudaArray* a, size_t, size_t, size_t n, cudaMemcpyKind) { memcpy(d, a->data, n); return cudaSuccess; }
enum cudaTextureReadMode { cudaReadModeElementType = 0 };
template <class T, int D, cudaTextureReadMode M> struct texture { const cudaArray* arr = nullptr; };
namespace cusim {
struct Surface { char* data; size_t pitch; };
}
template <class T> static inline void surf2Dwrite(T v, cudaSurfaceObject_t s, int xbytes, int y)
{
    cusim::Surface* sf = (cusim::Surface*)(uintptr_t)s;
    memcpy(sf->data + (size_t)y * sf->pitch + xbytes, &v, sizeof(T));
}
template <class T, cudaTextureReadMode M> static inline cudaError cudaBindTextureToArray(texture<T, 2, M>& t, const cudaArray* a) { t.arr = a; return cudaSuccess; }
template <class T, cudaTextureReadMode M> static inline cudaError cudaUnbindTexture(texture<T, 2, M>& t) { t.arr = nullptr; return cudaSuccess; }
template <class T, cudaTextureReadMode M> static inline T tex2D(const texture<T, 2, M>& t, float x, float y)
{
    // unnormalised coordinates, point sampling, clamp addressing (the defaults of a texture reference)
    int xi = (int)floorf(x), yi = (int)floorf(y);
    xi = xi < 0 ? 0 : (xi >= t.arr->width ? t.arr->width - 1 : xi);
    yi = yi < 0 ? 0 : (yi >= t.arr->height ? t.arr->height - 1 : yi);
    return ((const T*)t.arr->data)[(size_t)yi * t.arr->width + xi];
}

// ---- execution model -------------------------------------------------------------------------------------------------
extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 32;
void __syncthreads();

namespace cusim {
void run_grid(dim3 grid, dim3 block, bool fibers, void (*entry)(void*), void* arg);
template <class F> static void thunk(void* p) { (*(F*)p)(); }
template <class F> static inline void launch(dim3 grid, dim3 block, bool fibers, F f) { run_grid(grid, block, fibers, &thunk<F>, &f); }
}
