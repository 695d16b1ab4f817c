// ref_api.cpp — flat C entry points over the reference's OWN host functions (Core/Cuda/cudafuncs.cuh), compiled from
// /root/reference/Core/Cuda/{reduce.cu,cudafuncs.cu,containers/device_memory.cpp} under the CPU SIMT emulator in
// include/cusim.h.  TEST INFRASTRUCTURE ONLY: used to pin oracle/*.c (tests/test_cpu_refpin.py) and to generate
// tests/golden/ref_v1.npz (tests/golden/make_ref_golden.py).  Each function uploads plain arrays into the
// reference's DeviceArray containers, calls the reference function and downloads the result.
#include "cudafuncs.cuh"

namespace {
template <class T> void up(DeviceArray2D<T>& d, const void* h, int rows, int cols) { d.create(rows, cols); d.upload(h, (size_t)cols * sizeof(T), rows, cols); }
template <class T> void down(const DeviceArray2D<T>& d, void* h) { d.download(h, (size_t)d.cols() * sizeof(T)); }
mat33 m33(const float* r) { mat33 m; memcpy(m.data, r, 36); return m; }
float3 v3(const float* t) { return make_float3(t[0], t[1], t[2]); }
}  // namespace

extern "C" {

// cudafuncs.cu:136 createVMap
void ref_create_vmap(const float* depth, int cols, int rows, float fx, float fy, float cx, float cy, float cutoff, float* vmap)
{
    DeviceArray2D<float> d, v; DeviceArray2D<unsigned char> mask;
    up(d, depth, rows, cols); mask.create(rows, cols);
    createVMap(CameraModel(fx, fy, cx, cy), d, mask, v, cutoff, 0);
    down(v, vmap);
}
// cudafuncs.cu:191 createNMap
void ref_create_nmap(const float* vmap, int cols, int rows, float* nmap)
{
    DeviceArray2D<float> v, n; up(v, vmap, rows * 3, cols);
    createNMap(v, n);
    down(n, nmap);
}
// cudafuncs.cu:251 tranformMaps, called in place like RGBDOdometry.cpp:171 does (source and destination are the same maps)
void ref_transform_maps(float* vmap, float* nmap, int cols, int rows, const float* R, const float* t)
{
    DeviceArray2D<float> v, n; up(v, vmap, rows * 3, cols); up(n, nmap, rows * 3, cols);
    tranformMaps(v, n, m33(R), v3(t), v, n);
    down(v, vmap); down(n, nmap);
}
// cudafuncs.cu:313 copyMaps (float4-per-pixel inputs)
void ref_copy_maps(const float* v4, const float* n4, int cols, int rows, float* vmap, float* nmap)
{
    DeviceArray<float> vs, ns; vs.create((size_t)rows * cols * 4); ns.create((size_t)rows * cols * 4);
    vs.upload(v4, (size_t)rows * cols * 4); ns.upload(n4, (size_t)rows * cols * 4);
    DeviceArray2D<float> vd(rows * 3, cols), nd(rows * 3, cols);
    copyMaps(vs, ns, vd, nd);
    down(vd, vmap); down(nd, nmap);
}
// cudafuncs.cu:437/442 resizeVMap / resizeNMap
void ref_resize_map(const float* in, int cols, int rows, int normalize, float* out)
{
    DeviceArray2D<float> i, o; up(i, in, rows * 3, cols);
    if (normalize) resizeNMap(i, o); else resizeVMap(i, o);
    down(o, out);
}
// cudafuncs.cu:615 verticesToDepth
void ref_vertices_to_depth(const float* v4, int cols, int rows, float cutoff, float* depth)
{
    DeviceArray<float> vs; vs.create((size_t)rows * cols * 4); vs.upload(v4, (size_t)rows * cols * 4);
    DeviceArray2D<float> d(rows, cols);
    verticesToDepth(vs, d, cutoff);
    down(d, depth);
}
// cudafuncs.cu:510 pyrDownGaussF
void ref_pyrdown_gauss_f32(const float* src, int cols, int rows, float* dst)
{
    DeviceArray2D<float> s, d(rows / 2, cols / 2); up(s, src, rows, cols);
    pyrDownGaussF(s, d);
    down(d, dst);
}
// cudafuncs.cu:566 pyrDownUcharGauss
void ref_pyrdown_gauss_u8(const unsigned char* src, int cols, int rows, unsigned char* dst)
{
    DeviceArray2D<unsigned char> s, d(rows / 2, cols / 2); up(s, src, rows, cols);
    pyrDownUcharGauss(s, d);
    down(d, dst);
}
// cudafuncs.cu:641 imageBGRToIntensity (the texture array holds the frame's 4-byte pixels)
void ref_rgba_to_intensity(const unsigned char* rgba, int cols, int rows, unsigned char* dst)
{
    cudaArray arr{(void*)rgba, cols, rows};
    DeviceArray2D<unsigned char> d(rows, cols);
    imageBGRToIntensity(&arr, d);
    down(d, dst);
}
// cudafuncs.cu:685 computeDerivativeImages
void ref_sobel(const unsigned char* src, int cols, int rows, short* dx, short* dy)
{
    DeviceArray2D<unsigned char> s; up(s, src, rows, cols);
    DeviceArray2D<short> x(rows, cols), y(rows, cols);
    computeDerivativeImages(s, x, y);
    down(x, dx); down(y, dy);
}
// cudafuncs.cu:738 projectToPointCloud (intrinsics of level 0, the function scales them by `level`)
void ref_project_cloud(const float* depth, int cols, int rows, float fx, float fy, float cx, float cy, int level, float* cloud3)
{
    DeviceArray2D<float> d; up(d, depth, rows, cols);
    DeviceArray2D<float3> c(rows, cols);
    CameraModel intr(fx, fy, cx, cy);
    projectToPointCloud(d, c, intr, level);
    down(c, cloud3);
}

// reduce.cu:425 icpStep.  out29 = the 29 reduced floats as laid out in JtJJtrSE3; A/b/residual as the reference unpacks them
void ref_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                  const float* tprev, float fx, float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                  float dist_thres, float angle_thres, int cols, int rows, int threads, int blocks, float* A36, float* b6,
                  float* residual2, float* err_surface)
{
    DeviceArray2D<float> vc, nc, vp, np; up(vc, vmap_curr, rows * 3, cols); up(nc, nmap_curr, rows * 3, cols);
    up(vp, vmap_g_prev, rows * 3, cols); up(np, nmap_g_prev, rows * 3, cols);
    DeviceArray<JtJJtrSE3> sum, out; sum.create(65536); out.create(1);  // RGBDOdometry.cpp sizes these the same way (sumDataSE3 / outDataSE3)
    cusim::Surface sf{(char*)err_surface, (size_t)cols * 4};
    icpStep(m33(Rcurr), v3(tcurr), vc, nc, m33(Rprev_inv), v3(tprev), CameraModel(fx, fy, cx, cy), vp, np, dist_thres, angle_thres,
            sum, out, A36, b6, residual2, threads, blocks, err_surface ? (cudaSurfaceObject_t)(uintptr_t)&sf : 0);
}

// reduce.cu:893 computeRgbResidual.  corres: cols*rows DataTerm records (16 bytes each)
void ref_rgb_residual(float min_scale, const short* dIdx, const short* dIdy, const float* last_depth, const float* next_depth,
                      const unsigned char* last_image, const unsigned char* next_image, void* corres, float max_depth_delta,
                      const float* kt, const float* krkinv, int cols, int rows, int threads, int blocks, int* sigma_sum, int* count,
                      float* err_surface)
{
    DeviceArray2D<short> dx, dy; up(dx, dIdx, rows, cols); up(dy, dIdy, rows, cols);
    DeviceArray2D<float> ld, nd; up(ld, last_depth, rows, cols); up(nd, next_depth, rows, cols);
    DeviceArray2D<unsigned char> li, ni, lm(rows, cols), nm(rows, cols); up(li, last_image, rows, cols); up(ni, next_image, rows, cols);
    // residualKernel / rgbKernel index the DataTerm image LINEARLY (corresImg.data[k], reduce.cu:862 / :524), ignoring the row
    // pitch, so it is moved as one linear block here (identical to a 2-D copy when the pitch equals the row size, as for 640 columns)
    DeviceArray2D<DataTerm> c(rows, cols);
    memset(c.ptr(0), 0, (size_t)cols * rows * sizeof(DataTerm));
    DeviceArray<int2> sum; sum.create(65536);
    cusim::Surface sf{(char*)err_surface, (size_t)cols * 4};
    computeRgbResidual(min_scale, dx, dy, ld, nd, li, ni, lm, nm, c, sum, max_depth_delta, v3(kt), m33(krkinv), *sigma_sum, *count,
                       threads, blocks, err_surface ? (cudaSurfaceObject_t)(uintptr_t)&sf : 0, 0);
    memcpy(corres, c.ptr(0), (size_t)cols * rows * sizeof(DataTerm));
}

// reduce.cu:635 rgbStep
void ref_rgb_step(const void* corres, float sigma, const float* cloud3, float fx, float fy, const short* dIdx, const short* dIdy,
                  float sobel_scale, int cols, int rows, int threads, int blocks, float* A36, float* b6)
{
    DeviceArray2D<DataTerm> c(rows, cols);
    memcpy(c.ptr(0), corres, (size_t)cols * rows * sizeof(DataTerm));  // linear, see ref_rgb_residual
    DeviceArray2D<float3> cl; up(cl, cloud3, rows, cols);
    DeviceArray2D<short> dx, dy; up(dx, dIdx, rows, cols); up(dy, dIdy, rows, cols);
    DeviceArray<JtJJtrSE3> sum, out; sum.create(65536); out.create(1);
    rgbStep(c, sigma, cl, fx, fy, dx, dy, sobel_scale, sum, out, A36, b6, threads, blocks);
}

// reduce.cu:1118 so3Step
void ref_so3_step(const unsigned char* last_image, const unsigned char* next_image, const float* image_basis, const float* kinv,
                  const float* krlr, int cols, int rows, int threads, int blocks, float* A9, float* b3, float* residual2)
{
    DeviceArray2D<unsigned char> li, ni; up(li, last_image, rows, cols); up(ni, next_image, rows, cols);
    DeviceArray<JtJJtrSO3> sum, out; sum.create(65536); out.create(1);
    so3Step(li, ni, m33(image_basis), m33(kinv), m33(krlr), sum, out, A9, b3, residual2, threads, blocks);
}

int ref_sizeof_dataterm() { return (int)sizeof(DataTerm); }
}
