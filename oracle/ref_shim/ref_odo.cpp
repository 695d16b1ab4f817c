// ref_odo.cpp -- flat C entry points over the reference's OWN RGBDOdometry class (Core/Utils/RGBDOdometry.{h,cpp} +
// Core/Utils/OdometryProvider.h, compiled as they lie), TEST INFRASTRUCTURE ONLY.  The class runs on the reference's CUDA kernels
// under the CPU SIMT emulator (include/cusim.h) with the Eigen stand-in of eigen_fixed/ (its header states what that does and does
// not prove) and host-memory stand-ins for the OpenGL textures.  This file is appended to the generated copy of RGBDOdometry.cpp
// by build_ref.py, so it sees the class definition.  Used by tests/test_cpu_refpin.py to pin oracle/orc_track.c's
// orc_odom_get_incremental_transformation and to generate tests/golden/ref_odo_v1.npz.
#include <vector>

namespace {
struct RefOdo {
    RGBDOdometry* od;
    int w, h;
};
}  // namespace

extern "C" {

void* ref_odo_create(int w, int h, float cx, float cy, float fx, float fy)
{
    return new RefOdo{new RGBDOdometry(w, h, cx, cy, fx, fy), w, h};
}
void ref_odo_destroy(void* p)
{
    RefOdo* r = (RefOdo*)p;
    delete r->od; delete r;
}
// Model::initICP / performTracking call sequence (Model.cpp:319-389): the same five initialisers, the same arguments
void ref_odo_init_first_rgb(void* p, const unsigned char* rgba)
{
    RefOdo* r = (RefOdo*)p;
    GPUTexture t((void*)rgba, r->w, r->h);
    r->od->initFirstRGB(&t);
}
void ref_odo_init_icp_model(void* p, const float* v4, const float* n4, float cutoff, const float* pose_row_major)
{
    RefOdo* r = (RefOdo*)p;
    GPUTexture tv((void*)v4, r->w, r->h), tn((void*)n4, r->w, r->h);
    Eigen::Matrix4f pose;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) pose(i, j) = pose_row_major[i * 4 + j];
    r->od->initICPModel(&tv, &tn, cutoff, pose);
}
void ref_odo_init_rgb_model(void* p, const unsigned char* rgba)
{
    RefOdo* r = (RefOdo*)p;
    GPUTexture t((void*)rgba, r->w, r->h);
    r->od->initRGBModel(&t);
}
void ref_odo_init_icp(void* p, const float* const* depth_pyr, float cutoff)
{
    RefOdo* r = (RefOdo*)p;
    std::vector<DeviceArray2D<float> > depth(RGBDOdometry::NUM_PYRS);
    std::vector<DeviceArray2D<unsigned char> > mask(RGBDOdometry::NUM_PYRS);
    for (int i = 0; i < RGBDOdometry::NUM_PYRS; i++) {
        const int rows = r->h >> i, cols = r->w >> i;
        depth[i].create(rows, cols); depth[i].upload(depth_pyr[i], (size_t)cols * sizeof(float), rows, cols);
        mask[i].create(rows, cols);  // zeros: the frame-wide mask pyramid of CoFusion.cpp:205-209 with maskID 0
    }
    r->od->initICP(depth, mask, cutoff);
}
void ref_odo_init_rgb(void* p, const unsigned char* rgba)
{
    RefOdo* r = (RefOdo*)p;
    GPUTexture t((void*)rgba, r->w, r->h);
    r->od->initRGB(&t);
}
// stats: lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count; lastA row-major [36], lastb [6]
void ref_odo_track(void* p, float* trans, float* rot_row_major, int rgb_only, float icp_weight, int pyramid, int fast_odom, int so3,
                   float* icp_err_surface, float* stats6, double* lastA36, double* lastb6)
{
    RefOdo* r = (RefOdo*)p;
    Eigen::Vector3f t(trans[0], trans[1], trans[2]);
    Eigen::Matrix<float, 3, 3, Eigen::RowMajor> R;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R(i, j) = rot_row_major[i * 3 + j];
    cusim::Surface sf{(char*)icp_err_surface, (size_t)r->w * 4};
    r->od->getIncrementalTransformation(t, R, rgb_only != 0, icp_weight, pyramid != 0, fast_odom != 0, so3 != 0,
                                        icp_err_surface ? (cudaSurfaceObject_t)(uintptr_t)&sf : 0, 0);
    for (int i = 0; i < 3; i++) { trans[i] = t(i); for (int j = 0; j < 3; j++) rot_row_major[i * 3 + j] = R(i, j); }
    if (stats6) {
        stats6[0] = r->od->lastICPError; stats6[1] = r->od->lastICPCount; stats6[2] = r->od->lastRGBError; stats6[3] = r->od->lastRGBCount;
        stats6[4] = r->od->lastSO3Error; stats6[5] = r->od->lastSO3Count;
    }
    if (lastA36) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) lastA36[i * 6 + j] = r->od->lastA(i, j);
    if (lastb6) for (int i = 0; i < 6; i++) lastb6[i] = r->od->lastb(i);
}

}  // extern "C"
