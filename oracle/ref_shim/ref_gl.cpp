// ref_gl.cpp — harness that EXECUTES the reference's own surfel shaders (compiled to C++ by build_ref.py, namespaces sh_*)
// behind entry points with the signatures of oracle/orc_surfel.c (orc_X -> ref_X).  TEST INFRASTRUCTURE ONLY.
//
// What comes from the reference: every line of shader code (data association, fusion, cleaning, projection, splat
// intersection, bootstrap, fill-in, bilateral filter).  What this file restates (the fixed-function OpenGL around the
// shaders, with the same driver-level choices oracle/orc_surfel.c documents in its header): draw order (column-major pixel
// grid, surfel order), the pass-through geometry shaders (data.geom, copy_unstable.geom, vertex_feedback.geom: emit when the
// flag is > 0), ordered transform feedback, point rasterisation (pixel = floor of the projected window coordinate, sprite of
// size s covers centres in [u-s/2, u+s/2)), GL_LESS depth test where the first primitive wins ties, sampler filter modes, and
// RGBA8 conversion.  The update maps are texDim x texDim with texDim = 1024 instead of 3072 (only the id -> texel map changes).
#include <vector>
#include <stdlib.h>

extern "C" {
#include "orc.h"
}

namespace glsl {
vec4 gl_Position, gl_FragCoord;
float gl_PointSize, gl_FragDepth;
int gl_VertexID;
bool gl_Discarded;
mat3 inverse(const mat3&) { abort(); }  // only reached in the deformation-graph branch of copy_unstable.vert (nodes > 0), dead here
}

namespace {
using namespace glsl;

// texcoord of pixel column/row i as the host builds it (Model.cpp:166-170)
float tex_coord(int i, int size) { return (float)((double)((float)i / (float)size) + 1.0 / (2.0 * (double)(float)size)); }
vec4 cam_inv(orc_cam c) { return vec4(c.cx, c.cy, (float)(1.0 / (double)c.fx), (float)(1.0 / (double)c.fy)); }
vec4 cam_f(orc_cam c) { return vec4(c.cx, c.cy, c.fx, c.fy); }
sampler2D tex_f(const float* d, int w, int h, int comps, bool linear) { sampler2D s; s.data = d; s.width = w; s.height = h; s.comps = comps; s.linear = linear; return s; }
sampler2D tex_u8(const uint8_t* d, int w, int h) { sampler2D s; s.bytes = d; s.width = w; s.height = h; s.comps = 4; return s; }
usampler2D tex_u32(const uint32_t* d, int w, int h) { usampler2D s; s.data = d; s.width = w; s.height = h; return s; }
usampler2D tex_mask(const uint8_t* d, int w, int h) { usampler2D s; s.bytes = d; s.width = w; s.height = h; return s; }
void put(float* o, vec4 a, vec4 b, vec4 c) { memcpy(o, a.d, 16); memcpy(o + 4, b.d, 16); memcpy(o + 8, c.d, 16); }
vec4 get(const float* s) { return vec4(s[0], s[1], s[2], s[3]); }
uint8_t unorm8(float v) { v = v < 0 ? 0 : (v > 1 ? 1 : v); return (uint8_t)glsl::round(v * 255.0f); }
constexpr int kTexDim = 1024;
}  // namespace
using namespace glsl;

extern "C" {

// depth_bilateral_metric.frag
void ref_bilateral(const float* depth, int cols, int rows, float maxD, float* out)
{
    namespace S = sh_depth_bilateral_metric_frag;
    S::gSampler = tex_f(depth, cols, rows, 1, false); S::cols = (float)cols; S::rows = (float)rows; S::maxD = maxD;
    S::gSampler.snap = true;  // the shader addresses neighbours at float(cx)/cols, i.e. at texel corners (oracle/orc_surfel.c header)
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            S::texcoord = vec2(((float)x + 0.5f) / (float)cols, ((float)y + 0.5f) / (float)rows);
            S::shader_main();
            out[y * cols + x] = S::FragColor;
        }
}

// vertex_feedback.vert + vertex_feedback.geom (FeedbackBuffer::compute)
int ref_vertex_feedback(const uint8_t* rgba, const float* depth, int cols, int rows, orc_cam cam, int time, float maxDepth, float* out)
{
    namespace S = sh_vertex_feedback_vert;
    S::gSampler = tex_f(depth, cols, rows, 1, false); S::cSampler = tex_u8(rgba, cols, rows);
    S::cam = cam_inv(cam); S::cols = (float)cols; S::rows = (float)rows; S::time = time; S::maxDepth = maxDepth;
    memset(out, 0, sizeof(float) * 12 * (size_t)cols * rows);
    int n = 0;
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) {
            S::texcoord = vec2(tex_coord(i, cols), tex_coord(j, rows));
            S::shader_main();
            if (S::zVal > 0) { put(out + (size_t)n * 12, S::vPosition, S::vColor, S::vNormRad); n++; }
        }
    return n;
}

// Model::initialise + init_unstable.vert: attributes 0/1 from the raw feedback, attribute 2 from the filtered one
int ref_model_initialise(const float* raw_fb, int raw_count, const float* filtered_fb, float* surfels)
{
    namespace S = sh_init_unstable_vert;
    for (int k = 0; k < raw_count; k++) {
        S::vPosition = get(raw_fb + (size_t)k * 12); S::vColor = get(raw_fb + (size_t)k * 12 + 4); S::vNormRad = get(filtered_fb + (size_t)k * 12 + 8);
        S::shader_main();
        put(surfels + (size_t)k * 12, S::vPosition0, S::vColor0, S::vNormRad0);
    }
    return raw_count;
}

// index_map.vert / index_map.frag (ModelProjection::predictIndices)
void ref_predict_indices(const float* surfels, int count, const float pose[16], orc_cam cam, int cols, int rows, float maxDepth,
                         int time, int timeDelta, uint32_t* index, float* vertConf4, float* colorTime4, float* normRad4)
{
    namespace V = sh_index_map_vert; namespace F = sh_index_map_frag;
    float t_inv[16]; orc_inverse_pose(pose, t_inv);
    V::t_inv = mat4::from_row_major(t_inv); V::cam = cam_f(cam); V::cols = (float)cols; V::rows = (float)rows; V::maxDepth = maxDepth;
    V::time = time; V::timeDelta = timeDelta;
    const size_t N = (size_t)cols * rows;
    std::vector<float> zbuf(N, maxDepth);
    memset(index, 0, 4 * N); memset(vertConf4, 0, 16 * N); memset(colorTime4, 0, 16 * N); memset(normRad4, 0, 16 * N);
    for (int id = 0; id < count; id++) {
        const float* s = surfels + (size_t)id * 12;
        V::vPosition = get(s); V::vColorTime = get(s + 4); V::vNormRad = get(s + 8); gl_VertexID = id;
        V::shader_main();
        if (gl_Position.x == -10.0f && gl_Position.y == -10.0f) continue;  // culled by the shader
        const vec4 ph = V::vPosition0;
        const float u = ((cam.fx * ph.x) / ph.z) + cam.cx, v = ((cam.fy * ph.y) / ph.z) + cam.cy;  // window coordinate
        if (!(u >= 0.0f && v >= 0.0f && u < (float)cols && v < (float)rows)) continue;           // clipped
        const size_t q = (size_t)floorf(v) * cols + (size_t)floorf(u);
        if (!(ph.z < zbuf[q])) continue;  // GL_LESS
        zbuf[q] = ph.z;
        F::vPosition0 = V::vPosition0; F::vColorTime0 = V::vColorTime0; F::vNormRad0 = V::vNormRad0; F::vertexId = V::vertexId;
        F::shader_main();
        index[q] = (uint32_t)F::FragColor;
        memcpy(vertConf4 + q * 4, F::vPosition1.d, 16); memcpy(colorTime4 + q * 4, F::vColorTime1.d, 16); memcpy(normRad4 + q * 4, F::vNormRad1.d, 16);
    }
}

// splat.vert + combo_splat.frag (ModelProjection::combinedPredict)
void ref_combined_predict(const float* surfels, int count, const float pose[16], orc_cam cam, int cols, int rows, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, uint8_t* image_rgba, float* vertexConf4,
                          float* normalRad4, uint16_t* time16)
{
    namespace V = sh_splat_vert; namespace F = sh_combo_splat_frag;
    float t_inv[16]; orc_inverse_pose(pose, t_inv);
    V::t_inv = mat4::from_row_major(t_inv); V::cam = cam_f(cam); V::cols = (float)cols; V::rows = (float)rows; V::maxDepth = maxDepth;
    V::confThreshold = confThreshold; V::time = time; V::maxTime = maxTime; V::timeDelta = timeDelta;
    F::cam = cam_f(cam); F::maxDepth = maxDepth;
    const size_t N = (size_t)cols * rows;
    std::vector<float> zbuf(N, maxDepth);
    memset(image_rgba, 0, 4 * N); memset(vertexConf4, 0, 16 * N); memset(normalRad4, 0, 16 * N); memset(time16, 0, 2 * N);
    for (int id = 0; id < count; id++) {
        const float* s = surfels + (size_t)id * 12;
        V::vPosition = get(s); V::vColor = get(s + 4); V::vNormRad = get(s + 8);
        V::shader_main();
        if (gl_Position.x == 1000.0f) continue;  // culled by the shader
        const float size = gl_PointSize;
        if (!(size > 0.0f) || !(size <= 4096.0f)) continue;  // degenerate / NaN sprites draw nothing
        const vec4 ph = V::position;
        const float u = ((cam.fx * ph.x) / ph.z) + cam.cx, v = ((cam.fy * ph.y) / ph.z) + cam.cy;
        if (!(u >= 0.0f && v >= 0.0f && u <= (float)cols && v <= (float)rows)) continue;  // points are clipped by their centre
        const float half = size * 0.5f;
        int x_lo = (int)ceilf(u - half - 0.5f), x_hi = (int)ceilf(u + half - 0.5f) - 1;
        int y_lo = (int)ceilf(v - half - 0.5f), y_hi = (int)ceilf(v + half - 0.5f) - 1;
        x_lo = x_lo < 0 ? 0 : x_lo; y_lo = y_lo < 0 ? 0 : y_lo; x_hi = x_hi > cols - 1 ? cols - 1 : x_hi; y_hi = y_hi > rows - 1 ? rows - 1 : y_hi;
        F::position = V::position; F::normRad = V::normRad; F::colTime = V::colTime;
        for (int py = y_lo; py <= y_hi; py++)
            for (int px = x_lo; px <= x_hi; px++) {
                gl_FragCoord = vec4((float)px + 0.5f, (float)py + 0.5f, 0, 1);
                gl_Discarded = false;
                F::shader_main();
                if (gl_Discarded) continue;
                const size_t q = (size_t)py * cols + px;
                const float z = F::vertexConf.z;  // gl_FragDepth = z/(2 maxDepth)+0.5 is monotonic in z
                if (!(z < zbuf[q])) continue;       // GL_LESS (NaN fails)
                zbuf[q] = z;
                for (int k = 0; k < 4; k++) image_rgba[q * 4 + k] = unorm8(F::image.d[k]);
                memcpy(vertexConf4 + q * 4, F::vertexConf.d, 16); memcpy(normalRad4 + q * 4, F::normalRadius.d, 16);
                time16[q] = (uint16_t)F::time;
            }
    }
}

// fill_vertex.frag, fill_normal.frag, fill_rgb.frag (FillIn::vertex / normal / image)
void ref_fill_in(const float* pred_vertex4, const float* pred_normal4, const uint8_t* pred_image, const float* depth, const uint8_t* rgba,
                 int cols, int rows, orc_cam cam, int passthrough_geom, int passthrough_rgb, float* out_vertex4, float* out_normal4,
                 uint8_t* out_image)
{
    namespace A = sh_fill_vertex_frag; namespace B = sh_fill_normal_frag; namespace C = sh_fill_rgb_frag;
    A::eSampler = tex_f(pred_vertex4, cols, rows, 4, false); A::rSampler = tex_f(depth, cols, rows, 1, false);
    A::cam = cam_inv(cam); A::cols = (float)cols; A::rows = (float)rows; A::passthrough = passthrough_geom;
    B::eSampler = tex_f(pred_normal4, cols, rows, 4, false); B::rSampler = tex_f(depth, cols, rows, 1, false);
    B::cam = cam_inv(cam); B::cols = (float)cols; B::rows = (float)rows; B::passthrough = passthrough_geom;
    C::eSampler = tex_u8(pred_image, cols, rows); C::rSampler = tex_u8(rgba, cols, rows); C::passthrough = passthrough_rgb;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const size_t q = (size_t)y * cols + x;
            const vec2 tc(((float)x + 0.5f) / (float)cols, ((float)y + 0.5f) / (float)rows);
            A::texcoord = tc; A::shader_main(); memcpy(out_vertex4 + q * 4, A::FragColor.d, 16);
            B::texcoord = tc; B::shader_main(); memcpy(out_normal4 + q * 4, B::FragColor.d, 16);
            C::texcoord = tc; C::shader_main();
            for (int k = 0; k < 4; k++) out_image[q * 4 + k] = unorm8(C::FragColor.d[k]);
        }
}

// resize.frag through GPUResize (Shaders/Resize.cpp: a full-target quad into a cols/20 x rows/20 RGB8 renderbuffer), then the count of
// CoFusion::requiresFillIn (CoFusion.cpp:547-565).  The source is ModelProjection's splatColorTexture (draw = false: NEAREST).
int ref_requires_fill_in(const uint8_t* pred_image, int cols, int rows, float ratio)
{
    namespace R = sh_resize_frag;
    const int dw = cols / 20, dh = rows / 20;
    R::eSampler = tex_u8(pred_image, cols, rows);
    int sum = 0;
    for (int j = 0; j < dh; j++)
        for (int i = 0; i < dw; i++) {
            R::texcoord = vec2(((float)i + 0.5f) / (float)dw, ((float)j + 0.5f) / (float)dh);
            R::shader_main();
            const uint8_t r = unorm8(R::FragColor.d[0]), g = unorm8(R::FragColor.d[1]), b = unorm8(R::FragColor.d[2]);
            sum += r > 0 && g > 0 && b > 0;
        }
    return float(sum) / float(dh * dw) < ratio;
}

// data.vert + data.geom + data.frag, then update.vert (Model::fuse)
void ref_fuse(const float* surfels_in, int count, const uint32_t* index, const float* vertConf4, const float* normRad4, const uint8_t* rgba,
              const float* depth_raw, const float* depth_filt, const uint8_t* mask, const float pose[16], orc_cam cam, int cols, int rows,
              int time, float weighting, int maskID, float maxDepth, float* surfels_out, float* new_unstable, int* n_new)
{
    namespace D = sh_data_vert; namespace DF = sh_data_frag; namespace U = sh_update_vert;
    D::cSampler = tex_u8(rgba, cols, rows);
    D::drSampler = tex_f(depth_raw, cols, rows, 1, false); D::drfSampler = tex_f(depth_filt, cols, rows, 1, false);
    D::indexSampler = tex_u32(index, cols, rows); D::maskSampler = tex_mask(mask, cols, rows);
    // the sparse prediction maps are LINEAR (the string-literal -> bool accident of ModelProjection.cpp:28-38)
    D::vertConfSampler = tex_f(vertConf4, cols, rows, 4, true); D::normRadSampler = tex_f(normRad4, cols, rows, 4, true);
    D::colorTimeSampler = tex_f(vertConf4, cols, rows, 4, true);  // declared, never sampled by data.vert
    D::cam = cam_inv(cam); D::cols = (float)cols; D::rows = (float)rows; D::scale = 1.0f; D::texDim = (float)kTexDim;
    D::pose = mat4::from_row_major(pose); D::maxDepth = maxDepth; D::time = (float)time; D::weighting = weighting; D::maskID = (uint)maskID;
    if (count > kTexDim * kTexDim) abort();
    const size_t T = (size_t)kTexDim * kTexDim;
    std::vector<float> upd_v(T * 4, 0.f), upd_c(T * 4, 0.f), upd_n(T * 4, 0.f);
    std::vector<unsigned char> written(T, 0);
    int nn = 0;
    for (int i = 0; i < cols; i++)
        for (int j = 0; j < rows; j++) {  // column-major draw order
            D::texcoord = vec2(tex_coord(i, cols), tex_coord(j, rows));
            D::shader_main();
            if (!(D::updateId > 0)) continue;  // data.geom
            if (D::updateId == 2) { put(new_unstable + (size_t)nn * 12, D::vPosition, D::vColor, D::vNormRad); nn++; }  // transform feedback
            // rasterise the point into the update maps: window = (ndc + 1) / 2 * texDim
            const int px = (int)floorf((gl_Position.x + 1.0f) * 0.5f * (float)kTexDim), py = (int)floorf((gl_Position.y + 1.0f) * 0.5f * (float)kTexDim);
            if (px < 0 || py < 0 || px >= kTexDim || py >= kTexDim) continue;  // (-10, -10): off screen
            const size_t q = (size_t)py * kTexDim + px;
            if (written[q]) continue;  // all fragments have depth 0: GL_LESS keeps the first
            DF::vPosition0 = D::vPosition; DF::vColor0 = D::vColor; DF::vNormRad0 = D::vNormRad; DF::updateId0 = D::updateId;
            DF::shader_main();
            written[q] = 1;
            memcpy(&upd_v[q * 4], DF::vPosition1.d, 16); memcpy(&upd_c[q * 4], DF::vColor1.d, 16); memcpy(&upd_n[q * 4], DF::vNormRad1.d, 16);
        }
    *n_new = nn;
    U::texDim = (float)kTexDim; U::time = time;
    U::vertSamp = tex_f(upd_v.data(), kTexDim, kTexDim, 4, false); U::colorSamp = tex_f(upd_c.data(), kTexDim, kTexDim, 4, false);
    U::normSamp = tex_f(upd_n.data(), kTexDim, kTexDim, 4, false);
    for (int id = 0; id < count; id++) {
        const float* s = surfels_in + (size_t)id * 12;
        U::vPosition = get(s); U::vColor = get(s + 4); U::vNormRad = get(s + 8); gl_VertexID = id;
        U::shader_main();
        put(surfels_out + (size_t)id * 12, U::vPosition0, U::vColor0, U::vNormRad0);
    }
}

// copy_unstable.vert + copy_unstable.geom (Model::clean): old surfels, then the new unstable ones
int ref_clean(const float* surfels_in, int count, const float* new_unstable, int n_new, const uint32_t* index, const float* vertConf4,
              const float* colorTime4, const float* depth_filt, const uint8_t* mask, const float pose[16], orc_cam cam, int cols, int rows,
              int time, float confThreshold, float outlierCoeff, int timeDelta, int maskID, float* surfels_out)
{
    namespace S = sh_copy_unstable_vert;
    float t_inv[16]; orc_inverse_pose(pose, t_inv);
    S::time = time; S::scale = 1.0f; S::outlierCoeff = outlierCoeff; S::t_inv = mat4::from_row_major(t_inv); S::cam = cam_f(cam);
    S::cols = (float)cols; S::rows = (float)rows; S::confThreshold = confThreshold;
    S::indexSampler = tex_u32(index, cols, rows); S::maskSampler = tex_mask(mask, cols, rows);
    S::vertConfSampler = tex_f(vertConf4, cols, rows, 4, true); S::colorTimeSampler = tex_f(colorTime4, cols, rows, 4, true);
    S::normRadSampler = tex_f(vertConf4, cols, rows, 4, true);  // declared, never sampled
    S::depthSamplerInput = tex_f(depth_filt, cols, rows, 1, false); S::depthSamplerPrediction = tex_f(depth_filt, cols, rows, 1, false);
    S::nodes = 0; S::nodeCols = 1; S::maxDepth = 0; S::timeDelta = timeDelta; S::isFern = 0; S::maskID = (uint)maskID;
    int n = 0;
    for (int pass = 0; pass < 2; pass++) {
        const float* src = pass ? new_unstable : surfels_in;
        const int m = pass ? n_new : count;
        for (int k = 0; k < m; k++) {
            S::vPos = get(src + (size_t)k * 12); S::vCol = get(src + (size_t)k * 12 + 4); S::vNormR = get(src + (size_t)k * 12 + 8);
            S::shader_main();
            if (S::test > 0) { put(surfels_out + (size_t)n * 12, S::vPosition, S::vColor, S::vNormRad); n++; }
        }
    }
    return n;
}

}  // extern "C"
