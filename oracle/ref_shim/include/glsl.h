// glsl.h — just enough of GLSL 3.30 in C++ for g++ to compile and EXECUTE the reference's own surfel shaders
// (/root/reference/Core/Shaders/*.vert, *.frag, *.glsl) where they lie.  TEST INFRASTRUCTURE ONLY (oracle/): it pins the
// C restatement in oracle/orc_surfel.c against the shader sources.  Nothing here is reference code.
//
// Semantics (the parts GLSL leaves to the implementation are fixed the same way oracle/orc_surfel.c fixes them, see its header):
//   * IEEE-754 single precision, no FMA contraction (-ffp-contract=off); build_ref.py suffixes unsuffixed literals with `f`
//     because a GLSL `1.0` is a 32-bit float;
//   * mat*vec row-wise, left to right; normalize(v) = v * (1/sqrt(dot(v,v))); length = sqrt(dot);
//   * exp / acos: the fixed polynomial forms of oracle/orc_math.h (GLSL does not specify their accuracy); round(): half away
//     from zero; pow(x, y) only appears with y == 2 in the compiled shaders and is evaluated as powf;
//   * textureLod / texture: level 0 only; NEAREST = texel floor(u*size) clamped to the edge, LINEAR = f32 bilinear weights from
//     u*size - 0.5 with edge clamp (which sampler is which is set by the harness from the reference's texture constructors).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

extern "C" {
#include "orc_math.h"
}

namespace glsl {

typedef unsigned int uint;
struct vec2; struct vec3; struct vec4;

// ---- swizzle proxies: share storage with the owning vector through an anonymous union ---------------------------------
template <class V, int N, int A, int B, int C = -1, int D = -1> struct swz {
    float d[4];  // only d[0..size of the owner) exist; the proxy is never constructed on its own
    operator V() const;
    swz& operator=(const V& v);
    swz& operator=(const swz& o) { return *this = (V)o; }
};

struct vec2 {
    union { struct { float x, y; }; float d[2]; swz<vec2, 2, 0, 1> xy; };
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
    explicit vec2(float a) : x(a), y(a) {}
    explicit vec2(const vec4& o);
    vec2(const vec2& o) : x(o.x), y(o.y) {}
    vec2& operator=(const vec2& o) { x = o.x; y = o.y; return *this; }
};
struct vec3 {
    union { struct { float x, y, z; }; float d[3]; swz<vec2, 2, 0, 1> xy; swz<vec3, 3, 0, 1, 2> xyz; };
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
    vec3(vec2 a, float c) : x(a.x), y(a.y), z(c) {}
    vec3(const vec3& o) : x(o.x), y(o.y), z(o.z) {}
    explicit vec3(const vec4& o);
    vec3& operator=(const vec3& o) { x = o.x; y = o.y; z = o.z; return *this; }
};
struct vec4 {
    union {
        struct { float x, y, z, w; };
        float d[4];
        swz<vec2, 2, 0, 1> xy; swz<vec2, 2, 2, 3> zw; swz<vec3, 3, 0, 1, 2> xyz;
    };
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float e) : x(a), y(b), z(c), w(e) {}
    explicit vec4(float a) : x(a), y(a), z(a), w(a) {}
    vec4(vec3 a, float e) : x(a.x), y(a.y), z(a.z), w(e) {}
    vec4(vec2 a, float c, float e) : x(a.x), y(a.y), z(c), w(e) {}
    vec4(vec2 a, vec2 b) : x(a.x), y(a.y), z(b.x), w(b.y) {}
    vec4(const vec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
    vec4& operator=(const vec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
    explicit operator float() const { return x; }  // float(textureLod(..)) takes the first component
};
inline vec3::vec3(const vec4& o) : x(o.x), y(o.y), z(o.z) {}
inline vec2::vec2(const vec4& o) : x(o.x), y(o.y) {}
struct uvec4 {
    uint x, y, z, w;
    explicit operator uint() const { return x; }
    explicit operator int() const { return (int)x; }
    explicit operator float() const { return (float)x; }
};

template <> inline swz<vec2, 2, 0, 1>::operator vec2() const { return vec2(d[0], d[1]); }
template <> inline swz<vec2, 2, 0, 1>& swz<vec2, 2, 0, 1>::operator=(const vec2& v) { d[0] = v.x; d[1] = v.y; return *this; }
template <> inline swz<vec2, 2, 2, 3>::operator vec2() const { return vec2(d[2], d[3]); }
template <> inline swz<vec2, 2, 2, 3>& swz<vec2, 2, 2, 3>::operator=(const vec2& v) { d[2] = v.x; d[3] = v.y; return *this; }
template <> inline swz<vec3, 3, 0, 1, 2>::operator vec3() const { return vec3(d[0], d[1], d[2]); }
template <> inline swz<vec3, 3, 0, 1, 2>& swz<vec3, 3, 0, 1, 2>::operator=(const vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; return *this; }

// ---- arithmetic (by value, so swizzle proxies convert implicitly) -------------------------------------------------------
#define GLSL_VEC_OPS(V, N)                                                                                                  \
    inline V operator+(V a, V b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; }                   \
    inline V operator-(V a, V b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; }                   \
    inline V operator*(V a, V b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.d[i]; return r; }                   \
    inline V operator/(V a, V b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] / b.d[i]; return r; }                   \
    inline V operator*(V a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * s; return r; }                    \
    inline V operator*(float s, V a) { V r; for (int i = 0; i < N; i++) r.d[i] = s * a.d[i]; return r; }                    \
    inline V operator/(V a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] / s; return r; }                    \
    inline V operator+(V a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + s; return r; }                    \
    inline V operator-(V a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] - s; return r; }                    \
    inline V operator-(V a) { V r; for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }                                \
    inline V& operator+=(V& a, V b) { a = a + b; return a; }                                                                \
    inline V& operator-=(V& a, V b) { a = a - b; return a; }                                                                \
    inline V& operator*=(V& a, float s) { a = a * s; return a; }                                                            \
    inline V& operator/=(V& a, float s) { a = a / s; return a; }                                                            \
    inline float dot(V a, V b) { float r = a.d[0] * b.d[0]; for (int i = 1; i < N; i++) r = r + a.d[i] * b.d[i]; return r; } \
    inline float length(V a) { return sqrtf(dot(a, a)); }                                                                   \
    inline float distance(V a, V b) { return length(a - b); }                                                               \
    inline V normalize(V a) { const float rn = 1.0f / sqrtf(dot(a, a)); return a * rn; }                                    \
    inline V abs(V a) { V r; for (int i = 0; i < N; i++) r.d[i] = fabsf(a.d[i]); return r; }
GLSL_VEC_OPS(vec2, 2)
GLSL_VEC_OPS(vec3, 3)
GLSL_VEC_OPS(vec4, 4)
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

inline float abs(float v) { return fabsf(v); }
inline int abs(int v) { return v < 0 ? -v : v; }
inline float sqrt(float v) { return sqrtf(v); }
inline float min(float a, float b) { return b < a ? b : a; }
inline float max(float a, float b) { return a < b ? b : a; }
inline int min(int a, int b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }
inline float min(int a, float b) { return min((float)a, b); }
inline float min(float a, int b) { return min(a, (float)b); }
inline float max(int a, float b) { return max((float)a, b); }
inline float max(float a, int b) { return max(a, (float)b); }
inline float exp(float v) { return orc_expf(v); }
inline float acos(float v) { return orc_acosf(v); }
inline float round(float v) { return (v < 0) ? -floorf(-v + 0.5f) : floorf(v + 0.5f); }
inline float floor(float v) { return floorf(v); }
inline float pow(float a, float b) { return powf(a, b); }
inline float clamp(float v, float lo, float hi) { return min(max(v, lo), hi); }

// ---- matrices: column-major like GLSL, m[c][r] -------------------------------------------------------------------------
struct mat4;
struct mat3 {
    vec3 c[3];
    mat3() {}
    mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; }
    explicit mat3(const mat4& m);
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
    vec4 c[4];
    mat4() {}
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
    static mat4 from_row_major(const float* m) { mat4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.c[j].d[i] = m[i * 4 + j]; return r; }
};
inline mat3::mat3(const mat4& m) { for (int j = 0; j < 3; j++) c[j] = vec3(m.c[j].x, m.c[j].y, m.c[j].z); }
inline vec4 operator*(const mat4& m, vec4 v)
{
    vec4 r;
    for (int i = 0; i < 4; i++) r.d[i] = ((m.c[0].d[i] * v.x + m.c[1].d[i] * v.y) + m.c[2].d[i] * v.z) + m.c[3].d[i] * v.w;
    return r;
}
inline vec3 operator*(const mat3& m, vec3 v)
{
    vec3 r;
    for (int i = 0; i < 3; i++) r.d[i] = (m.c[0].d[i] * v.x + m.c[1].d[i] * v.y) + m.c[2].d[i] * v.z;
    return r;
}
inline mat3 transpose(const mat3& m) { mat3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.c[i].d[j] = m.c[j].d[i]; return r; }
mat3 inverse(const mat3& m);  // harness (only used with rigid transforms; computed in f64 and rounded)

// ---- samplers ----------------------------------------------------------------------------------------------------------
struct sampler2D {
    const float* data = nullptr;  // `comps` floats per texel, or bytes when `u8`
    const uint8_t* bytes = nullptr;
    int width = 0, height = 0, comps = 4;
    bool linear = false;
    bool snap = false;  // NEAREST with the 8-bit sub-texel fixed point of texture units: a coordinate that is a texel corner up to f32 rounding (float(cx)/cols) addresses texel cx
    vec4 texel(int x, int y) const
    {
        vec4 r(0, 0, 0, 1);
        if (bytes) { for (int k = 0; k < comps; k++) r.d[k] = (float)bytes[((size_t)y * width + x) * comps + k] / 255.0f; }
        else { for (int k = 0; k < comps; k++) r.d[k] = data[((size_t)y * width + x) * comps + k]; }
        return r;
    }
};
struct usampler2D {
    const uint32_t* data = nullptr; const uint8_t* bytes = nullptr;
    int width = 0, height = 0;
};
inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int nearest_texel(float u, int size) { return iclamp((int)floorf(u * (float)size), 0, size - 1); }
inline int nearest_texel_snap(float u, int size) { return iclamp((int)floorf(rintf(u * (float)size * 256.0f) / 256.0f), 0, size - 1); }
inline vec4 textureLod(const sampler2D& s, vec2 uv, float)
{
    if (s.snap) return s.texel(nearest_texel_snap(uv.x, s.width), nearest_texel_snap(uv.y, s.height));
    if (!s.linear) return s.texel(nearest_texel(uv.x, s.width), nearest_texel(uv.y, s.height));
    const float fu = uv.x * (float)s.width - 0.5f, fv = uv.y * (float)s.height - 0.5f;
    const float x0f = floorf(fu), y0f = floorf(fv);
    const float wx = fu - x0f, wy = fv - y0f;
    const int x0 = iclamp((int)x0f, 0, s.width - 1), x1 = iclamp((int)x0f + 1, 0, s.width - 1);
    const int y0 = iclamp((int)y0f, 0, s.height - 1), y1 = iclamp((int)y0f + 1, 0, s.height - 1);
    const vec4 a = s.texel(x0, y0), b = s.texel(x1, y0), c = s.texel(x0, y1), d = s.texel(x1, y1);
    vec4 r;
    for (int k = 0; k < 4; k++) {
        const float top = a.d[k] * (1.0f - wx) + b.d[k] * wx;
        const float bot = c.d[k] * (1.0f - wx) + d.d[k] * wx;
        r.d[k] = top * (1.0f - wy) + bot * wy;
    }
    return r;
}
inline uvec4 textureLod(const usampler2D& s, vec2 uv, float)
{
    const size_t i = (size_t)nearest_texel(uv.y, s.height) * s.width + nearest_texel(uv.x, s.width);
    return uvec4{s.data ? s.data[i] : (uint)s.bytes[i], 0, 0, 1};
}
inline vec4 texture(const sampler2D& s, vec2 uv) { return textureLod(s, uv, 0.0f); }
inline vec4 texture2D(const sampler2D& s, vec2 uv) { return textureLod(s, uv, 0.0f); }
inline uvec4 texture(const usampler2D& s, vec2 uv) { return textureLod(s, uv, 0.0f); }

// ---- per-invocation built-ins (set / read by the harness) --------------------------------------------------------------
extern vec4 gl_Position, gl_FragCoord;
extern float gl_PointSize, gl_FragDepth;
extern int gl_VertexID;
extern bool gl_Discarded;

}  // namespace glsl
