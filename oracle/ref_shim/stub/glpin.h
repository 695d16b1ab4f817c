// glpin.h -- a RECORDING OpenGL for the text of Model::fuse and Model::clean (Core/Model/Model.cpp:408-697), TEST INFRASTRUCTURE ONLY.
//
// Those two methods are OpenGL plumbing: they bind a shader program, set its uniforms from the model's members and the call's
// arguments, bind textures to units and buffers to attributes / transform feedback, and draw.  WHICH value goes to WHICH uniform,
// WHICH texture to WHICH sampler, which buffer is read and which written -- the argument handling rounds 1-5 could only restate,
// because the class owns GL objects -- is exactly what that text decides.  build_ref.py pastes the text (cut out of Model.cpp at
// build time, never stored) into the frame-loop translation unit behind this header: every gl* call and Shader method below just
// RECORDS, and a draw call hands the recorded state to a handler (ref_cofusion.cpp) that runs the CPU oracle's pass with it.  The
// frame-loop fixtures (tests/golden/ref_cofusion_v1.json, ref_traj_v1.npz), which the HIP facade is held to bit for bit, are
// reproduced through this path: the facade's argument handling is thereby pinned to the reference's text (missing #6).
#pragma once
#include <Eigen/Core>
#include <stdint.h>

#include <functional>
#include <map>
#include <memory>
#include <string>

#include "Shaders/Uniform.h"   // the reference's own (header-only)

typedef unsigned int GLuint;
typedef int GLint;
typedef int GLsizei;
typedef unsigned int GLenum;
typedef unsigned char GLboolean;
typedef void GLvoid;
typedef unsigned int GLbitfield;
// (only the identity of the enumerants matters; GL_FLOAT / GL_LUMINANCE come from the GPUTexture stand-in)
enum {
    GL_VIEWPORT_BIT = 0x800, GL_COLOR_BUFFER_BIT = 0x4000, GL_DEPTH_BUFFER_BIT = 0x100, GL_ARRAY_BUFFER = 0x8892, GL_FALSE = 0,
    GL_TRANSFORM_FEEDBACK = 0x8E22, GL_TRANSFORM_FEEDBACK_BUFFER = 0x8C8E, GL_TEXTURE_2D = 0x0DE1, GL_POINTS = 0, GL_RASTERIZER_DISCARD = 0x8C89,
    GL_TRANSFORM_FEEDBACK_PRIMITIVES_WRITTEN = 0x8C88, GL_QUERY_RESULT = 0x8866,
    GL_TEXTURE0 = 0x84C0, GL_TEXTURE1, GL_TEXTURE2, GL_TEXTURE3, GL_TEXTURE4, GL_TEXTURE5, GL_TEXTURE6, GL_TEXTURE7
};

namespace glpin {
struct State {
    std::string program;                          // name of the bound Shader ("" = none)
    std::map<std::string, Uniform> uniforms;      // of the bound program (cleared when a program is bound)
    unsigned active_unit = 0;
    GLuint tex[16] = {0};                         // texture id per unit
    GLuint array_buffer = 0, tf_object = 0, tf_buffer = 0;
    bool in_feedback = false, in_query = false;
    unsigned query_result = 0;                    // what the handler says the queried draws wrote
    // kind 0: glDrawArrays(first, count) -> arg = count; kind 1: glDrawTransformFeedback(id) -> arg = id
    std::function<void(State&, int kind, unsigned arg)> on_draw;
    unsigned next_id = 1;
};
inline State& state() { static State s; return s; }
inline GLuint new_id() { return state().next_id++; }
}  // namespace glpin

class Shader {
  public:
    explicit Shader(const std::string& n) : name(n) {}
    void Bind() { glpin::state().program = name; glpin::state().uniforms.clear(); }
    void Unbind() { glpin::state().program.clear(); }
    void setUniform(const Uniform& u) { glpin::state().uniforms.erase(u.id); glpin::state().uniforms.emplace(u.id, u); }
    std::string name;
};

inline void glPushAttrib(GLbitfield) {}
inline void glPopAttrib() {}
inline void glViewport(GLint, GLint, GLsizei, GLsizei) {}
inline void glClearColor(float, float, float, float) {}
inline void glClear(GLbitfield) {}
inline void glEnableVertexAttribArray(GLuint) {}
inline void glDisableVertexAttribArray(GLuint) {}
inline void glVertexAttribPointer(GLuint, GLint, GLenum, GLboolean, GLsizei, const GLvoid*) {}
inline void glEnable(GLenum) {}
inline void glDisable(GLenum) {}
inline void glFinish() {}
inline void glBindBuffer(GLenum target, GLuint id) { if (target == GL_ARRAY_BUFFER) glpin::state().array_buffer = id; }
inline void glBindTransformFeedback(GLenum, GLuint id) { glpin::state().tf_object = id; }
inline void glBindBufferBase(GLenum target, GLuint, GLuint id) { if (target == GL_TRANSFORM_FEEDBACK_BUFFER) glpin::state().tf_buffer = id; }
inline void glActiveTexture(GLenum unit) { glpin::state().active_unit = unit - GL_TEXTURE0; }
inline void glBindTexture(GLenum, GLuint id) { glpin::state().tex[glpin::state().active_unit & 15u] = id; }
inline void glTexSubImage2D(GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const GLvoid*) {}
inline void glBeginTransformFeedback(GLenum) { glpin::state().in_feedback = true; }
inline void glEndTransformFeedback() { glpin::state().in_feedback = false; }
inline void glBeginQuery(GLenum, GLuint) { glpin::state().in_query = true; glpin::state().query_result = 0; }
inline void glEndQuery(GLenum) { glpin::state().in_query = false; }
inline void glGetQueryObjectuiv(GLuint, GLenum, GLuint* out) { *out = glpin::state().query_result; }
inline void glDrawArrays(GLenum, GLint, GLsizei count) { if (glpin::state().on_draw) glpin::state().on_draw(glpin::state(), 0, (unsigned)count); }
inline void glDrawTransformFeedback(GLenum, GLuint id) { if (glpin::state().on_draw) glpin::state().on_draw(glpin::state(), 1, id); }
