#!/usr/bin/env python3
"""Build oracle/_ref/libcofusion_ref.so: the reference's own CUDA sources, compiled where they lie under
/root/reference/Core/Cuda by g++ against the CPU SIMT emulator in oracle/ref_shim/include (TEST INFRASTRUCTURE ONLY).

g++ cannot parse the `kernel<<<grid, block>>>(args);` launch syntax, so each .cu file is passed through a purely
syntactic rewrite (launch sites -> cusim::launch(...), `static __shared__` -> `__shared__`) into a temporary file that is
deleted after compilation; nothing from /root/reference is copied into the repository (oracle/_ref/ is git-ignored and holds
only the resulting shared library).  Does nothing (exit 0) when /root/reference is absent, e.g. on the GPU box."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("COFUSION_REFERENCE", "/root/reference")
CUDA = os.path.join(REF, "Core", "Cuda")
OUT = os.path.join(os.path.dirname(HERE), "_ref")
LAUNCH = re.compile(r"^(\s*)([A-Za-z_][\w<>, ]*?)\s*<<\s*<\s*(.+?)\s*,\s*(.+?)\s*>>>\s*\((.*)\)\s*;\s*$")


def rewrite(text: str, fibers: bool) -> str:
    out = []
    for line in text.split("\n"):
        m = LAUNCH.match(line)
        if m and not line.lstrip().startswith("//"):
            ind, k, g, b, args = m.groups()
            line = f"{ind}cusim::launch(dim3({g}), dim3({b}), {'true' if fibers else 'false'}, [&]() {{ {k}({args}); }});"
        out.append(line.replace("static __shared__", "__shared__"))
    return "\n".join(out)


SHADERS = os.path.join(REF, "Core", "Shaders")
# the surfel passes of the hot path; geometry shaders are pass-through filters restated in ref_gl.cpp
SHADER_FILES = ["data.vert", "update.vert", "copy_unstable.vert", "index_map.vert", "index_map.frag", "splat.vert",
                "combo_splat.frag", "vertex_feedback.vert", "init_unstable.vert", "fill_vertex.frag", "fill_normal.frag",
                "fill_rgb.frag", "depth_bilateral_metric.frag", "data.frag", "resize.frag"]
FLOAT_LIT = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])")


def glsl_to_cpp(name: str) -> str:
    """Purely syntactic GLSL -> C++ rewrite of one shader: inline #include, drop #version / layout(...), turn the in / out / uniform
    interface into namespace-scope variables, suffix float literals with f (a GLSL 1.0 is a 32-bit float), rename main."""
    def load(fn):
        out = []
        for line in open(os.path.join(SHADERS, fn)).read().split("\n"):
            m = re.match(r'\s*#include\s+"([^"]+)"', line)
            if m:
                out.append(load(m.group(1)))
            elif not line.lstrip().startswith("#version"):
                out.append(line)
        return "\n".join(out)
    src = load(name)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    lines = []
    for line in src.split("\n"):
        code, _, comment = line.partition("//")
        code = re.sub(r"layout\s*\([^)]*\)\s*", "", code)
        code = re.sub(r"^(\s*)(?:flat\s+)?(?:in|out|uniform)\s+", r"\1", code)
        code = FLOAT_LIT.sub(lambda m: m.group(1) + "f", code)
        code = re.sub(r"\bvoid\s+main\s*\(\s*\)", "void shader_main()", code)
        code = re.sub(r"\bdiscard\s*;", "{ gl_Discarded = true; return; }", code)
        lines.append(code)
    ns = "sh_" + name.replace(".", "_")
    return f"namespace glsl {{ namespace {ns} {{\n" + "\n".join(lines) + f"\n}} }}  // namespace glsl::{ns}\n"


def main() -> int:
    if not os.path.isdir(CUDA):
        print(f"build_ref: {CUDA} not present, keeping any prebuilt oracle/_ref", file=sys.stderr)
        return 0
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(HERE)])  # liborc.so: the harness takes the host-side pose inverse from it
    san = ["-fsanitize=address", "-g"] if os.environ.get("COFUSION_REF_SANITIZE") else []   # debugging aid: LD_PRELOAD libasan.so
    flags = san + ["-O2", "-g0" if not san else "-g", "-std=c++14", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w",
             "-I", os.path.join(HERE, "include"), "-I", CUDA]
    objs = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, fibers in (("reduce.cu", True), ("cudafuncs.cu", False)):
            src = rewrite(open(os.path.join(CUDA, name)).read(), fibers)
            gen = os.path.join(tmp, name.replace(".cu", "_launch.cpp"))
            with open(gen, "w") as f:
                f.write(f'#line 1 "{os.path.join(CUDA, name)}"\n' + src)
            obj = os.path.join(tmp, name + ".o")
            subprocess.check_call(["g++", *flags, "-c", gen, "-o", obj])
            objs.append(obj)
        for path in (os.path.join(CUDA, "containers", "device_memory.cpp"), os.path.join(HERE, "cusim.cpp"),
                     os.path.join(HERE, "ref_api.cpp")):
            obj = os.path.join(tmp, os.path.basename(path) + ".o")
            subprocess.check_call(["g++", *flags, "-include", "cusim.h", "-c", path, "-o", obj])
            objs.append(obj)
        # surfel shaders + their harness in one translation unit
        gen = os.path.join(tmp, "gl_all.cpp")
        with open(gen, "w") as f:
            f.write('#include "glsl.h"\n' + "".join(glsl_to_cpp(n) for n in SHADER_FILES) + f'#include "{os.path.join(HERE, "ref_gl.cpp")}"\n')
        obj = os.path.join(tmp, "gl_all.o")
        subprocess.check_call(["g++", *flags, "-I", os.path.dirname(HERE), "-c", gen, "-o", obj])
        objs.append(obj)
        # the segmentation stage: Slic.cpp as it lies; Segmentation.cpp's text goes into a generated file under <tmp>/Segmentation/
        # so that its `#include "../Model/Model.h"` (OpenGL-backed class) resolves to the three-method stand-in copied to
        # <tmp>/Model/Model.h; gSLICr / densecrf / Eigen / OpenCV stand-ins come from include/.  The harness is appended.
        seg_dir = os.path.join(REF, "Core", "Segmentation")
        seg_flags = ["-I", os.path.join(HERE, "eigen_fixed")] + [f for f in flags if f != "-std=c++14"] + ["-std=c++17", "-I", seg_dir, "-I", os.path.join(REF, "Core"), "-I", os.path.join(os.path.dirname(HERE))]
        obj = os.path.join(tmp, "Slic.o")
        subprocess.check_call(["g++", *seg_flags, "-c", os.path.join(seg_dir, "Slic.cpp"), "-o", obj])
        objs.append(obj)
        os.makedirs(os.path.join(tmp, "Segmentation")); os.makedirs(os.path.join(tmp, "Model"))
        for rel_src, rel_dst in ((os.path.join("Model", "Model.h"), "Model.h"), ("GPUTexture.h", "GPUTexture.h"), ("glpin.h", "glpin.h")):
            with open(os.path.join(tmp, "Model", rel_dst), "w") as f:   # (the stand-in's own includes resolve beside it, not to Core/GPUTexture.h)
                f.write(open(os.path.join(HERE, "stub", rel_src)).read())
        gen = os.path.join(tmp, "Segmentation", "Segmentation_gen.cpp")
        with open(gen, "w") as f:
            f.write(f'#line 1 "{os.path.join(seg_dir, "Segmentation.cpp")}"\n' + open(os.path.join(seg_dir, "Segmentation.cpp")).read() +
                    f'\n#include "{os.path.join(HERE, "ref_seg.cpp")}"\n#include "{os.path.join(HERE, "ref_cc.cpp")}"\n')  # ref_cc: the
            # header-only connected-components pass on its own (ConnectedLabels.hpp defines non-inline functions: one TU only)
        obj = os.path.join(tmp, "Segmentation.o")
        subprocess.check_call(["g++", *seg_flags, "-c", gen, "-o", obj])
        objs.append(obj)
        # the odometry host code: RGBDOdometry.{h,cpp} as generated copies under <tmp>/Utils/ so that their `#include "../GPUTexture.h"`
        # (a Pangolin OpenGL texture) and "Stopwatch.h" resolve to the stand-ins copied beside them; OdometryProvider.h, GPUConfig.h
        # and the CUDA headers are read where they lie.  Eigen: the fixed-size stand-in of eigen_fixed/ (FIRST on the include path, the
        # dynamic-matrix stand-in of include/Eigen is for the segmentation sources).  The harness is appended.
        utils_dir = os.path.join(REF, "Core", "Utils")
        os.makedirs(os.path.join(tmp, "Utils"))
        for src_name, dst_name in (("GPUTexture.h", "GPUTexture.h"), (os.path.join("Utils", "Stopwatch.h"), os.path.join("Utils", "Stopwatch.h"))):
            with open(os.path.join(tmp, dst_name), "w") as f:
                f.write(open(os.path.join(HERE, "stub", src_name)).read())
        with open(os.path.join(tmp, "Utils", "RGBDOdometry.h"), "w") as f:
            f.write(f'#line 1 "{os.path.join(utils_dir, "RGBDOdometry.h")}"\n' + open(os.path.join(utils_dir, "RGBDOdometry.h")).read())
        gen = os.path.join(tmp, "Utils", "RGBDOdometry_gen.cpp")
        with open(gen, "w") as f:
            f.write(f'#line 1 "{os.path.join(utils_dir, "RGBDOdometry.cpp")}"\n' + open(os.path.join(utils_dir, "RGBDOdometry.cpp")).read() +
                    f'\n#include "{os.path.join(HERE, "ref_odo.cpp")}"\n')
        odo_flags = san + ["-O2", "-g0" if not san else "-g", "-std=c++14", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-DCUSIM_HOST_TU", "-I", os.path.join(HERE, "eigen_fixed"),
                     "-I", os.path.join(HERE, "include"), "-I", utils_dir, "-I", CUDA, "-include", "cusim.h", "-include", "string", "-include", "sstream",
                     "-include", "iostream", "-include", "limits"]
        obj = os.path.join(tmp, "RGBDOdometry.o")
        subprocess.check_call(["g++", *odo_flags, "-c", gen, "-o", obj])
        objs.append(obj)
        # the frame loop: the reference's own function bodies out of Core/CoFusion.cpp (cut at build time, never stored) behind
        # stub/CoFusionPin.h, followed by the harness; every pass runs on the oracle (stub/Model/Model.h)
        cofusion_cpp = open(os.path.join(REF, "Core", "CoFusion.cpp")).read().split("\n")
        wanted = ["SegmentationResult CoFusion::performSegmentation(", "bool CoFusion::processFrame(", "void CoFusion::predict(",
                  "bool CoFusion::requiresFillIn(", "void CoFusion::spawnObjectModel(", "void CoFusion::moveNewModelToList(",
                  "ModelListIterator CoFusion::inactivateModel(", "unsigned char CoFusion::getNextModelID("]
        pieces = []
        for sig in wanted:
            start = next(i for i, l in enumerate(cofusion_cpp) if l.startswith(sig))
            end = next(i for i in range(start, len(cofusion_cpp)) if cofusion_cpp[i] == "}")
            pieces.append(f'#line {start + 1} "{os.path.join(REF, "Core", "CoFusion.cpp")}"\n' + "\n".join(cofusion_cpp[start:end + 1]))
        # ... and, since round 6, the four Model methods the loop drives per model (Core/Model/Model.cpp, cut the same way): their OpenGL
        # calls are recorded by stub/glpin.h and the draw handler of ref_cofusion.cpp runs the oracle's pass with what was recorded
        model_cpp_path = os.path.join(REF, "Core", "Model", "Model.cpp")
        model_lines = open(model_cpp_path).read().split("\n")
        for sig in ("void Model::initICP(", "void Model::performTracking(", "void Model::fuse(", "void Model::clean("):
            start = next(i for i, l in enumerate(model_lines) if l.startswith(sig))
            end = next(i for i in range(start, len(model_lines)) if model_lines[i] == "}")
            pieces.append(f'#line {start + 1} "{model_cpp_path}"\n' + "\n".join(model_lines[start:end + 1]))
        gen = os.path.join(tmp, "CoFusion_gen.cpp")
        with open(gen, "w") as f:
            f.write('#include "CoFusionPin.h"\n' + "\n".join(pieces) + f'\n#include "{os.path.join(HERE, "ref_cofusion.cpp")}"\n')
        cf_flags = (["-fsanitize=address", "-g", "-O1", "-fno-inline"] if os.environ.get("COFUSION_REF_SANITIZE") else ["-O2", "-g0"]) + [ "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-DCUSIM_HOST_TU", "-DRGBDOdometry=PinOdometry",
                    "-I", os.path.join(HERE, "stub"), "-I", os.path.join(HERE, "eigen_fixed"), "-I", os.path.join(HERE, "include"),
                    "-I", os.path.join(REF, "Core"), "-I", os.path.join(REF, "Core", "Segmentation"), "-I", os.path.dirname(HERE), "-include", "cusim.h"]
        obj = os.path.join(tmp, "CoFusion.o")
        subprocess.check_call(["g++", *cf_flags, "-c", gen, "-o", obj])
        objs.append(obj)
        # Model::computeFusionWeight + Model::rodrigues2: their text out of Core/Model/Model.cpp (cut at build time, never stored), as members
        # of a class that declares what they use (round 5; ref_weight.cpp says what the JacobiSVD stand-in is)
        model_cpp = open(os.path.join(REF, "Core", "Model", "Model.cpp")).read().split("\n")
        pieces = []
        for sig in ("float Model::computeFusionWeight(", "Eigen::Vector3f Model::rodrigues2("):
            start = next(i for i, l in enumerate(model_cpp) if l.startswith(sig))
            end = next(i for i in range(start, len(model_cpp)) if model_cpp[i] == "}")
            pieces.append(f'#line {start + 1} "{os.path.join(REF, "Core", "Model", "Model.cpp")}"\n' + "\n".join(model_cpp[start:end + 1]))
        gen = os.path.join(tmp, "ModelWeight_gen.cpp")
        with open(gen, "w") as f:
            f.write('#include <Eigen/Core>\n#include <algorithm>\n#include <cmath>\n#include <math.h>\n'
                    'namespace Eigen {\n'
                    'enum { ComputeFullU = 1, ComputeFullV = 2 };\n'
                    'template <class M> struct JacobiSVD {   // stand-in: see ref_weight.cpp\n'
                    '    M u, v;\n'
                    '    JacobiSVD(const M& m, int) : u(m) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v(i, j) = (i == j) ? 1.0f : 0.0f; }\n'
                    '    const M& matrixU() const { return u; }\n'
                    '    const M& matrixV() const { return v; }\n'
                    '};\n}\n'
                    'class WeightPinModel {\n'
                    '  public:\n'
                    '    Eigen::Matrix4f pose, lastPose;\n'
                    '    const Eigen::Matrix4f& getPose() const { return pose; }\n'
                    '    Eigen::Matrix4f getLastTransform() const { return getPose().inverse() * lastPose; }   // Model.h:216\n'
                    '    float computeFusionWeight(float weightMultiplier) const;\n'
                    '    static Eigen::Vector3f rodrigues2(const Eigen::Matrix3f& matrix);\n'
                    '};\n'
                    '#define Model WeightPinModel\n' + "\n".join(pieces) + f'\n#undef Model\n#include "{os.path.join(HERE, "ref_weight.cpp")}"\n')
        obj = os.path.join(tmp, "ModelWeight.o")
        subprocess.check_call(["g++", "-O2", "-g0", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w", "-I", os.path.join(HERE, "eigen_fixed"), "-c", gen, "-o", obj])
        objs.append(obj)
        orc_dir = os.path.join(os.path.dirname(HERE), "_build")  # orc_inverse_pose (host-side pose inverse) comes from the oracle
        subprocess.check_call(["g++", "-shared", *(["-fsanitize=address"] if os.environ.get("COFUSION_REF_SANITIZE") else []), "-o", os.path.join(OUT, "libcofusion_ref.so"), *objs, "-L", orc_dir, "-lorc",
                               "-Wl,-rpath,$ORIGIN/../_build"])
    print(os.path.join(OUT, "libcofusion_ref.so"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
