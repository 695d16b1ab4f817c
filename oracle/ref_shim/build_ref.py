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


def main() -> int:
    if not os.path.isdir(CUDA):
        print(f"build_ref: {CUDA} not present, keeping any prebuilt oracle/_ref", file=sys.stderr)
        return 0
    os.makedirs(OUT, exist_ok=True)
    flags = ["-O2", "-g0", "-std=c++14", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w",
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
        subprocess.check_call(["g++", "-shared", "-o", os.path.join(OUT, "libcofusion_ref.so"), *objs])
    print(os.path.join(OUT, "libcofusion_ref.so"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
