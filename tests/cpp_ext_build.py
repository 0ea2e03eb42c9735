"""Builds tests/cpp/vsl_torch_ext.cpp (the libtorch header binding as a Python extension) with g++."""
import os
import subprocess
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "vsl_torch_ext.cpp")
OUT = os.path.join(ROOT, "tests", "cpp", "vsl_torch_ext.so")
PKG = os.path.join(ROOT, "monocular-depth-estimation_amd")


def build_ext(force=False):
    import torch
    deps = [SRC, os.path.join(ROOT, "include", "vsloss_torch.h"), os.path.join(ROOT, "include", "vsloss.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    tdir = os.path.dirname(torch.__file__)
    cmd = ["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           "-DTORCH_EXTENSION_NAME=vsl_torch_ext", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()),
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(tdir, "include"),
           "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-I", "/opt/rocm/include",
           "-I", sysconfig.get_paths()["include"], SRC, "-o", OUT,
           "-L", os.path.join(tdir, "lib"), "-ltorch_python", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
           "-L", PKG, "-lvsloss", "-Wl,-rpath,$ORIGIN/../../monocular-depth-estimation_amd",
           "-Wl,-rpath," + os.path.join(tdir, "lib")]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_ext(force=True))
