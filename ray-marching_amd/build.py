"""Builds the native libraries in-tree (no torch, no cmake):

  librm_hip.so   hipcc --offload-arch=gfx950   kernels + the C ABI of include/rm_abi.h
  librm_host.so  g++                           host-side mirror (CSGNode, camera) + include/rm_host.h

`python -m ray_marching_amd.build` or __graft_entry__.build() run this; hipcc cross-compiles
for gfx950 without a GPU.  The .so files are git-ignored but travel with the tree.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")

HIP_SO = os.path.join(PKG, "librm_hip.so")
HOST_SO = os.path.join(PKG, "librm_host.so")

# -ffp-contract=off: the arithmetic contract forbids fused multiply-add.
# -fno-slp-vectorize: packed-f32 (v_pk_*) code plus its register shuffles measured 3 % slower.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
             "-fno-slp-vectorize",
             "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
              "-fvisibility=hidden", "-Wall", "-Wextra"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(*dirs):
    out = []
    for d in dirs:
        for base, _, files in os.walk(d):
            out += [os.path.join(base, f) for f in files if f.endswith((".h", ".hpp", ".hip", ".cpp"))]
    return out


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, verbose=False, extra=()):
    srcs = _sources(CSRC, INCLUDE)
    if not force and not _newer(HIP_SO, srcs):
        return HIP_SO
    cmd = [hipcc_path()] + HIP_FLAGS + list(extra) + ["-I", INCLUDE, "-I", CSRC, "-o", HIP_SO,
                                                      os.path.join(CSRC, "rm_abi.hip")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HIP_SO


def build_host(force=False, verbose=False):
    src = os.path.join(CSRC, "rm_host.cpp")
    if not os.path.exists(src):
        return None
    srcs = _sources(CSRC, INCLUDE)
    if not force and not _newer(HOST_SO, srcs):
        return HOST_SO
    cmd = [os.environ.get("CXX", "g++")] + HOST_FLAGS + ["-I", INCLUDE, "-I", CSRC, "-o", HOST_SO, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_SO


DROPIN_EXE = os.path.join(PKG, "dropin_test")


def build_cpp_dropin(force=False, verbose=False):
    """tests/cpp/dropin_test.cpp: the C++ host mirror linked against librm_hip.so (host-only C++)."""
    src = os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp")
    build_hip(False, verbose)
    if not force and not _newer(DROPIN_EXE, _sources(CSRC, INCLUDE) + [src]):
        return DROPIN_EXE
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-ffp-contract=off", "-I", INCLUDE, "-I", CSRC,
           "-o", DROPIN_EXE, src, "-L", PKG, "-lrm_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return DROPIN_EXE


def build_all(force=False, verbose=False):
    out = build_hip(force, verbose), build_host(force, verbose)
    build_cpp_dropin(force, verbose)
    return out


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
