"""Build the native libraries in-tree (so the .so files travel to the GPU box with the snapshot).

  popsift_amd/lib/libpopsift_hip.so   C-ABI + HIP kernels for gfx950 (hipcc)
  popsift_amd/lib/libpopsift.so       C++14 host library: PopSift / SiftJob / Config / Features (g++)
  popsift_amd/lib/popsift-testdriver        small C++ driver over the C++ API (raw frames; used by tests)
  popsift_amd/lib/popsift-demo        the command line extractor (PGM/PPM in, output-features.txt out; reference main.cpp)
  popsift_amd/lib/popsift-match       the MatchingMode tool (reference match.cpp)

hipcc cross-compiles gfx950 without a GPU.  Objects are cached by source mtime.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")

HIP_SOURCES = ["pyramid.hip", "pyramid_tile.hip", "pyramid_alt.hip", "extrema.hip", "orient_desc.hip", "gridfilter.hip", "match.hip", "util.hip", "api.hip"]
HIP_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # no implicit fused multiply-add: the arithmetic order is part of the parity contract
    "-ffp-contract=off",
    "-Wall", "-Wno-unused-function",
    "-I", os.path.join(ROOT, "include"), "-I", os.path.join(CSRC, "hip"),
]

HOST_SOURCES = ["popsift.cpp", "sift_conf.cpp", "features.cpp", "device_prop.cpp", "popsift_c.cpp", "log_dump.cpp"]
HOST_FLAGS = ["-O2", "-std=c++14", "-fPIC", "-Wall", "-pthread",
              "-I", os.path.join(ROOT, "include"), "-I", os.path.join(CSRC, "include")]


def _hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return p.stdout


def _headers(d):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith((".h", ".hpp"))]
    return out


def build_hip(verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = _headers(os.path.join(CSRC, "hip")) + _headers(os.path.join(ROOT, "include"))
    jobs = []
    objs = []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, "hip", s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if _newer(obj, [src] + hdrs):
            jobs.append([_hipcc()] + HIP_FLAGS + ["-c", src, "-o", obj])
    with cf.ThreadPoolExecutor(max_workers=max(1, min(4, len(jobs) or 1))) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out.strip():
                print(out)
    so = os.path.join(LIBDIR, "libpopsift_hip.so")
    if jobs or not os.path.exists(so):
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    return so


def build_host(verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hostdir = os.path.join(CSRC, "host")
    if not os.path.isdir(hostdir) or not all(os.path.exists(os.path.join(hostdir, s)) for s in HOST_SOURCES):
        return None
    hdrs = _headers(os.path.join(CSRC, "include")) + _headers(os.path.join(ROOT, "include"))
    objs, rebuilt = [], False
    for s in HOST_SOURCES:
        src = os.path.join(hostdir, s)
        obj = os.path.join(OBJDIR, "host_" + s.replace(".cpp", ".o"))
        objs.append(obj)
        if _newer(obj, [src] + hdrs):
            _run(["g++"] + HOST_FLAGS + ["-c", src, "-o", obj])
            rebuilt = True
    so = os.path.join(LIBDIR, "libpopsift.so")
    if rebuilt or not os.path.exists(so):
        _run(["g++", "-shared", "-fPIC", "-pthread", "-o", so] + objs +
             ["-L", LIBDIR, "-lpopsift_hip", "-ldl", "-Wl,-rpath,$ORIGIN"])
    demo_src = os.path.join(hostdir, "testdriver_main.cpp")
    demo = os.path.join(LIBDIR, "popsift-testdriver")
    if os.path.exists(demo_src) and (_newer(demo, [demo_src, so] + hdrs)):
        _run(["g++"] + HOST_FLAGS + [demo_src, "-o", demo, "-L", LIBDIR, "-lpopsift", "-lpopsift_hip",
                                      "-Wl,-rpath,$ORIGIN"])
    # command line tools with the reference's option surface (src/application/main.cpp, match.cpp)
    appdir = os.path.join(hostdir, "app")
    for exe, srcs in (("popsift-demo", ["main.cpp", "pgmread.cpp"]), ("popsift-match", ["match.cpp", "pgmread.cpp"])):
        paths = [os.path.join(appdir, f) for f in srcs]
        out = os.path.join(LIBDIR, exe)
        if all(os.path.exists(f) for f in paths) and _newer(out, paths + [so, os.path.join(appdir, "options.h")] + hdrs):
            _run(["g++"] + HOST_FLAGS + paths + ["-o", out, "-L", LIBDIR, "-lpopsift", "-lpopsift_hip", "-Wl,-rpath,$ORIGIN"])
    return so


def build_all(verbose=False):
    hip = build_hip(verbose)
    host = build_host(verbose)
    return hip, host


if __name__ == "__main__":
    print(build_all(verbose="-v" in sys.argv))
