"""Build the native libraries in-tree (so the .so files travel to the GPU box with the snapshot).

  popsift_amd/lib/libpopsift_hip.so   C-ABI + HIP kernels for gfx950 (hipcc)
  popsift_amd/lib/libpopsift.so       C++14 host library: PopSift / SiftJob / Config / Features (g++)
  popsift_amd/lib/popsift-testdriver        small C++ driver over the C++ API (raw frames; used by tests)
  popsift_amd/lib/popsift-demo        the command line extractor (PGM/PPM in, output-features.txt out; reference main.cpp)
  popsift_amd/lib/popsift-match       the MatchingMode tool (reference match.cpp)

hipcc cross-compiles gfx950 without a GPU.  Objects are cached by CONTENT: popsift_amd/build/manifest.json records, per object,
the SHA-1 of its source, of every header it may include and of its command line; an object is rebuilt when that hash
differs (file times say nothing after a checkout).  `python -m popsift_amd.build -v` prints what was rebuilt and what was
reused; POPSIFT_BUILD_FORCE=1 rebuilds everything; manifest.json also records when each object was built.
"""
import concurrent.futures as cf
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")

HIP_SOURCES = ["pyramid.hip", "pyramid_tile.hip", "pyramid_alt.hip", "pyramid_fixed.hip", "pyramid_interp.hip", "extrema.hip", "orient_desc.hip", "gridfilter.hip", "match.hip", "util.hip", "api.hip"]
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


MANIFEST = os.path.join(OBJDIR, "manifest.json")
_manifest = None
LAST_BUILD = {"rebuilt": [], "reused": []}       # what the last build_all() did (for __graft_entry__.build())


def _load_manifest():
    global _manifest
    if _manifest is None:
        try:
            _manifest = json.load(open(MANIFEST))
        except Exception:
            _manifest = {}
    return _manifest


def _digest(cmd, files):
    h = hashlib.sha1(" ".join(cmd).encode())
    for f in sorted(files):
        if os.path.exists(f):
            h.update(f.encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def _stale(target, cmd, files):
    """True when `target` must be (re)built: missing, forced, or its recorded content hash differs"""
    key = os.path.relpath(target, HERE)
    d = _digest(cmd, files)
    m = _load_manifest()
    stale = os.environ.get("POPSIFT_BUILD_FORCE") == "1" or not os.path.exists(target) or m.get(key, {}).get("sha1") != d
    (LAST_BUILD["rebuilt"] if stale else LAST_BUILD["reused"]).append(key)
    return stale, key, d


def _record(key, digest):
    m = _load_manifest()
    m[key] = {"sha1": digest, "built": time.strftime("%Y-%m-%dT%H:%M:%S")}
    os.makedirs(OBJDIR, exist_ok=True)
    json.dump(m, open(MANIFEST, "w"), indent=1, sort_keys=True)


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
        cmd = [_hipcc()] + HIP_FLAGS + ["-c", src, "-o", obj]
        stale, key, dg = _stale(obj, cmd, [src] + hdrs)
        if stale:
            jobs.append((cmd, key, dg))
    with cf.ThreadPoolExecutor(max_workers=max(1, min(4, len(jobs) or 1))) as ex:
        for out, (cmd, key, dg) in zip(ex.map(_run, [j[0] for j in jobs]), jobs):
            _record(key, dg)
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
        cmd = ["g++"] + HOST_FLAGS + ["-c", src, "-o", obj]
        stale, key, dg = _stale(obj, cmd, [src] + hdrs + _headers(hostdir))
        if stale:
            _run(cmd)
            _record(key, dg)
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
    LAST_BUILD["rebuilt"], LAST_BUILD["reused"] = [], []
    hip = build_hip(verbose)
    host = build_host(verbose)
    if verbose:
        print("rebuilt: %s" % (", ".join(LAST_BUILD["rebuilt"]) or "nothing"))
        print("reused (content hash unchanged): %s" % (", ".join(LAST_BUILD["reused"]) or "nothing"))
    return hip, host


if __name__ == "__main__":
    print(build_all(verbose="-v" in sys.argv))
